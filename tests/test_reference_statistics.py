"""The only numbers the reference itself records for any path of SURVEY.md section 8: the comment at the end of game.py
(game.py:103-119) -- 10 000 `Game.start` games of a GreedyPlayer against `GreedyPlayer(player_num=2, stochastic=True)`:

    Counter({1: 5172, 2: 4675, None: 153})
    Counter({1: 5233, 2: 4594, None: 173})

What the IMPORTED REFERENCE does today (tests/golden/greedy_stats.json, made by `oracle/harness/gen_golden.py greedy_stats`: 10 000
Game.start games per seating on the substituted draws, winner and number of moves of every game):

    greedy      vs greedy              {1: 5174, 2: 4685, None: 141}      <- the record, to sampling error
    greedy      vs stochastic greedy   {1: 9275, 2:  301, None: 424}
    stochastic  vs greedy              {1:  342, 2: 9212, None: 446}

i.e. the record is reproduced by two NON-stochastic GreedyPlayers (whose choice among the best moves is itself a random draw,
player.py:122); with the stochastic policy as the file defines it now (player.py:77-97: a forward move drawn with probability
proportional to its distance) the deterministic player wins nine games in ten -- the comment predates that policy.  So:
  * every one of the 30 000 reference games is reproduced exactly (winner, number of moves) by the C oracle and by the HIP engine;
  * the frequencies of 200 000 more games per seating on the GPU (25 seeds = the reference's `np.random.seed()` per game) lie within
    a stated binomial interval of the reference-made sample, and for greedy-vs-greedy of the record in game.py itself."""
import json
import math

import numpy as np
import pytest

import oracle_ffi as orc

RECORD = [{1: 5172, 2: 4675, None: 153}, {1: 5233, 2: 4594, None: 173}]      # game.py:115-116
N_REF = 10000
Z = 3.9            # two-sided 1e-4 per comparison


def _sd(p, n1, n2):
    return math.sqrt(p * (1 - p) * (1.0 / n1 + 1.0 / n2))


def _within(count, n, ref_count, n_ref, what):
    """two-sample binomial interval: |p_ours - p_ref| <= Z sqrt(p (1 - p) (1/n + 1/n_ref)), p = the pooled frequency"""
    p_o, p_r = count / n, ref_count / n_ref
    lim = Z * _sd((count + ref_count) / (n + n_ref), n, n_ref)
    assert abs(p_o - p_r) <= lim, '%s: %.4f here against %.4f in the reference (allowed +-%.4f)' % (what, p_o, p_r, lim)


@pytest.fixture(scope='module')
def stats(golden_dir):
    doc = json.load(open(golden_dir + '/greedy_stats.json'))
    assert doc['n'] == N_REF and sorted(doc['seatings']) == ['gg', 'gs', 'sg']
    return doc


def test_the_record_in_game_py_is_the_greedy_vs_greedy_statistic(stats):
    """pure data: the reference-made samples against the comment in game.py"""
    gg, gs = stats['seatings']['gg']['counts'], stats['seatings']['gs']['counts']
    for rec in RECORD:
        for k in (1, 2, None):
            _within(gg[str(k)], N_REF, rec[k], N_REF, 'greedy vs greedy, %s' % k)
        # ... and NOT the statistic of the seating the comment names: more than 40 standard deviations away
        assert abs(gs['1'] - rec[1]) / N_REF > 40 * _sd(0.5, N_REF, N_REF)


def _oracle_games(stats, seating):
    ev = [orc.EV_GREEDY_STOCHASTIC if s == 's' else orc.EV_GREEDY for s in seating]
    first, n = stats['first_game'], stats['n']
    return [orc.arena_game(stats['seed'], first + g, 1, ev[0], ev[1], True, False) for g in range(n)]


def test_oracle_replays_every_reference_game(stats):
    for seating, d in stats['seatings'].items():
        games = _oracle_games(stats, seating)
        assert [g['winner'] or 0 for g in games] == d['winner'], seating
        assert [g['n_moves'] for g in games] == d['moves'], seating


def _gpu_counts(n, seeds, seating):
    from chinesecheckersagent_amd import greedy
    c = {1: 0, 2: 0, None: 0}
    for s in range(seeds):
        r = greedy.greedy_vs_greedy(n, seed=977 + 31 * s, first_game=s * n, stochastic=(seating[0] == 's', seating[1] == 's'))
        for k in c:
            c[k] += r[k]
    return c


@pytest.mark.gpu
def test_gpu_replays_every_reference_game_and_its_frequencies(stats):
    from chinesecheckersagent_amd import _lib, engine
    G1, G2, S1, S2 = _lib.GREEDY_P1, _lib.GREEDY_P2, _lib.GREEDY_STOCHASTIC_P1, _lib.GREEDY_STOCHASTIC_P2
    for seating, d in stats['seatings'].items():
        n = stats['n']
        bits = G1 | G2 | (S1 if seating[0] == 's' else 0) | (S2 if seating[1] == 's' else 0)
        e = engine.SelfPlayEngine(n_slots=n, sims=1, seed=stats['seed'], first_game=stats['first_game'], max_games=n, log_capacity=1,
                                  arena=True, greedy=bits)
        try:
            for _ in range(256):
                e.play_plies(0, 32)
                if (e.slots()['status'] != _lib.ST_RUNNING).all():
                    break
            res = e.results()
        finally:
            e.close()
        status = res['status'].astype(np.int64)
        assert not (status == _lib.ST_ERROR).any()
        assert np.array_equal(np.where(status <= 2, status, 0), np.array(d['winner'])), seating
        assert np.array_equal(res['n_plies'].astype(np.int64), np.array(d['moves'])), seating
        # 200 000 further games on other seeds: the frequencies
        per, seeds = 8000, 25
        c = _gpu_counts(per, seeds, seating)
        assert c[1] + c[2] + c[None] == per * seeds
        for k in (1, 2, None):
            _within(c[k], per * seeds, d['counts'][str(k)], N_REF, '%s, %s' % (seating, k))
            if seating == 'gg':
                for rec in RECORD:
                    _within(c[k], per * seeds, rec[k], N_REF, 'greedy vs greedy against the record in game.py, %s' % k)
