"""The CPU oracle against the reference's own search and driver on substituted draws:
rows T1-T4, S2, S3, O1 of SURVEY.md §8a (tests/golden/tree.json, games.json, rng.json)."""
import ctypes as C
import hashlib
import json
import struct

import numpy as np
import pytest

import oracle_ffi as orc


def bits(x):
    return struct.unpack('<Q', struct.pack('<d', float(x)))[0]


def frombits(b):
    return struct.unpack('<d', struct.pack('<Q', b))[0]


@pytest.fixture(scope='module')
def tree(golden_dir):
    return json.load(open(golden_dir + '/tree.json'))


@pytest.fixture(scope='module')
def games(golden_dir):
    return json.load(open(golden_dir + '/games.json'))


def test_search_cases(tree):
    seed = tree['seed']
    assert len(tree['cases']) >= 90
    terminals = 0
    for i, c in enumerate(tree['cases']):
        o = orc.search(c['pos12'], c['last'], c['player'], seed, c['game'], c['nplies'], c['sims'],
                       c['tau'] != 1, c['evaluator'])
        k = o.n_root
        tag = 'case %d (ev=%d sims=%d tau=%s start=%s)' % (i, c['evaluator'], c['sims'], c['tau'], c['start'])
        assert k == len(c['N']), tag
        assert [o.id[j] for j in range(k)] == c['cid'] and [o.dest[j] for j in range(k)] == c['dest'], tag
        assert [o.N[j] for j in range(k)] == c['N'], 'visit counts differ: ' + tag
        assert [bits(o.W[j]) for j in range(k)] == c['W'], 'W bits differ: ' + tag
        assert [bits(o.Q[j]) for j in range(k)] == c['Q'], 'Q bits differ: ' + tag
        assert [bits(o.P[j]) for j in range(k)] == c['P'], 'P (prior + Dirichlet noise) bits differ: ' + tag
        pi = np.array(o.pi[:])
        nz = [int(j) for j in np.nonzero(pi)[0]]
        assert nz == c['pi_idx'] and [bits(pi[j]) for j in nz] == c['pi_bits'], 'pi bits differ: ' + tag
        assert [o.chosen_id, o.chosen_dest] == c['chosen'], tag
        assert o.evals == c['evals'] and o.nodes == c['nodes'] and o.edges == c['edges'], tag
        assert o.digest == c['tree_sha'], 'whole-tree digest differs: ' + tag
        terminals += o.terminals
    assert terminals > 100        # the crafted near-win roots exercise terminal backups (MCTS.py:81-90)


def _check_games(games):
    seed = games['seed']
    seen = set()
    for g in games['games']:
        ev = g['evaluator']
        o = (orc.selfplay(seed, g['game'], g['sims'], ev[0], g['randomised'], evaluator2=ev[1]) if isinstance(ev, list)
             else orc.selfplay(seed, g['game'], g['sims'], ev, g['randomised']))
        status = {orc.ST_WON_P1: 'won', orc.ST_WON_P2: 'won', orc.ST_DISCARD_REPETITION: 'repetition',
                  orc.ST_DISCARD_NO_PROGRESS: 'no_progress'}[o['status']]
        tag = 'game %d' % g['game']
        assert status == g['status'], tag
        seen.add(status)
        assert [[int(x) for x in p] for p in o['plies']] == g['plies'], 'move sequence differs: ' + tag
        assert o['evals'] == g['evals'], tag
        if status == 'won':
            assert o['reward'] == g['reward'], tag
            assert [hashlib.sha256(np.asarray(p, dtype='<f8').tobytes()).hexdigest()[:16] for p in o['pi']] == g['pi_sha'], tag
            assert [[int(x) for x in p] for p in o['hist_pos12']] == g['hist_pos12'], tag
            # O1: utils.convert_to_train_data -- board_x via C1, pi_y, v_y alternating from p1's reward
            n = len(o['pi'])
            bx = np.stack([orc.planes(o['hist_pos12'][i], o['hist_last'][i], 1 + i % 2).reshape(7, 7, 7)
                           for i in range(n)]).astype('<f8')
            vy = [o['reward'] * (1 if i % 2 == 0 else -1) for i in range(n)]
            assert hashlib.sha256(bx.tobytes()).hexdigest() == g['o1']['board_x_sha'], tag
            assert hashlib.sha256(o['pi'].astype('<f8').tobytes()).hexdigest() == g['o1']['pi_y_sha'], tag
            assert vy == g['o1']['v_y'], tag
    return seen


def test_selfplay_games(games):
    assert _check_games(games) == {'won', 'repetition', 'no_progress'}


def test_selfplay_games_at_the_default_simulation_count(golden_dir):
    """a second set of whole reference games (175 simulations per move: config.py:35; normal, randomised, two models,
    all three table evaluators) that only this CPU restatement is checked against"""
    doc = json.load(open(golden_dir + '/games_cpu.json'))
    assert len(doc['games']) >= 6 and {g['sims'] for g in doc['games']} >= {175}
    assert 'won' in _check_games(doc)


def test_draw_spec_known_answers(golden_dir):
    doc = json.load(open(golden_dir + '/rng.json'))
    ka, seed = doc['known'], doc['seed']
    L = orc.lib()
    for x, y in ka['mix64']:
        assert L.orc_mix64(x) == y
    for k, y in ka['rng']:
        assert L.orc_rng(*k) == y
    for u, n, y in ka['choice']:
        assert L.orc_choice(u, n) == y
    for x, y in ka['det_log']:
        assert bits(L.orc_det_log(frombits(x))) == y
    for x, y in ka['det_exp']:
        assert bits(L.orc_det_exp(frombits(x))) == y
    for g, ply, e, y in ka['gamma']:
        assert bits(L.orc_gamma_small(seed, g, ply, e, 0.03)) == y
    for g, ply, k, ys in ka['dirichlet']:
        out = (C.c_double * k)()
        L.orc_dirichlet(seed, g, ply, k, 0.03, out)
        assert [bits(v) for v in out] == ys
    for g, ys in ka['pick_distinct']:
        out = (C.c_int * 12)()
        L.orc_pick_distinct(seed, g, 49, 12, 0, out)
        assert list(out) == ys
    L.orc_forward_eval.argtypes = L.orc_hash_eval.argtypes
    for pos12, player, key, ps, v in ka['hash_eval']:
        a = np.array(pos12, dtype=np.uint8)
        pa = a.ctypes.data_as(C.POINTER(C.c_uint8))
        assert L.orc_state_key(pa, player) == key
        p = (C.c_double * 294)()
        vv = C.c_float()
        L.orc_hash_eval(pa, player, p, C.byref(vv))
        assert [bits(p[i]) for i in range(8)] + [bits(p[293])] == ps and bits(vv.value) == v
    for pos12, player, ps, v in ka['forward_eval']:
        a = np.array(pos12, dtype=np.uint8)
        p = (C.c_double * 294)()
        vv = C.c_float()
        L.orc_forward_eval(a.ctypes.data_as(C.POINTER(C.c_uint8)), player, p, C.byref(vv))
        assert [bits(p[i]) for i in range(8)] + [bits(p[293])] == ps and bits(vv.value) == v
    for g, ply, vec, y in ka['sample_index']:
        arr = (C.c_double * 294)(*[frombits(b) for b in vec])
        assert L.orc_sample_index(L.orc_rng(seed, g, ply, 0, 0, 4), arr, 294) == y


def test_arena_games(golden_dir):
    """next-3: Game.start between two AiPlayers (no root noise, no pre-expansion, tau rule on
    total moves, its own repetition / move-limit rules) against the reference's games"""
    doc = json.load(open(golden_dir + '/arena.json'))
    outcomes = set()
    for g in doc['games']:
        o = orc.arena_game(doc['seed'], g['game'], g['sims'], g['ev1'], g['ev2'], g['tau'] != 1, g['enforce'])
        tag = 'arena game %d' % g['game']
        assert [[int(a), int(b)] for a, b in o['moves']] == g['moves'], tag
        assert (o['winner'] or None) == g['winner'] and o['evals'] == g['evals'], tag
        outcomes.add(o['status'])
    assert {1, 2, 4} <= outcomes          # wins of both sides and the enforced move limit


def test_greedy_policy_games_and_arena(golden_dir):
    """next-4: GreedyPlayer.decide_move, GreedyDataGenerator.generate_play and Game.start with greedy
    seats against the reference (tests/golden/greedy.json)"""
    doc = json.load(open(golden_dir + '/greedy.json'))
    seed = doc['seed']
    assert len(doc['policy']) > 400
    sizes = set()
    for c in doc['policy']:
        got = orc.greedy_best(c['pos12'], c['player'])
        assert got == c['best'], c
        sizes.add(len(got))
    assert len(sizes) >= 3                    # single best move up to several moves sharing the row rule
    seen = set()
    for g in doc['games']:
        o = orc.greedy_game(seed, g['game'], g['randomised'], g['random_start'], stuck_limit=doc['limit'])
        tag = 'greedy game %d' % g['game']
        assert o['moves'] == g['moves'] and o['reward'] == g['reward'] and o['stuck'] == g['stuck'], tag
        assert len(o['rows']) == len(g['rows']), tag
        for (pos12, last, player, idx), r in zip(o['rows'], g['rows']):
            assert [int(x) for x in pos12] == r['pos12'] and [int(x) for x in last] == r['last'], tag
            assert sorted(idx) == r['idx'] and bits(1.0 / len(idx)) == r['p'], tag
        seen.add((g['randomised'], g['random_start'], g['stuck']))
    assert {(False, False, False), (False, True, False), (True, False, False)} <= seen
    seats = {'a': None, 'g': orc.EV_GREEDY}
    for g in doc['arena']:
        e1 = seats[g['p1']] if g['p1'] == 'g' else g['ev']
        e2 = seats[g['p2']] if g['p2'] == 'g' else g['ev']
        o = orc.arena_game(seed, g['game'], max(g['sims'], 1), e1, e2, True, g['enforce'])
        tag = 'greedy arena game %d' % g['game']
        assert [[int(a), int(b)] for a, b in o['moves']] == g['moves'], tag
        assert (o['winner'] or None) == g['winner'] and o['evals'] == g['evals'], tag


def test_stochastic_greedy_player_against_the_reference(golden_dir):
    """GreedyPlayer(stochastic=True) (player.py:77-97): 480 single decisions and 8 whole Game.start games with such seats
    (against the deterministic greedy player, itself, and AiPlayers) made by the reference (tests/golden/greedy_stochastic.json)"""
    doc = json.load(open(golden_dir + '/greedy_stochastic.json'))
    seed = doc['seed']
    for c in doc['decisions']:
        assert list(orc.greedy_stochastic_move(c['pos12'], c['player'], seed, c['game'], c['ply'])) == c['move'], c
    code = {'g': orc.EV_GREEDY, 's': orc.EV_GREEDY_STOCHASTIC}
    assert {(g['p1'], g['p2']) for g in doc['arena']} >= {('s', 'g'), ('g', 's'), ('s', 's'), ('a', 's'), ('s', 'a')}
    for g in doc['arena']:
        o = orc.arena_game(seed, g['game'], max(g['sims'], 1), code.get(g['p1'], g['ev']), code.get(g['p2'], g['ev']), True, g['enforce'])
        tag = 'stochastic greedy arena game %d' % g['game']
        assert [[int(a), int(b)] for a, b in o['moves']] == g['moves'], tag
        assert (o['winner'] or None) == g['winner'] and o['evals'] == g['evals'], tag


def test_pow_overflow_at_det_tau_is_an_error_as_in_the_reference():
    """MCTS.py:132: pow(N, 1/tau) with tau = 0.01 leaves float64 at N = 1210 and Python raises OverflowError.  Near-win
    roots under 2000 simulations put more than 1209 visits on one edge: the restatement reports the error instead of
    pi = inf/inf; with tau = 1, or 1209 simulations, the same search is fine"""
    seed = 99
    found = 0
    for game in range(40):
        pos = orc.near_win_pos12(seed, game, 1)
        o = orc.search(pos, orc.NO_LAST, 1, seed, game, 20, 2000, False, 0)
        if max(o.N[j] for j in range(o.n_root)) < 1210:
            continue
        found += 1
        with pytest.raises(RuntimeError, match='-3'):
            orc.search(pos, orc.NO_LAST, 1, seed, game, 20, 2000, True, 0)
        ok = orc.search(pos, orc.NO_LAST, 1, seed, game, 20, 1209, True, 0)
        assert abs(sum(ok.pi[:]) - 1.0) < 1e-12
        if found == 3:
            break
    assert found == 3
