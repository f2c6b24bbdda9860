"""next-4: GreedyPlayer / GreedyDataGenerator / greedy arena seats on the GPU against the
reference's outputs (tests/golden/greedy.json) and against the CPU oracle."""
import json

import numpy as np
import pytest

import oracle_ffi as orc
from test_gpu_api import TableModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def doc(golden_dir):
    return json.load(open(golden_dir + '/greedy.json'))


def test_greedy_policy_kernel(doc, golden_dir):
    from chinesecheckersagent_amd import _lib, rules
    pos = [c['pos12'] for c in doc['policy']]
    pl = [c['player'] for c in doc['policy']]
    want = [c['best'] for c in doc['policy']]
    # plus every position of the rules fixture, both sides, against the oracle
    g = np.load(golden_dir + '/rules.npz')
    extra = g['pos12'][::3]
    for p12 in extra:
        for side in (1, 2):
            pos.append([int(x) for x in p12]); pl.append(side); want.append(orc.greedy_best(p12, side))
    states = _lib.pack_states(np.array(pos, dtype=np.uint8), None)
    best, count = rules.greedy_best(rules.to_device_states(states), np.array(pl, dtype=np.uint8))
    best, count = best.cpu().numpy(), count.cpu().numpy()
    sizes = set()
    for i, w in enumerate(want):
        assert int(count[i]) == len(w) and [[int(a), int(b)] for a, b in best[i, :count[i]]] == w, 'position %d' % i
        sizes.add(len(w))
    assert len(want) > 2000 and max(sizes) >= 4


def _check_game(hist, reward, o, tag):
    assert reward == o['reward'] and len(hist) == len(o['rows']), tag
    for (bv, pi), (pos12, last, player, idx) in zip(hist, o['rows']):
        assert [int(x) for x in bv.pos12] == [int(x) for x in pos12], tag
        assert [int(x) for x in bv.last4] == [int(x) for x in last], tag
        nz = sorted(int(i) for i in np.nonzero(pi)[0])
        assert nz == sorted(idx) and (pi[nz] == 1.0 / len(idx)).all(), tag


def test_greedy_generator_games(doc):
    from chinesecheckersagent_amd import greedy
    seed = doc['seed']
    for g in doc['games']:                                   # the reference's games, one at a time
        hist, reward = greedy.generate_greedy_games(1, g['randomised'], g['random_start'], seed=seed, first_game=g['game'],
                                                    stuck_limit=doc['limit'])[0]
        tag = 'greedy game %d' % g['game']
        assert reward == g['reward'] and len(hist) == len(g['rows']), tag
        for (bv, pi), r in zip(hist, g['rows']):
            assert [int(x) for x in bv.pos12] == r['pos12'] and [int(x) for x in bv.last4] == r['last'], tag
            nz = [int(i) for i in np.nonzero(pi)[0]]
            assert nz == r['idx'] and orc_bits(pi[nz[0]]) == r['p'], tag
    for randomised, random_start, first in ((False, False, 30000), (True, False, 20170), (False, True, 31000)):
        n = 384                                              # a batch against the oracle (20178 is a stuck game)
        out = greedy.generate_greedy_games(n, randomised, random_start, seed=seed, first_game=first, stuck_limit=200)
        stuck = 0
        for k, (hist, reward) in enumerate(out):
            o = orc.greedy_game(seed, first + k, randomised, random_start, 200)
            _check_game(hist, reward, o, 'batch game %d' % (first + k))
            stuck += o['stuck']
        if randomised:
            assert stuck >= 1


def orc_bits(x):
    import struct
    return struct.unpack('<Q', struct.pack('<d', float(x)))[0]


def test_greedy_arena_seats(doc):
    from chinesecheckersagent_amd import _lib, engine
    seed = doc['seed']
    bits = {('a', 'g'): _lib.GREEDY_P2, ('g', 'a'): _lib.GREEDY_P1, ('g', 'g'): _lib.GREEDY_P1 | _lib.GREEDY_P2}
    for g in doc['arena']:
        e = engine.SelfPlayEngine(n_slots=1, sims=max(g['sims'], 1), seed=seed, first_game=g['game'], max_games=1, log_capacity=1024,
                                  arena=True, enforce_move_limit=g['enforce'], greedy=bits[(g['p1'], g['p2'])])
        for _ in range(64):
            e.play_plies(g['ev'], 16)
            if e.slots()['status'][0] != 0:
                break
        res = e.results()[0]
        e.close()
        tag = 'greedy arena game %d' % g['game']
        winner = int(res['status']) if int(res['status']) in (1, 2) else None
        assert winner == g['winner'] and int(res['n_plies']) == len(g['moves']) and int(res['expansions']) == g['evals'], tag
    # alternating seats (ai_vs_greedy.py:47-48) in one batch, against the oracle: even ids ai = player one
    n, first, sims = 6, 9900, 8
    e = engine.SelfPlayEngine(n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, log_capacity=4096, arena=True,
                              greedy=_lib.GREEDY_P2 | _lib.GREEDY_ALTERNATE)
    for _ in range(64):
        e.play_plies(_lib.EVAL_FORWARD, 16)
        if (e.slots()['status'] != 0).all():
            break
    res = e.results()
    e.close()
    for k in range(n):
        ai_first = (first + k) % 2 == 0
        o = orc.arena_game(seed, first + k, sims, 2 if ai_first else orc.EV_GREEDY, orc.EV_GREEDY if ai_first else 2, True, False)
        assert int(res['n_plies'][k]) == len(o['moves']) and int(res['expansions'][k]) == o['evals'], k
        assert (int(res['status'][k]) if int(res['status'][k]) in (1, 2) else 0) == (o['winner'] or 0), k


def test_stochastic_greedy_seats(golden_dir):
    """GreedyPlayer(stochastic=True) seats of Game.start (player.py:77-97, game.py:111) on the GPU: the reference's eight games
    (against the deterministic greedy player, itself and AiPlayers: tests/golden/greedy_stochastic.json) move for move through
    the result table, and 96 more games of every seating against the oracle; greedy.greedy_vs_greedy(stochastic=...)"""
    from chinesecheckersagent_amd import _lib, engine, greedy
    doc = json.load(open(golden_dir + '/greedy_stochastic.json'))
    seed = doc['seed']
    G1, G2, S1, S2 = _lib.GREEDY_P1, _lib.GREEDY_P2, _lib.GREEDY_STOCHASTIC_P1, _lib.GREEDY_STOCHASTIC_P2
    bits = {('s', 'g'): G1 | G2 | S1, ('g', 's'): G1 | G2 | S2, ('s', 's'): G1 | G2 | S1 | S2, ('a', 's'): G2 | S2, ('s', 'a'): G1 | S1}
    for g in doc['arena']:
        e = engine.SelfPlayEngine(n_slots=1, sims=max(g['sims'], 1), seed=seed, first_game=g['game'], max_games=1, log_capacity=1024,
                                  arena=True, enforce_move_limit=g['enforce'], greedy=bits[(g['p1'], g['p2'])])
        for _ in range(64):
            e.play_plies(g['ev'], 16)
            if e.slots()['status'][0] != 0:
                break
        res, final = e.results()[0], e.slots()
        e.close()
        tag = 'stochastic greedy arena game %d' % g['game']
        winner = int(res['status']) if int(res['status']) in (1, 2) else None
        assert winner == g['winner'] and int(res['n_plies']) == len(g['moves']) and int(res['expansions']) == g['evals'], tag
        # the final position is the one the reference's move list leads to
        pos, last, player = orc.initial_pos12(), orc.NO_LAST.copy(), 1
        for cid, dest in g['moves']:
            pos, last, _ = orc.step(pos, last, player, cid, dest)
            player = 3 - player
        assert [int(x) for x in final['state'][0]['pos'].reshape(12)] == [int(x) for x in pos], tag
    # every seating against the oracle, 32 games each in one batch
    code = {'g': orc.EV_GREEDY, 's': orc.EV_GREEDY_STOCHASTIC}
    for (p1, p2), b in (('s', 'g'), bits[('s', 'g')]), (('g', 's'), bits[('g', 's')]), (('s', 's'), bits[('s', 's')]):
        n, first = 32, 13000
        e = engine.SelfPlayEngine(n_slots=n, sims=1, seed=seed, first_game=first, max_games=n, log_capacity=1, arena=True, greedy=b)
        for _ in range(64):
            e.play_plies(0, 32)
            if (e.slots()['status'] != 0).all():
                break
        res, final = e.results(), e.slots()
        e.close()
        for k in range(n):
            o = orc.arena_game(seed, first + k, 1, code[p1], code[p2], True, False)
            assert int(res['n_plies'][k]) == len(o['moves']) and (int(res['status'][k]) if int(res['status'][k]) in (1, 2) else 0) == (o['winner'] or 0), (p1, p2, k)
            pos, last, player = orc.initial_pos12(), orc.NO_LAST.copy(), 1
            for cid, dest in o['moves']:
                pos, last, _ = orc.step(pos, last, player, cid, dest)
                player = 3 - player
            assert [int(x) for x in final['state'][k]['pos'].reshape(12)] == [int(x) for x in pos], (p1, p2, k)
    c = greedy.greedy_vs_greedy(64, seed=seed, first_game=14000, stochastic=(False, True))
    assert c[1] + c[2] + c[None] == 64 and c[1] + c[2] > 0


def test_greedy_api():
    from chinesecheckersagent_amd import greedy, selfplay
    m = TableModel(2)
    r = greedy.agent_greedy_match(m, 4, sims=8, seed=5, first_game=100)
    assert r is m or r == 'greedy' or r is None
    c = greedy.greedy_vs_greedy(64, seed=5)
    assert c[1] + c[2] + c[None] == 64 and c[1] > 0 and c[2] > 0
    selfplay.set_seed(11, first_game=500)
    gen = greedy.GreedyDataGenerator()
    h1, r1 = gen.generate_play()
    h2, r2 = gen.generate_play()
    assert r1 in (1, -1, 0) and len(h1) > 10 and len(h2) > 10
    o = orc.greedy_game(11, 501, False, False, 200)
    _check_game(h2, r2, o, 'second generate_play = game 501')


def test_train_on_greedy_miniature(golden_dir, tmp_path):
    """train_on_greedy.train in miniature: the three kinds of greedy games, sampling, one epoch, weights file"""
    import os
    from chinesecheckersagent_amd import greedy
    from chinesecheckersagent_amd.model import ResidualCNN
    games = greedy.generate_greedy_training_games(40, seed=3)
    assert len(games) == 40 and all(r in (1, -1, 0) and len(h) > 0 for h, r in games)
    path = greedy.train_on_greedy(120, golden_dir + '/good_model.h5', 7, save_dir=str(tmp_path), epochs=1, seed=3)
    assert path.endswith('greedy-model0007-weights.h5') and os.path.exists(path)
    m = ResidualCNN()
    m.load_weights(path)                                      # the file is a loadable weights file
