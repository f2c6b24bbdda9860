"""next-3: the arena (Game.start between two AiPlayers) on the GPU tree against the
reference's games (tests/golden/arena.json)."""
import json

import numpy as np
import pytest

import oracle_ffi as orc
from test_gpu_api import TableModel

pytestmark = pytest.mark.gpu


def _follow(moves, st, meta):
    pos, last, player = orc.initial_pos12(), orc.NO_LAST.copy(), 1
    assert len(meta) == len(moves)
    for i, (cid, dest) in enumerate(moves):
        assert [int(x) for x in st[i]['pos'].reshape(12)] == [int(x) for x in pos] and int(meta[i]['player']) == player
        pos, last, _ = orc.step(pos, last, player, cid, dest)
        player = 3 - player


def test_fused_arena_games(golden_dir):
    from chinesecheckersagent_amd import _lib, engine
    doc = json.load(open(golden_dir + '/arena.json'))
    n = 0
    for g in doc['games']:
        if g['ev1'] != g['ev2']:
            continue
        e = engine.SelfPlayEngine(n_slots=1, sims=g['sims'], seed=doc['seed'], first_game=g['game'], max_games=1,
                                  log_capacity=1024, arena=True, arena_det_tau=(g['tau'] != 1), enforce_move_limit=g['enforce'])
        for _ in range(64):
            e.play_plies(g['ev1'], 16)
            if e.slots()['status'][0] != 0:
                break
        res = e.results()[0]
        st, meta, pi = e.log()
        e.close()
        tag = 'arena game %d' % g['game']
        winner = int(res['status']) if int(res['status']) in (1, 2) else None
        assert winner == g['winner'], tag
        assert int(res['n_plies']) == len(g['moves']) and int(res['expansions']) == g['evals'], tag
        order = np.argsort(meta['ply'])
        _follow(g['moves'], st[order], meta[order])
        n += 1
    assert n >= 4


def test_arena_api_two_models(golden_dir):
    from chinesecheckersagent_amd import arena
    doc = json.load(open(golden_dir + '/arena.json'))
    g = [x for x in doc['games'] if x['ev1'] != x['ev2']][0]
    b = arena.BatchArena(TableModel(g['ev1']), TableModel(g['ev2']), 1, sims=g['sims'], seed=doc['seed'], first_game=g['game'],
                         tree_tau=g['tau'], enforce_move_limit=g['enforce'])
    winners, res = b.run()
    st, meta, pi = b.eng.log()
    b.close()
    assert winners[0] == g['winner'] and int(res['n_plies'][0]) == len(g['moves']) and int(res['expansions'][0]) == g['evals']
    order = np.argsort(meta['ply'])
    _follow(g['moves'], st[order], meta[order])
    # agent_match / evaluate with the net: well-formed results
    w1, w2, d = arena.evaluate(golden_dir + '/good_model.h5', golden_dir + '/good_model.h5', 4, enforce_move_limit=True, sims=8, seed=3)
    assert w1 + w2 + d == 4
    r = arena.agent_match(TableModel(2), TableModel(2), 2, sims=8, seed=doc['seed'], first_game=7000)
    assert r is None or isinstance(r, TableModel)


def test_arena_graph_replay_and_shared_model(golden_dir):
    """the net arena's simulation steps replayed from a captured hipGraph, and one forward per step when both seats share
    a model: the same games, move for move, as plain launches with two separately loaded copies of the model"""
    import torch
    from chinesecheckersagent_amd import arena
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN(); m.load_weights(golden_dir + '/good_model.h5')
    m2 = ResidualCNN(); m2.load_weights(golden_dir + '/good_model.h5')
    out = []
    for models, graph in (((m, None), True), ((m, m2), False)):
        b = arena.BatchArena(models[0], models[1], 6, sims=13, seed=11, first_game=300, enforce_move_limit=True, alternate=True,
                             use_graph=graph)
        for _ in range(12):
            b.play_move()
        torch.cuda.synchronize()
        assert (b._graph is not None) == graph and (b.m2 is b.m1) == graph
        st, meta, pi = b.eng.log()
        order = np.lexsort((meta['ply'], meta['game']))
        out.append((st[order].tobytes(), pi[order].tobytes(), b.eng.counters()))
        b.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert out[0][2]['errors'] == 0 and out[0][2]['samples'] == 6 * 12
