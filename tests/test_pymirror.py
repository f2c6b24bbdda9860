"""oracle/pymirror.py (the reference-shaped pure-Python baseline of bench.py's cpu_baseline leg, SURVEY.md §8d form 1)
is pinned to the C oracle -- and through it to the reference's golden trees -- on identical substituted draws."""
import os
import sys

import numpy as np

import oracle_ffi as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import pymirror  # noqa: E402
from pymirror import spec  # noqa: E402


def _record(node):
    pos12 = np.array(node.pos.cells12(), dtype=np.uint8)
    last = [255] * 4
    for k, (src, dst) in enumerate(list(node.pos.trail)[::-1][:2]):
        last[2 * k], last[2 * k + 1] = src[0] * 7 + src[1], dst[0] * 7 + dst[1]
    return pos12, np.array(last, dtype=np.uint8)


def test_move_lists_equal_the_oracle():
    for game in range(40):
        node = pymirror.random_opening(11, game, plies=game % 9)
        pos12, _ = _record(node)
        for side in (1, 2):
            want = [(int(i), int(d)) for i, d in orc.movegen(pos12, side)]
            got = [(node.pos.who[side][src], d[0] * 7 + d[1]) for src, ds in node.pos.legal_moves(side).items() for d in ds]
            assert got == want


def test_searched_plies_equal_the_oracle():
    seed = 20261003
    for game, kind, sims, tau in ((3, spec.EVAL_UNIFORM, 40, 1.0), (4, spec.EVAL_HASH, 60, 1.0), (5, spec.EVAL_FORWARD, 50, 0.01)):
        node = pymirror.random_opening(seed, game)
        model = pymirror.TableEvaluator(kind)
        for ply in (6, 7):
            pos12, last = _record(node)
            o = orc.search(pos12, last, node.mover, seed, game, ply, sims, tau != 1.0, kind)
            nxt, pi, tree = pymirror.make_move(node, model, tau, seed, game, ply, sims)
            assert [e.stats['N'] for e in node.edges] == [o.N[j] for j in range(o.n_root)]
            assert [float(e.stats['W']) for e in node.edges] == [o.W[j] for j in range(o.n_root)]
            assert np.array_equal(pi, np.array(o.pi[:]))
            moved = [k for k in range(12) if nxt.pos.cells12()[k] != pos12[k]]
            assert moved == [(node.mover - 1) * 6 + o.chosen_id] and nxt.pos.cells12()[moved[0]] == o.chosen_dest
            node = nxt
        assert 0 < model.calls <= 2 * (sims + 1)                 # one call per non-terminal expansion (MCTS.py:93)
