"""Reference-SHAPED pure-Python / NumPy mirror of the self-play search (CPU baseline form 1).

TEST INFRASTRUCTURE / REPORTED BASELINE ONLY -- imported by tests/ and by bench.py's `cpu_baseline` leg, never
by the product package.  The reference's own files cannot travel to the GPU box, so the CPU number printed next
to the GPU's is taken with this file: an independent restatement that keeps the reference's algorithmic SHAPE --
a NumPy 7x7x3 array plus dict look-ups per position, one object per tree node and per edge with a dict of
statistics, `copy.deepcopy` of the whole position for every child (MCTS.py:104: 85 % of the reference's time),
two Python passes over all edges per selection level -- so that its speed tracks the reference's.  It is pinned
two ways: (1) results: with the four draw sites on the stream of oracle/harness/spec.py it reproduces the C
oracle (and through it the reference's golden trees) bit for bit -- tests/test_pymirror.py; (2) speed: timed
against the imported reference on the same seeded work in the build container -- oracle/harness/time_reference.py
--mirror; the ratio is in CALIBRATION below and is printed with every baseline figure.

What mirrors what (file:line under /root/reference):
    Position                 board.py:9-57      grid / where / who / trail
    Position.jumps_from      board.py:166-211   recursive mirror-hop search, direction order N,E,SE,S,W,NW
    Position.moves_of        board.py:139-162   walks, then hops with the checker lifted
    Position.legal_moves     board.py:215-222
    Position.move            board.py:226-250
    Position.winner          board.py:89-111
    planes                   utils.py:101-160
    TreeNode / TreeEdge      MCTS.py:13-37
    Search.descend           MCTS.py:49-76
    Search.grow_and_backup   MCTS.py:79-118
    Search.run               MCTS.py:121-153
    make_move                selfplay.py:107-133
"""
import copy
import math
import os
import sys
from collections import deque

import numpy as np

_HARNESS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'harness')
if _HARNESS not in sys.path:
    sys.path.insert(0, _HARNESS)
import spec  # noqa: E402  (the draw stream and the table evaluators; pure Python)

SIDE = 7
PER_SIDE = 6
HISTORY_PLANES = 3
TRAIL_LEN = 16
C_PUCT = 3.5
TIE_EPS = 1e-5
NOISE_ALPHA = 0.03
NOISE_SHARE = 0.25
COMPASS = ((-1, 0), (0, 1), (1, 1), (1, 0), (0, -1), (-1, -1))
START_CELLS = {1: ((6, 0), (5, 0), (6, 1), (4, 0), (5, 1), (6, 2)),
               2: ((0, 6), (1, 6), (0, 5), (2, 6), (1, 5), (0, 4))}

# mirror expansions/s divided by the REFERENCE's expansions/s, same seeded work (400 simulations per move, uniform
# table evaluator), same core, build container; measured by oracle/harness/time_reference.py --mirror
CALIBRATION = {
    # same interpreter (CPython 3.9 / NumPy 1.26), same core, identical trees, runs alternated: 0.88, 1.00, 1.22
    'mirror_over_reference_same_interpreter': 1.0,
    # the mirror under the GPU box's interpreter (CPython 3.10 / NumPy 2.2: 234-261 /s) over the reference under
    # the only interpreter it runs on here (CPython 3.9: 177-206 /s): what converts a mirror figure taken on the
    # GPU box into an estimate of the reference there
    'mirror_py310_over_reference_py39': 1.3,
    'reference_exp_per_s_build_container': 189.0,
    'where': 'build container (8 shared vCPUs: +-20 % run to run), one core, 400 simulations per move, uniform table evaluator',
}


def inside(r, c):
    return 0 <= r < SIDE and 0 <= c < SIDE


class Position(object):
    """a game position held the way the reference holds it: array + two dict look-ups + a trail of moves"""

    def __init__(self, cells12=None):
        self.grid = np.zeros((SIDE, SIDE, HISTORY_PLANES), dtype='uint8')
        self.where = [None, {}, {}]                 # id -> (r, c)
        self.who = [None, {}, {}]                   # (r, c) -> id
        self.trail = deque()
        for side in (1, 2):
            for i in range(PER_SIDE):
                rc = START_CELLS[side][i] if cells12 is None else divmod(int(cells12[(side - 1) * PER_SIDE + i]), SIDE)
                self.grid[rc[0], rc[1], 0] = side
                self.where[side][i] = rc
                self.who[side][rc] = i

    def winner(self):
        now = self.grid[:, :, 0]
        first, second = True, True
        for k in range(SIDE - 3, SIDE):
            if first and not np.array_equal(now.diagonal(k), [1] * (SIDE - k)):
                first = False
            if second and not np.array_equal(now.diagonal(-k), [2] * (SIDE - k)):
                second = False
            if not (first or second):
                return 0
        return 1 if first else 2

    def jumps_from(self, found, seen, at):
        r0, c0 = at
        for dr, dc in COMPASS:
            gap = 1
            r, c = r0 + dr, c0 + dc
            usable = True
            while True:                              # first occupied cell along the line
                if not inside(r, c):
                    usable = False
                    break
                if self.grid[r, c, 0] != 0:
                    break
                gap += 1
                r += dr
                c += dc
            if not usable:
                continue
            for _ in range(gap):                     # as many empty cells again behind it
                r += dr
                c += dc
                if not inside(r, c) or self.grid[r, c, 0] != 0:
                    usable = False
                    break
            if not usable or seen[r, c] == 1:
                continue
            found.append((r, c))
            seen[r][c] = 1
            self.jumps_from(found, seen, (r, c))

    def moves_of(self, side, at):
        found = [at]
        seen = np.zeros((SIDE, SIDE), dtype='uint8')
        seen[at] = 1
        for dr, dc in COMPASS:
            r, c = at[0] + dr, at[1] + dc
            if inside(r, c) and self.grid[r, c, 0] == 0:
                found.append((r, c))
                seen[r, c] = 1
        self.grid[at[0], at[1], 0] = 0
        self.jumps_from(found, seen, at)
        self.grid[at[0], at[1], 0] = side
        found.remove(at)
        return found

    def legal_moves(self, side):
        out = {}
        for at in self.where[side].values():
            out[at] = self.moves_of(side, at)
        return out

    def move(self, side, src, dst):
        now = np.copy(self.grid[:, :, 0])
        now[src], now[dst] = now[dst], now[src]
        for i, at in self.where[side].items():
            if at == src:
                self.where[side][i] = dst
                break
        self.who[side][dst] = self.who[side].pop(src)
        self.grid = np.concatenate((np.expand_dims(now, axis=2), self.grid[:, :, :HISTORY_PLANES - 1]), axis=2)
        if len(self.trail) == TRAIL_LEN:
            self.trail.popleft()
        self.trail.append((src, dst))
        return self.winner()

    def cells12(self):
        return [self.where[s][i][0] * SIDE + self.where[s][i][1] for s in (1, 2) for i in range(PER_SIDE)]


def planes(pos, mover):
    """the 7x7x7 input of the evaluator, rebuilt ply by ply from the trail as utils.py:101-160 does"""
    out = np.zeros((SIDE, SIDE, HISTORY_PLANES * 2 + 1))
    other = 3 - mover
    mine = np.zeros((SIDE, SIDE))
    theirs = np.zeros((SIDE, SIDE))
    for i, rc in pos.where[mover].items():
        mine[rc] = i + 1
    for i, rc in pos.where[other].items():
        theirs[rc] = i + 1
    out[:, :, 0], out[:, :, 1] = mine, theirs
    last, moved = len(pos.trail) - 1, other
    for back in range(1, HISTORY_PLANES):
        if not np.any(pos.grid[:, :, back]):
            break
        src, dst = pos.trail[last]
        layer = mine if moved == mover else theirs
        layer[dst], layer[src] = layer[src], layer[dst]
        last -= 1
        moved = 3 - moved
        out[:, :, back * 2], out[:, :, back * 2 + 1] = np.copy(mine), np.copy(theirs)
    if mover == 2:
        out[:, :, HISTORY_PLANES * 2] = np.ones((SIDE, SIDE))
    return out


class TableEvaluator(object):
    """duck-typed `model` (MCTS.py:93): predict(planes) -> (p float64[294], v 0-d float32); decodes the position from
    the planes like the harness' stub model does, so the encode step is paid as in the reference"""

    def __init__(self, kind=spec.EVAL_UNIFORM):
        self.kind = kind
        self.calls = 0

    def predict(self, x):
        self.calls += 1
        mover = 2 if x[0, 0, 6] == 1 else 1
        cells = [0] * 12
        for ch, side in ((0, mover), (1, 3 - mover)):
            rs, cs = np.nonzero(x[:, :, ch])
            for r, c in zip(rs, cs):
                cells[(side - 1) * PER_SIDE + int(x[r, c, ch]) - 1] = int(r) * SIDE + int(c)
        if self.kind == spec.EVAL_UNIFORM:
            p, v = spec.uniform_eval()
        elif self.kind == spec.EVAL_HASH:
            p, v = spec.hash_eval(cells, mover)
        else:
            p, v = spec.forward_eval(cells, mover)
        return np.array(p, dtype='float64'), np.array(v, dtype='float32')


class TreeNode(object):
    def __init__(self, pos, mover):
        self.pos = pos
        self.mover = mover
        self.edges = []
        self.pi = np.zeros(PER_SIDE * SIDE * SIDE, dtype='float64')


class TreeEdge(object):
    def __init__(self, parent, child, prior, src, dst):
        self.parent, self.child = parent, child
        self.mover = parent.mover
        self.src, self.dst = src, dst
        self.stats = {'N': 0, 'W': 0, 'Q': 0, 'P': prior}


class Search(object):
    def __init__(self, root, model, sims, tau, key):
        self.root, self.model, self.sims, self.tau = root, model, sims, tau
        self.seed, self.game, self.ply = key
        self.sim = 0
        self.depth_sum = 0

    def descend(self):
        trail, node, level = [], self.root, 0
        while node.edges:
            visits = 0
            for e in node.edges:
                visits += e.stats['N']
            best, ties = float('-inf'), []
            for e in node.edges:
                u = C_PUCT * e.stats['P'] * np.sqrt(visits) / (1. + e.stats['N'])
                qu = e.stats['Q'] + u
                if qu > best:
                    best, ties = qu, [e]
                elif math.fabs(qu - best) < TIE_EPS:
                    ties.append(e)
            # random.choice(ties) on the substituted stream (a draw is consumed even for a single candidate, as
            # random.choice does in the harness)
            pick = ties[spec.choice_index(spec.rng(self.seed, self.game, self.ply, self.sim, level, spec.P_SELECT), len(ties))]
            trail.append(pick)
            node = pick.child
            level += 1
        self.depth_sum += level
        return node, trail

    def grow_and_backup(self, leaf, trail):
        won = leaf.pos.winner()
        if won:
            for e in trail:
                sign = -1 if e.mover == leaf.mover else 1
                e.stats['N'] += 1
                e.stats['W'] += 1 * sign
                e.stats['Q'] = e.stats['W'] / float(e.stats['N'])
            return
        p, v = self.model.predict(planes(leaf.pos, leaf.mover))
        v = float(v)                                  # legacy NumPy widened the float32 scalar to float64
        for src, dests in leaf.pos.legal_moves(leaf.mover).items():
            cid = leaf.pos.who[leaf.mover][src]
            for dst in dests:
                idx = cid * SIDE * SIDE + dst[0] * SIDE + dst[1]
                nxt = copy.deepcopy(leaf.pos)
                nxt.move(leaf.mover, src, dst)
                child = TreeNode(nxt, 3 - leaf.mover)
                leaf.edges.append(TreeEdge(leaf, child, p[idx], src, dst))
        for e in trail:
            sign = 1 if e.mover == leaf.mover else -1
            e.stats['N'] += 1
            e.stats['W'] += v * sign
            e.stats['Q'] = e.stats['W'] / float(e.stats['N'])

    def run(self):
        for i in range(self.sims):
            self.sim = i
            leaf, trail = self.descend()
            self.grow_and_backup(leaf, trail)
        root = self.root
        for e in root.edges:
            cid = root.pos.who[root.mover][e.src]
            root.pi[cid * SIDE * SIDE + e.dst[0] * SIDE + e.dst[1]] = pow(e.stats['N'], 1. / self.tau)
        root.pi /= np.sum(root.pi)
        idx = spec.sample_index(spec.rng(self.seed, self.game, self.ply, 0, 0, spec.P_SAMPLE), [float(x) for x in root.pi])
        cid, cell = idx // (SIDE * SIDE), idx % (SIDE * SIDE)
        src, dst = root.pos.where[root.mover][cid], divmod(cell, SIDE)
        for e in root.edges:
            if e.src == src and e.dst == dst:
                return root.pi, e
        raise AssertionError('the sampled action has no edge')


def make_move(root, model, tau, seed, game, ply, sims):
    """selfplay.py:107-133 -> (next root, pi, the Search object)"""
    tree = Search(root, model, sims, tau, (seed, game, ply))
    tree.grow_and_backup(root, [])
    noise = spec.dirichlet(seed, game, ply, len(root.edges), NOISE_ALPHA)
    for i, e in enumerate(root.edges):
        e.stats['P'] *= (1. - NOISE_SHARE)
        e.stats['P'] += NOISE_SHARE * noise[i]
    pi, chosen = tree.run()
    nxt = chosen.child
    nxt.edges = []
    return copy.deepcopy(nxt), pi, tree


def random_opening(seed, game, plies=6):
    """selfplay.make_random_move (selfplay.py:83-104) x plies on the substituted stream: a checker that can move,
    then one of its destinations"""
    node = TreeNode(Position(), 1)
    for ply in range(plies):
        moves = node.pos.legal_moves(node.mover)
        starts = list(moves.keys())
        draw = 0
        start = starts[spec.choice_index(spec.rng(seed, game, ply, draw, 0, spec.P_OPENING), len(starts))]
        draw += 1
        while not moves[start]:
            start = starts[spec.choice_index(spec.rng(seed, game, ply, draw, 0, spec.P_OPENING), len(starts))]
            draw += 1
        end = moves[start][spec.choice_index(spec.rng(seed, game, ply, draw, 0, spec.P_OPENING), len(moves[start]))]
        nxt = copy.deepcopy(node.pos)
        nxt.move(node.mover, start, end)
        node = TreeNode(nxt, 3 - node.mover)
    return node


class NumpyNetEvaluator(object):
    """duck-typed `model` (model.py:21-24) on the NumPy float32 restatement of the net (oracle/net_oracle.py): the evaluator of
    BASELINE config 1 ("good_model.h5 on CPU NumPy"); batch of one per call, as MCTS.py:93 calls it"""

    def __init__(self, weights):
        import net_oracle
        self.net, self.weights, self.calls = net_oracle, weights, 0

    def predict(self, x):
        self.calls += 1
        logits, v = self.net.forward(self.weights, np.asarray(x, dtype=np.float32)[None], dtype=np.float32)
        return self.net.softmax64(logits[0]), np.float32(v[0])


def bench_plies(seed, game, sims, plies, kind=spec.EVAL_UNIFORM, model=None, t_stop=None):
    """`plies` searched plies of one game after the six opening plies (fewer if the game ends, or once time.time() passes
    t_stop); returns evaluator calls (= node-expansions)"""
    import time
    model = TableEvaluator(kind) if model is None else model
    calls0 = model.calls
    node = random_opening(seed, game)
    for ply in range(6, 6 + plies):
        if node.pos.winner() or (t_stop is not None and time.time() >= t_stop):
            break
        node, _, _ = make_move(node, model, 1.0 if ply <= 16 else 0.01, seed, game, ply, sims)
    return model.calls - calls0


if __name__ == '__main__':
    import time
    sims = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    plies = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    t0 = time.time()
    n = bench_plies(20261003, 123, sims, plies)
    dt = time.time() - t0
    print('mirror: %d evaluator calls in %.1f s = %.0f node-expansions/s at %d sims/move' % (n, dt, n / dt, sims))
