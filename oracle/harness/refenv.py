"""Import the reference (read-only, /root/reference) into THIS process with its four
random draw sites replaced by the counter-based stream of spec.py.

TEST INFRASTRUCTURE ONLY; runs in the build container under /opt/conda/bin/python3.9
(NumPy 1.26: same legacy scalar promotion as the reference's NumPy 1.14).
Nothing of the reference is copied: it is imported where it lies, and only the
inputs/outputs it produces are written out as fixtures by gen_golden.py.
"""
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'


class _Anything(types.ModuleType):
    """Stub module: any attribute is a do-nothing callable/class (keras / tensorflow are not
    installed; the reference only needs the names to exist at import time, model.py:3-7)."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)

        class _Stub(object):
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return self

            def __getattr__(self, n):
                return _Stub()
        _Stub.__name__ = name
        return _Stub


def _install_stubs():
    for name in ['keras', 'keras.regularizers', 'keras.optimizers', 'keras.models', 'keras.layers',
                 'keras.utils', 'keras.utils.vis_utils', 'keras.backend',
                 'keras.backend.tensorflow_backend', 'tensorflow']:
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)


_install_stubs()
if REF not in sys.path:
    sys.path.insert(0, REF)

import numpy as np            # noqa: E402
import spec                   # noqa: E402

import board as ref_board     # noqa: E402
import utils as ref_utils     # noqa: E402
import MCTS as ref_mcts       # noqa: E402
import selfplay as ref_selfplay   # noqa: E402
import config as ref_config   # noqa: E402
import game as ref_game       # noqa: E402
import player as ref_player   # noqa: E402
import data_generators as ref_datagen   # noqa: E402


class Ctx(object):
    """Where we are in the stream key space; the patched draw sites read it."""
    seed = 0
    game = 0
    ply = 0          # index of the ply being decided (0-based, counts the random opening plies)
    sim = 0          # simulation index within the ply (0-based)
    level = 0        # selection depth within the simulation
    draw = 0         # draw counter inside make_random_move
    log = None       # optional list collecting (purpose, key..., result)


ctx = Ctx()


class _MCTSRandom(object):
    """stands in for the `random` module inside MCTS.py (only .choice is used, MCTS.py:72)."""

    @staticmethod
    def choice(seq):
        u = spec.rng(ctx.seed, ctx.game, ctx.ply, ctx.sim, ctx.level, spec.P_SELECT)
        ctx.level += 1
        return seq[spec.choice_index(u, len(seq))]


class _SelfplayRandom(object):
    """stands in for `random` inside selfplay.py: .seed() is neutralised (selfplay.py:88),
    .choice() reads the opening stream (selfplay.py:95-98)."""

    @staticmethod
    def seed(*a):
        return None

    @staticmethod
    def choice(seq):
        u = spec.rng(ctx.seed, ctx.game, ctx.ply, ctx.draw, 0, spec.P_OPENING)
        ctx.draw += 1
        return seq[spec.choice_index(u, len(seq))]


class _NpRandom(object):
    @staticmethod
    def dirichlet(alpha):
        k = len(alpha)
        a = float(alpha[0])
        return np.array(spec.dirichlet(ctx.seed, ctx.game, ctx.ply, k, a), dtype='float64')

    @staticmethod
    def choice(a, size=None, replace=True, p=None):
        if p is not None:                                     # MCTS.py:140
            u = spec.rng(ctx.seed, ctx.game, ctx.ply, 0, 0, spec.P_SAMPLE)
            idx = spec.sample_index(u, [float(x) for x in p])
            return a[idx]
        assert replace is False                               # board.py:69
        n = int(a)
        return np.array(spec.pick_distinct(ctx.seed, ctx.game, n, int(size)))


class _NpShim(object):
    """numpy with .random replaced; everything else forwards."""
    random = _NpRandom()

    def __getattr__(self, name):
        return getattr(np, name)


ref_mcts.random = _MCTSRandom()
ref_mcts.np = _NpShim()
ref_selfplay.random = _SelfplayRandom()
ref_selfplay.np = _NpShim()
ref_board.np = _NpShim()

# count simulations: wrap moveToLeaf so ctx.sim/ctx.level follow MCTS.search's loop (MCTS.py:123-125)
_orig_move_to_leaf = ref_mcts.MCTS.moveToLeaf


def _move_to_leaf(self):
    ctx.level = 0
    out = _orig_move_to_leaf(self)
    return out


_orig_expand = ref_mcts.MCTS.expandAndBackUp


def _expand(self, leaf, breadcrumbs):
    out = _orig_expand(self, leaf, breadcrumbs)
    if getattr(self, '_ccsp_in_search', False):
        ctx.sim += 1
    return out


_orig_search = ref_mcts.MCTS.search


def _search(self):
    ctx.sim = 0
    self._ccsp_in_search = True
    try:
        return _orig_search(self)
    finally:
        self._ccsp_in_search = False


ref_mcts.MCTS.moveToLeaf = _move_to_leaf
ref_mcts.MCTS.expandAndBackUp = _expand
ref_mcts.MCTS.search = _search


def set_sims(n):
    """sims/move can only be set through the default argument (SURVEY.md §5 caveat, MCTS.py:41)."""
    ref_mcts.MCTS.__init__.__defaults__ = (ref_config.C_PUCT, int(n), ref_config.TREE_TAU)


class TableModel(object):
    """duck-typed evaluator (MCTS.py:93 contract: predict(x[7,7,7]) -> (p f64[294], v 0-d f32)).
    Recovers the position from the planes (ch0/ch1 hold checker id+1, ch6 = player-2 flag;
    utils.py:101-160) and returns the spec's table evaluator."""

    def __init__(self, kind):
        self.kind = kind
        self.calls = 0

    def predict(self, x):
        self.calls += 1
        if self.kind == spec.EVAL_UNIFORM:
            p, v = spec.uniform_eval()
        else:
            player = 2 if x[0, 0, 6] == 1 else 1
            cur = {}
            opp = {}
            for r in range(7):
                for c in range(7):
                    if x[r, c, 0] != 0:
                        cur[int(x[r, c, 0]) - 1] = r * 7 + c
                    if x[r, c, 1] != 0:
                        opp[int(x[r, c, 1]) - 1] = r * 7 + c
            p1, p2 = (cur, opp) if player == 1 else (opp, cur)
            pos12 = [p1[i] for i in range(6)] + [p2[i] for i in range(6)]
            p, v = (spec.hash_eval if self.kind == spec.EVAL_HASH else spec.forward_eval)(pos12, player)
        return np.array(p, dtype='float64'), np.array([[v]], dtype='float32').squeeze()


def pos12_of(b):
    return [b.checkers_pos[1][i][0] * 7 + b.checkers_pos[1][i][1] for i in range(6)] + \
           [b.checkers_pos[2][i][0] * 7 + b.checkers_pos[2][i][1] for i in range(6)]


def last_moves_of(b):
    """(from1,to1,from2,to2) cell indices of hist_moves[-1], [-2] as to_model_input uses them
    (utils.py:135-155): 255 where the matching history plane is still all-zero."""
    out = [255, 255, 255, 255]
    h = list(b.hist_moves)
    for ch in (1, 2):
        if not b.board[:, :, ch].any():
            break
        mv = h[len(h) - ch]
        out[(ch - 1) * 2] = mv[0][0] * 7 + mv[0][1]
        out[(ch - 1) * 2 + 1] = mv[1][0] * 7 + mv[1][1]
    return out


# ---- next-4: GreedyPlayer (player.py:67-129) and GreedyDataGenerator (data_generators.py:14-80)
class _GreedyRandom(object):
    """stands in for `random` inside player.py and data_generators.py.  A list of ((row, col), (row, col)) move
    pairs is the greedy choice among the filtered best moves (player.py:122, data_generators.py:51): one draw
    keyed by the ply.  Anything else is the generator's random start (data_generators.py:33-37), which is
    selfplay.make_random_move's rule on the opening stream."""

    @staticmethod
    def seed(*a):
        return None

    @staticmethod
    def choice(seq):
        first = seq[0]
        if isinstance(first, tuple) and len(first) == 2 and isinstance(first[0], tuple):
            u = spec.rng(ctx.seed, ctx.game, ctx.ply, 0, 0, spec.P_GREEDY)
            return seq[spec.choice_index(u, len(seq))]
        u = spec.rng(ctx.seed, ctx.game, ctx.ply, ctx.draw, 0, spec.P_OPENING)
        ctx.draw += 1
        return seq[spec.choice_index(u, len(seq))]


ref_player.random = _GreedyRandom()
ref_datagen.random = _GreedyRandom()


class _PlayerNpRandom(object):
    """stands in for np.random inside player.py: GreedyPlayer(stochastic=True) draws the index of a forward move with
    probability proportional to its forward distance (player.py:94-96: np.random.choice(len(forward_moves), p=prior)) -- the ply's
    ONE greedy draw, read through spec.sample_index (cumulative sums, / last, first one above u) like MCTS.py:140's"""

    @staticmethod
    def choice(a, size=None, replace=True, p=None):
        assert p is not None and isinstance(a, int)
        u = spec.rng(ctx.seed, ctx.game, ctx.ply, 0, 0, spec.P_GREEDY)
        return spec.sample_index(u, [float(x) for x in p])


class _PlayerNpShim(object):
    random = _PlayerNpRandom()

    def __getattr__(self, name):
        return getattr(np, name)


ref_player.np = _PlayerNpShim()


class FakeClock(object):
    """data_generators.py:61 declares a game stuck after STUCK_TIME_LIMIT (0.1 s) of WALL CLOCK.  The restatement
    counts plies instead: now() advances 0.1 s / limit per call (one call when the game starts, one after every
    ply without a winner), so the reference gives up right after ply limit + 1 -- exactly the ply rule."""

    def __init__(self, limit):
        import datetime as _dt
        self._dt = _dt
        self.limit = int(limit)
        assert 100000 % self.limit == 0
        self.calls = 0

    def now(self):
        t = self._dt.datetime(2020, 1, 1) + self._dt.timedelta(microseconds=self.calls * (100000 // self.limit))
        self.calls += 1
        return t
