"""ctypes binding of oracle/libccsp_oracle.so -- the CPU checker.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only (never by the product package)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, 'oracle', 'libccsp_oracle.so')

NACT = 294
MAXMV = 126


class SearchOut(C.Structure):
    _fields_ = [('n_root', C.c_int), ('N', C.c_int * MAXMV), ('W', C.c_double * MAXMV), ('Q', C.c_double * MAXMV),
                ('P', C.c_double * MAXMV), ('id', C.c_uint8 * MAXMV), ('dest', C.c_uint8 * MAXMV),
                ('pi', C.c_double * NACT), ('chosen_id', C.c_int), ('chosen_dest', C.c_int),
                ('evals', C.c_long), ('terminals', C.c_long), ('nodes', C.c_long), ('edges', C.c_long),
                ('digest', C.c_uint64), ('max_depth', C.c_int), ('sum_depth', C.c_long)]


class GameOut(C.Structure):
    _fields_ = [('status', C.c_int), ('reward', C.c_int), ('n_plies', C.c_int), ('n_hist', C.c_int),
                ('evals', C.c_long), ('terminals', C.c_long), ('n_searched', C.c_int)]


EVAL_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_double),
                      C.POINTER(C.c_float), C.c_void_p)

ST_WON_P1, ST_WON_P2, ST_DISCARD_REPETITION, ST_DISCARD_NO_PROGRESS, ST_ERROR = 1, 2, 3, 4, 5


def build():
    src = os.path.join(ROOT, 'oracle', 'ccsp_oracle.c')
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        u8p = C.POINTER(C.c_uint8)
        L.orc_mix64.restype = C.c_uint64; L.orc_mix64.argtypes = [C.c_uint64]
        L.orc_rng.restype = C.c_uint64
        L.orc_rng.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_choice.restype = C.c_uint32; L.orc_choice.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_det_log.restype = C.c_double; L.orc_det_log.argtypes = [C.c_double]
        L.orc_det_exp.restype = C.c_double; L.orc_det_exp.argtypes = [C.c_double]
        L.orc_gamma_small.restype = C.c_double
        L.orc_gamma_small.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double]
        L.orc_dirichlet.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.orc_sample_index.restype = C.c_int
        L.orc_sample_index.argtypes = [C.c_uint64, C.POINTER(C.c_double), C.c_int]
        L.orc_pick_distinct.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_state_key.restype = C.c_uint64; L.orc_state_key.argtypes = [u8p, C.c_int]
        L.orc_hash_eval.argtypes = [u8p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float)]
        L.orc_initial_pos12.argtypes = [u8p]
        L.orc_randomised_pos12.argtypes = [C.c_uint64, C.c_uint64, u8p]
        L.orc_near_win_pos12.argtypes = [C.c_uint64, C.c_uint64, C.c_int, u8p]
        L.orc_movegen.restype = C.c_int; L.orc_movegen.argtypes = [u8p, C.c_int, u8p]
        L.orc_step.restype = C.c_int
        L.orc_step.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, u8p, u8p, u8p]
        L.orc_check_win.restype = C.c_int; L.orc_check_win.argtypes = [u8p]
        L.orc_progress.restype = C.c_int; L.orc_progress.argtypes = [u8p, C.c_int]
        L.orc_planes.argtypes = [u8p, u8p, C.c_int, u8p]
        L.orc_encode_index.restype = C.c_int; L.orc_encode_index.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_decode_index.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_random_move.restype = C.c_int
        L.orc_random_move.argtypes = [u8p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_search.restype = C.c_int
        L.orc_search.argtypes = [u8p, u8p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.POINTER(SearchOut)]
        L.orc_selfplay.restype = C.c_int
        L.orc_selfplay.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_int, u8p, u8p, u8p, u8p, C.POINTER(C.c_double), C.POINTER(GameOut)]
        L.orc_bench_plies.restype = C.c_long
        L.orc_bench_plies.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


class ArenaOut(C.Structure):
    _fields_ = [('winner', C.c_int), ('n_moves', C.c_int), ('evals', C.c_long), ('status', C.c_int)]


def arena_game(seed, game, sims, ev1, ev2, det_tau_initial=True, enforce_move_limit=False, max_moves=2048):
    L = lib()
    L.orc_arena_game.restype = C.c_int
    L.orc_arena_game.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_uint8), C.POINTER(ArenaOut)]
    moves = np.zeros((max_moves, 2), dtype=np.uint8)
    out = ArenaOut()
    L.orc_arena_game(seed, game, sims, ev1, ev2, int(det_tau_initial), int(enforce_move_limit), max_moves,
                     moves.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(out))
    return dict(winner=out.winner, status=out.status, n_moves=out.n_moves, evals=out.evals, moves=moves[:out.n_moves].copy())


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


NO_LAST = np.full(4, 255, dtype=np.uint8)


def movegen(pos12, player):
    a, pa = _u8(pos12)
    out = np.zeros((MAXMV, 2), dtype=np.uint8)
    n = lib().orc_movegen(pa, int(player), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out[:n].copy()


def step(pos12, last4, player, cid, dest, want_board=False):
    a, pa = _u8(pos12)
    l, pl = _u8(last4)
    npos = np.zeros(12, dtype=np.uint8)
    nlast = np.zeros(4, dtype=np.uint8)
    nb = np.zeros((7, 7, 3), dtype=np.uint8)
    w = lib().orc_step(pa, pl, int(player), int(cid), int(dest), npos.ctypes.data_as(C.POINTER(C.c_uint8)),
                       nlast.ctypes.data_as(C.POINTER(C.c_uint8)), nb.ctypes.data_as(C.POINTER(C.c_uint8)))
    return (npos, nlast, w, nb) if want_board else (npos, nlast, w)


def check_win(pos12):
    a, pa = _u8(pos12)
    return lib().orc_check_win(pa)


def progress(pos12, player):
    a, pa = _u8(pos12)
    return lib().orc_progress(pa, int(player))


def planes(pos12, last4, player):
    a, pa = _u8(pos12)
    l, pl = _u8(last4)
    out = np.zeros(343, dtype=np.uint8)
    lib().orc_planes(pa, pl, int(player), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def random_move(pos12, player, seed, game, ply):
    a, pa = _u8(pos12)
    cid, dest = C.c_int(), C.c_int()
    ok = lib().orc_random_move(pa, int(player), seed, game, ply, C.byref(cid), C.byref(dest))
    return (cid.value, dest.value) if ok else None


def initial_pos12():
    out = np.zeros(12, dtype=np.uint8)
    lib().orc_initial_pos12(out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def randomised_pos12(seed, game):
    out = np.zeros(12, dtype=np.uint8)
    lib().orc_randomised_pos12(seed, game, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def near_win_pos12(seed, game, who):
    out = np.zeros(12, dtype=np.uint8)
    lib().orc_near_win_pos12(seed, game, who, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def search(pos12, last4, player, seed, game, ply, sims, det_tau, evaluator, fn=None):
    a, pa = _u8(pos12)
    l, pl = _u8(last4)
    out = SearchOut()
    cb = C.cast(fn, C.c_void_p) if fn is not None else None
    rc = lib().orc_search(pa, pl, int(player), seed, game, ply, sims, int(det_tau), evaluator, cb, None, C.byref(out))
    if rc:
        raise RuntimeError('orc_search failed: %d' % rc)
    return out


def selfplay(seed, game, sims, evaluator, randomised=False, fn=None, max_plies=1024, evaluator2=-1):
    ply_moves = np.zeros((max_plies, 3), dtype=np.uint8)
    hp = np.zeros((max_plies, 12), dtype=np.uint8)
    hl = np.zeros((max_plies, 4), dtype=np.uint8)
    hpl = np.zeros(max_plies, dtype=np.uint8)
    pi = np.zeros((max_plies, NACT), dtype=np.float64)
    out = GameOut()
    u8 = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint8))
    cb = C.cast(fn, C.c_void_p) if fn is not None else None
    lib().orc_selfplay(seed, game, sims, evaluator, int(randomised), int(evaluator2), cb, None, max_plies, u8(ply_moves), u8(hp), u8(hl),
                       u8(hpl), pi.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out))
    n, h = out.n_plies, out.n_hist
    won = out.status in (ST_WON_P1, ST_WON_P2)
    a = h if won else out.n_searched                     # a discarded game hands nothing back, but its searched plies are in the buffers
    return dict(status=out.status, reward=out.reward, plies=ply_moves[:n].copy(), hist_pos12=hp[:h].copy(),
                hist_last=hl[:h].copy(), hist_player=hpl[:h].copy(), pi=pi[:h].copy(), evals=out.evals,
                terminals=out.terminals, n_searched=out.n_searched, searched_pos12=hp[:a].copy(), searched_pi=pi[:a].copy())


EV_GREEDY = 100          # a GreedyPlayer seat in arena_game (next-4)
EV_GREEDY_STOCHASTIC = 101   # GreedyPlayer(stochastic=True)


def greedy_stochastic_move(pos12, player, seed, game, ply):
    """GreedyPlayer(stochastic=True).decide_move: (id, dest), or None without a legal move"""
    L = lib()
    L.orc_greedy_stochastic_move.restype = C.c_int
    L.orc_greedy_stochastic_move.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    a = np.ascontiguousarray(pos12, dtype=np.uint8)
    cid, dest = C.c_int(), C.c_int()
    ok = L.orc_greedy_stochastic_move(a.ctypes.data_as(C.POINTER(C.c_uint8)), int(player), seed, game, ply, C.byref(cid), C.byref(dest))
    return (cid.value, dest.value) if ok else None


def greedy_best(pos12, player):
    """GreedyPlayer.decide_move(training=True): [(id, dest)] of the filtered best moves, in order"""
    L = lib()
    a = np.ascontiguousarray(pos12, dtype=np.uint8)
    out = np.zeros((126, 2), dtype=np.uint8)
    L.orc_greedy_best.restype = C.c_int
    n = L.orc_greedy_best(a.ctypes.data_as(C.POINTER(C.c_uint8)), int(player), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return [[int(x), int(y)] for x, y in out[:n]]


class _GreedyOut(C.Structure):
    _fields_ = [('status', C.c_int), ('reward', C.c_int), ('n_plies', C.c_int), ('n_hist', C.c_int), ('stuck', C.c_int)]


def greedy_game(seed, game, randomised=False, random_start=False, stuck_limit=200, max_plies=1024):
    """GreedyDataGenerator.generate_play: dict(status, reward, stuck, moves[(id, dest)], rows[(pos12, last, player, [idx])])
    -- rows as generate_play returns them (first 3 dropped for randomised boards, first 43 kept when stuck)"""
    L = lib()
    mv = np.zeros((max_plies, 2), dtype=np.uint8)
    hp = np.zeros((max_plies, 12), dtype=np.uint8)
    hl = np.zeros((max_plies, 4), dtype=np.uint8)
    hpl = np.zeros(max_plies, dtype=np.uint8)
    hn = np.zeros(max_plies, dtype=np.int32)
    hi = np.zeros((max_plies, 32), dtype=np.int32)
    out = _GreedyOut()
    p8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    L.orc_greedy_game.restype = C.c_int
    L.orc_greedy_game.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                                  C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_GreedyOut)]
    L.orc_greedy_game(seed, game, int(randomised), int(random_start), stuck_limit, max_plies, p8(mv), p8(hp), p8(hl), p8(hpl), pi(hn), pi(hi),
                      C.byref(out))
    rows = [(hp[i].copy(), hl[i].copy(), int(hpl[i]), [int(x) for x in hi[i, :hn[i]]]) for i in range(out.n_hist)]
    if randomised and not out.stuck:
        rows = rows[3:]                                  # data_generators.py:77-78
    return dict(status=out.status, reward=out.reward, stuck=bool(out.stuck), moves=[[int(a), int(b)] for a, b in mv[:out.n_plies]], rows=rows)
