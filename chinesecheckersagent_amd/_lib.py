"""ctypes binding of libccsp.so (include/ccsp.h).  There is no CPU fallback: if the library is
missing, or no MI355X is visible when a GPU entry point is used, this raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CCSP_LIB') or os.path.join(HERE, 'libccsp.so')        # (CCSP_LIB: an experimental build, tools/ only)

NUM_ACTIONS = 294
MAX_MOVES = 126
PLANES = 343
NO_MOVE = 255

OK, EINVAL, ENOMEM, EHIP, ENODEVICE, ESTATE = 0, -1, -2, -3, -4, -5
ST_RUNNING, ST_WON_P1, ST_WON_P2, ST_DISCARD_REPETITION, ST_DISCARD_NO_PROGRESS, ST_ERROR, ST_IDLE = range(7)
EVAL_UNIFORM, EVAL_HASH, EVAL_FORWARD, EVAL_ROLLOUT, EVAL_EXTERNAL = range(5)
MODE_SELFPLAY, MODE_ARENA, MODE_GREEDY_DATA = range(3)
GREEDY_P1, GREEDY_P2, GREEDY_ALTERNATE, GREEDY_RANDOM_START, GREEDY_STOCHASTIC_P1, GREEDY_STOCHASTIC_P2 = 1, 2, 4, 8, 16, 32
GREEDY_MAX = 32
(CNT_EXPANSIONS, CNT_TERMINAL_SIMS, CNT_SIMS, CNT_PLIES, CNT_MCTS_PLIES, CNT_GAMES_WON, CNT_GAMES_DISCARDED,
 CNT_SUM_DEPTH, CNT_SUM_CHILDREN, CNT_SELECT_EDGES, CNT_SAMPLES, CNT_ERRORS) = range(12)
CNT_CACHE_HITS = 15
CNT_COUNT = 16
ADVANCE_REUSE, ADVANCE_LOG_GUARD, ADVANCE_STAGGER, ADVANCE_DEBUG, ADVANCE_OVERLAPPED = 1, 2, 4, 8, 16
REQUEST_MOVES = 128
CNT_NAMES = ['expansions', 'terminal_sims', 'sims', 'plies', 'mcts_plies', 'games_won', 'games_discarded',
             'sum_depth', 'sum_children', 'select_edges', 'samples', 'errors', 'cache_hits']
CNT_INDEX = {name: (i if i < 12 else 15) for i, name in enumerate(CNT_NAMES)}        # cache_hits = CCSP_CNT_CACHE_HITS (15)

STATE_DTYPE = np.dtype([('occ', '<u8', (2,)), ('pos', 'u1', (2, 6)), ('last', 'u1', (4,))])
META_DTYPE = np.dtype([('game', '<u8'), ('ply', '<u4'), ('player', 'u1'), ('pad', 'u1', (3,))])
RESULT_DTYPE = np.dtype([('status', 'u1'), ('reward', 'i1'), ('n_plies', '<u2'), ('n_samples', '<u4'),
                         ('expansions', '<u8')])
# one record of the free-running path's request buffer (ccsp_request)
REQUEST_DTYPE = np.dtype([('state', STATE_DTYPE), ('kind', '<u4'), ('reserved0', '<u4', (2,)), ('player', '<u4'), ('k', '<u4'),
                          ('reserved1', '<u4', (3,))])
assert STATE_DTYPE.itemsize == 32 and META_DTYPE.itemsize == 16 and RESULT_DTYPE.itemsize == 16 and REQUEST_DTYPE.itemsize == 64


class Config(C.Structure):
    _fields_ = [('n_slots', C.c_int32), ('sims', C.c_int32), ('randomised', C.c_int32), ('auto_restart', C.c_int32),
                ('seed', C.c_uint64), ('first_game', C.c_uint64), ('game_stride', C.c_uint64), ('max_games', C.c_uint64),
                ('log_capacity', C.c_uint64), ('device', C.c_int32), ('max_plies', C.c_int32),
                ('mode', C.c_int32), ('arena_det_tau', C.c_int32), ('enforce_move_limit', C.c_int32), ('greedy', C.c_int32),
                ('stuck_limit', C.c_int32), ('pad', C.c_int32)]


class CcspError(RuntimeError):
    pass


_lib = None

_VP = C.c_void_p
_SIGS = {
    'ccsp_device_count': (C.c_int, []),
    'ccsp_version': (C.c_char_p, []),
    'ccsp_last_hip_error': (C.c_char_p, []),
    'ccsp_pack_states': (C.c_int, [_VP, _VP, C.c_int, _VP]),
    'ccsp_movegen': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, _VP, _VP]),
    'ccsp_movegen_packed': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, _VP, _VP]),
    'ccsp_step': (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, _VP, _VP, _VP]),
    'ccsp_encode': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP]),
    'ccsp_greedy_best': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, _VP]),
    'ccsp_debug_movegen_stack_cap': (C.c_int, [C.c_int]),
    'ccsp_debug_plies_per_launch': (C.c_int, [C.c_int]),
    'ccsp_net_plain_size': (C.c_int, []),
    'ccsp_net_packed_size': (C.c_int, []),
    'ccsp_net_pack': (C.c_int, [_VP, _VP]),
    'ccsp_net_forward': (C.c_int, [_VP, _VP, C.c_int, _VP, _VP, _VP, _VP]),
    'ccsp_debug_net_shape': (C.c_int, [C.c_int]),
    'ccsp_create': (_VP, [C.POINTER(Config), C.POINTER(C.c_int)]),
    'ccsp_destroy': (C.c_int, [_VP]),
    'ccsp_reset': (C.c_int, [_VP, _VP]),
    'ccsp_set_positions': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'ccsp_play_plies': (C.c_int, [_VP, C.c_int, C.c_int, _VP]),
    'ccsp_ply_begin': (C.c_int, [_VP, _VP, _VP]),
    'ccsp_root_expand': (C.c_int, [_VP, _VP, _VP, _VP]),
    'ccsp_select': (C.c_int, [_VP, _VP, _VP]),
    'ccsp_expand_backup': (C.c_int, [_VP, _VP, _VP, _VP]),
    'ccsp_expand_backup_select': (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    'ccsp_ply_end': (C.c_int, [_VP, _VP]),
    'ccsp_enable_tree_reuse': (C.c_int, [_VP]),
    'ccsp_advance': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int, _VP]),
    'ccsp_boundary': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int, _VP]),
    'ccsp_encode_requests': (C.c_int, [_VP, C.c_int, _VP, _VP]),
    'ccsp_gather_priors': (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, _VP]),
    'ccsp_debug_table_eval': (C.c_int, [C.c_int, _VP, _VP, C.c_int, _VP, _VP, _VP]),
    'ccsp_net_forward_requests': (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, _VP, _VP]),
    'ccsp_set_advance_limits': (C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    'ccsp_set_stagger_span': (C.c_int, [_VP, C.c_int]),
    'ccsp_debug_advance_budget': (C.c_int, [C.c_int]),
    'ccsp_debug_advance_time_cap': (C.c_int, [C.c_int]),
    'ccsp_debug_advance_deadline': (C.c_int, [C.c_int]),
    'ccsp_debug_read': (C.c_int, [_VP, _VP, C.c_int]),
    'ccsp_debug_read_slots': (C.c_int, [_VP, _VP]),
    'ccsp_read_counters': (C.c_int, [_VP, _VP]),
    'ccsp_read_visit_histogram': (C.c_int, [_VP, _VP]),
    'ccsp_read_slots': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    'ccsp_log_size': (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    'ccsp_log_clear': (C.c_int, [_VP, _VP]),
    'ccsp_log_device_ptrs': (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    'ccsp_read_log': (C.c_int, [_VP, C.c_uint64, C.c_uint64, _VP, _VP, _VP]),
    'ccsp_read_results': (C.c_int, [_VP, C.c_uint64, C.c_uint64, _VP]),
    'ccsp_read_root': (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int), _VP, _VP, _VP, _VP]),
    'ccsp_debug_tree_digest': (C.c_int, [_VP, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
}
EXPORTS = sorted(_SIGS)


def lib():
    """Load libccsp.so.  Raises if it has not been built (python -m chinesecheckersagent_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CcspError('libccsp.so is not built: run `python -m chinesecheckersagent_amd.build` '
                            '(the self-play path has no CPU fallback)')
        # torch first: its wheel bundles its own HIP runtime, and a process must end up with ONE
        # libamdhip64 (ours is resolved against whichever is already loaded)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_blocking_sync = {}


def prefer_blocking_sync(device=None):
    """hipSetDeviceFlags(hipDeviceScheduleBlockingSync): a host thread that waits for the GPU (the harvest's synchronisation, a rank of
    the N-GPU generator between two graph launches) SLEEPS instead of spinning.  Measured on the GPU box (tools/host_spin_probe.py): a
    rank of the generator keeps 2.0 host cores busy by default -- one of them this spin -- and 1.0 with the flag, at the same
    throughput; eight ranks on a 16-core quota leave the converter threads no core otherwise.  Only BEFORE the process touches the
    GPU (launch.init_rank and bench.py do): on a process whose GPU is already initialised this is a no-op (see below).
    CCSP_NO_BLOCKING_SYNC=1 leaves the runtime's default.  -> the runtime's return code, or None if not attempted."""
    key = -1 if device is None else int(device)          # the flags are per DEVICE: a run on device N sets them there, once
    if key in _blocking_sync or os.environ.get('CCSP_NO_BLOCKING_SYNC') == '1':
        return _blocking_sync.get(key)
    try:
        import torch
        if torch.cuda.is_initialized():
            # NEVER on a process that already uses the GPU.  Measured in round 5 (tools/diag_hang.py, gpurun_out/r5g4_diag.log): changing
            # the scheduling flags of an ACTIVE device leaves the runtime waiting for ever inside the next hipFree (torch's
            # empty_cache() in front of a graph capture) -- the rank entry points (launch.init_rank, bench.py) set the flag before
            # their first GPU call; a process that comes here later keeps the runtime's default (a spinning wait: one more host core)
            _blocking_sync[key] = None
            return None
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))   # the runtime torch has loaded: the process's only one
        prev = C.c_int(-1)
        if device is not None:
            if hip.hipGetDevice(C.byref(prev)) != 0:
                prev.value = -1
            hip.hipSetDevice(int(device))                       # the flags are the current device's
        _blocking_sync[key] = int(hip.hipSetDeviceFlags(4))
        if device is not None and prev.value >= 0 and prev.value != int(device):
            hip.hipSetDevice(prev.value)                        # the caller's current device stays current
    except Exception:
        _blocking_sync[key] = -1
    return _blocking_sync[key]


def check(rc, what=''):
    if rc != OK:
        names = {EINVAL: 'EINVAL', ENOMEM: 'ENOMEM', EHIP: 'EHIP', ENODEVICE: 'ENODEVICE', ESTATE: 'ESTATE'}
        detail = lib().ccsp_last_hip_error().decode() if rc == EHIP else ''
        raise CcspError('%s failed: %s %s' % (what or 'ccsp call', names.get(rc, rc), detail))


def require_gpu():
    if lib().ccsp_device_count() <= 0:
        raise CcspError('no HIP device visible: the self-play path runs on MI355X only (no CPU fallback)')


def pack_states(pos12, last4=None):
    """host helper: (n,12) cells [+ (n,4) last moves] -> structured array of 32-byte records"""
    pos12 = np.ascontiguousarray(pos12, dtype=np.uint8).reshape(-1, 12)
    n = len(pos12)
    out = np.zeros(n, dtype=STATE_DTYPE)
    lp = None
    if last4 is not None:
        last4 = np.ascontiguousarray(last4, dtype=np.uint8).reshape(-1, 4)
        assert len(last4) == n
        lp = last4.ctypes.data
    check(lib().ccsp_pack_states(pos12.ctypes.data, lp, n, out.ctypes.data), 'ccsp_pack_states')
    return out
