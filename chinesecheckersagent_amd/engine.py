"""Host-side handle on the GPU-resident self-play engine (ccsp_ctx of include/ccsp.h).

PyTorch is used for device memory and streams only; every computation happens in libccsp.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (CNT_COUNT, CNT_NAMES, META_DTYPE, NUM_ACTIONS, PLANES, RESULT_DTYPE, STATE_DTYPE, check)


def _stream_ptr(stream=None):
    """hipStream_t of torch's current stream (kernels are stream-ordered with torch work)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


class SelfPlayEngine(object):
    """`n_slots` concurrent games on one GPU.  Game ids are first_game + k * game_stride."""

    def __init__(self, n_slots, sims, seed, first_game=0, game_stride=1, max_games=None, log_capacity=None,
                 randomised=False, auto_restart=False, device=0, max_plies=0, arena=False, arena_det_tau=True,
                 enforce_move_limit=False, greedy=0, greedy_data=False, stuck_limit=0):
        """greedy: _lib.GREEDY_* bits (GreedyPlayer seats of the arena / random start of the generator);
        greedy_data: GreedyDataGenerator mode (no search; play_plies() runs whole plies in one kernel)"""
        _lib.require_gpu()
        self.L = _lib.lib()
        self.n_slots, self.sims = int(n_slots), int(sims)
        self.max_games = int(max_games if max_games is not None else n_slots)
        self.log_capacity = int(log_capacity if log_capacity is not None else self.max_games * 512)
        cfg = _lib.Config(n_slots=self.n_slots, sims=self.sims, randomised=int(bool(randomised)),
                          auto_restart=int(bool(auto_restart)), seed=int(seed), first_game=int(first_game),
                          game_stride=int(game_stride), max_games=self.max_games, log_capacity=self.log_capacity,
                          device=int(device), max_plies=int(max_plies),
                          mode=_lib.MODE_GREEDY_DATA if greedy_data else (_lib.MODE_ARENA if arena else _lib.MODE_SELFPLAY),
                          arena_det_tau=int(bool(arena_det_tau)), enforce_move_limit=int(bool(enforce_move_limit)),
                          greedy=int(greedy), stuck_limit=int(stuck_limit), pad=0)
        err = C.c_int(0)
        self.ctx = self.L.ccsp_create(C.byref(cfg), C.byref(err))
        if not self.ctx:
            check(err.value or _lib.EHIP, 'ccsp_create')
        self.first_game, self.game_stride, self.device = int(first_game), int(game_stride), int(device)

    def close(self):
        if getattr(self, 'ctx', None):
            self.L.ccsp_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    # ---- control -----------------------------------------------------------------------------------
    def reset(self, stream=None):
        check(self.L.ccsp_reset(self.ctx, _stream_ptr(stream)), 'ccsp_reset')

    def set_positions(self, states, player, game, ply, det_tau, stream=None):
        n = self.n_slots
        states = np.ascontiguousarray(states, dtype=STATE_DTYPE)
        player = np.ascontiguousarray(player, dtype=np.uint8)
        game = np.ascontiguousarray(game, dtype=np.uint64)
        ply = np.ascontiguousarray(ply, dtype=np.uint32)
        det_tau = np.ascontiguousarray(det_tau, dtype=np.uint8)
        assert len(states) == len(player) == len(game) == len(ply) == len(det_tau) == n
        check(self.L.ccsp_set_positions(self.ctx, states.ctypes.data, player.ctypes.data, game.ctypes.data,
                                        ply.ctypes.data, det_tau.ctypes.data, _stream_ptr(stream)), 'ccsp_set_positions')

    def play_plies(self, evaluator, n_plies, stream=None):
        check(self.L.ccsp_play_plies(self.ctx, int(evaluator), int(n_plies), _stream_ptr(stream)), 'ccsp_play_plies')

    # stepped path: tensors are torch CUDA tensors owned by the caller
    def ply_begin(self, planes, stream=None):
        assert planes.is_cuda and planes.dtype.is_floating_point and planes.numel() == self.n_slots * PLANES
        check(self.L.ccsp_ply_begin(self.ctx, planes.data_ptr(), _stream_ptr(stream)), 'ccsp_ply_begin')

    def root_expand(self, p, v, stream=None):
        self._check_pv(p, v)
        check(self.L.ccsp_root_expand(self.ctx, p.data_ptr(), v.data_ptr(), _stream_ptr(stream)), 'ccsp_root_expand')

    def select(self, planes, stream=None):
        check(self.L.ccsp_select(self.ctx, planes.data_ptr(), _stream_ptr(stream)), 'ccsp_select')

    def expand_backup(self, p, v, stream=None):
        self._check_pv(p, v)
        check(self.L.ccsp_expand_backup(self.ctx, p.data_ptr(), v.data_ptr(), _stream_ptr(stream)), 'ccsp_expand_backup')

    def expand_backup_select(self, p, v, planes, stream=None):
        """expand_backup(p, v) and the next simulation's select(planes) in one launch"""
        self._check_pv(p, v)
        check(self.L.ccsp_expand_backup_select(self.ctx, p.data_ptr(), v.data_ptr(), planes.data_ptr(), _stream_ptr(stream)),
              'ccsp_expand_backup_select')

    def ply_end(self, stream=None):
        check(self.L.ccsp_ply_end(self.ctx, _stream_ptr(stream)), 'ccsp_ply_end')

    # free-running stepped path
    def enable_tree_reuse(self):
        """the second tree pool ccsp_advance(reuse=True) needs (an allocation: call it before capturing a graph)"""
        check(self.L.ccsp_enable_tree_reuse(self.ctx), 'ccsp_enable_tree_reuse')

    def set_stagger_span(self, boundary_calls):
        """boundary(stagger=True): the number of boundary calls over which the slots' first games begin (default: `sims`)"""
        check(self.L.ccsp_set_stagger_span(self.ctx, int(min(boundary_calls, 65535))), 'ccsp_set_stagger_span')

    def request_buffers(self):
        """the caller-owned hand-off buffers of the free-running path, zero-filled: (req [n_slots, 64] uint8 = ccsp_request records,
        moves [n_slots, 128] int16 = the requests' legal moves, pk [n_slots, 128] float64 = a compact answer to start with, v [n_slots])"""
        import torch
        dev = torch.device('cuda', self.device)
        return (torch.zeros((self.n_slots, 64), dtype=torch.uint8, device=dev),
                torch.zeros((self.n_slots, _lib.REQUEST_MOVES), dtype=torch.int16, device=dev),
                torch.zeros((self.n_slots, _lib.REQUEST_MOVES), dtype=torch.float64, device=dev),
                torch.zeros(self.n_slots, dtype=torch.float32, device=dev))

    def set_advance_limits(self, budget=-1, time_cap=-1, deadline=-1):
        """this context's limits of ccsp_advance (ticks of 10 ns; a negative value leaves that one alone)"""
        check(self.L.ccsp_set_advance_limits(self.ctx, int(budget), int(time_cap), int(deadline)), 'ccsp_set_advance_limits')

    def advance(self, pk, v, req, moves, model_sel=None, reuse=False, log_guard=False, stream=None, debug=False):
        """slots in a search: take the compact answer (pk, v) to the leaf they asked about, go on to their next request (req, moves out)"""
        self._check_hand_off(pk, v, req, moves)
        flags = (_lib.ADVANCE_REUSE if reuse else 0) | (_lib.ADVANCE_LOG_GUARD if log_guard else 0) | (_lib.ADVANCE_DEBUG if debug else 0)
        check(self.L.ccsp_advance(self.ctx, pk.data_ptr(), v.data_ptr(), req.data_ptr(), moves.data_ptr(),
                                  model_sel.data_ptr() if model_sel is not None else None, flags, _stream_ptr(stream)), 'ccsp_advance')

    def boundary(self, pk, v, req, moves, model_sel=None, reuse=False, log_guard=False, stagger=False, stream=None, debug=False, overlapped=False):
        """slots between two searches: root expansion from the answer, or the finished ply's move and rules and the next ply's root"""
        self._check_hand_off(pk, v, req, moves)
        flags = ((_lib.ADVANCE_REUSE if reuse else 0) | (_lib.ADVANCE_LOG_GUARD if log_guard else 0) | (_lib.ADVANCE_STAGGER if stagger else 0) |
                 (_lib.ADVANCE_DEBUG if debug else 0) | (_lib.ADVANCE_OVERLAPPED if overlapped else 0))
        check(self.L.ccsp_boundary(self.ctx, pk.data_ptr(), v.data_ptr(), req.data_ptr(), moves.data_ptr(),
                                   model_sel.data_ptr() if model_sel is not None else None, flags, _stream_ptr(stream)), 'ccsp_boundary')

    def _check_hand_off(self, pk, v, req, moves):
        import torch
        n, m = self.n_slots, _lib.REQUEST_MOVES
        assert pk.is_cuda and pk.dtype == torch.float64 and pk.is_contiguous() and pk.numel() == n * m
        assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and v.numel() == n
        assert req.is_cuda and req.dtype == torch.uint8 and req.is_contiguous() and req.numel() == n * 64
        assert moves.is_cuda and moves.dtype == torch.int16 and moves.is_contiguous() and moves.numel() == n * m

    def _check_pv(self, p, v):
        import torch
        assert p.is_cuda and p.dtype == torch.float64 and p.is_contiguous() and p.numel() == self.n_slots * NUM_ACTIONS
        assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and v.numel() == self.n_slots

    # ---- read-back -----------------------------------------------------------------------------------
    def counters(self):
        out = np.zeros(CNT_COUNT, dtype=np.uint64)
        check(self.L.ccsp_read_counters(self.ctx, out.ctypes.data), 'ccsp_read_counters')
        return {name: int(out[_lib.CNT_INDEX[name]]) for name in CNT_NAMES}

    def debug_read(self, clear=True):
        out = np.zeros(64, dtype=np.uint64)
        check(self.L.ccsp_debug_read(self.ctx, out.ctypes.data, int(clear)), 'ccsp_debug_read')
        return [int(x) for x in out]

    def debug_read_slots(self):
        out = np.zeros((self.n_slots, 20), dtype=np.uint64)
        check(self.L.ccsp_debug_read_slots(self.ctx, out.ctypes.data), 'ccsp_debug_read_slots')
        return out

    def raw_counters(self):
        out = np.zeros(CNT_COUNT, dtype=np.uint64)
        check(self.L.ccsp_read_counters(self.ctx, out.ctypes.data), 'ccsp_read_counters')
        return [int(x) for x in out]

    def visit_histogram(self):
        out = np.zeros(NUM_ACTIONS, dtype=np.uint64)
        check(self.L.ccsp_read_visit_histogram(self.ctx, out.ctypes.data), 'ccsp_read_visit_histogram')
        return out

    def slots(self):
        n = self.n_slots
        status = np.zeros(n, dtype=np.uint8)
        ply = np.zeros(n, dtype=np.uint32)
        game = np.zeros(n, dtype=np.uint64)
        state = np.zeros(n, dtype=STATE_DTYPE)
        player = np.zeros(n, dtype=np.uint8)
        check(self.L.ccsp_read_slots(self.ctx, status.ctypes.data, ply.ctypes.data, game.ctypes.data, state.ctypes.data,
                                     player.ctypes.data), 'ccsp_read_slots')
        return dict(status=status, ply=ply, game=game, state=state, player=player)

    def log_size(self):
        n = C.c_uint64(0)
        check(self.L.ccsp_log_size(self.ctx, C.byref(n)), 'ccsp_log_size')
        return int(n.value)

    def log(self, first=0, n=None):
        n = self.log_size() - first if n is None else n
        state = np.zeros(n, dtype=STATE_DTYPE)
        meta = np.zeros(n, dtype=META_DTYPE)
        pi = np.zeros((n, NUM_ACTIONS), dtype=np.float64)
        if n:
            check(self.L.ccsp_read_log(self.ctx, first, n, state.ctypes.data, meta.ctypes.data, pi.ctypes.data), 'ccsp_read_log')
        return state, meta, pi

    def host_log_buffers(self, rows=None):
        """page-locked host arrays (state, meta, pi) of `rows` rows (default: the log's capacity) for log_into(): a device-to-host
        copy into pinned memory runs at PCIe speed, into fresh pageable NumPy memory at a fraction of it"""
        import torch
        rows = self.log_capacity if rows is None else int(rows)

        def pinned(dtype, shape):
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            raw = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=True)
            return raw.numpy()[:n].view(dtype).reshape(shape), raw
        st, k0 = pinned(STATE_DTYPE, (rows,))
        meta, k1 = pinned(META_DTYPE, (rows,))
        pi, k2 = pinned(np.float64, (rows, NUM_ACTIONS))
        return dict(state=st, meta=meta, pi=pi, rows=rows, keep=(k0, k1, k2))

    def log_into(self, bufs, first=0):
        """the log rows from `first` on, copied into the arrays of host_log_buffers(); -> (state, meta, pi) views of the rows read"""
        n = self.log_size() - first
        assert n <= bufs['rows']
        if n:
            check(self.L.ccsp_read_log(self.ctx, first, n, bufs['state'].ctypes.data, bufs['meta'].ctypes.data, bufs['pi'].ctypes.data),
                  'ccsp_read_log')
        return bufs['state'][:n], bufs['meta'][:n], bufs['pi'][:n]

    def log_clear(self, stream=None):
        """forget the rows read so far (stream-ordered): the log is empty again"""
        check(self.L.ccsp_log_clear(self.ctx, _stream_ptr(stream)), 'ccsp_log_clear')

    def log_device_ptrs(self):
        s, m, p = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.L.ccsp_log_device_ptrs(self.ctx, C.byref(s), C.byref(m), C.byref(p)), 'ccsp_log_device_ptrs')
        return s.value, m.value, p.value

    def results(self, first=0, n=None):
        n = self.max_games - first if n is None else n
        out = np.zeros(n, dtype=RESULT_DTYPE)
        if n:
            check(self.L.ccsp_read_results(self.ctx, first, n, out.ctypes.data), 'ccsp_read_results')
        return out

    def read_root(self, slot):
        k = C.c_int(0)
        N = np.zeros(126, dtype=np.uint32)
        W = np.zeros(126, dtype=np.float64)
        P = np.zeros(126, dtype=np.float64)
        mv = np.zeros(126, dtype=np.uint16)
        check(self.L.ccsp_read_root(self.ctx, int(slot), C.byref(k), N.ctypes.data, W.ctypes.data, P.ctypes.data,
                                    mv.ctypes.data), 'ccsp_read_root')
        k = k.value
        return dict(N=N[:k], W=W[:k], P=P[:k], mv=mv[:k])

    def tree_digest(self, slot):
        d, n, e = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self.L.ccsp_debug_tree_digest(self.ctx, int(slot), C.byref(d), C.byref(n), C.byref(e)), 'ccsp_debug_tree_digest')
        return d.value, n.value, e.value
