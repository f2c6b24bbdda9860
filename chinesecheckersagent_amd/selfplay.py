"""The drop-in boundary of the path: the reference's self-play API on top of the
GPU-resident engine.

    selfplay(model1, model2=None, randomised=False)                  selfplay.py:11-80
    generate_self_play(worker_id, model_path, num_self_play, ...)    train.py:27-67
    generate_self_play_in_parallel(model_path, num_self_play, num_workers, model2_path=None)    train.py:71-105, one rank per GPU
    selfplay_batch(...) / generate_train_data(...)                   many games in <= 4096 restarting slots (SelfPlayRun)

`model` is anything with the batched evaluator interface of model.ResidualCNN
(`evaluate_batch(x[G,7,7,7] f32 cuda) -> (p f64 [G,294], v f32 [G])`); the reference's duck-typed
`.predict` objects are accepted too (wrapped, batch 1 at a time -- slow, for plumbing tests only).
Per simulated step ALL games are advanced by one HIP select kernel, ONE batched forward of the net and
one HIP expand/backup kernel.  Randomness is the counter-based stream keyed by (seed, game id, ...)
(oracle/harness/spec.py), so a game's result does not depend on which batch or GPU played it.
"""
import numpy as np

from . import _lib
from .board import BoardView
from .config import MCTS_SIMULATIONS, NUM_ACTIONS
from .engine import SelfPlayEngine

_default_seed = [20261003]
_next_game = [0]


def set_seed(seed, first_game=0):
    """the reference reseeds from OS entropy per worker (train.py:38-39); here the stream is explicit"""
    _default_seed[0] = int(seed)
    _next_game[0] = int(first_game)


class _PredictAdapter(object):
    """wraps a reference-style model (predict(x[7,7,7]) -> (p, v), model.py:21-24) into the batched interface"""

    def __init__(self, model):
        self.model = model

    def evaluate_batch(self, x):
        import torch
        xs = x.detach().cpu().numpy().reshape(-1, 7, 7, 7)
        p = np.zeros((len(xs), NUM_ACTIONS))
        v = np.zeros(len(xs), dtype=np.float32)
        for i, xi in enumerate(xs):
            if not xi.any():
                continue                                # slot not searching (opening ply / finished): row is ignored
            pi, vi = self.model.predict(xi.astype(np.float64))
            p[i], v[i] = pi, vi
        return torch.from_numpy(p).to(x.device), torch.from_numpy(v).to(x.device)


def _batched(model):
    return model if hasattr(model, 'evaluate_batch') else _PredictAdapter(model)


def _on_device(model):
    """a batched evaluator that lives on the GPU (model.ResidualCNN), as opposed to a reference-style `.predict` object wrapped by
    _PredictAdapter, which evaluates on the HOST one position at a time: nothing to capture in a hipGraph, and -- not being known to be a
    pure function of the position -- nothing whose calls tree reuse may skip (selfplay.py:130-133 re-evaluates the subtree)"""
    m = _batched(model)
    return hasattr(m, 'model') and not isinstance(m, _PredictAdapter)


_malloc_tuned = [False]


def tune_host_allocator():
    """glibc serves every large array from a fresh mmap and gives it back on free: the harvest's copies, the float64 planes and the
    gathered pi rows of every conversion are then first-touch page faults -- measured 11 us per 2.7-KB row against 0.3 us once the heap
    keeps its pages (mmap threshold 1 GB, no trimming).  Process-level state: called by the rank entry points (bench.py, launch.init_rank; the rank
    processes of the N-GPU generator spend a host core per GPU on conversion otherwise) and by SelfPlayRun only under CCSP_MALLOPT=1; CCSP_NO_MALLOPT=1 leaves the process's allocator alone."""
    import os
    if _malloc_tuned[0] or os.environ.get('CCSP_NO_MALLOPT') == '1':
        return
    _malloc_tuned[0] = True
    try:
        import ctypes
        libc = ctypes.CDLL('libc.so.6')
        libc.mallopt(-3, 1 << 30)              # M_MMAP_THRESHOLD
        libc.mallopt(-1, (1 << 31) - 1)        # M_TRIM_THRESHOLD
        libc.mallopt(-2, 1 << 28)              # M_TOP_PAD
    except Exception:                          # not glibc: nothing to tune
        pass


def _strict():
    """CCSP_STRICT=1 (bench.py sets it): a performance path that cannot be taken is an error, not a silent fallback"""
    import os
    return os.environ.get('CCSP_STRICT') == '1'


def _warn(msg):
    import sys
    sys.stderr.write('chinesecheckersagent_amd: ' + msg + '\n')


def _check_hot_path(model1, model2, n_slots, free_running):
    """The delivered path is the hand-written one: a CUDA float32 ResidualCNN on the fused HIP kernel (backend 'hip'), and from 1024
    slots on the free-running kernels.  A model that would silently leave it -- backend='torch' / precision='fp64' (MIOpen modules), a
    wrapped model without the batched interface at a batch size where that matters, free_running=False forced at >= 1024 slots -- is
    reported once on stderr and is an ERROR under CCSP_STRICT=1 (bench.py sets it).  Reference-style `.predict`
    objects at small sizes (plumbing tests, selfplay.py:155-175) are what they are and pass."""
    problems = []
    for name, m in (('model1', model1), ('model2', model2)):
        if m is None:
            continue
        backend, device = getattr(m, 'backend', None), getattr(m, 'device', None)
        if backend is not None and getattr(device, 'type', None) == 'cuda' and backend != 'hip':
            problems.append("%s runs on backend=%r, precision=%r: the PyTorch modules, not the fused HIP kernel (ResidualCNN(precision='fp32', "
                            "backend='auto') is the delivered evaluator)" % (name, backend, getattr(m, 'precision', None)))
        if backend is None and n_slots >= 1024:
            problems.append('%s has no batched evaluator (a reference-style .predict object): %d slots would be evaluated one position at a '
                            'time on the host' % (name, n_slots))
    if free_running is False and n_slots >= 1024:
        problems.append('free_running=False with %d slots: the lock-step kernels, not the free-running path bench.py measures' % n_slots)
    if not problems:
        return
    msg = 'SelfPlayRun leaves the delivered hot path: ' + '; '.join(problems)
    if _strict():
        raise _lib.CcspError(msg + ' (CCSP_STRICT=1)')
    if msg not in _warned:
        _warned.add(msg)
        _warn(msg)


_warned = set()


class BatchSelfPlay(object):
    """n_slots concurrent games through the stepped path (external evaluator).

    Two ways of stepping them (same games, bit for bit):
      lock-step (default)   every slot plays the same ply's same simulation: per ply ply_begin -> net -> root_expand -> sims x
                            [net -> expand_backup_select] -> ply_end; the form the reference-made tree fixtures are replayed in
      free_running=True     ccsp_advance: every slot at its own simulation of its own ply -- [net -> advance] for ever; simulations that
                            end in a won leaf never wait for the net, and with `reuse` (default when there is ONE model) positions the
                            previous ply's tree already holds below the move that was played are expanded from that tree instead of
                            asking the net again (selfplay.py:130-133 discards it).  play_ply() is then sims + 1 such steps: a ply's
                            worth of evaluator launches, in which a slot plays 1.3-1.5 plies.  The delivered mode (SelfPlayRun)."""

    def __init__(self, model1, model2=None, n_slots=1, sims=MCTS_SIMULATIONS, seed=None, first_game=0, game_stride=1,
                 max_games=None, randomised=False, auto_restart=False, device=0, log_capacity=None, use_graph=True,
                 free_running=False, reuse=None, log_guard=False, stagger=False, stagger_span=None):
        import torch
        self.torch = torch
        self.m1 = _batched(model1)
        self.m2 = _batched(model2) if model2 is not None else None
        self.randomised = bool(randomised)
        self.eng = SelfPlayEngine(n_slots=n_slots, sims=sims, seed=_default_seed[0] if seed is None else seed,
                                  first_game=first_game, game_stride=game_stride, max_games=max_games,
                                  log_capacity=log_capacity, randomised=randomised, auto_restart=auto_restart, device=device)
        dev = torch.device('cuda', device)
        self.planes = torch.zeros((n_slots, 7, 7, 7), dtype=torch.float32, device=dev)
        self.n_slots, self.sims = n_slots, sims
        # one simulation step (select kernel -> net forward -> f64 softmax -> expand/backup kernel) is captured
        # once into a hipGraph and replayed `sims` times per ply: the step is launch-bound otherwise
        # (a reference-style .predict object evaluates on the HOST: nothing to capture)
        self.use_graph = bool(use_graph) and _on_device(self.m1) and (self.m2 is None or _on_device(self.m2))
        self._graph = None
        self._root_is_p2 = torch.zeros(n_slots, dtype=torch.bool, device=dev)
        self.free_running = bool(free_running)
        self.reuse = self.free_running and (self.m2 is None if reuse is None else bool(reuse))
        if self.reuse and self.m2 is not None:
            raise ValueError('tree reuse needs ONE model: the previous ply of a two-model game was searched with the other one (selfplay.py:30,59)')
        self.log_guard = bool(log_guard)
        self.stagger = bool(stagger)
        if self.stagger and stagger_span:                  # rounds over which the slots' FIRST games begin (default: one ply's worth); a span
            # of a whole game's worth of rounds puts a restarting run into its steady state -- games ending at an even rate -- from the
            # moment the last slot has started, instead of after several generations of games that began together
            self.eng.set_stagger_span(max(1, int(stagger_span) // (self.BOUNDARY_EVERY if n_slots >= 1024 else 1)))
        self.steps = 0                                     # free-running: [net -> advance] steps taken
        if self.free_running:
            if self.reuse:
                self.eng.enable_tree_reuse()
            self._model_sel = torch.zeros(n_slots, dtype=torch.uint8, device=dev) if self.m2 is not None else None
            # the hand-off of the free-running path (include/ccsp.h): request records + their move lists out, compact answers in;
            # _p0 / _v0: the first call's answer to no request
            self._req, self._moves, self._p0, self._v0 = self.eng.request_buffers()
            self._started = False
            self._round_no = 0
            self.net_events = None                         # bench.py: [(start, end)] HIP events around the evaluator launch of the plain (uncaptured) steps

    def _evaluate(self, root_is_p2):
        p, v = self.m1.evaluate_batch(self.planes)
        if self.m2 is not None:
            # selfplay.py:30,59: model1 moves for player one, model2 for player two; the whole search of a
            # ply uses the mover's model (make_move(root, model, ...), selfplay.py:36)
            p2, v2 = self.m2.evaluate_batch(self.planes)
            p = self.torch.where(root_is_p2[:, None], p2, p)
            v = self.torch.where(root_is_p2, v2, v)
        return p.contiguous(), v.contiguous()

    def _capture(self, root_is_p2, p, v):
        """several simulation steps -> one hipGraph (one hipGraphLaunch, and its host / front-end cost, per `unroll` steps).
        A failed capture is reported (and is an error under CCSP_STRICT=1); the ply then runs on plain launches."""
        torch, e = self.torch, self.eng
        selected = False
        keep = []
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                     # warm-up on a side stream (allocator)
                for _ in range(2):
                    self._evaluate(root_is_p2)
            torch.cuda.current_stream().wait_stream(s)
            self._unroll = max(k for k in (25, 20, 16, 10, 8, 5, 4, 2, 1) if self.sims % k == 0)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):                      # capture only: nothing executes here
                e.select(self.planes)
                selected = True
                for i in range(self._unroll):              # [evaluate -> expand/backup + the next selection] in one tree launch
                    gp, gv = self._evaluate(root_is_p2)
                    if i + 1 < self._unroll:
                        e.expand_backup_select(gp, gv, self.planes)
                    else:
                        e.expand_backup(gp, gv)
                        selected = False
                    keep.append((gp, gv))
            self._graph = g
            self._graph_out = keep                         # keep the captured outputs alive
        except Exception as ex:
            self.use_graph = False
            self._graph = None
            if selected:                                   # close the half-captured step on the host side
                e.expand_backup(p, v)                      # (device side: no pending leaf -> no-op)
            if _strict():
                raise
            _warn('hipGraph capture of the simulation steps failed (%r): this batch runs on plain launches' % (ex,))

    # ---- free-running stepping ---------------------------------------------------------------------------------------------
    FREE_UNROLL = 25         # [net -> advance | boundary] rounds per captured hipGraph
    BOUNDARY_EVERY = 6       # ccsp_boundary in every k-th round only (a slot between two searches then waits up to k - 1 rounds;
                             # measured at 4096 x 400 in steady state: k = 1 / 2 / 4 / 6 / 8 / 12 -> 17.1 / 17.4 / 18.24 / 18.32 / 18.26 / 18.15 M
                             # node-expansions/s: beside an evaluator launch the boundary kernel's few long waves cost the round 30-40 us)
    DEBUG = False            # diagnostic tallies of ccsp_advance / ccsp_boundary in counters 12-14 (tools/bench_free.py --debug)
    SIDE_STREAM = False      # ccsp_boundary on a stream of its own beside the next evaluator launch (measured: hipGraphs with forks
                             # stop overlapping the two half-batches' graphs; kept for experiments)

    def set_advance_limits(self, budget=-1, time_cap=-1, deadline=-1):
        """ccsp_set_advance_limits for this batch.  The limits travel BY VALUE in every ccsp_advance launch, so a captured hipGraph keeps
        the ones it was captured with: the graph is dropped here and captured again by the next play_steps (the same goes for a change of
        BOUNDARY_EVERY, whose cadence is laid down at capture time).  Results do not depend on the limits."""
        self.eng.set_advance_limits(budget, time_cap, deadline)
        if self.free_running and self._graph is not None:
            self.torch.cuda.synchronize(self.planes.device)
            self._graph = None
            self._graph_out = None

    def set_advance_limits(self, budget=-1, time_cap=-1, deadline=-1):
        """ccsp_set_advance_limits for this batch.  The limits travel BY VALUE in every ccsp_advance launch, so a captured hipGraph keeps
        the ones it was captured with: the graph is dropped here and captured again by the next play_steps (the same goes for a change of
        BOUNDARY_EVERY, whose cadence is laid down at capture time).  Results do not depend on the limits."""
        self.eng.set_advance_limits(budget, time_cap, deadline)
        if self.free_running and self._graph is not None:
            self.torch.cuda.synchronize(self.planes.device)
            self._graph = None
            self._graph_out = None

    def _answer(self, m):
        """(pk, v): model m's compact answers to the requests on the table"""
        if hasattr(m, 'evaluate_requests'):
            return m.evaluate_requests(self._req, self._moves)
        from .model import evaluate_requests_with
        return evaluate_requests_with(m.evaluate_batch, self._req, self._moves)

    def _evaluate_free(self):
        p, v = self._answer(self.m1)
        if self.m2 is not None:                            # the request names the model: player two's searches ask model2 (selfplay.py:30,36,59)
            p2, v2 = self._answer(self.m2)
            sel = self._model_sel.bool()
            p = self.torch.where(sel[:, None], p2, p)
            v = self.torch.where(sel, v2, v)
        return p.contiguous(), v.contiguous()

    def _round(self, p, v, capturing=False):
        """the tree's share of a step, after the evaluator launch that produced (p, v) has been issued on the current stream:
        ccsp_advance there (the slots in a search), ccsp_boundary on a side stream (the few slots between two searches: the end of a
        ply is long -- pi, sampling, rules, Dirichlet noise -- and runs beside the NEXT evaluator launch instead of in front of it)"""
        torch, e = self.torch, self.eng
        if not self.SIDE_STREAM:
            e.advance(p, v, self._req, self._moves, self._model_sel, reuse=self.reuse, log_guard=self.log_guard, debug=self.DEBUG)
            self._round_no += 1
            if self._round_no % (self.BOUNDARY_EVERY if self.n_slots >= 1024 else 1) == 0:      # (a small batch is latency-bound: no waiting there)
                e.boundary(p, v, self._req, self._moves, self._model_sel, reuse=self.reuse, log_guard=self.log_guard, stagger=self.stagger, debug=self.DEBUG)
            return
        cur = torch.cuda.current_stream()
        cur.wait_stream(self._side)                        # the previous round's boundary work: done before this round's advance
        e.advance(p, v, self._req, self._moves, self._model_sel, reuse=self.reuse, log_guard=self.log_guard, debug=self.DEBUG)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            e.boundary(p, v, self._req, self._moves, self._model_sel, reuse=self.reuse, log_guard=self.log_guard, stagger=self.stagger, debug=self.DEBUG,
                       overlapped=True)
        if not capturing:
            p.record_stream(self._side)
            v.record_stream(self._side)

    def _capture_free(self):
        torch = self.torch
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                     # warm-up on a side stream (allocator)
                for _ in range(2):
                    self._evaluate_free()
            torch.cuda.current_stream().wait_stream(s)
            self._unroll = self.FREE_UNROLL
            keep = []
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):                      # capture only: nothing executes here
                for _ in range(self._unroll):
                    gp, gv = self._evaluate_free()
                    self._round(gp, gv, capturing=True)
                    keep.append((gp, gv))
                torch.cuda.current_stream().wait_stream(self._side)      # the side stream joins before the capture ends
            self._graph, self._graph_out = g, keep
        except Exception as ex:
            self.use_graph = False
            self._graph = None
            if _strict():
                raise
            _warn('hipGraph capture of the free-running steps failed (%r): this batch runs on plain launches' % (ex,))

    def play_steps(self, n):
        """n steps of [one batched forward of the net -> ccsp_advance]"""
        assert self.free_running
        torch = self.torch
        if not self._started:
            self._side = torch.cuda.Stream(device=self.planes.device)
            self._round(self._p0, self._v0)                # nothing is pending yet: every slot starts its game
            self._started = True
        if self.use_graph and self._graph is None:
            self._capture_free()
        done = 0
        if self._graph is not None:
            while n - done >= self._unroll:
                self._graph.replay()
                done += self._unroll
        while done < n:
            if self.net_events is not None:                # the evaluator launch as it runs in the pipeline, timed on its own stream
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                p, v = self._evaluate_free()
                ev1.record()
                self._round(p, v)
                ev2 = torch.cuda.Event(enable_timing=True)
                ev2.record()
                self.net_events.append((ev0, ev1, ev2))       # net launch | the round's tree kernels
            else:
                p, v = self._evaluate_free()
                self._round(p, v)
            done += 1
        torch.cuda.current_stream().wait_stream(self._side)
        self.steps += n

    def play_ply(self):
        """one ply of every running slot: random opening move, or root expansion + sims x
        (select -> net -> expand/backup) + pi + move.  free_running: sims + 1 steps of [net -> advance] instead -- the same number of
        evaluator launches, in which every slot gets as far as it gets"""
        if self.free_running:
            return self.play_steps(self.sims + 1)
        e = self.eng
        e.ply_begin(self.planes)
        self._root_is_p2.copy_(self.planes[:, 0, 0, 6] == 1)
        root_is_p2 = self._root_is_p2
        p, v = self._evaluate(root_is_p2)
        e.root_expand(p, v)
        if self.use_graph and self._graph is None:
            self._capture(root_is_p2, p, v)
        if self._graph is not None:
            for _ in range(self.sims // self._unroll):
                self._graph.replay()
        else:
            e.select(self.planes)
            for i in range(self.sims):
                p, v = self._evaluate(root_is_p2)
                if i + 1 < self.sims:
                    e.expand_backup_select(p, v, self.planes)
                else:
                    e.expand_backup(p, v)
        e.ply_end()

    def run_to_completion(self, max_plies=2048):
        for i in range(max_plies):
            self.play_ply()
            if i % 8 == 7 and (self.eng.slots()['status'] == _lib.ST_RUNNING).sum() == 0:
                break
        return self.collect()

    HOST_RING = 4            # page-locked host copies of the log in rotation: a harvest's arrays stay valid for the next three

    def harvest(self):
        """the rows logged since the last harvest and the result table as it stands, as ONE part:
        [(state, meta, pi, results, first_game, game_stride)]; the device log is empty again afterwards.  The row arrays are
        views of page-locked host buffers used in rotation: valid until HOST_RING - 1 further harvests have been taken."""
        e = self.eng
        if not hasattr(self, '_ring'):
            self._ring, self._ring_at = [], 0
        k = self._ring_at % self.HOST_RING
        if k >= len(self._ring):
            self._ring.append(e.host_log_buffers())
        bufs = self._ring[k]
        self._ring_at += 1
        st, meta, pi = e.log_into(bufs)
        e.log_clear()
        return [(st, meta, pi, e.results(), e.first_game, e.game_stride)]

    def _check_log_complete(self):
        """a full sample log drops rows (the engine ends such a game with status ERROR, counts an error and leaves the slot
        out of play): games with missing plies never reach the training data"""
        c = self.eng.counters()
        if c['errors']:
            raise _lib.CcspError('%d engine errors (sample log full after %d of %d rows, or a game ended in ERROR): '
                                 'raise log_capacity, or harvest() more often' % (c['errors'], self.eng.log_size(), self.eng.log_capacity))

    def collect(self, allow_errors=False):
        """per finished game, in game-id order: (play_history, p1_reward) or (None, None) -- the
        return value of selfplay() (selfplay.py:45-47, 72-80).  A game that ended in ERROR raises, or with
        allow_errors=True is handed back as ('error', status) beside the whole games."""
        e = self.eng
        if not allow_errors:
            self._check_log_complete()
        st, meta, pi = e.log()
        return _games_from_rows(st, meta, pi, e.results(), e.first_game, e.game_stride, self.randomised, allow_errors)

    def collect_train_data(self, allow_errors=False):
        """utils.convert_to_train_data(self.collect()) as arrays, without building a Python object per position
        (utils.log_to_train_data): (board_x [N,7,7,7] f64, pi_y [N,294] f64, v_y [N] int64) of the games won so far"""
        from . import utils
        e = self.eng
        if not allow_errors:
            self._check_log_complete()
        st, meta, pi = e.log()
        return utils.log_to_train_data(st, meta, pi, e.results(), first_game=e.first_game, game_stride=e.game_stride,
                                       randomised=self.randomised)

    def close(self):
        self.eng.close()


def _games_from_rows(st, meta, pi, res, first_game, game_stride, randomised, allow_errors=False):
    """sample rows + result table -> [(play_history, p1_reward) | (None, None) | ('unfinished' | 'error', status)] by game index"""
    order = np.lexsort((meta['ply'], meta['game']))
    by_game = {}
    for r in order:
        by_game.setdefault(int(meta['game'][r]), []).append(r)
    out = []
    for k in range(len(res)):
        game = first_game + k * game_stride
        status = int(res['status'][k])
        if status in (_lib.ST_WON_P1, _lib.ST_WON_P2):
            rows = by_game.get(game, [])
            if randomised:
                rows = rows[3:]                         # selfplay.py:76-78
            hist = [(BoardView(st[r]), pi[r].copy()) for r in rows]
            out.append((hist, int(res['reward'][k])))
        elif status in (_lib.ST_DISCARD_REPETITION, _lib.ST_DISCARD_NO_PROGRESS):
            out.append((None, None))
        elif status == _lib.ST_ERROR:
            if not allow_errors:
                raise _lib.CcspError('game %d ended in ERROR status' % game)
            out.append(('error', status))
        else:
            out.append(('unfinished', status))
    return out


class PipelinedSelfPlay(object):
    """The same batch as `n_parts` BatchSelfPlay halves on their own HIP streams (game ids interleaved), so that the
    select / expand-backup kernels and launch gaps of one half run under the evaluator kernel of the other: the
    evaluator fills the GPU with one workgroup per CU at 2048 positions, the tree kernels need almost nothing.
    4096 games x 400 simulations with good_model.h5: 12.5 -> 13.4 M node-expansions/s with two parts (four are slower).
    Same interface as BatchSelfPlay (play_ply / run_to_completion / harvest / collect / close)."""

    def __init__(self, model1, model2=None, n_slots=2, n_parts=2, first_game=0, game_stride=1, max_games=None, log_capacity=None,
                 auto_restart=False, **kw):
        import torch
        max_games = n_slots if max_games is None else max_games
        # part i plays ids first_game + (i + k n_parts) game_stride, k < ceil((max_games - i) / n_parts): with restarts every part
        # draws from its own id budget
        assert n_slots % n_parts == 0 and (auto_restart or max_games == n_slots)
        self.torch = torch
        self.n_parts, self.n_slots = n_parts, n_slots
        per = n_slots // n_parts
        self.parts = [BatchSelfPlay(model1, model2, n_slots=per, first_game=first_game + i * game_stride,
                                    game_stride=game_stride * n_parts, max_games=max(1, (max_games - i + n_parts - 1) // n_parts),
                                    auto_restart=auto_restart,
                                    log_capacity=None if log_capacity is None else (log_capacity + n_parts - 1) // n_parts, **kw)
                      for i in range(n_parts)]
        self.streams = [torch.cuda.Stream() for _ in range(n_parts)]
        self.max_games = max_games
        self._forked = False

    def play_ply(self):
        cur = self.torch.cuda.current_stream()
        if all(b.free_running for b in self.parts):
            # Free-running parts: the rounds of a ply's worth are handed to the parts' streams ALTERNATELY, one captured graph at a time
            # (a whole ply of part 0 first would leave part 1's stream empty for as long as the host takes to enqueue it: the halves
            # would overlap for part of every ply only), and the streams are not joined ply by ply -- every read-back of the engines
            # (harvest, counters, collect) synchronises the device itself.
            if not self._forked:
                for st in self.streams:
                    st.wait_stream(cur)
                self._forked = True
            n, unroll = self.parts[0].sims + 1, BatchSelfPlay.FREE_UNROLL
            done = 0
            while done < n:
                k = min(unroll, n - done)
                for b, st in zip(self.parts, self.streams):
                    with self.torch.cuda.stream(st):
                        b.play_steps(k)
                done += k
            return
        for b, st in zip(self.parts, self.streams):
            st.wait_stream(cur)
            with self.torch.cuda.stream(st):
                b.play_ply()
        for st in self.streams:
            cur.wait_stream(st)

    def join(self):
        """the parts' streams joined into the current one (free-running parts are not joined ply by ply)"""
        cur = self.torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)

    def counters(self):
        tot = {}
        for b in self.parts:
            for k, v in b.eng.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def running(self):
        return any((b.eng.slots()['status'] == _lib.ST_RUNNING).any() for b in self.parts)

    def run_to_completion(self, max_plies=2048):
        for i in range(max_plies):
            self.play_ply()
            if i % 8 == 7 and not self.running():
                break
        return self.collect()

    def harvest(self):
        self.join()
        return [h for b in self.parts for h in b.harvest()]

    def collect(self, allow_errors=False):
        outs = [b.collect(allow_errors) for b in self.parts]
        n = min(self.max_games, sum(len(o) for o in outs))
        return [outs[j % self.n_parts][j // self.n_parts] for j in range(n)]

    def collect_train_data(self, allow_errors=False):
        """the parts' samples, concatenated (training order is shuffled anyway)"""
        outs = [b.collect_train_data(allow_errors) for b in self.parts]
        return tuple(np.concatenate([o[i] for o in outs]) for i in range(3))

    def close(self):
        for b in self.parts:
            b.close()


class GameStore(object):
    """Host side of a harvested run: the sample rows of the games still being played, the result of every game that
    has ended (by GLOBAL game index j: id = first_game + j * game_stride) and what became of the finished games' rows --
    kept as records (the object path: selfplay()'s play_history) and / or handed on as (board_x, pi_y, v_y) chunks.
    A harvest's rows stay together as one batch with an index sorted by game, so that taking the rows of the games that
    have just ended costs a binary search per batch -- not a pass over every row still waiting."""

    def __init__(self, n_games, first_game, game_stride, randomised, keep_records=True):
        self.n_games, self.first_game, self.game_stride = int(n_games), int(first_game), int(game_stride)
        self.randomised = bool(randomised)
        self.results = np.zeros(self.n_games, dtype=_lib.RESULT_DTYPE)
        self.results['status'] = 0xFF                                    # not finished yet (ccsp_reset's fill)
        self.keep_records = keep_records
        self._batches = []                                               # harvests with rows of games that have not ended yet
        self._new = []                                                   # indices of the games that ended since the last take
        self._records = []                                               # rows of finished (won) games (object path)
        self.rows_seen = 0

    def add(self, parts):
        """parts = the harvest() of a batch: the rows become a batch of their own (copied: harvest() hands out views of host
        buffers in rotation), the result tables are merged by global index"""
        for st, meta, pi, res, first, stride in parts:
            j0 = (first - self.first_game) // self.game_stride              # part-local index k <-> global index j0 + k * step
            step = stride // self.game_stride
            j = j0 + np.arange(len(res)) * step
            ended = (res['status'] != 0xFF) & (j < self.n_games)
            j = j[ended]
            fresh = self.results['status'][j] == 0xFF
            self.results[j[fresh]] = res[ended][fresh]
            if fresh.any():
                self._new.append(j[fresh])
            if len(meta):
                g = (np.asarray(meta['game'], dtype=np.int64) - self.first_game) // self.game_stride
                order = np.argsort(g, kind='stable')                        # a game's rows of this harvest stay in log (= ply) order
                self._batches.append(dict(st=np.array(st), meta=np.array(meta), pi=np.array(pi), order=order, g=g[order], live=len(meta)))
                self.rows_seen += len(meta)

    def finished(self):
        return bool((self.results['status'] != 0xFF).all())

    def n_finished(self):
        return int((self.results['status'] != 0xFF).sum())

    def _take_rows(self, games_all, games_won):
        """rows of the games `games_won` out of every batch (both sorted index arrays); the rows of `games_all` (won and
        discarded) leave the batches' accounts"""
        out = []
        keep = []
        for b in self._batches:
            lo, hi = np.searchsorted(b['g'], games_all, 'left'), np.searchsorted(b['g'], games_all, 'right')
            b['live'] -= int((hi - lo).sum())
            lo, hi = np.searchsorted(b['g'], games_won, 'left'), np.searchsorted(b['g'], games_won, 'right')
            cnt = hi - lo
            tot = int(cnt.sum())
            if tot:
                idx = np.repeat(lo - (np.cumsum(cnt) - cnt), cnt) + np.arange(tot)
                rows = np.sort(b['order'][idx])                              # log order within the batch
                out.append((b['st'][rows], b['meta'][rows], b['pi'][rows]))
            if b['live'] > 0:
                if b['live'] * 4 < len(b['g']) and len(b['g']) > 1024:       # mostly consumed: keep the live rows only
                    alive = self.results['status'][b['g']] == 0xFF
                    rows = np.sort(b['order'][alive])
                    g = b['g'][alive]
                    st, meta, pi = b['st'][rows], b['meta'][rows], b['pi'][rows]
                    g2 = (np.asarray(meta['game'], dtype=np.int64) - self.first_game) // self.game_stride
                    order = np.argsort(g2, kind='stable')
                    b = dict(st=st, meta=meta, pi=pi, order=order, g=g2[order], live=len(meta))
                keep.append(b)
        self._batches = keep
        return out

    def take_finished(self):
        """(board_x, pi_y, v_y, game id per row) of the games that have ended since the last call (won games only,
        utils.convert_to_train_data's rows and labels); their rows leave the store.  None if nothing ended."""
        from . import utils
        if not self._new:
            return None
        ended = np.sort(np.concatenate(self._new))
        self._new = []
        st = self.results['status'][ended]
        won = ended[(st == _lib.ST_WON_P1) | (st == _lib.ST_WON_P2)]
        rows = self._take_rows(ended, won)
        if not rows:
            return None
        d = tuple(np.concatenate([x[i] for x in rows]) for i in range(3)) if len(rows) > 1 else rows[0]
        if self.keep_records:
            self._records.append(d)
        return utils.log_to_train_data(d[0], d[1], d[2], self.results, first_game=self.first_game, game_stride=self.game_stride,
                                       randomised=self.randomised, return_games=True)

    def games(self, allow_errors=False):
        """[(play_history, p1_reward) | (None, None)] in game-id order (the object path)"""
        assert self.keep_records
        self.take_finished()
        rows = list(self._records)
        if rows:
            st, meta, pi = (np.concatenate([x[i] for x in rows]) for i in range(3))
        else:
            st, meta, pi = (np.zeros(0, dtype=_lib.STATE_DTYPE), np.zeros(0, dtype=_lib.META_DTYPE), np.zeros((0, NUM_ACTIONS)))
        return _games_from_rows(st, meta, pi, self.results, self.first_game, self.game_stride, self.randomised, allow_errors)


MAX_SLOTS = 4096          # concurrent games per GPU (BASELINE.json); more games than this restart in the slots that come free
MIN_GAMES_PER_RANK = 256  # a cohort smaller than this per rank is LATENCY-bound (a round's length does not depend on how many slots it
                          # carries: one slot, 23 or 180 take the same time per step), so spreading it over more GPUs buys nothing:
                          # config 5's real cohort (config.py:57-58: 180 games per iteration) at 800 simulations takes 6.52 s in 180 slots
                          # of ONE GPU and 5.91 s as eight ranks' shares of 23 -- 1.10 x for 8 x the GPUs (profiles/r6_small_cohort.txt).


def selfplay_ranks(n_games, world, min_games_per_rank=None):
    """ranks that PLAY a cohort of n_games when `world` are there: min(world, ceil(n_games / MIN_GAMES_PER_RANK)) -- rank r below that
    number plays ids r, r + ranks, ...; the others sit the self-play out (and join whatever comes after it: the DDP fit, the arena).
    A game's record depends on its id alone, so the rows are the same whatever the number."""
    m = MIN_GAMES_PER_RANK if min_games_per_rank is None else max(1, int(min_games_per_rank))
    return max(1, min(int(world), (int(n_games) + m - 1) // m))

HARVEST_EVERY = 2         # steps between two harvests of the sample log (a harvest costs the run 0.3 % at 8 and no more at 2; the
                          # shorter the interval, the less is left to convert when a run ends: at the end of a 20-step region 22 ms
                          # against 37-51 ms with 4 -- +0.5-1 % of node-expansions/s over such a region, measured in round 5)


class SelfPlayRun(object):
    """`n_games` self-play games on one GPU the way bench.py measures them: min(n_games, max_slots) game slots, a slot that
    finishes its game starts the slot's next one by itself (ids first_game + j * game_stride, j < n_games; a game's record is
    a function of its id alone), two half-batches on their own streams when the batch is large, the sample log harvested
    every `harvest_every` plies -- so device memory is n_slots tree pools + n_slots x (harvest_every + 1) log rows whatever
    n_games is, and the log cannot overflow.
    `sink(board_x, pi_y, v_y, game_ids)` (optional; e.g. a TrainDataSink) receives the training rows of the games that ended,
    harvest by harvest, from a worker thread that converts while the GPU plays on."""

    def __init__(self, model1, model2=None, n_games=1, sims=MCTS_SIMULATIONS, seed=None, randomised=False, first_game=0,
                 game_stride=1, device=0, max_slots=MAX_SLOTS, harvest_every=HARVEST_EVERY, use_graph=True, keep_records=True,
                 sink=None, n_parts=None, free_running=None, reuse=None, stagger_span=None, off_path_ok=False):
        # (host allocator tuning is the PROCESS's business: the rank entry points -- worker.py, bench.py -- call tune_host_allocator();
        # an embedding process opts in with CCSP_MALLOPT=1)
        import os
        if os.environ.get('CCSP_MALLOPT') == '1':
            tune_host_allocator()
        _lib.prefer_blocking_sync(device)                  # (this run's device; a no-op once the process has initialised the GPU)
        n_games = int(n_games)
        if not off_path_ok:                                # (off_path_ok: a deliberate comparison, e.g. bench.py's lock-step variant)
            _check_hot_path(model1, model2, min(n_games, int(max_slots)), free_running)
        n_slots = max(1, min(n_games, int(max_slots)))
        if n_parts is None:
            # two half-batches from 1024 slots on (measured in steady state at 400 simulations, M node-expansions/s, one part lock-step /
            # one part free-running / two parts lock-step / two parts free-running: 512 slots 8.1 / 7.1 / 5.7 / 6.4; 1024 slots
            # 10.5 / 10.4 / 10.9 / 11.6; 2048 slots - / - / 12.7 / 15.8: an evaluator launch of up to 512 positions takes 46 us, one of
            # up to 1024 76 us -- two halves of 512 overlap their tree work with each other's launch AND get the shorter launch)
            n_parts = 2 if (n_slots >= 1024 and _on_device(model1)) else 1
        n_slots -= n_slots % n_parts
        self.n_games, self.n_slots, self.harvest_every = n_games, n_slots, int(harvest_every)
        kw = dict(sims=sims, seed=seed, first_game=first_game, game_stride=game_stride, max_games=n_games, randomised=randomised,
                  auto_restart=True, device=device, use_graph=use_graph)
        cap = n_slots * (self.harvest_every + 1)
        if free_running is None:
            # a batch that does not fill the GPU is latency-bound: a lock-step round (net -> one tree kernel) is shorter than a
            # free-running one (net -> advance -> boundary) -- measured in steady state at 800 simulations with the <1,8> evaluator shape,
            # k node-expansions/s lock-step / free-running: 1 slot 22.7 / 24.9, 8 slots 166 / 173, 24 slots 488 / 489, 64 slots 1283 /
            # 1194, 256 slots 4954 / 4219 (a sixth to a fifth of the free-running expansions come from the previous tree; its round is
            # 48 us against 43 for one slot, 72 against 52 for 256) -- so: free-running for a handful of slots (one game of selfplay():
            # the reused positions are worth more than the longer round) and from about a thousand on (4096: +30 %), lock-step between
            # -- for ON-DEVICE evaluators only: a host-side `.predict` object stays on lock-step (called exactly as often as the reference
            # calls it, no encode / gather round trip per round) unless free_running / reuse is asked for explicitly
            free_running = (n_slots >= 1024 or (n_slots <= 8 and model2 is None)) and _on_device(model1) and (model2 is None or _on_device(model2))
        if free_running and hasattr(_batched(model1), 'model'):
            # slots run at their own pace (BatchSelfPlay): between two harvests a slot plays up to ~1.5 plies per `play_ply`; a slot that
            # could find the log full waits for the harvest (log_guard) instead of losing a row
            kw.update(free_running=True, reuse=reuse, log_guard=True, stagger=n_games > n_slots, stagger_span=stagger_span)      # (restarting slots: a long run)
            cap = n_slots * (2 * self.harvest_every + 2)
        self.free_running = bool(kw.get('free_running'))
        if n_parts > 1:
            self.b = PipelinedSelfPlay(model1, model2, n_slots=n_slots, n_parts=n_parts, log_capacity=cap, **kw)
            self.counters = self.b.counters
        else:
            self.b = BatchSelfPlay(model1, model2, n_slots=n_slots, log_capacity=cap, **kw)
            self.counters = self.b.eng.counters
        self.store = GameStore(n_games, first_game, game_stride, randomised, keep_records=keep_records)
        self.sink = sink
        self.plies = 0
        self._since = 0
        self._worker = self._queue = None
        self._worker_error = []
        if sink is not None:
            import queue
            import threading
            self._queue = queue.Queue(maxsize=2)      # (+ the one in work: fewer than BatchSelfPlay.HOST_RING harvests alive)
            self._worker = threading.Thread(target=self._drain, daemon=True)
            self._worker.start()

    # the worker thread: merge a harvest into the store, convert the finished games' rows, hand them to the sink
    def _drain(self):
        while True:
            parts = self._queue.get()
            try:
                if parts is None:
                    return
                self._absorb(parts)
            except Exception as ex:                      # reported by the main thread at the next harvest / at flush()
                self._worker_error.append(ex)
            finally:
                self._queue.task_done()

    def _absorb(self, parts):
        self.store.add(parts)
        if self.sink is not None:
            chunk = self.store.take_finished()
            if chunk is not None and len(chunk[2]):
                self.sink(*chunk)

    def play_ply(self):
        self.b.play_ply()
        self.plies += 1
        self._since += 1
        if self._since >= self.harvest_every:
            self.harvest()

    def harvest(self):
        parts = self.b.harvest()                          # synchronises, copies the rows out, empties the device log
        self._since = 0
        if self._worker_error:
            raise self._worker_error[0]
        if self._queue is not None:
            self._queue.put(parts)
        else:
            self._absorb(parts)

    def drain(self):
        """harvest what is left and wait until the worker has converted it: the store (and the sink) then hold everything
        played so far; the run can go on afterwards"""
        self.harvest()
        if self._queue is not None:
            self._queue.join()
            if self._worker_error:
                raise self._worker_error[0]

    def flush(self):
        """drain() and retire the worker thread (end of the run)"""
        self.harvest()
        if self._queue is not None:
            self._queue.put(None)
            self._worker.join()
            self._worker = self._queue = None
            if self._worker_error:
                raise self._worker_error[0]

    def run(self, max_plies=None):
        """play until every game has a result"""
        limit = max_plies if max_plies is not None else 1100 * ((self.n_games + self.n_slots - 1) // self.n_slots) + 64
        while self.plies < limit:
            for _ in range(self.harvest_every):
                self.b.play_ply()
            self.plies += self.harvest_every
            self.harvest()
            if self._queue is None and self.store.finished():
                break
            if self._queue is not None and not self.b_running():
                break
        self.flush()
        return self

    def b_running(self):
        if hasattr(self.b, 'running'):
            return self.b.running()
        return bool((self.b.eng.slots()['status'] == _lib.ST_RUNNING).any())

    def games(self, allow_errors=False):
        return self.store.games(allow_errors)

    def errors(self):
        return self.counters()['errors']

    def close(self):
        if self._queue is not None:
            self._queue.put(None)
            self._worker.join()
            self._worker = self._queue = None
        self.b.close()


class TrainDataSink(object):
    """where a SelfPlayRun's finished games go, as utils.convert_to_train_data's rows (board_x, pi_y, v_y; + the game id of every
    row).  Without a path the chunks are kept and arrays() / save() hand them over; with a path they are STREAMED into the training
    file (utils.save_train_data's datasets board_x / pi_y / v_y, chunked along the first axis: h5lite.StreamWriter) while the run
    goes on -- host memory stays bounded, the file is whole when close() returns."""

    def __init__(self, path=None, chunk_rows=4096, with_games=False):
        self.chunks = []
        self.rows = 0
        self.discard = False              # rows arriving are dropped (bench.py: games that ended before the timed region)
        self.writer = None
        self.with_games = bool(with_games)       # a fourth dataset `game`: the id of every row's game (the N-GPU generator merges by it)
        if path is not None:
            self.open(path, chunk_rows)

    def open(self, path, chunk_rows=4096):
        from .h5lite import StreamWriter
        specs = [('board_x', (7, 7, 7), '<f8'), ('pi_y', (NUM_ACTIONS,), '<f8'), ('v_y', (), '<i8')]
        if self.with_games:
            specs.append(('game', (), '<i8'))
        self.writer = StreamWriter(path, specs, chunk_rows=chunk_rows)

    def __call__(self, board_x, pi_y, v_y, games):
        if self.discard:
            return
        if self.writer is not None:
            self.writer.append([board_x, pi_y, v_y] + ([np.asarray(games, dtype=np.int64)] if self.with_games else []))
        else:
            self.chunks.append((board_x, pi_y, v_y, games))
        self.rows += len(v_y)

    def close(self):
        """finish the streamed file; -> its path (None without one)"""
        if self.writer is None:
            return None
        path = self.writer.close()
        self.writer = None
        return path

    def arrays(self, canonical=True, with_games=False):
        """canonical: games in id order (what convert_to_train_data(selfplay_batch(...)) gives) instead of the order they ended in"""
        if not self.chunks:
            out = (np.zeros((0, 7, 7, 7)), np.zeros((0, NUM_ACTIONS)), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
        else:
            out = tuple(np.concatenate([c[i] for c in self.chunks]) for i in range(4))
            if canonical:
                o = np.argsort(out[3], kind='stable')
                out = tuple(x[o] for x in out)
        return out if with_games else out[:3]

    def save(self, version, directory=None):
        """utils.save_train_data (utils.py:48-56) of everything collected"""
        from . import utils
        from .config import SAVE_TRAIN_DATA_DIR
        bx, py, vy = self.arrays()
        return utils.save_train_data(bx, py, vy, version, SAVE_TRAIN_DATA_DIR if directory is None else directory)


def selfplay_batch(model1, model2=None, n_games=1, sims=MCTS_SIMULATIONS, seed=None, randomised=False,
                   first_game=0, game_stride=1, device=0, max_slots=MAX_SLOTS, harvest_every=HARVEST_EVERY):
    """n_games games; returns [(play_history, p1_reward) | (None, None)] in game-id order.  The games run in
    min(n_games, max_slots) slots with restarts (SelfPlayRun): the steady-state mode bench.py measures."""
    run = SelfPlayRun(model1, model2, n_games=n_games, sims=sims, seed=seed, randomised=randomised, first_game=first_game,
                      game_stride=game_stride, device=device, max_slots=max_slots, harvest_every=harvest_every)
    try:
        return run.run().games()
    finally:
        run.close()


def generate_train_data(model1, model2=None, n_games=1, sims=MCTS_SIMULATIONS, seed=None, randomised=False, first_game=0,
                        game_stride=1, device=0, max_slots=MAX_SLOTS, harvest_every=HARVEST_EVERY, out_path=None):
    """utils.convert_to_train_data(selfplay_batch(...)) without a Python object per position: the finished games' rows are
    converted harvest by harvest while the GPU plays on.  -> (board_x [N,7,7,7] f64, pi_y [N,294] f64, v_y [N] int64,
    summary dict); the same rows in the same order as utils.convert_to_train_data(generate_self_play's list).
    With out_path the rows are streamed into that training file instead (datasets board_x / pi_y / v_y, games in the order they
    ended) and (out_path, summary) is returned: host memory stays bounded whatever n_games is."""
    sink = TrainDataSink(path=out_path)
    run = SelfPlayRun(model1, model2, n_games=n_games, sims=sims, seed=seed, randomised=randomised, first_game=first_game,
                      game_stride=game_stride, device=device, max_slots=max_slots, harvest_every=harvest_every,
                      keep_records=False, sink=sink)
    try:
        run.run()
        st = run.store.results['status']
        summary = dict(games=n_games, won=int(((st == _lib.ST_WON_P1) | (st == _lib.ST_WON_P2)).sum()),
                       discarded=int(((st == _lib.ST_DISCARD_REPETITION) | (st == _lib.ST_DISCARD_NO_PROGRESS)).sum()),
                       errors=int((st == _lib.ST_ERROR).sum()), plies=run.plies, rows=sink.rows, counters=run.counters())
        if out_path is not None:
            return sink.close(), summary
        return sink.arrays() + (summary,)
    finally:
        if sink.writer is not None:
            sink.writer.abort()
        run.close()


def selfplay(model1, model2=None, randomised=False, sims=MCTS_SIMULATIONS, seed=None, game_id=None):
    """selfplay.py:11-80: one game -> (play_history, p1_reward), or (None, None) if the game was discarded"""
    if game_id is None:
        game_id = _next_game[0]
        _next_game[0] += 1
    return selfplay_batch(model1, model2, n_games=1, sims=sims, seed=seed, randomised=randomised, first_game=game_id)[0]


def _load_models(model_path, model2_path=None, device=None):
    from .model import ResidualCNN
    model = ResidualCNN(device=device)
    model2 = None
    if model_path is not None:
        model.load_weights(model_path)
        if model2_path is not None:
            model2 = ResidualCNN(device=device)
            model2.load_weights(model2_path)
    return model, model2


def generate_self_play(worker_id, model_path, num_self_play, model2_path=None, sims=MCTS_SIMULATIONS, seed=None,
                       first_game=None, game_stride=1, device=0):
    """train.py:27-67: load the model(s) and return [(play_history, p1_reward)] of the games that were
    not discarded.  The games are played on this worker's GPU in up to MAX_SLOTS concurrent slots; worker w of W
    plays game ids first_game + k * game_stride (generate_self_play_in_parallel passes w and W)."""
    model, model2 = _load_models(model_path, model2_path, device='cuda:%d' % device)
    if first_game is None:
        first_game = _next_game[0]
        _next_game[0] += num_self_play * game_stride
    games = selfplay_batch(model, model2, n_games=num_self_play, sims=sims, seed=seed, first_game=first_game,
                           game_stride=game_stride, device=device)
    return [(h, r) for h, r in games if h is not None and r is not None and not isinstance(h, str)]


def generate_self_play_in_parallel(model_path, num_self_play, num_workers, model2_path=None, sims=MCTS_SIMULATIONS, seed=None,
                                   first_game=None, randomised=False, devices=None, as_arrays=False, out_dir=None, max_slots=MAX_SLOTS,
                                   return_summary=False, timeout=None, max_steps=None, with_games=False, min_games_per_rank=None):
    """train.generate_self_play_in_parallel (train.py:71-105) with GPUs for workers: `num_workers` rank processes, one per
    MI355X (devices[r], default r), are started from THIS process -- which never touches the GPU -- and play the ids
    first_game + j, j < num_self_play, sharded j mod num_workers; their counters and visit histograms meet in one RCCL
    all-reduce.  Returns the reference's list [(play_history, p1_reward)] of the games that were not discarded, in game-id
    order (as_arrays=True: utils.convert_to_train_data of that list as (board_x, pi_y, v_y) arrays, no object per position);
    with return_summary=True also the all-reduced summary {'counters': ..., 'visit_histogram': ...}.
    timeout (seconds): ranks still running after it are terminated (then killed) and the call raises, instead of waiting for ever on
    a rank that stalls in a collective or a wedged kernel.
    min_games_per_rank (default MIN_GAMES_PER_RANK = 256): fewer rank processes are started for a small cohort -- selfplay_ranks(): a batch
    that does not fill a GPU is latency-bound, more GPUs do not shorten it; the summary's `world` says how many played.
    max_steps: every rank stops after that many steps (of sims + 1 rounds of all its slots) and hands back the games that have ENDED by
    then -- a bounded rehearsal of a shape too long to play out (BASELINE config 4 on one device); with_games (as_arrays): the game id of
    every row as a fourth array."""
    import json
    import os
    import sys
    import tempfile
    from . import launch, utils
    if first_game is None:
        first_game = _next_game[0]
        _next_game[0] += num_self_play
    seed = _default_seed[0] if seed is None else seed
    num_workers = selfplay_ranks(num_self_play, num_workers, min_games_per_rank)          # (small cohorts: fewer ranks, the same games)
    if devices is not None:
        devices = list(devices)[:num_workers]
    tmp = None
    if out_dir is None:
        tmp = tempfile.TemporaryDirectory(prefix='ccsp-selfplay-')
        out_dir = tmp.name
    try:
        argv = [sys.executable, '-m', 'chinesecheckersagent_amd.worker', 'selfplay', '--games', str(num_self_play), '--sims', str(sims),
                '--seed', str(seed), '--first-game', str(first_game), '--max-slots', str(max_slots), '--out', out_dir]
        if model_path is not None:
            argv += ['--model', model_path]
        if model2_path is not None:
            argv += ['--model2', model2_path]
        if randomised:
            argv += ['--randomised']
        if max_steps:
            argv += ['--max-steps', str(int(max_steps))]
        if as_arrays:
            argv += ['--arrays']                          # the ranks stream (board_x, pi_y, v_y, game) into files: their memory stays bounded
        extra = {'PYTHONPATH': os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                               ([os.environ['PYTHONPATH']] if os.environ.get('PYTHONPATH') else []))}
        if devices is not None and len(set(devices)) < len(devices):
            extra['CCSP_ONE_DEVICE'] = '1'                 # several ranks on one device: gloo carries the summary
        rc = launch.run_ranks(argv, num_workers, devices=devices, extra_env=extra, timeout=timeout)
        if rc:
            raise _lib.CcspError('generate_self_play_in_parallel: %s' % ('timed out after %s s' % timeout if rc == 124 else 'a rank process failed (exit code %d)' % rc))
        if as_arrays:
            from .h5lite import H5File
            parts = []
            for r in range(num_workers):
                path = os.path.join(out_dir, 'selfplay-rank%d.h5' % r)
                if os.path.exists(path):
                    f = H5File(path)
                    parts.append(tuple(f.get(k) for k in ('board_x', 'pi_y', 'v_y', 'game')))
            if parts:
                bx, py, vy, gid = (np.concatenate([x[i] for x in parts]) for i in range(4))
                o = np.argsort(gid, kind='stable')          # games by id; a game's rows stay in ply order
                out = (bx[o], py[o], vy[o]) + ((gid[o],) if with_games else ())
            else:
                out = (np.zeros((0, 7, 7, 7)), np.zeros((0, NUM_ACTIONS)), np.zeros(0, dtype=np.int64)) + ((np.zeros(0, dtype=np.int64),) if with_games else ())
            if return_summary:
                with open(os.path.join(out_dir, 'summary.json')) as f:
                    return out, json.load(f)
            return out
        results = np.zeros(num_self_play, dtype=_lib.RESULT_DTYPE)
        results['status'] = 0xFF
        rows = []
        for r in range(num_workers):
            path = os.path.join(out_dir, 'selfplay-rank%d.npz' % r)
            if not os.path.exists(path):                   # a rank with no game to play
                continue
            z = np.load(path)
            res = z['results']
            results[r + np.arange(len(res)) * num_workers] = res
            rows.append((z['state'], z['meta'], z['pi']))
        st, meta, pi = (np.concatenate([x[i] for x in rows]) for i in range(3)) if rows else \
            (np.zeros(0, dtype=_lib.STATE_DTYPE), np.zeros(0, dtype=_lib.META_DTYPE), np.zeros((0, NUM_ACTIONS)))
        games = _games_from_rows(st, meta, pi, results, first_game, 1, randomised)       # the object path: one Python object per position
        out = [(h, r) for h, r in games if h is not None and r is not None and not isinstance(h, str)]
        if return_summary:
            with open(os.path.join(out_dir, 'summary.json')) as f:
                return out, json.load(f)
        return out
    finally:
        if tmp is not None:
            tmp.cleanup()


def bench_net_plies(n_slots, sims, plies=2, weights=None, precision='fp32'):
    """config 3 timing helper for bench.py: node-expansions/s of the stepped path
    with the policy/value net (good_model.h5 when present, else random-initialised weights)."""
    import os
    import time
    import torch
    from .model import ResidualCNN
    if weights is None:
        cand = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'good_model.h5')
        weights = cand if os.path.exists(cand) else None
    model = ResidualCNN(precision=precision)
    if weights:
        model.load_weights(weights)
    parts = 2 if (n_slots >= 2048 and n_slots % 2 == 0) else 1
    if parts > 1:
        b = PipelinedSelfPlay(model, n_slots=n_slots, n_parts=parts, sims=sims, log_capacity=n_slots * (plies + 4))
        counters = b.counters
    else:
        b = BatchSelfPlay(model, n_slots=n_slots, sims=sims, max_games=n_slots, log_capacity=n_slots * (plies + 4))
        counters = b.eng.counters
    for _ in range(6):
        b.play_ply()                                   # opening plies: no search
    b.play_ply()                                       # one searched ply as warm-up
    torch.cuda.synchronize()
    c0 = counters()
    t0 = time.time()
    for _ in range(plies):
        b.play_ply()
    torch.cuda.synchronize()
    dt = time.time() - t0
    c1 = counters()
    b.close()
    ex = c1['expansions'] - c0['expansions']
    return {'node_expansions_per_s': ex / dt, 'ms_per_ply': dt / plies * 1e3, 'ms_per_sim_step': dt / plies / (sims + 1) * 1e3,
            'net_tflops': ex * 6483264 / dt / 1e12, 'precision': precision,
            'weights': os.path.basename(weights) if weights else 'random-init',
            'backend': model.backend,
            'streams': parts,
            'workload': '%d games x %d sims, policy/value net (%s), stepped path: select kernel -> net -> expand/backup kernel per simulation, 25 steps per hipGraph, %d half-batches on their own streams' % (n_slots, sims, 'fused fp32-MFMA HIP kernel' if model.backend == 'hip' else 'PyTorch-ROCm modules', parts)}


if __name__ == '__main__':
    # selfplay.py:155-175: one game with the given weights, the first positions printed
    import argparse
    ap = argparse.ArgumentParser(description='one self-play game on the GPU (python -m chinesecheckersagent_amd.selfplay <weights.h5>)')
    ap.add_argument('model_path')
    ap.add_argument('--sims', type=int, default=MCTS_SIMULATIONS)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--show', type=int, default=8, help='positions to print')
    a = ap.parse_args()
    from .model import ResidualCNN
    model = ResidualCNN()
    model.load_weights(a.model_path)
    history, reward = selfplay(model, sims=a.sims, seed=a.seed)
    if history is None:
        print('the game was discarded (repetition or no progress)')
    else:
        print('%d searched plies, reward for player one: %d' % (len(history), reward))
        for i, (board, pi) in enumerate(history[:a.show]):
            board.visualise(cur_player=1 + i % 2)

