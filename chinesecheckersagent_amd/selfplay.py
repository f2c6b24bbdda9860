"""The drop-in boundary of the path: the reference's self-play API on top of the
GPU-resident engine.

    selfplay(model1, model2=None, randomised=False)                  selfplay.py:11-80
    generate_self_play(worker_id, model_path, num_self_play, ...)    train.py:27-67
    selfplay_batch(...)                                              many games as one batch (new)

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


class BatchSelfPlay(object):
    """n_slots concurrent games through the stepped path (external evaluator)."""

    def __init__(self, model1, model2=None, n_slots=1, sims=MCTS_SIMULATIONS, seed=None, first_game=0, game_stride=1,
                 max_games=None, randomised=False, auto_restart=False, device=0, log_capacity=None, use_graph=True):
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
        self.use_graph = bool(use_graph) and hasattr(self.m1, 'model') and (self.m2 is None or hasattr(self.m2, 'model'))
        self._graph = None
        self._root_is_p2 = torch.zeros(n_slots, dtype=torch.bool, device=dev)

    def _evaluate(self, root_is_p2):
        p, v = self.m1.evaluate_batch(self.planes)
        if self.m2 is not None:
            # selfplay.py:30,59: model1 moves for player one, model2 for player two; the whole search of a
            # ply uses the mover's model (make_move(root, model, ...), selfplay.py:36)
            p2, v2 = self.m2.evaluate_batch(self.planes)
            p = self.torch.where(root_is_p2[:, None], p2, p)
            v = self.torch.where(root_is_p2, v2, v)
        return p.contiguous(), v.contiguous()

    def play_ply(self):
        """one ply of every running slot: random opening move, or root expansion + sims x
        (select -> net -> expand/backup) + pi + move"""
        e = self.eng
        torch = self.torch
        e.ply_begin(self.planes)
        self._root_is_p2.copy_(self.planes[:, 0, 0, 6] == 1)
        root_is_p2 = self._root_is_p2
        p, v = self._evaluate(root_is_p2)
        e.root_expand(p, v)
        if self.use_graph and self._graph is None:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):                     # warm-up on a side stream (allocator, MIOpen find)
                    for _ in range(2):
                        self._evaluate(root_is_p2)
                torch.cuda.current_stream().wait_stream(s)
                # several simulation steps per graph: one hipGraphLaunch (and its host/front-end cost) per
                # `unroll` steps instead of per step
                self._unroll = max(k for k in (25, 20, 16, 10, 8, 5, 4, 2, 1) if self.sims % k == 0)
                g = torch.cuda.CUDAGraph()
                selected = False
                keep = []
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
            except Exception:
                self.use_graph = False
                self._graph = None
                if selected:                                   # close the half-captured step on the host side
                    e.expand_backup(p, v)                      # (device side: no pending leaf -> no-op)
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
            if i % 8 == 7 and (self.eng.slots()['status'] != _lib.ST_RUNNING).all():
                break
        return self.collect()

    def _check_log_complete(self):
        """a full sample log drops rows (the engine ends such games with status ERROR and counts an error): games
        with missing plies must never reach the training data, so collecting from such a run raises"""
        c = self.eng.counters()
        if c['errors']:
            raise _lib.CcspError('%d engine errors (sample log full after %d of %d rows, or a game ended in ERROR): '
                                 'raise log_capacity' % (c['errors'], self.eng.log_size(), self.eng.log_capacity))

    def collect(self):
        """per finished game, in game-id order: (play_history, p1_reward) or (None, None) -- the
        return value of selfplay() (selfplay.py:45-47, 72-80)"""
        e = self.eng
        self._check_log_complete()
        st, meta, pi = e.log()
        res = e.results()
        order = np.lexsort((meta['ply'], meta['game']))
        by_game = {}
        for r in order:
            by_game.setdefault(int(meta['game'][r]), []).append(r)
        out = []
        for k in range(len(res)):
            game = e.first_game + k * e.game_stride
            status = int(res['status'][k])
            if status in (_lib.ST_WON_P1, _lib.ST_WON_P2):
                rows = by_game.get(game, [])
                if self.randomised:
                    rows = rows[3:]                         # selfplay.py:76-78
                hist = [(BoardView(st[r]), pi[r].copy()) for r in rows]
                out.append((hist, int(res['reward'][k])))
            elif status in (_lib.ST_DISCARD_REPETITION, _lib.ST_DISCARD_NO_PROGRESS):
                out.append((None, None))
            elif status == _lib.ST_ERROR:
                raise _lib.CcspError('game %d ended in ERROR status' % game)
            else:
                out.append(('unfinished', status))
        return out

    def collect_train_data(self):
        """utils.convert_to_train_data(self.collect()) as arrays, without building a Python object per position
        (utils.log_to_train_data): (board_x [N,7,7,7] f64, pi_y [N,294] f64, v_y [N] int64) of the games won so far"""
        from . import utils
        e = self.eng
        self._check_log_complete()
        st, meta, pi = e.log()
        return utils.log_to_train_data(st, meta, pi, e.results(), first_game=e.first_game, game_stride=e.game_stride,
                                       randomised=self.randomised)

    def close(self):
        self.eng.close()


class PipelinedSelfPlay(object):
    """The same batch as `n_parts` BatchSelfPlay halves on their own HIP streams (game ids interleaved), so that the
    select / expand-backup kernels and launch gaps of one half run under the evaluator kernel of the other: the
    evaluator fills the GPU with one workgroup per CU at 2048 positions, the tree kernels need almost nothing.
    4096 games x 400 simulations with good_model.h5: 12.5 -> 13.4 M node-expansions/s with two parts (four are slower).
    Same interface as BatchSelfPlay (play_ply / run_to_completion / collect / close)."""

    def __init__(self, model1, model2=None, n_slots=2, n_parts=2, first_game=0, game_stride=1, max_games=None, log_capacity=None,
                 auto_restart=False, **kw):
        import torch
        max_games = n_slots if max_games is None else max_games
        # part i plays ids first_game + (i + k n_parts) game_stride: with restarts every part draws from its own id budget
        assert n_slots % n_parts == 0 and max_games % n_parts == 0 and (auto_restart or max_games == n_slots)
        self.torch = torch
        self.n_parts, self.n_slots = n_parts, n_slots
        per = n_slots // n_parts
        self.parts = [BatchSelfPlay(model1, model2, n_slots=per, first_game=first_game + i * game_stride,
                                    game_stride=game_stride * n_parts, max_games=max_games // n_parts, auto_restart=auto_restart,
                                    log_capacity=None if log_capacity is None else log_capacity // n_parts, **kw)
                      for i in range(n_parts)]
        self.streams = [torch.cuda.Stream() for _ in range(n_parts)]

    def play_ply(self):
        cur = self.torch.cuda.current_stream()
        for b, st in zip(self.parts, self.streams):
            st.wait_stream(cur)
            with self.torch.cuda.stream(st):
                b.play_ply()
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

    def collect(self):
        outs = [b.collect() for b in self.parts]
        return [outs[j % self.n_parts][j // self.n_parts] for j in range(sum(len(o) for o in outs))]

    def collect_train_data(self):
        """the parts' samples, concatenated (training order is shuffled anyway)"""
        outs = [b.collect_train_data() for b in self.parts]
        return tuple(np.concatenate([o[i] for o in outs]) for i in range(3))

    def close(self):
        for b in self.parts:
            b.close()


def selfplay_batch(model1, model2=None, n_games=1, sims=MCTS_SIMULATIONS, seed=None, randomised=False,
                   first_game=0, game_stride=1, device=0):
    """n_games games as one batch; returns [(play_history, p1_reward) | (None, None)] in game-id order"""
    if n_games >= 2048 and n_games % 2 == 0 and hasattr(_batched(model1), 'model'):
        b = PipelinedSelfPlay(model1, model2, n_slots=n_games, n_parts=2, sims=sims, seed=seed, first_game=first_game,
                              game_stride=game_stride, randomised=randomised, device=device, log_capacity=n_games * 512)
        try:
            return b.run_to_completion()
        finally:
            b.close()
    b = BatchSelfPlay(model1, model2, n_slots=n_games, sims=sims, seed=seed, first_game=first_game,
                      game_stride=game_stride, max_games=n_games, randomised=randomised, device=device,
                      log_capacity=n_games * 512)
    try:
        return b.run_to_completion()
    finally:
        b.close()


def selfplay(model1, model2=None, randomised=False, sims=MCTS_SIMULATIONS, seed=None, game_id=None):
    """selfplay.py:11-80: one game -> (play_history, p1_reward), or (None, None) if the game was discarded"""
    if game_id is None:
        game_id = _next_game[0]
        _next_game[0] += 1
    return selfplay_batch(model1, model2, n_games=1, sims=sims, seed=seed, randomised=randomised, first_game=game_id)[0]


def generate_self_play(worker_id, model_path, num_self_play, model2_path=None, sims=MCTS_SIMULATIONS, seed=None):
    """train.py:27-67: load the model(s) and return [(play_history, p1_reward)] of the games that were
    not discarded.  The games are played as ONE batch on this worker's GPU; worker w of W plays game ids
    w-1, w-1+W, ... when the caller passes seeds/ids accordingly (see parallel use in INTEGRATION.md)."""
    from .model import ResidualCNN
    model = ResidualCNN()
    model2 = None
    if model_path is not None:
        model.load_weights(model_path)
        if model2_path is not None:
            model2 = ResidualCNN()
            model2.load_weights(model2_path)
    first = _next_game[0]
    _next_game[0] += num_self_play
    games = selfplay_batch(model, model2, n_games=num_self_play, sims=sims, seed=seed, first_game=first)
    return [(h, r) for h, r in games if h is not None and r is not None and h != 'unfinished']


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

