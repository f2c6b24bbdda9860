"""next-2: the training step on generated (board_x, pi_y, v_y) -- train.train (train.py:109-146)
with the model and losses of model.py:58-87 / loss.py:3-4 on PyTorch(-ROCm):

    SGD(lr=1e-4, momentum=0.9, nesterov=True)                      model.py:83, config.py:55
    loss = softmax_cross_entropy_with_logits(pi, policy logits)     loss.py:4       (weight 1, config.py:43)
         + mean_squared_error(z, value)                             model.py:82
         + 6e-3 * sum(kernel**2) over every Conv2D / Dense kernel   regularizers.l2(REG_CONST), model.py:60, config.py:54
    fit(batch_size=32, epochs=5, validation_split=0.05, shuffle=True)   train.py:139-143
    BatchNormalization(momentum=0.99, epsilon=1e-3) in training mode (Keras defaults, model.py:63)

PARITY UNPINNED: Keras/TensorFlow are not installed and the reference holds no training vectors, so this
follows the documented Keras 2.1.6 semantics; its step is checked against a hand-derived NumPy float64 restatement of those
semantics (oracle/train_oracle.py, tests/test_train.py), not against Keras itself.  Multi-GPU: one
process per GPU; every rank takes its share of each GLOBAL batch of 32, BatchNormalization statistics are all-reduced over the
ranks (the step is the one a single GPU takes on the whole batch, moving statistics included) and DistributedDataParallel
(RCCL) averages the 999,408-byte gradient each step.
Weights are read from and written to the reference's own `versionNNNN-weights.h5` layout (h5lite).
"""
import os
import time

import numpy as np

from .config import INPUT_DIM, NUM_ACTIONS, NUM_FILTERS
from .h5lite import H5File, write_keras_weights

REG_CONST = 6e-3
LEARNING_RATE = 0.0001
BATCH_SIZE = 32
EPOCHS = 5
SAVE_WEIGHTS_DIR = 'saved-weights/'
MODEL_PREFIX = 'version'


def keras_layer_names():
    """the 105 layer names of the reference model in Keras' order (model.layers), with their weight names"""
    out = [('input_1', [])]

    def conv(i):
        return ('conv2d_%d' % i, ['conv2d_%d/kernel:0' % i, 'conv2d_%d/bias:0' % i])

    def bn(i):
        return ('batch_normalization_%d' % i, ['batch_normalization_%d/%s:0' % (i, n) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance')])
    out += [conv(1), bn(1), ('activation_1', [])]
    c, a = 2, 2
    for blk in range(1, 10):
        out += [conv(c), bn(c), ('activation_%d' % a, []), conv(c + 1), bn(c + 1), ('activation_%d' % (a + 1), []),
                conv(c + 2), bn(c + 2), ('add_%d' % blk, []), ('activation_%d' % (a + 2), [])]
        c += 3
        a += 3
    out += [conv(30), conv(29), bn(30), bn(29), ('activation_30', []), ('activation_29', []), ('flatten_2', []), ('flatten_1', []),
            ('dense_1', ['dense_1/kernel:0', 'dense_1/bias:0']), ('policy_head', ['policy_head/kernel:0', 'policy_head/bias:0']),
            ('value_head', ['value_head/kernel:0', 'value_head/bias:0'])]
    return out


def _build(filters=NUM_FILTERS):
    import torch
    nn = torch.nn

    class BatchNorm(nn.BatchNorm2d):
        """nn.BatchNorm2d whose TRAINING statistics are taken over the rows of every rank (sync = the process group is
        initialised and Trainer(ddp=True) switched it on): mean and biased variance of the global batch through one
        differentiable all-reduce, the moving variance from the unbiased global one -- exactly what ONE process computes
        on the whole batch (to float32 rounding), so N ranks with 32 / N rows each take the single-GPU step (with plain per-rank
        BatchNorm the statistics would come from 4 rows at N = 8).  Works over RCCL and over gloo (CPU tests)."""
        sync = False

        def forward(self, x):
            if not (self.training and self.sync):
                return super().forward(x)
            import torch.distributed as dist
            import torch.distributed.nn.functional as dfn
            # ONE collective per layer and pass: sums of (x - m0) and (x - m0)^2 with the moving mean m0 as the common shift (the
            # same on every rank; it keeps E[d^2] - E[d]^2 well conditioned) and the row count, all in one vector
            c = x.shape[1]
            d0 = x - self.running_mean.detach()[None, :, None, None]
            cnt_local = torch.full((1,), float(x.shape[0] * x.shape[2] * x.shape[3]), dtype=x.dtype, device=x.device)
            tot = dfn.all_reduce(torch.cat([d0.sum(dim=(0, 2, 3)), (d0 * d0).sum(dim=(0, 2, 3)), cnt_local]))
            cnt = tot[2 * c:].detach()
            dm = tot[:c] / cnt
            mean = self.running_mean.detach() + dm
            var = (tot[c:2 * c] / cnt - dm * dm).clamp_min(0.0)
            xc = x - mean[None, :, None, None]
            with torch.no_grad():
                m = self.momentum
                self.running_mean.mul_(1 - m).add_(mean.detach(), alpha=m)
                self.running_var.mul_(1 - m).add_(var.detach() * (cnt / (cnt - 1)), alpha=m)
                self.num_batches_tracked += 1
            y = xc * torch.rsqrt(var + self.eps)[None, :, None, None]
            return y * self.weight[None, :, None, None] + self.bias[None, :, None, None]

    def bn(c):                                   # Keras momentum 0.99 <-> torch momentum 0.01
        return BatchNorm(c, eps=1e-3, momentum=0.01)

    class TrainNet(nn.Module):
        """the reference graph with explicit BatchNorm layers (the inference module of model.py folds them)"""

        def __init__(self):
            super().__init__()
            f = filters
            self.convs = nn.ModuleDict()
            self.bns = nn.ModuleDict()

            def add(i, cin, cout, k, pad):
                self.convs[str(i)] = nn.Conv2d(cin, cout, k, padding=pad)
                self.bns[str(i)] = bn(cout)
            add(1, INPUT_DIM[2], f, 3, 0)
            i = 2
            for _ in range(9):
                add(i, f, 32, 1, 0); add(i + 1, 32, 32, 3, 1); add(i + 2, 32, f, 1, 0)
                i += 3
            add(29, f, 16, 1, 0)
            add(30, f, 1, 1, 0)
            self.policy_head = nn.Linear(400, NUM_ACTIONS)
            self.dense_1 = nn.Linear(25, 32)
            self.value_head = nn.Linear(32, 1)

        def cb(self, x, i):
            return self.bns[str(i)](self.convs[str(i)](x))

        def forward(self, x):                    # x [N,7,7,7] (row, col, channel)
            x = torch.relu(self.cb(x.permute(0, 3, 1, 2), 1))
            i = 2
            for _ in range(9):
                y = torch.relu(self.cb(x, i))
                y = torch.relu(self.cb(y, i + 1))
                x = torch.relu(self.cb(y, i + 2) + x)
                i += 3
            p = torch.relu(self.cb(x, 29)).permute(0, 2, 3, 1).flatten(1)       # Flatten over (h, w, c) as Keras does
            logits = self.policy_head(p)
            v = torch.relu(self.cb(x, 30)).permute(0, 2, 3, 1).flatten(1)
            v = torch.tanh(self.value_head(torch.relu(self.dense_1(v))))
            return logits, v[:, 0]

        def kernels(self):
            """every tensor that carries kernel_regularizer=l2(REG_CONST) in the reference"""
            return [c.weight for c in self.convs.values()] + [self.policy_head.weight, self.dense_1.weight, self.value_head.weight]
    return TrainNet()


class Trainer(object):
    def __init__(self, device=None, ddp=False):
        import torch
        self.torch = torch
        if device is None:
            device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.device = torch.device(device)
        self.net = _build().to(self.device)
        self.ddp = None
        if ddp:
            for b in self.net.bns.values():
                b.sync = True                        # global-batch statistics: identical on every rank, nothing to broadcast
            self.ddp = torch.nn.parallel.DistributedDataParallel(self.net, device_ids=[self.device.index] if self.device.type == 'cuda' else None,
                                                                 broadcast_buffers=False)
        self.opt = torch.optim.SGD(self.net.parameters(), lr=LEARNING_RATE, momentum=0.9, nesterov=True)

    # ---- weights in the reference's file layout ------------------------------------------------------------
    def load_weights(self, path):
        torch = self.torch
        from .model import read_keras_weights
        w = read_keras_weights(path)                  # canonical layer names, whole-model files too

        def g(layer, name):
            return torch.from_numpy(np.ascontiguousarray(w['%s/%s/%s:0' % (layer, layer, name)]))
        with torch.no_grad():
            for i, conv in self.net.convs.items():
                conv.weight.copy_(g('conv2d_' + i, 'kernel').permute(3, 2, 0, 1))          # HWIO -> OIHW
                conv.bias.copy_(g('conv2d_' + i, 'bias'))
                b = self.net.bns[i]
                b.weight.copy_(g('batch_normalization_' + i, 'gamma')); b.bias.copy_(g('batch_normalization_' + i, 'beta'))
                b.running_mean.copy_(g('batch_normalization_' + i, 'moving_mean'))
                b.running_var.copy_(g('batch_normalization_' + i, 'moving_variance'))
            for name in ('policy_head', 'dense_1', 'value_head'):
                lin = getattr(self.net, name)
                lin.weight.copy_(g(name, 'kernel').t()); lin.bias.copy_(g(name, 'bias'))

    def state_as_keras(self):
        """{ 'conv2d_1/kernel:0': ndarray, ... } in Keras layouts"""
        out = {}
        for i, conv in self.net.convs.items():
            out['conv2d_%s/kernel:0' % i] = conv.weight.detach().permute(2, 3, 1, 0).cpu().numpy().copy()
            out['conv2d_%s/bias:0' % i] = conv.bias.detach().cpu().numpy().copy()
            b = self.net.bns[i]
            out['batch_normalization_%s/gamma:0' % i] = b.weight.detach().cpu().numpy().copy()
            out['batch_normalization_%s/beta:0' % i] = b.bias.detach().cpu().numpy().copy()
            out['batch_normalization_%s/moving_mean:0' % i] = b.running_mean.detach().cpu().numpy().copy()
            out['batch_normalization_%s/moving_variance:0' % i] = b.running_var.detach().cpu().numpy().copy()
        for name in ('policy_head', 'dense_1', 'value_head'):
            lin = getattr(self.net, name)
            out['%s/kernel:0' % name] = lin.weight.detach().t().cpu().numpy().copy()
            out['%s/bias:0' % name] = lin.bias.detach().cpu().numpy().copy()
        return out

    def save_weights(self, save_dir, prefix, version):
        """Model.save_weights (model.py:33-37): '{dir}/{prefix}{version:0>4}-weights.h5' in Keras' own layout"""
        os.makedirs(save_dir, exist_ok=True)
        st = self.state_as_keras()
        layers = [(ln, [(wn, st[wn].astype(np.float32)) for wn in wns]) for ln, wns in keras_layer_names()]
        path = '{}/{}{:0>4}-weights.h5'.format(save_dir, prefix, version)
        tmp = '%s.tmp%d' % (path, os.getpid())
        write_keras_weights(tmp, layers)
        os.replace(tmp, path)                         # readers see the old file or the whole new one
        return path

    # ---- loss and step -----------------------------------------------------------------------------------------
    def loss(self, logits, v, pi, z):
        torch = self.torch
        policy = -(pi * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()               # loss.py:4, averaged by Keras
        value = ((v - z) ** 2).mean()                                                      # 'mean_squared_error'
        reg = sum((k ** 2).sum() for k in self.net.kernels()) * REG_CONST                  # l2 on every kernel
        return policy + value + reg, policy, value, reg

    def step(self, x, pi, z, global_rows=None):
        """one optimisation step.  Under ddp a rank passes ITS rows of the global batch and the batch's size: its data terms
        are weighted len(x) * world / global_rows, so that DistributedDataParallel's mean over ranks is the mean over the
        global batch whatever the shares (regulariser: the same on every rank)."""
        self.net.train()
        model = self.ddp if self.ddp is not None else self.net
        logits, v = model(x)
        total, policy, value, reg = self.loss(logits, v, pi, z)
        if self.ddp is not None and global_rows is not None:
            import torch.distributed as dist
            total = (policy + value) * (len(x) * dist.get_world_size() / float(global_rows)) + reg
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        self.opt.step()
        return float(total.detach()), float(policy.detach()), float(value.detach()), float(reg.detach())

    def _capture_step(self, batch_size):
        """One optimisation step on a full batch -- forward, the three loss terms, backward, the SGD update: some four hundred small
        kernels -- captured ONCE as a hipGraph on static input tensors and replayed per batch: the step of this 250 k-parameter net is
        launch-bound in eager mode (8 ms; the GPU work itself is well under 2).  Lazy initialisation (convolution algorithm choice,
        the optimiser's momentum buffers) is driven by three throw-away steps whose effect is undone before the capture: parameters
        and BatchNorm statistics are put back, the momentum buffers zeroed (SGD's first step from a zero buffer = its first step)."""
        torch = self.torch
        dev = self.device
        self._gx = torch.zeros((batch_size, 7, 7, 7), dtype=torch.float32, device=dev)
        self._gpi = torch.full((batch_size, NUM_ACTIONS), 1.0 / NUM_ACTIONS, dtype=torch.float32, device=dev)
        self._gz = torch.zeros(batch_size, dtype=torch.float32, device=dev)
        keep = {k: v.clone() for k, v in self.net.state_dict().items()}
        had_state = len(self.opt.state) > 0
        opt_keep = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.opt.state.items()} if had_state else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if quiet is not None:
            quiet(False)                               # (the warm-up runs on a side stream on purpose)
        t_first = time.time()
        try:
            with torch.cuda.stream(side):
                for _ in range(3):
                    self.step(self._gx, self._gpi, self._gz)
            side.synchronize()
            # (the first of them carries the process's lazy initialisation: MIOpen's choice -- on a fresh machine, BUILD -- of the
            # convolution kernels of this net's 31 layer shapes, forward and backward; seconds on a cold kernel cache, none on a warm one)
            self.fit_timings['first_steps_s'] = time.time() - t_first
        except BaseException:
            if quiet is not None:
                quiet(True)
            raise
        finally:
            # whatever happened to the throw-away steps (all-zero batch), training starts from the weights that were loaded: the eager
            # fallback of fit() must not inherit them
            torch.cuda.current_stream().wait_stream(side)
            with torch.no_grad():
                self.net.load_state_dict(keep)
                for p_, st in self.opt.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(opt_keep[id(p_)][k]) if (had_state and id(p_) in opt_keep and k in opt_keep[id(p_)]) else v.zero_()
        self.net.train()
        t_cap = time.time()
        try:
            g = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                logits, v = self.net(self._gx)
                total, policy, value, reg = self.loss(logits, v, self._gpi, self._gz)
                total.backward()
                self.opt.step()
        finally:
            if quiet is not None:
                quiet(True)                            # the warning is back on whether the capture worked or not
        self.fit_timings['capture_s'] = time.time() - t_cap
        self._graph, self._graph_loss, self._graph_bs = g, total, batch_size

    def fit(self, board_x, pi_y, v_y, batch_size=BATCH_SIZE, epochs=EPOCHS, validation_split=0.05, seed=0, use_graph=True):
        """keras fit semantics: the LAST validation_split of the arrays is held out (before shuffling), the
        rest is reshuffled every epoch; returns per-epoch (train loss, val loss).  On one GPU the full batches are replays of a
        captured hipGraph of the step (_capture_step); a ragged last batch, DistributedDataParallel and the CPU run eagerly."""
        torch = self.torch
        # wall seconds of this call by phase (bench.py's config 5 reports them): host -> device copies, the three throw-away steps in
        # front of the capture (lazy initialisation), the capture itself, the epochs
        self.fit_timings = {'to_device_s': 0.0, 'first_steps_s': 0.0, 'capture_s': 0.0, 'epochs_s': 0.0, 'steps': 0, 'graph': False}
        t_fit = time.time()
        x = torch.as_tensor(np.asarray(board_x), dtype=torch.float32)
        pi = torch.as_tensor(np.asarray(pi_y), dtype=torch.float32)
        z = torch.as_tensor(np.asarray(v_y), dtype=torch.float32)
        n = len(x)
        split = int(n * (1.0 - validation_split))
        xt, pt, zt = x[:split].to(self.device), pi[:split].to(self.device), z[:split].to(self.device)
        xv, pv, zv = x[split:].to(self.device), pi[split:].to(self.device), z[split:].to(self.device)
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)
        self.fit_timings['to_device_s'] = time.time() - t_fit
        gen = torch.Generator().manual_seed(seed)
        hist = []
        graph = None
        if use_graph and self.ddp is None and self.device.type == 'cuda' and split >= batch_size:
            try:
                if getattr(self, '_graph', None) is None or self._graph_bs != batch_size:
                    self._capture_step(batch_size)
                graph = self._graph
            except Exception as ex:
                from .selfplay import _strict, _warn
                self._graph = None
                if _strict():
                    raise
                _warn('hipGraph capture of the training step failed (%r): eager steps' % (ex,))
        self.fit_timings['graph'] = graph is not None
        t_ep = time.time()
        for _ in range(epochs):
            perm = torch.randperm(split, generator=gen).to(self.device)
            tot, cnt = 0.0, 0
            self.fit_timings['steps'] += (split + batch_size - 1) // batch_size
            tot_dev = torch.zeros((), dtype=torch.float64, device=self.device) if graph is not None else None
            if graph is not None:                       # the epoch's rows in their shuffled order, gathered once: a step is three copies + a replay
                xe, pe, ze = xt[perm], pt[perm], zt[perm]
            for i in range(0, split, batch_size):
                idx = perm[i:i + batch_size]
                if graph is not None and len(idx) == batch_size:
                    self._gx.copy_(xe[i:i + batch_size]); self._gpi.copy_(pe[i:i + batch_size]); self._gz.copy_(ze[i:i + batch_size])
                    graph.replay()
                    tot_dev += self._graph_loss.detach().double() * batch_size        # (no host read-back inside the epoch)
                    cnt += batch_size
                    continue
                if self.ddp is not None:
                    # every rank draws the SAME permutation (same seed) and takes its own share of each global batch (rows
                    # rank, rank + world, ...: shares differ by at most one row and step() weights them, so the averaged
                    # gradient is the global batch's); a batch with fewer rows than ranks is skipped on every rank alike
                    import torch.distributed as dist
                    world, rank = dist.get_world_size(), dist.get_rank()
                    n_global = len(idx)
                    if n_global < world:
                        continue
                    idx = idx[rank::world]
                    l = self.step(xt[idx], pt[idx], zt[idx], global_rows=n_global)[0]
                else:
                    l = self.step(xt[idx], pt[idx], zt[idx])[0]
                tot += l * len(idx); cnt += len(idx)
            val = None
            if len(xv):
                self.net.eval()
                with torch.no_grad():
                    val = float(self.loss(*self.net(xv), pv, zv)[0])
            if tot_dev is not None:
                tot += float(tot_dev)
            hist.append((tot / max(cnt, 1), val))
        self.fit_timings['epochs_s'] = time.time() - t_ep          # (the float(...) read-backs above have synchronised)
        return hist


last_fit_timings = {}          # Trainer.fit_timings of the last train() call of this process


def warm_up(device=None, batch_size=BATCH_SIZE):
    """The process's lazy initialisation of the training step, on its own: a throw-away Trainer takes the three steps in front of a
    capture (MIOpen chooses -- on a machine with a cold kernel cache, BUILDS -- the forward and backward kernels of the net's layer
    shapes) and captures the step graph once.  Seconds on a fresh machine (measured on fresh GPU boxes: ~5 s of an 11-s first fit), a
    fraction of one afterwards; a loop that wants its first iteration's `train` to cost what every later one costs calls this first.
    -> wall seconds."""
    t0 = time.time()
    t = Trainer(device=device)
    if t.device.type == 'cuda':
        t.fit_timings = {}
        t._capture_step(batch_size)
        t.torch.cuda.synchronize(t.device)
    return time.time() - t0


def train(model_path, board_x, pi_y, v_y, data_retention, version, save_dir=SAVE_WEIGHTS_DIR, device=None, seed=0, ddp=False):
    """train.train (train.py:109-146): load weights, keep a random `data_retention` fraction of the samples
    (train.py:134-137), fit, save 'saved-weights/version{version:0>4}-weights.h5'.  Returns the path.
    ddp=True (one process per GPU, torch.distributed initialised by the caller): every rank calls this with the SAME
    arrays and seed; fit() gives each rank its share of every global batch, BatchNormalization statistics are global, so the
    replicas stay identical -- moving statistics included.  Rank 0 alone writes the file (temporary name, then os.replace);
    every rank returns from the barrier behind it, so the path is readable when this returns."""
    t = Trainer(device=device, ddp=ddp)
    if model_path is not None:
        t.load_weights(model_path)
    n = len(v_y)
    rng = np.random.RandomState(seed)
    keep = rng.choice(n, int(data_retention * n), replace=False)
    bx, py, vy = np.asarray(board_x)[keep], np.asarray(pi_y)[keep], np.asarray(v_y)[keep]
    t.fit(bx, py, vy, seed=seed)
    last_fit_timings.clear()
    last_fit_timings.update(t.fit_timings)             # (evolve reports them: bench.py's config 5)
    if not ddp:
        return t.save_weights(save_dir, MODEL_PREFIX, version)
    import torch.distributed as dist
    path = '{}/{}{:0>4}-weights.h5'.format(save_dir, MODEL_PREFIX, version)
    if dist.get_rank() == 0:
        path = t.save_weights(save_dir, MODEL_PREFIX, version)
    dist.barrier()
    return path


# ---- the training loop around the path (train.py:235-352) -------------------------------------------------------
NUM_SELF_PLAY = 180               # config.py:57
EVAL_GAMES = 24                   # config.py:40
PAST_ITER_COUNT = 1               # config.py:51
DEF_DATA_RETENTION_RATE = 0.5     # config.py:52


def version_of_filename(filename):
    """the 4-digit iteration number in a weights file name ('version0016-weights.h5', 'greedy-model0003.h5'), -1 if there is
    none: what the reference resumes from (utils.py:12-19, train.py:390-394)"""
    import re
    m = re.search(r'(%s|%s)([0-9]{4})(-weights|)\.h5' % (MODEL_PREFIX, 'greedy-model'), filename)
    return int(m.group(2)) if m else -1


def get_weights_path_from_version(version, save_dir=SAVE_WEIGHTS_DIR):
    """train.py:360-361"""
    return '{}/{}{:0>4}-weights.h5'.format(save_dir.rstrip('/'), MODEL_PREFIX, version)


def combine_prev_iters_train_data(board_x, pi_y, v_y, iteration_count, directory=None):
    """train.py:321-352: this iteration's samples plus the files of the PAST_ITER_COUNT iterations before it (this
    iteration's own file is one of them when it was just saved -- the reference reads range(it - PAST, it)).
    -> (board_x, pi_y, v_y, number of sources used)"""
    from .config import SAVE_TRAIN_DATA_DIR, SAVE_TRAIN_DATA_PREF
    directory = SAVE_TRAIN_DATA_DIR if directory is None else directory
    bx, py, vy = [], [], []
    if len(board_x) > 0 and len(pi_y) > 0 and len(v_y) > 0:
        bx.append(np.asarray(board_x)); py.append(np.asarray(pi_y)); vy.append(np.asarray(v_y))
    for i in range(iteration_count - PAST_ITER_COUNT, iteration_count):
        if i < 0:
            continue
        filename = os.path.join(directory, '%s%d.h5' % (SAVE_TRAIN_DATA_PREF, i))
        if not os.path.exists(filename):
            continue
        f = H5File(filename)
        bx.append(np.array(f.get('board_x'))); py.append(np.array(f.get('pi_y'))); vy.append(np.array(f.get('v_y')))
    if bx:
        return np.vstack(bx), np.vstack(py), np.hstack(vy), len(bx)
    return [], [], [], 0


def evaluate(best_model, cur_model, num_games=EVAL_GAMES, sims=None, seed=None, first_game=0, dist=None, device=0):
    """train.evaluate / evaluate_in_parallel (train.py:150-231): num_games with alternating colours and the 100-move
    limit, all as one batch per GPU (sharded over the ranks of `dist`); returns the number of games cur_model won."""
    from . import arena
    from .config import MCTS_SIMULATIONS
    w_best, w_cur, draws = arena.evaluate(best_model, cur_model, num_games, enforce_move_limit=True,
                                          sims=MCTS_SIMULATIONS if sims is None else sims, seed=seed, first_game=first_game,
                                          dist=dist, device=device)
    return w_cur


def _selfplay_shard(p1, p2, num_self_play, sims, seed, game0, dist, device, data_dir, iteration_count, min_games_per_rank=None):
    """this rank's share of an iteration's self-play games -> the iteration's (board_x, pi_y, v_y) on EVERY rank, rows in
    game-id order (what one GPU playing all the games would return), plus (games kept, games played, expansions)."""
    import torch
    from . import selfplay as sp
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    # a small cohort is latency-bound: min(world, ceil(games / 256)) ranks play it (selfplay.selfplay_ranks), the others wait at the
    # all-reduce below and join the fit and the arena -- config 5's real 180 games: one rank (profiles/r6_small_cohort.txt)
    active = sp.selfplay_ranks(num_self_play, world, min_games_per_rank)
    mine = len(range(rank, num_self_play, active)) if rank < active else 0
    sink = sp.TrainDataSink()
    kept = played = expansions = 0
    if mine > 0:
        m1, m2 = sp._load_models(p1, p2, device='cuda:%d' % device)
        run = sp.SelfPlayRun(m1, m2, n_games=mine, sims=sims, seed=seed, first_game=game0 + rank, game_stride=active, device=device,
                             keep_records=False, sink=sink)
        try:
            run.run()
            st = run.store.results['status']
            from . import _lib
            kept = int(((st == _lib.ST_WON_P1) | (st == _lib.ST_WON_P2)).sum())
            played = mine
            c = run.counters()
            expansions = c['expansions']
            if c['errors']:
                raise _lib.CcspError('%d self-play games ended in ERROR' % c['errors'])
        finally:
            run.close()
    bx, py, vy, gid = sink.arrays(canonical=True, with_games=True)
    if dist is None:
        return bx, py, vy, kept, played, expansions
    # the ranks' rows meet through files of the (node-local) data directory: rows carry their game id, the union is sorted by it
    from .launch import coll_device
    shard_dir = os.path.join(data_dir, '.shards')
    os.makedirs(shard_dir, exist_ok=True)
    mine_path = os.path.join(shard_dir, 'iter%d-rank%d.npz' % (iteration_count, rank))
    tmp = mine_path + '.tmp.npz'
    np.savez(tmp, board_x=bx, pi_y=py, v_y=vy, game=gid)
    os.replace(tmp, mine_path)
    t = torch.tensor([kept, played, expansions], dtype=torch.int64, device=coll_device(dist))
    dist.all_reduce(t)                                    # (also the barrier behind which every shard file is complete)
    kept, played, expansions = (int(x) for x in t.cpu())
    parts = [np.load(os.path.join(shard_dir, 'iter%d-rank%d.npz' % (iteration_count, r))) for r in range(world)]
    bx = np.concatenate([z['board_x'] for z in parts]); py = np.concatenate([z['pi_y'] for z in parts])
    vy = np.concatenate([z['v_y'] for z in parts]); gid = np.concatenate([z['game'] for z in parts])
    o = np.argsort(gid, kind='stable')
    dist.barrier()                                        # everyone has read the shards
    if rank == 0:
        for r in range(world):
            os.remove(os.path.join(shard_dir, 'iter%d-rank%d.npz' % (iteration_count, r)))
    return bx[o], py[o], vy[o], kept, played, expansions


def evolve(cur_model_path, other_opponent_for_selfplay=None, iteration_count=0, best_model=None, iterations=None,
           num_self_play=NUM_SELF_PLAY, eval_games=EVAL_GAMES, sims=None, seed=None, data_dir=None, weights_dir=SAVE_WEIGHTS_DIR,
           log=print, dist=None, device=0, timings=None, selfplay_min_games_per_rank=None):
    """train.evolve (train.py:235-317): self-play -> convert / augment / save -> pool with the previous iteration ->
    train -> (if a best model is tracked) gate the new weights on more than int(0.55 * eval_games) wins.
    The reference loops forever; `iterations` bounds it (None = forever).

    One GPU (dist=None): an iteration's games run in up to 4096 restarting slots on `device`, the gate's games as one batch.
    N GPUs: every rank calls this with the same arguments and its own `device` after torch.distributed is initialised
    (launch.init_rank; `python -m chinesecheckersagent_amd.train --gpus N` starts the ranks): rank r plays the self-play
    games r, r + N, ... and the arena games r, r + N, ..., the ranks' rows are merged in game-id order (the SAME arrays a
    single GPU produces), the fit runs under DistributedDataParallel with each rank's share of every batch, and the gate's win
    counts are all-reduced -- every rank takes the same decision.  Rank 0 writes the data and weight files.  A cohort of fewer than
    256 games per rank is played by fewer ranks (selfplay.selfplay_ranks; selfplay_min_games_per_rank overrides the 256): latency-bound.
    `timings` (a list) receives one dict of wall seconds per phase and iteration.
    Returns (cur_model_path, best_model, iteration_count)."""
    import time
    from . import utils
    from .config import MCTS_SIMULATIONS, SAVE_TRAIN_DATA_DIR
    from .selfplay import selfplay_ranks as sp_ranks
    sims = MCTS_SIMULATIONS if sims is None else sims
    data_dir = SAVE_TRAIN_DATA_DIR if data_dir is None else data_dir
    rank = dist.get_rank() if dist is not None else 0
    if rank != 0:
        log = lambda *a, **k: None                                      # noqa: E731
    done = 0
    # game ids (= draw streams) of an iteration follow from its NUMBER: a run that is re-invoked at iteration k (the CLI resumes with
    # iteration_count = version + 1) with an unchanged self-play model -- e.g. the best model kept after a rejected gate -- would
    # otherwise replay iteration 0's ids bit for bit and pool duplicates of its games with the previous iteration's data.  (The
    # reference reseeds every worker from OS entropy, train.py:38-39.)
    game0 = iteration_count * (num_self_play + eval_games)
    while iterations is None or done < iterations:
        tm = {'iteration': iteration_count}
        t0 = time.time()
        # generate plays (train.py:246-262): current model vs another, or the best model alone, or the current alone
        if other_opponent_for_selfplay is not None:
            p1, p2 = cur_model_path, other_opponent_for_selfplay
        elif best_model is not None:
            p1, p2 = best_model, None
        else:
            p1, p2 = cur_model_path, None
        board_x, pi_y, v_y, kept, played, expansions = _selfplay_shard(p1, p2, num_self_play, sims, seed, game0, dist, device,
                                                                       data_dir, iteration_count, selfplay_min_games_per_rank)
        if dist is not None:
            tm['selfplay_ranks'] = sp_ranks(num_self_play, dist.get_world_size(), selfplay_min_games_per_rank)
        game0 += num_self_play
        tm['selfplay_s'] = time.time() - t0
        tm['selfplay_games'], tm['selfplay_games_kept'], tm['selfplay_expansions'] = played, kept, expansions
        log('iteration %d: %d of %d self-play games kept' % (iteration_count, kept, num_self_play))
        # prepare data (train.py:268-285)
        t0 = time.time()
        board_x, pi_y, v_y = utils.augment_train_data(list(board_x), list(pi_y), list(v_y))
        board_x, pi_y, v_y = np.array(board_x), np.array(pi_y), np.array(v_y)
        if len(board_x) > 0 and len(pi_y) > 0 and len(v_y) > 0 and rank == 0:
            utils.save_train_data(board_x, pi_y, v_y, version=iteration_count, directory=data_dir)
        if dist is not None:
            dist.barrier()                                              # the iteration's file is there for every rank
        board_x, pi_y, v_y, used = combine_prev_iters_train_data(board_x, pi_y, v_y, iteration_count, directory=data_dir)
        tm['data_s'] = time.time() - t0
        tm['samples'] = int(len(v_y))
        log('iteration %d: self-play %.1f s, data %.1f s, %d samples to fit' % (iteration_count, tm['selfplay_s'], tm['data_s'], tm['samples']))
        if used == 0:
            log('no training data for iteration %d, re-iterating' % iteration_count)
            done += 1
            if timings is not None:
                timings.append(tm)
            continue
        retention = min(1. / used, DEF_DATA_RETENTION_RATE)            # train.py:288
        # train (train.py:294-304; in this process -- there is no TensorFlow session to keep out of it)
        t0 = time.time()
        cur_model_path = train(cur_model_path, board_x, pi_y, v_y, retention, iteration_count, save_dir=weights_dir,
                               device='cuda:%d' % device if dist is not None else None, ddp=dist is not None)
        tm['train_s'] = time.time() - t0
        # where train_s goes: the fit's lazy initialisation (first steps: MIOpen picks -- on a cold cache builds -- its kernels), the
        # capture of the step graph, the epochs themselves; the rest is loading / saving weights and the retention draw
        tm['train_first_steps_s'] = last_fit_timings.get('first_steps_s', 0.0)
        tm['train_capture_s'] = last_fit_timings.get('capture_s', 0.0)
        tm['train_epochs_s'] = last_fit_timings.get('epochs_s', 0.0)
        tm['train_steps'] = last_fit_timings.get('steps', 0)
        log('iteration %d: fit %.1f s (%d steps, epochs %.1f s)' % (iteration_count, tm['train_s'], tm['train_steps'], tm['train_epochs_s']))
        # evaluate (train.py:308-314)
        if best_model is not None:
            t0 = time.time()
            wins = evaluate(best_model, cur_model_path, eval_games, sims=sims, seed=seed, first_game=game0, dist=dist, device=device)
            game0 += eval_games
            tm['arena_s'] = time.time() - t0
            tm['arena_wins'] = wins
            if wins > int(0.55 * eval_games):
                best_model = cur_model_path
                log('now using %s as the best model (%d/%d wins)' % (best_model, wins, eval_games))
            else:
                log('iteration %d is not better (%d/%d wins); retaining %s' % (iteration_count, wins, eval_games, best_model))
        if timings is not None:
            timings.append(tm)
        iteration_count += 1
        done += 1
    return cur_model_path, best_model, iteration_count


def evolve_rank(config_path):
    """a rank process of evolve_in_parallel (worker.py `evolve`): joins the process group, runs evolve() on its GPU, rank 0
    writes the result next to the config file"""
    import json
    from .launch import init_rank
    cfg = json.load(open(config_path))
    rank, world, local, dist = init_rank()
    timings = []
    cur, best, it = evolve(dist=dist, device=local, timings=timings, **cfg['evolve'])
    if rank == 0:
        with open(cfg['result'] + '.tmp', 'w') as f:
            json.dump(dict(cur_model_path=cur, best_model=best, iteration_count=it, timings=timings, world=world,
                           backend=dist.get_backend() if dist is not None else None), f)
        os.replace(cfg['result'] + '.tmp', cfg['result'])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def evolve_in_parallel(n_gpus, cur_model_path, devices=None, work_dir=None, timeout=None, **kw):
    """train.evolve over `n_gpus` MI355X (BASELINE config 5): starts one rank process per GPU from THIS process, which never
    touches the GPU, and returns (cur_model_path, best_model, iteration_count, timings) of the run.  kw = evolve()'s arguments.
    timeout (seconds): ranks still running after it are terminated (then killed) and the call raises."""
    import json
    import sys
    import tempfile
    from . import launch
    tmp = None
    if work_dir is None:
        tmp = tempfile.TemporaryDirectory(prefix='ccsp-evolve-')
        work_dir = tmp.name
    try:
        cfg = os.path.join(work_dir, 'evolve-config.json')
        result = os.path.join(work_dir, 'evolve-result.json')
        kw = dict(kw, cur_model_path=cur_model_path)
        kw.pop('log', None)
        with open(cfg, 'w') as f:
            json.dump(dict(evolve=kw, result=result), f)
        extra = {'PYTHONPATH': os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                               ([os.environ['PYTHONPATH']] if os.environ.get('PYTHONPATH') else []))}
        if devices is not None and len(set(devices)) < len(devices):
            extra['CCSP_ONE_DEVICE'] = '1'
        rc = launch.run_ranks([sys.executable, '-m', 'chinesecheckersagent_amd.worker', 'evolve', '--config', cfg], n_gpus,
                              devices=devices, extra_env=extra, timeout=timeout)
        if rc:
            raise RuntimeError('evolve_in_parallel: %s' % ('timed out after %s s' % timeout if rc == 124 else 'a rank process failed (exit code %d)' % rc))
        r = json.load(open(result))
        return r['cur_model_path'], r['best_model'], r['iteration_count'], r['timings']
    finally:
        if tmp is not None:
            tmp.cleanup()


if __name__ == '__main__':
    # python -m chinesecheckersagent_amd.train [-c weights.h5] [-b best.h5] [-p opponent.h5] --gpus N --iterations K ...
    # (the flags of the reference's train.py:368-380 plus the sizes it keeps in config.py)
    import argparse
    ap = argparse.ArgumentParser(description='self-play -> train -> arena loop on N MI355X (train.py:235-317)')
    ap.add_argument('-c', '--cur_model_path', default=None)
    ap.add_argument('-b', '--best_model_path', default=None)
    ap.add_argument('-p', '--opponent', default=None)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--iterations', type=int, default=1)
    ap.add_argument('--num-self-play', type=int, default=NUM_SELF_PLAY)
    ap.add_argument('--eval-games', type=int, default=EVAL_GAMES)
    ap.add_argument('--sims', type=int, default=None)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--data-dir', default=None)
    ap.add_argument('--weights-dir', default=SAVE_WEIGHTS_DIR)
    a = ap.parse_args()
    it0 = 0
    if a.cur_model_path is not None:
        it0 = version_of_filename(a.cur_model_path) + 1                  # train.py:390-394
    cur, best, it, timings = evolve_in_parallel(a.gpus, a.cur_model_path, other_opponent_for_selfplay=a.opponent, iteration_count=it0,
                                                best_model=a.best_model_path, iterations=a.iterations, num_self_play=a.num_self_play,
                                                eval_games=a.eval_games, sims=a.sims, seed=a.seed, data_dir=a.data_dir,
                                                weights_dir=a.weights_dir)
    for tm in timings:
        print(tm)
    print('current model: %s; best model: %s; next iteration: %d' % (cur, best, it))
