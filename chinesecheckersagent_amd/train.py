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
process per GPU, DistributedDataParallel (RCCL) averages the 999,408-byte gradient each step.
Weights are read from and written to the reference's own `versionNNNN-weights.h5` layout (h5lite).
"""
import os

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

    def bn(c):                                   # Keras momentum 0.99 <-> torch momentum 0.01
        return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)

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
            self.ddp = torch.nn.parallel.DistributedDataParallel(self.net, device_ids=[self.device.index] if self.device.type == 'cuda' else None)
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
        if not os.path.exists(save_dir):
            os.makedirs(save_dir)
        st = self.state_as_keras()
        layers = [(ln, [(wn, st[wn].astype(np.float32)) for wn in wns]) for ln, wns in keras_layer_names()]
        path = '{}/{}{:0>4}-weights.h5'.format(save_dir, prefix, version)
        write_keras_weights(path, layers)
        return path

    # ---- loss and step -----------------------------------------------------------------------------------------
    def loss(self, logits, v, pi, z):
        torch = self.torch
        policy = -(pi * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()               # loss.py:4, averaged by Keras
        value = ((v - z) ** 2).mean()                                                      # 'mean_squared_error'
        reg = sum((k ** 2).sum() for k in self.net.kernels()) * REG_CONST                  # l2 on every kernel
        return policy + value + reg, policy, value, reg

    def step(self, x, pi, z):
        self.net.train()
        model = self.ddp if self.ddp is not None else self.net
        logits, v = model(x)
        total, policy, value, reg = self.loss(logits, v, pi, z)
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        self.opt.step()
        return float(total.detach()), float(policy.detach()), float(value.detach()), float(reg.detach())

    def fit(self, board_x, pi_y, v_y, batch_size=BATCH_SIZE, epochs=EPOCHS, validation_split=0.05, seed=0):
        """keras fit semantics: the LAST validation_split of the arrays is held out (before shuffling), the
        rest is reshuffled every epoch; returns per-epoch (train loss, val loss)"""
        torch = self.torch
        x = torch.as_tensor(np.asarray(board_x), dtype=torch.float32)
        pi = torch.as_tensor(np.asarray(pi_y), dtype=torch.float32)
        z = torch.as_tensor(np.asarray(v_y), dtype=torch.float32)
        n = len(x)
        split = int(n * (1.0 - validation_split))
        xt, pt, zt = x[:split].to(self.device), pi[:split].to(self.device), z[:split].to(self.device)
        xv, pv, zv = x[split:].to(self.device), pi[split:].to(self.device), z[split:].to(self.device)
        gen = torch.Generator().manual_seed(seed)
        hist = []
        for _ in range(epochs):
            perm = torch.randperm(split, generator=gen).to(self.device)
            tot, cnt = 0.0, 0
            for i in range(0, split, batch_size):
                idx = perm[i:i + batch_size]
                if self.ddp is not None:
                    # every rank draws the SAME permutation (same seed) and takes its own equal share of each global batch:
                    # the gradient DistributedDataParallel averages is then the mean over the global batch, not N copies
                    # of the same work; what does not divide by the world size is dropped from that batch
                    import torch.distributed as dist
                    world, rank = dist.get_world_size(), dist.get_rank()
                    idx = idx[:len(idx) // world * world][rank::world]
                    if len(idx) == 0:
                        continue
                l = self.step(xt[idx], pt[idx], zt[idx])[0]
                tot += l * len(idx); cnt += len(idx)
            val = None
            if len(xv):
                self.net.eval()
                with torch.no_grad():
                    val = float(self.loss(*self.net(xv), pv, zv)[0])
            hist.append((tot / max(cnt, 1), val))
        return hist


def train(model_path, board_x, pi_y, v_y, data_retention, version, save_dir=SAVE_WEIGHTS_DIR, device=None, seed=0, ddp=False):
    """train.train (train.py:109-146): load weights, keep a random `data_retention` fraction of the samples
    (train.py:134-137), fit, save 'saved-weights/version{version:0>4}-weights.h5'.  Returns the path.
    ddp=True (one process per GPU, torch.distributed initialised by the caller): every rank calls this with the SAME
    arrays and seed; fit() gives each rank its share of every batch, the replicas stay identical, rank 0's file is the result."""
    t = Trainer(device=device, ddp=ddp)
    if model_path is not None:
        t.load_weights(model_path)
    n = len(v_y)
    rng = np.random.RandomState(seed)
    keep = rng.choice(n, int(data_retention * n), replace=False)
    bx, py, vy = np.asarray(board_x)[keep], np.asarray(pi_y)[keep], np.asarray(v_y)[keep]
    t.fit(bx, py, vy, seed=seed)
    return t.save_weights(save_dir, MODEL_PREFIX, version)


# ---- the training loop around the path (train.py:235-352) -------------------------------------------------------
NUM_SELF_PLAY = 180               # config.py:57
EVAL_GAMES = 24                   # config.py:40
PAST_ITER_COUNT = 1               # config.py:51
DEF_DATA_RETENTION_RATE = 0.5     # config.py:52


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


def evaluate(best_model, cur_model, num_games=EVAL_GAMES, sims=None, seed=None, first_game=0):
    """train.evaluate / evaluate_in_parallel (train.py:150-231): num_games with alternating colours and the 100-move
    limit, all as one batch on the GPU; returns the number of games cur_model won."""
    from . import arena
    from .config import MCTS_SIMULATIONS
    w_best, w_cur, draws = arena.evaluate(best_model, cur_model, num_games, enforce_move_limit=True,
                                          sims=MCTS_SIMULATIONS if sims is None else sims, seed=seed, first_game=first_game)
    return w_cur


def evolve(cur_model_path, other_opponent_for_selfplay=None, iteration_count=0, best_model=None, iterations=None,
           num_self_play=NUM_SELF_PLAY, eval_games=EVAL_GAMES, sims=None, seed=None, data_dir=None, weights_dir=SAVE_WEIGHTS_DIR,
           log=print):
    """train.evolve (train.py:235-317): self-play -> convert / augment / save -> pool with the previous iteration ->
    train -> (if a best model is tracked) gate the new weights on more than int(0.55 * eval_games) wins.
    The reference loops forever; `iterations` bounds it (None = forever).  One iteration's `num_self_play` games are
    ONE batch on this GPU (generate_self_play_in_parallel's 12 workers, train.py:71-105), the gate's games another.
    Returns (cur_model_path, best_model, iteration_count)."""
    from . import selfplay as sp
    from . import utils
    from .config import MCTS_SIMULATIONS, SAVE_TRAIN_DATA_DIR
    from .model import ResidualCNN
    sims = MCTS_SIMULATIONS if sims is None else sims
    data_dir = SAVE_TRAIN_DATA_DIR if data_dir is None else data_dir
    done = 0
    game0 = 0
    while iterations is None or done < iterations:
        # generate plays (train.py:246-262): current model vs another, or the best model alone, or the current alone
        if other_opponent_for_selfplay is not None:
            p1, p2 = cur_model_path, other_opponent_for_selfplay
        elif best_model is not None:
            p1, p2 = best_model, None
        else:
            p1, p2 = cur_model_path, None
        m1 = ResidualCNN()
        if p1 is not None:
            m1.load_weights(p1)
        m2 = None
        if p2 is not None:
            m2 = ResidualCNN()
            m2.load_weights(p2)
        games = sp.selfplay_batch(m1, m2, n_games=num_self_play, sims=sims, seed=seed, first_game=game0)
        game0 += num_self_play
        games = [(h, r) for h, r in games if h is not None and r is not None and h != 'unfinished']
        log('iteration %d: %d of %d self-play games kept' % (iteration_count, len(games), num_self_play))
        # prepare data (train.py:268-285)
        board_x, pi_y, v_y = utils.convert_to_train_data(games)
        board_x, pi_y, v_y = utils.augment_train_data(board_x, pi_y, v_y)
        board_x, pi_y, v_y = np.array(board_x), np.array(pi_y), np.array(v_y)
        if len(board_x) > 0 and len(pi_y) > 0 and len(v_y) > 0:
            utils.save_train_data(board_x, pi_y, v_y, version=iteration_count, directory=data_dir)
        board_x, pi_y, v_y, used = combine_prev_iters_train_data(board_x, pi_y, v_y, iteration_count, directory=data_dir)
        if used == 0:
            log('no training data for iteration %d, re-iterating' % iteration_count)
            done += 1
            continue
        retention = min(1. / used, DEF_DATA_RETENTION_RATE)            # train.py:288
        # train (train.py:294-304; in this process -- there is no TensorFlow session to keep out of it)
        cur_model_path = train(cur_model_path, board_x, pi_y, v_y, retention, iteration_count, save_dir=weights_dir)
        # evaluate (train.py:308-314)
        if best_model is not None:
            wins = evaluate(best_model, cur_model_path, eval_games, sims=sims, seed=seed, first_game=game0)
            game0 += eval_games
            if wins > int(0.55 * eval_games):
                best_model = cur_model_path
                log('now using %s as the best model (%d/%d wins)' % (best_model, wins, eval_games))
            else:
                log('iteration %d is not better (%d/%d wins); retaining %s' % (iteration_count, wins, eval_games, best_model))
        iteration_count += 1
        done += 1
    return cur_model_path, best_model, iteration_count
