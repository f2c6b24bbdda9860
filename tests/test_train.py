"""next-2: the training step.  Parity with Keras is UNPINNED (no Keras here); what is
checked: the training graph equals the inference graph on the reference's weights, the loss terms are what
train.py / model.py / loss.py define, the nesterov update, the weight-file layout (against the reference file
through the real HDF5 library when available), and the DDP path on gloo with world_size 2."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def net(golden_dir):
    return np.load(golden_dir + '/net.npz')


def test_training_graph_equals_inference_graph(net, golden_dir):
    import torch
    from chinesecheckersagent_amd.train import Trainer
    t = Trainer(device='cpu')
    t.load_weights(golden_dir + '/good_model.h5')
    t.net.eval()
    x = torch.from_numpy(net['planes'][:64].astype(np.float32))
    with torch.no_grad():
        logits, v = t.net(x)
    assert np.abs(logits.double().numpy() - net['logits_good_model'][:64]).max() < 3e-5
    assert np.abs(v.double().numpy() - net['v_good_model'][:64]).max() < 1e-5


def test_loss_terms_and_nesterov_step(net, golden_dir):
    import torch
    from chinesecheckersagent_amd import train as T
    torch.manual_seed(0)
    t = T.Trainer(device='cpu')
    t.load_weights(golden_dir + '/good_model.h5')
    x = torch.from_numpy(net['planes'][:32].astype(np.float32))
    pi = torch.softmax(torch.randn(32, 294), dim=1)
    z = torch.tensor([1.0, -1.0] * 16)
    t.net.eval()
    with torch.no_grad():
        logits, v = t.net(x)
        total, policy, value, reg = t.loss(logits, v, pi, z)
    # the three terms, restated in numpy float64
    lg = logits.double().numpy()
    lsm = lg - lg.max(1, keepdims=True)
    lsm = lsm - np.log(np.exp(lsm).sum(1, keepdims=True))
    assert abs(float(policy) - float(-(pi.double().numpy() * lsm).sum(1).mean())) < 1e-5
    assert abs(float(value) - float(((v.double().numpy() - z.numpy()) ** 2).mean())) < 1e-6
    k2 = sum(float((k.detach().double() ** 2).sum()) for k in t.net.kernels())
    assert len(t.net.kernels()) == 33 and abs(float(reg) - 6e-3 * k2) < 1e-4 * max(1.0, 6e-3 * k2)
    # one nesterov step on one tensor: w' = w - lr * (g + m * (m * 0 + g)) = w - lr * (1 + m) * g at the first step
    w0 = t.net.value_head.weight.detach().clone()
    t.net.train()
    lg2, v2 = t.net(x)
    tot = t.loss(lg2, v2, pi, z)[0]
    t.opt.zero_grad()
    tot.backward()
    g = t.net.value_head.weight.grad.detach().clone()
    t.opt.step()
    assert torch.allclose(t.net.value_head.weight.detach(), w0 - T.LEARNING_RATE * (1 + 0.9) * g, atol=1e-9)
    # a few steps lower the loss on the same batch; BatchNorm running statistics move with momentum 0.99
    rm0 = t.net.bns['1'].running_mean.clone()
    losses = [t.step(x, pi, z)[0] for _ in range(8)]
    assert losses[-1] < losses[0]
    assert not torch.equal(rm0, t.net.bns['1'].running_mean)


def test_weight_file_layout_roundtrip(golden_dir, tmp_path):
    from chinesecheckersagent_amd import train as T
    from chinesecheckersagent_amd.h5lite import H5File
    from chinesecheckersagent_amd.model import ResidualCNN
    assert len(T.keras_layer_names()) == 105
    t = T.Trainer(device='cpu')
    t.load_weights(golden_dir + '/good_model.h5')
    path = t.save_weights(str(tmp_path), T.MODEL_PREFIX, 17)
    assert path.endswith('version0017-weights.h5')
    a, b = dict(H5File(golden_dir + '/good_model.h5').walk()), dict(H5File(path).walk())
    assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)           # float32 values unchanged
    ResidualCNN(device='cpu').load_weights(path)                                      # and the product loads it
    conda = '/opt/conda/bin/python3.9'
    if os.path.exists(conda) and os.path.exists('/root/reference/good_model.h5'):     # real HDF5 library, reference layout
        code = ("import h5py,sys; a=h5py.File('/root/reference/good_model.h5','r'); b=h5py.File(sys.argv[1],'r'); "
                "assert list(a.attrs['layer_names'])==list(b.attrs['layer_names']); "
                "assert all(list(a[n.decode()].attrs['weight_names'])==list(b[n.decode()].attrs['weight_names']) for n in a.attrs['layer_names']); "
                "print('same layout')")
        assert 'same layout' in subprocess.check_output([conda, '-c', code, path]).decode()


def test_train_function_and_fit(golden_dir, tmp_path):
    from chinesecheckersagent_amd import train as T
    g = np.load(golden_dir + '/net.npz')
    n = 80
    bx = g['planes'][:n].astype(np.float64)
    rng = np.random.RandomState(0)
    py = rng.dirichlet(np.ones(294) * 0.1, size=n)
    vy = np.array([1, -1] * (n // 2))
    path = T.train(golden_dir + '/good_model.h5', bx, py, vy, 0.5, 3, save_dir=str(tmp_path), device='cpu')
    assert os.path.exists(path) and path.endswith('version0003-weights.h5')
    t = T.Trainer(device='cpu')
    hist = t.fit(bx, py, vy, epochs=2)
    assert len(hist) == 2 and hist[0][1] is not None


def _ddp_rank(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from chinesecheckersagent_amd import train as T
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(1)                                   # same initial weights on every rank
    t = T.Trainer(device='cpu', ddp=True)
    g = torch.Generator().manual_seed(100 + rank)          # different data per rank
    x = torch.randint(0, 7, (16, 7, 7, 7), generator=g).float()
    pi = torch.softmax(torch.randn(16, 294, generator=g), dim=1)
    z = torch.ones(16)
    for _ in range(2):
        t.step(x, pi, z)
    # fit(): the same arrays on every rank; each rank must take its own half of every global batch of 8
    gc = torch.Generator().manual_seed(7)
    xs = torch.randint(0, 7, (42, 7, 7, 7), generator=gc).float().numpy()
    ps = torch.softmax(torch.randn(42, 294, generator=gc), dim=1).numpy()
    zs = np.ones(42, dtype=np.int64)
    seen = []
    step0 = t.step
    t.step = lambda a, b, c, **kw: (seen.append(a.clone()), step0(a, b, c, **kw))[1]
    t.fit(xs, ps, zs, batch_size=8, epochs=1, validation_split=0.0, seed=5)
    rows = torch.cat(seen)
    q.put((rank, float(t.net.policy_head.weight.sum()), float(t.net.convs['5'].weight.abs().sum()), [len(a) for a in seen],
           float(rows.sum())))
    dist.destroy_process_group()


def test_ddp_gloo_world2_keeps_replicas_identical():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000
    ps = [ctx.Process(target=_ddp_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=60) for _ in ps)
    [p.join(60) for p in ps]
    assert res[0][1:3] == res[1][1:3]                      # gradient all-reduce: both replicas took the same steps
    # 42 samples in global batches of 8: 5 x 4 per rank, then 2 -> 1 per rank; the two ranks saw DIFFERENT rows
    assert res[0][3] == res[1][3] == [4, 4, 4, 4, 4, 1] and res[0][4] != res[1][4]


def _fit_data(n=43):
    import torch
    gc = torch.Generator().manual_seed(7)
    xs = torch.randint(0, 7, (n, 7, 7, 7), generator=gc).float().numpy()
    ps = torch.softmax(torch.randn(n, 294, generator=gc), dim=1).numpy()
    zs = (torch.randint(0, 2, (n,), generator=gc) * 2 - 1).numpy().astype(np.int64)
    return xs, ps, zs


def _ddp_fit_rank(rank, world, port, q, save_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from chinesecheckersagent_amd import train as T
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    xs, ps, zs = _fit_data()
    torch.manual_seed(1)
    t = T.Trainer(device='cpu', ddp=True)
    t.fit(xs, ps, zs, batch_size=8, epochs=2, validation_split=0.0, seed=5)
    state = {k: v.copy() for k, v in t.state_as_keras().items()}
    # train(): rank 0 alone writes the file, every rank returns behind the barrier with a readable path
    torch.manual_seed(1)
    path = T.train(None, xs, ps, zs, 1.0, 7, save_dir=save_dir, device='cpu', seed=5, ddp=True)
    t2 = T.Trainer(device='cpu')
    t2.load_weights(path)
    q.put((rank, state, {k: v.copy() for k, v in t2.state_as_keras().items()}, sorted(os.listdir(save_dir))))
    dist.destroy_process_group()


def test_ddp_world2_fit_equals_the_single_process_fit(tmp_path):
    """under ddp the BatchNormalization statistics are those of the GLOBAL batch and unequal shares are weighted,
    so two ranks take the step one process takes on the whole batch -- weights AND moving statistics; train(ddp=True) leaves ONE
    whole file written by rank 0."""
    import torch
    import torch.multiprocessing as mp
    from chinesecheckersagent_amd import train as T
    xs, ps, zs = _fit_data()                               # 43 rows, batches of 8: the last one has 3 rows = shares of 2 and 1
    torch.manual_seed(1)
    single = T.Trainer(device='cpu')
    single.fit(xs, ps, zs, batch_size=8, epochs=2, validation_split=0.0, seed=5)
    want = single.state_as_keras()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000
    save_dir = str(tmp_path / 'w')
    ps_ = [ctx.Process(target=_ddp_fit_rank, args=(r, 2, port, q, save_dir)) for r in range(2)]
    [p.start() for p in ps_]
    res = sorted((q.get(timeout=300) for _ in ps_), key=lambda r: r[0])
    [p.join(60) for p in ps_]
    for name, w in want.items():
        for rank, state, _, _ in res:
            scale = max(1.0, float(np.abs(w).max()))
            assert np.abs(state[name] - w).max() < 2e-5 * scale, (name, rank, np.abs(state[name] - w).max())
        assert np.array_equal(res[0][1][name], res[1][1][name]), name          # replicas bit-identical, moving statistics included
    moved = [n for n in want if 'moving_mean' in n and np.abs(want[n]).max() > 0]
    assert len(moved) == 30                                                     # the statistics did move
    # the file: one, whole, the same on both ranks, equal to rank 0's replica after the SAME fit from the same initial weights
    assert res[0][3] == res[1][3] == ['version0007-weights.h5']
    for name in want:
        assert np.array_equal(res[0][2][name], res[1][2][name])


def test_iteration_pooling_and_paths(tmp_path):
    """train.combine_prev_iters_train_data (train.py:321-352) and get_weights_path_from_version (360-361) on files
    written by utils.save_train_data: host-side only"""
    from chinesecheckersagent_amd import train as tr, utils
    rng = np.random.RandomState(3)
    d = str(tmp_path / 'data')

    def fake(n):
        return rng.rand(n, 7, 7, 7), rng.rand(n, 294), rng.randint(-1, 2, size=n)
    b0, p0, v0 = fake(6)
    b1, p1, v1 = fake(4)
    utils.save_train_data(b0, p0, v0, version=0, directory=d)
    utils.save_train_data(b1, p1, v1, version=1, directory=d)
    # iteration 1 with its own fresh samples: those + the file of iteration 0 (PAST_ITER_COUNT = 1)
    bx, py, vy, used = tr.combine_prev_iters_train_data(b1, p1, v1, 1, directory=d)
    assert used == 2 and len(bx) == 10 and np.array_equal(bx[:4], b1) and np.array_equal(bx[4:], b0)
    assert np.array_equal(py[4:], p0) and np.array_equal(vy, np.hstack([v1, v0])) and vy.dtype == np.int64
    # nothing new and no earlier file: no data
    assert tr.combine_prev_iters_train_data([], [], [], 0, directory=d) == ([], [], [], 0)
    # a missing file is skipped (train.py:334-336)
    bx, py, vy, used = tr.combine_prev_iters_train_data([], [], [], 3, directory=d)
    assert used == 0
    bx, py, vy, used = tr.combine_prev_iters_train_data([], [], [], 2, directory=d)
    assert used == 1 and np.array_equal(bx, b1)
    assert tr.get_weights_path_from_version(17) == 'saved-weights/version0017-weights.h5'
    assert min(1. / 2, tr.DEF_DATA_RETENTION_RATE) == 0.5 and tr.NUM_SELF_PLAY == 180 and tr.EVAL_GAMES == 24


def test_against_keras_step_when_present(golden_dir):
    """pins next-2 once somebody has run oracle/harness/gen_keras_train_golden.py where Keras exists: one optimisation step
    on the same batch from the same weights -- total loss and every updated tensor (BatchNorm moving statistics included)"""
    path = golden_dir + '/train_keras.npz'
    if not os.path.exists(path):
        pytest.skip('tests/golden/train_keras.npz absent: the training step stays parity-unpinned at the Keras boundary '
                    '(make it with oracle/harness/gen_keras_train_golden.py on a machine with Keras 2.1.6)')
    import torch
    from chinesecheckersagent_amd import train as T
    k = np.load(path)
    t = T.Trainer(device='cpu')
    t.load_weights(golden_dir + '/good_model.h5')
    total, policy, value, reg = t.step(torch.from_numpy(k['x'].astype(np.float32)), torch.from_numpy(k['pi'].astype(np.float32)),
                                       torch.from_numpy(k['z'].astype(np.float32)))
    assert abs(total - float(k['losses'][0])) < 1e-4 * max(1.0, abs(float(k['losses'][0])))
    st = t.state_as_keras()
    worst = 0.0
    for name in k.files:
        if not name.startswith('after/'):
            continue
        key = name[len('after/'):]
        key = key if key in st else key.split('/', 1)[-1]                # Keras names weights 'layer/kernel:0'
        assert key in st, name
        worst = max(worst, float(np.abs(st[key] - k[name]).max()))
    assert worst < 1e-5, worst


def test_step_equals_hand_derived_numpy_restatement(net, golden_dir):
    """next-2: two optimisation steps of the product's trainer (PyTorch autograd, run in float64 here) against oracle/train_oracle.py
    -- the same step derived by hand in NumPy float64 (training-mode BatchNorm, the three loss terms, every gradient, Keras' Nesterov
    update, the moving statistics): losses, all 186 tensors after each step"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import train_oracle
    from chinesecheckersagent_amd import train as T
    t = T.Trainer(device='cpu')
    t.load_weights(golden_dir + '/good_model.h5')
    t.net.double()
    rs = np.random.RandomState(5)
    x = net['planes'][:24].reshape(-1, 7, 7, 7).astype(np.float64)
    pi = rs.dirichlet(np.ones(294) * 0.1, size=24)
    z = rs.choice([-1.0, 1.0], size=24)
    w = {k: v.astype(np.float64) for k, v in t.state_as_keras().items()}
    assert len(w) == 186
    vel = {}
    for it in range(2):
        got = t.step(torch.from_numpy(x), torch.from_numpy(pi), torch.from_numpy(z))
        want, w, vel, grads = train_oracle.step(w, vel, x, pi, z)
        assert np.allclose(got, want, rtol=1e-10, atol=1e-12), (it, got, want)
        st = t.state_as_keras()
        assert set(st) == set(w)
        for k in sorted(w):
            scale = max(1e-12, float(np.abs(w[k]).max()))
            assert np.abs(st[k] - w[k]).max() < 1e-10 * max(1.0, scale), (it, k, float(np.abs(st[k] - w[k]).max()))
    # the step is not a no-op: gradients and updates are far above the comparison's resolution
    assert max(float(np.abs(g).max()) for g in grads.values()) > 1e-3
    t0 = T.Trainer(device='cpu'); t0.load_weights(golden_dir + '/good_model.h5')
    w0 = t0.state_as_keras()
    moved = [float(np.abs(w[k] - w0[k]).max()) for k in sorted(w)]
    # (the 30 convolution biases sit in front of a BatchNormalization: their gradient is zero)
    assert sum(m > 1e-9 for m in moved) == 186 - 30 and max(moved) > 1e-6
