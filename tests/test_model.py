"""Row N1: the PyTorch evaluator against the float64 restatement's vectors (tests/golden/net.npz,
made by oracle/harness/gen_net_golden.py) and the weight loading (h5lite).  Tolerance: logits
within 1e-5 (BASELINE.json north_star); "parity unpinned" at the Keras boundary (oracle/net_oracle.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

LOGIT_TOL = 1e-5        # absolute, on logits in [-15, 23] (BASELINE.json north_star)
# DEVIATION from north_star, stated in DESIGN.md section 5 and set FROM EVIDENCE (round 4): a float32 evaluation (the reference's own
# Keras floatx) of this 30-layer net differs from a float64 one by more than 1e-5 on a few logits in 10^5 whatever computes it and in
# whatever order it adds (six segmentations of the accumulation chains measured: profiles/r4_n1_wide_variants.json).  The fused HIP
# kernel on the widened fixture -- 4096 self-play positions x the reference's three weight files = 3 612 672 logits: max 2.00e-5,
# 215 logits >= 1e-5 (6.0e-5 of all), mean 6.1e-7 / 8.5e-7 / 3.2e-7 per weight file; on the 256-position fixture (75 264 logits)
# max 1.11e-5, 4 above.  The bars: every logit within 1.25 x the measured maximum (NOT round 3's 3e-5), >= 99.99 % within 1e-5 on the wide
# fixture over all three files (>= 99.98 % for each: 99.988 % measured for good_model.h5); values within 1e-5 outright; fp64 mode 1e-9.  The float32 evaluators that are NOT the product path (the PyTorch module on
# the CPU / through MIOpen, the float32 NumPy restatement: max 2.1e-5 on the small fixture already) keep the loose cap.
FP32_CAP_WIDE, FP32_FRACTION_WIDE = 2.5e-5, 0.9998        # (per weight file: 0.99988 / 0.99995 / 0.999996 measured)
FP32_CAP_HIP_SMALL = 1.4e-5
FP32_FRACTION, FP32_CAP = 0.999, 3e-5


def _dist(a, b):
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    return {'max': float(d.max()), 'mean': float(d.mean()), 'frac_within_1e-5': float((d < LOGIT_TOL).mean()), 'n_above_1e-5': int((d >= LOGIT_TOL).sum())}


@pytest.fixture(scope='module')
def net(golden_dir):
    return np.load(golden_dir + '/net.npz')


def _check(model, net, name, n):
    import torch
    x = torch.from_numpy(net['planes'][:n].astype(np.float32)).to(model.device)
    logits, v = model.predict_batch(x)
    logits, v = logits.double().cpu().numpy(), v.double().cpu().numpy()
    d = np.abs(logits - net['logits_' + name][:n])
    if model.precision == 'fp64':
        assert d.max() < 1e-9 and np.abs(v - net['v_' + name][:n]).max() < 1e-9
    else:
        cap = FP32_CAP_HIP_SMALL if getattr(model, 'backend', '') == 'hip' else FP32_CAP
        assert (d < LOGIT_TOL).mean() >= FP32_FRACTION and d.max() < cap, (d.max(), (d < LOGIT_TOL).mean())
        assert np.abs(v - net['v_' + name][:n]).max() < LOGIT_TOL
    p, _ = model.evaluate_batch(x)
    import net_oracle
    assert np.abs(p.cpu().numpy() - net_oracle.softmax64(net['logits_' + name][:n])).max() < 1e-5
    assert np.abs(p.sum(dim=1).cpu().numpy() - 1).max() < 1e-12


def test_cpu_forward_matches_float64_restatement(net, golden_dir):
    from chinesecheckersagent_amd.model import ResidualCNN
    for precision in ('fp32', 'fp64'):
        m = ResidualCNN(device='cpu', precision=precision)
        m.load_weights(golden_dir + '/good_model.h5')
        _check(m, net, 'good_model', 256)
    # the reference's predict() contract (model.py:21-24): p float64 softmaxed, v 0-d float32
    p, v = m.predict(net['planes'][0].astype(np.float64))
    assert p.dtype == np.float64 and p.shape == (294,) and v.dtype == np.float32 and v.shape == ()
    assert abs(p.sum() - 1) < 1e-12


def test_other_weight_files(net, golden_dir):
    """the reference's other two weight files (data fixtures since round 4) through the CPU module"""
    from chinesecheckersagent_amd.model import ResidualCNN
    for name in ('good_model2', 'version0016-weights'):
        m = ResidualCNN(device='cpu')
        m.load_weights(golden_dir + '/%s.h5' % name)
        _check(m, net, name, 32)


def test_oracle_restatement_reproduces_its_fixture(net, golden_dir):
    """the float64 restatement itself, run here on weights read by h5lite (not h5py) = the fixture"""
    import net_oracle
    from chinesecheckersagent_amd.h5lite import H5File
    w = dict(H5File(golden_dir + '/good_model.h5').walk())
    assert len(w) == 186 and sum(a.size for a in w.values()) == 249852          # SURVEY.md §2 "Weights"
    logits, v = net_oracle.forward(w, net['planes'][:16].reshape(-1, 7, 7, 7))
    assert np.array_equal(logits, net['logits_good_model'][:16]) or np.abs(logits - net['logits_good_model'][:16]).max() < 1e-12
    # policy mass on legal moves at the start position (0.98 for player 2)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_ffi as orc
    pos = orc.initial_pos12()
    planes = orc.planes(pos, orc.NO_LAST, 2).reshape(1, 7, 7, 7)
    p = net_oracle.softmax64(net_oracle.forward(w, planes)[0])[0]
    legal = [int(a) * 49 + int(b) for a, b in orc.movegen(pos, 2)]
    assert p[legal].sum() > 0.9


def test_h5lite_writer_roundtrip_and_h5py(tmp_path):
    from chinesecheckersagent_amd import h5lite
    bx = np.random.RandomState(0).rand(9, 7, 7, 7)
    pi = np.random.RandomState(1).rand(9, 294)
    vy = np.array([1, -1] * 4 + [1], dtype=np.int64)
    path = str(tmp_path / 'data-for-iter-3.h5')
    h5lite.write_datasets(path, [('board_x', bx), ('pi_y', pi), ('v_y', vy)])
    back = dict(h5lite.H5File(path).walk())
    assert np.array_equal(back['board_x'], bx) and np.array_equal(back['pi_y'], pi) and np.array_equal(back['v_y'], vy)
    assert back['v_y'].dtype == np.int64 and back['board_x'].dtype == np.float64
    conda = '/opt/conda/bin/python3.9'
    if os.path.exists(conda):           # the real HDF5 library must read what we write (build container only)
        code = ("import h5py,sys; f=h5py.File(sys.argv[1],'r'); "
                "print(sorted(f.keys()), f['board_x'].shape, f['pi_y'].dtype, f['v_y'].dtype, int(f['v_y'][...].sum()))")
        out = subprocess.check_output([conda, '-c', code, path]).decode()
        assert "['board_x', 'pi_y', 'v_y'] (9, 7, 7, 7) float64 int64 1" in out


def test_h5lite_stream_writer_chunked_files(tmp_path):
    """h5lite.StreamWriter: rows appended piece by piece end up in CHUNKED datasets (a version-1 B-tree of raw data chunks per
    dataset, up to three levels here) that our reader and the real HDF5 library both read back exactly -- the file the
    self-play generator streams finished games into (selfplay.TrainDataSink(path=...))"""
    from chinesecheckersagent_amd import h5lite
    rng = np.random.RandomState(0)
    conda = '/opt/conda/bin/python3.9'
    for n, cr in ((0, 16), (5, 16), (16, 16), (1000, 16), (70000, 16), (300, 4096)):
        path = str(tmp_path / ('s_%d_%d.h5' % (n, cr)))
        w = h5lite.StreamWriter(path, [('board_x', (7, 7, 7), '<f8'), ('pi_y', (294,), '<f8'), ('v_y', (), '<i8')], chunk_rows=cr)
        bx, py, vy = rng.rand(n, 7, 7, 7), rng.rand(n, 294), rng.randint(-1, 2, size=n).astype(np.int64)
        i = 0
        while i < n:
            k = min(n - i, rng.randint(1, 50) if n < 5000 else rng.randint(1, 5000))
            w.append([bx[i:i + k], py[i:i + k], vy[i:i + k]])
            i += k
        assert not os.path.exists(path)                  # the file appears under its name when it is whole
        w.close()
        f = h5lite.H5File(path)
        assert np.array_equal(f.get('board_x'), bx) and np.array_equal(f.get('pi_y'), py) and np.array_equal(f.get('v_y'), vy)
        if os.path.exists(conda):
            np.savez(path + '.npz', bx=bx, py=py, vy=vy)
            code = ("import h5py,sys,numpy as np; f=h5py.File(sys.argv[1],'r'); z=np.load(sys.argv[1]+'.npz'); "
                    "assert f['board_x'].chunks[0]==int(sys.argv[2]); "
                    "assert np.array_equal(np.array(f.get('board_x')),z['bx']) and np.array_equal(f['pi_y'][:],z['py']) "
                    "and np.array_equal(f['v_y'][:],z['vy']) and f['v_y'].dtype==np.int64; print('same')")
            assert 'same' in subprocess.check_output([conda, '-W', 'ignore', '-c', code, path, str(cr)]).decode()


@pytest.mark.gpu
def test_gpu_forward_matches_float64_restatement(net, golden_dir):
    import torch
    from chinesecheckersagent_amd.model import ResidualCNN
    assert torch.cuda.is_available()
    for precision, backend in (('fp32', 'hip'), ('fp32', 'torch'), ('fp64', 'torch')):
        m = ResidualCNN(device='cuda', precision=precision, backend=backend)
        m.load_weights(golden_dir + '/good_model.h5')
        _check(m, net, 'good_model', 256)
    # the fused kernel on ragged batch sizes (tail workgroup) agrees with itself on the full batch
    m = ResidualCNN(device='cuda', backend='hip')
    m.load_weights(golden_dir + '/good_model.h5')
    x = torch.from_numpy(net['planes'][:256].astype(np.float32)).cuda()
    full, vfull = m.predict_batch(x)
    for n in (1, 7, 9, 250):
        part, vpart = m.predict_batch(x[:n].contiguous())
        assert torch.equal(part, full[:n]) and torch.equal(vpart, vfull[:n])


def test_load_weights_by_layer_order_and_whole_model_files(net, golden_dir, tmp_path):
    """Keras matches layers by order and also reads whole-model files: a copy of good_model.h5 with every auto-numbered
    layer renamed (conv2d_31.., batch_normalization_31.., dense_2) under 'model_weights/' loads to the same network"""
    from chinesecheckersagent_amd.h5lite import H5File, write_keras_weights
    from chinesecheckersagent_amd.model import ResidualCNN
    src = dict(H5File(golden_dir + '/good_model.h5').walk())
    layers = {}
    for k, v in src.items():
        layer, _, wname = k.split('/')
        layers.setdefault(layer, []).append((wname, np.array(v)))

    def renamed(layer):
        for pref in ('conv2d_', 'batch_normalization_'):
            if layer.startswith(pref):
                return '%s%d' % (pref, int(layer[len(pref):]) + 30)
        return 'dense_2' if layer == 'dense_1' else layer
    order = sorted(layers, key=lambda n: (n.rstrip('0123456789'), int(n[len(n.rstrip('0123456789')):] or 0)))
    out = [(renamed(l), [('%s/%s' % (renamed(l), wn), arr) for wn, arr in layers[l]]) for l in order]
    path = str(tmp_path / 'whole.h5')
    write_keras_weights(path, out, root='model_weights')
    a, b = ResidualCNN(device='cpu'), ResidualCNN(device='cpu')
    a.load_weights(golden_dir + '/good_model.h5')
    b.load_weights(path)
    for pa, pb in zip(a.model.parameters(), b.model.parameters()):
        assert (pa == pb).all()


def _fp32_restatement(golden_dir, net):
    import net_oracle
    from chinesecheckersagent_amd.model import read_keras_weights
    w = read_keras_weights(golden_dir + '/good_model.h5')
    return net_oracle.forward(w, net['planes'][:256], dtype=np.float32)


def test_float32_distances(net, golden_dir):
    """three-way distances on the 256 fixture positions (75 264 logits): float32 restatement vs float64 restatement, and
    the product's float32 evaluation (CPU module here; the HIP kernel in the -m gpu twin) vs both"""
    from chinesecheckersagent_amd.model import ResidualCNN
    import torch
    l32, v32 = _fp32_restatement(golden_dir, net)
    assert l32.dtype == np.float32
    ref64 = net['logits_good_model'][:256]
    a = _dist(l32, ref64)
    assert a['frac_within_1e-5'] >= FP32_FRACTION and a['max'] < FP32_CAP, a
    m = ResidualCNN(device='cpu')
    m.load_weights(golden_dir + '/good_model.h5')
    lp, vp = m.predict_batch(torch.from_numpy(net['planes'][:256].astype(np.float32)))
    b, c = _dist(lp.numpy(), ref64), _dist(lp.numpy(), l32)
    assert b['max'] < FP32_CAP and c['max'] < FP32_CAP and b['mean'] < 2e-6 and c['mean'] < 2e-6, (b, c)
    assert b['max'] <= 1.5 * a['max']            # the product is no further from float64 than float32 arithmetic itself is


@pytest.mark.gpu
def test_gpu_evaluation_is_a_function_of_the_position_alone(net, golden_dir):
    """an evaluation must not depend on the batch size, on the position's slot in the batch, or on the workgroup's row tile
    the position's rows fall into (the k-split tile included): the same 3 positions at every slot of batches of several
    sizes, for both workgroup shapes of the fused kernel -- every copy bit-identical"""
    import torch
    from chinesecheckersagent_amd import _lib
    from chinesecheckersagent_amd.model import ResidualCNN
    L = _lib.lib()
    m = ResidualCNN(device='cuda', backend='hip')
    m.load_weights(golden_dir + '/good_model.h5')
    base = torch.from_numpy(net['planes'][:3].astype(np.float32)).cuda()
    ref64 = net['logits_good_model'][:3]
    try:
        want = None                                    # ONE expectation for every shape: the shapes agree bit for bit, too
        for shape in (8, 4, 2, 1, 0):                  # 0 = chosen by batch size: <1,8> up to 256 positions, <2,8> up to 512, <4,4> up to 1024, <8,8> beyond
            assert L.ccsp_debug_net_shape(shape) == shape
            for n in (1, 3, 16, 19, 51) + ((256, 257, 512, 515, 1024, 1027) if shape == 0 else ()):
                idx = torch.arange(n, device='cuda') % 3
                lg, v = m.predict_batch(base[idx].contiguous())
                p, _ = m.evaluate_batch(base[idx].contiguous())
                for k in range(3):
                    rows = (idx == k).nonzero().flatten()
                    if len(rows) == 0:
                        continue
                    if want is None:
                        want = [None, None, None]
                    if want[k] is None:
                        want[k] = (lg[rows[0]].clone(), v[rows[0]].clone(), p[rows[0]].clone())
                    for r in rows:
                        assert torch.equal(lg[r], want[k][0]) and torch.equal(v[r], want[k][1]) and torch.equal(p[r], want[k][2]), (shape, n, int(r))
            d = np.abs(torch.stack([w[0] for w in want]).double().cpu().numpy() - ref64)
            assert d.max() < FP32_CAP_HIP_SMALL
    finally:
        L.ccsp_debug_net_shape(0)


@pytest.mark.gpu
def test_gpu_float32_distances(net, golden_dir):
    """the fused HIP kernel against the float64 AND the float32 restatements"""
    import json
    import torch
    from chinesecheckersagent_amd.model import ResidualCNN
    l32, v32 = _fp32_restatement(golden_dir, net)
    ref64 = net['logits_good_model'][:256]
    m = ResidualCNN(device='cuda', backend='hip')
    m.load_weights(golden_dir + '/good_model.h5')
    lh, vh = m.predict_batch(torch.from_numpy(net['planes'][:256].astype(np.float32)).cuda())
    lh, vh = lh.cpu().numpy(), vh.cpu().numpy()
    doc = {'hip_fp32_vs_fp64_restatement': _dist(lh, ref64), 'hip_fp32_vs_fp32_restatement': _dist(lh, l32),
           'fp32_restatement_vs_fp64_restatement': _dist(l32, ref64),
           'v_hip_vs_fp64': _dist(vh, net['v_good_model'][:256]), 'north_star_tolerance': LOGIT_TOL, 'logits': int(lh.size)}
    print('N1 small fixture: ' + json.dumps(doc))          # (tools/n1_wide.py writes the numbers kept under profiles/)
    k = 'hip_fp32_vs_fp64_restatement'
    assert doc[k]['frac_within_1e-5'] >= FP32_FRACTION and doc[k]['max'] < FP32_CAP_HIP_SMALL and doc[k]['mean'] < 1e-6, doc
    k = 'hip_fp32_vs_fp32_restatement'                     # two float32 evaluations of different order: the loose cap
    assert doc[k]['frac_within_1e-5'] >= FP32_FRACTION and doc[k]['max'] < FP32_CAP and doc[k]['mean'] < 2e-6, doc
    assert doc['v_hip_vs_fp64']['max'] < LOGIT_TOL
    assert doc['hip_fp32_vs_fp64_restatement']['max'] <= 1.5 * doc['fp32_restatement_vs_fp64_restatement']['max']


def test_against_keras_vectors_when_present(net, golden_dir):
    """pins row N1 once somebody has run oracle/harness/gen_keras_net_golden.py where Keras exists"""
    path = golden_dir + '/net_keras.npz'
    if not os.path.exists(path):
        pytest.skip('tests/golden/net_keras.npz absent: row N1 stays parity-unpinned at the Keras boundary '
                    '(make it with oracle/harness/gen_keras_net_golden.py on a machine with Keras 2.1.6)')
    import torch
    from chinesecheckersagent_amd.model import ResidualCNN
    k = np.load(path)
    assert np.array_equal(k['planes'], net['planes'])
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    m = ResidualCNN(device=dev)
    m.load_weights(golden_dir + '/good_model.h5')
    lp, vp = m.predict_batch(torch.from_numpy(net['planes'][:256].astype(np.float32)).to(dev))
    d = _dist(lp.cpu().numpy(), k['logits_good_model'][:256])
    assert d['frac_within_1e-5'] >= FP32_FRACTION and d['max'] < FP32_CAP, d
    assert _dist(net['logits_good_model'][:256], k['logits_good_model'][:256])['max'] < FP32_CAP     # the float64 oracle itself
    assert np.abs(vp.cpu().numpy() - k['v_good_model'][:256]).max() < LOGIT_TOL


# ---- round 4: the widened fixture (tests/golden/net_wide.npz) ------------------------------------------------------------------

def _unfolded_torch_float64(weights, planes):
    """THIRD derivation of the graph, sharing no code with oracle/net_oracle.py or with the product's BatchNorm folding:
    torch.nn.functional.conv2d (cross-correlation, NCHW / OIHW) + F.batch_norm(eps=1e-3) as its own step, float64, CPU"""
    import torch
    import torch.nn.functional as F

    def t(layer, name):
        return torch.from_numpy(np.asarray(weights['%s/%s/%s:0' % (layer, layer, name)], dtype=np.float64))

    def conv_bn(x, i, pad):
        y = F.conv2d(x, t('conv2d_%d' % i, 'kernel').permute(3, 2, 0, 1).contiguous(), t('conv2d_%d' % i, 'bias'), padding=pad)
        bn = 'batch_normalization_%d' % i
        return F.batch_norm(y, t(bn, 'moving_mean'), t(bn, 'moving_variance'), t(bn, 'gamma'), t(bn, 'beta'), training=False, eps=1e-3)
    x = torch.from_numpy(np.asarray(planes, dtype=np.float64)).permute(0, 3, 1, 2).contiguous()
    x = F.relu(conv_bn(x, 1, 0))
    i = 2
    for _ in range(9):
        y = F.relu(conv_bn(x, i, 0))
        y = F.relu(conv_bn(y, i + 1, 1))
        x = F.relu(conv_bn(y, i + 2, 0) + x)
        i += 3
    p = F.relu(conv_bn(x, 29, 0)).permute(0, 2, 3, 1).reshape(len(x), -1)            # Keras Flatten of an NHWC tensor
    logits = p @ t('policy_head', 'kernel') + t('policy_head', 'bias')
    v = F.relu(conv_bn(x, 30, 0)).permute(0, 2, 3, 1).reshape(len(x), -1)
    v = torch.tanh(F.relu(v @ t('dense_1', 'kernel') + t('dense_1', 'bias')) @ t('value_head', 'kernel') + t('value_head', 'bias'))
    return logits.numpy(), v[:, 0].numpy()


def test_wide_fixture_three_derivations_agree(golden_dir):
    """the float64 restatement recomputed here (weights through h5lite) lands on the fixture (made with h5py in the build container) on
    the 512-position subset for all three weight files, and the unfolded torch derivation lands on it too"""
    import net_wide_ref
    from chinesecheckersagent_amd.model import read_keras_weights
    z = net_wide_ref.fixture()
    assert z['planes'].shape == (4096, 343) and len(np.unique(z['planes'], axis=0)) == 4096
    assert set(np.unique(z['player'])) == {1, 2} and z['ply'].max() > 100 and set(np.unique(z['kind'])) == {0, 1, 2}
    import oracle_ffi as orc
    for i in (0, 1, 777, 4095):                                # the stored planes are the planes of the stored positions
        assert np.array_equal(orc.planes(z['pos12'][i], z['last'][i], int(z['player'][i])), z['planes'][i])
    ref = net_wide_ref.reference_logits(subset_only=True)       # (asserts 1e-11 against the fixture's vectors)
    x = z['planes'][z['sub']].reshape(-1, 7, 7, 7)
    for name in net_wide_ref.NAMES:
        lg, v = _unfolded_torch_float64(read_keras_weights(golden_dir + '/%s.h5' % name), x)
        assert np.abs(lg - z['logits_' + name]).max() < 1e-9 and np.abs(v - z['v_' + name]).max() < 1e-10, name
        assert np.abs(lg - ref['logits_' + name]).max() < 1e-9


def test_policy_mass_on_legal_moves_whole_fixture():
    """semantic check of the graph reading over ALL 4096 positions and all three weight files: the softmax mass on the position's
    legal moves (fixture: legal_mass_<name>, computed by the restatement) is several times what an unrelated reading gives --
    a wrong flatten order, a transposed board or a wrong action codec all land on chance = (legal moves) / 294"""
    import net_oracle
    import net_wide_ref
    import oracle_ffi as orc
    from chinesecheckersagent_amd.model import read_keras_weights
    z = net_wide_ref.fixture()
    sub = z['sub']
    legal = [np.array([int(a) * 49 + int(b) for a, b in orc.movegen(z['pos12'][i], int(z['player'][i]))]) for i in sub]
    chance = float(np.mean([len(l) for l in legal])) / 294.0
    assert 0.08 < chance < 0.2
    for name, floor in (('good_model', 0.33), ('good_model2', 0.25), ('version0016-weights', 0.32)):
        mass = z['legal_mass_' + name]
        assert mass.shape == (4096,) and mass.mean() > floor and mass.mean() > 2.2 * chance, (name, mass.mean(), chance)
        # the stored masses are those of the stored logits (subset), with the legal moves generated here
        p = net_oracle.softmax64(z['logits_' + name])
        here = np.array([p[k, legal[k]].sum() for k in range(len(sub))])
        assert np.abs(here - mass[sub]).max() < 1e-12
        # an unrelated reading of the same numbers: action index read as id * 49 + col * 7 + row -> chance
        pt = p.reshape(-1, 6, 7, 7).transpose(0, 1, 3, 2).reshape(-1, 294)
        wrong = np.array([pt[k, legal[k]].sum() for k in range(len(sub))])
        assert wrong.mean() < 1.5 * chance < here.mean()
    # (c, h, w) flatten order in front of the policy dense layer instead of Keras' (h, w, c): chance
    w = read_keras_weights(os.path.join(net_wide_ref.GOLD, 'good_model.h5'))
    x = z['planes'][sub[:128]].reshape(-1, 7, 7, 7).astype(np.float64)
    net_oracle._DT[0] = np.float64
    t = np.maximum(net_oracle.conv_bn(x, w, 1, 'valid'), 0.0)
    i = 2
    for _ in range(9):
        y = np.maximum(net_oracle.conv_bn(t, w, i, 'valid'), 0.0)
        y = np.maximum(net_oracle.conv_bn(y, w, i + 1, 'same'), 0.0)
        t = np.maximum(net_oracle.conv_bn(y, w, i + 2, 'valid') + t, 0.0)
        i += 3
    pc = np.maximum(net_oracle.conv_bn(t, w, 29, 'valid'), 0.0)
    K, b = net_oracle._w(w, 'policy_head', 'kernel'), net_oracle._w(w, 'policy_head', 'bias')
    good = net_oracle.softmax64(pc.reshape(len(pc), -1) @ K + b)
    bad = net_oracle.softmax64(pc.transpose(0, 3, 1, 2).reshape(len(pc), -1) @ K + b)
    m_good = np.mean([good[k, legal[k]].sum() for k in range(128)])
    m_bad = np.mean([bad[k, legal[k]].sum() for k in range(128)])
    assert m_bad < 1.6 * chance and m_good > 2.2 * chance, (m_good, m_bad, chance)


@pytest.mark.gpu
def test_gpu_all_three_weight_files_on_the_wide_fixture(golden_dir):
    """the fused HIP kernel on 4096 self-play positions x the reference's three weight files = 3 612 672 logits against the float64
    restatement (recomputed here, held to the fixture's digests): the bars are set from the measured distances (header)"""
    import torch
    import net_wide_ref
    from chinesecheckersagent_amd.model import ResidualCNN
    z = net_wide_ref.fixture()
    ref = net_wide_ref.reference_logits(cache='/tmp/n1_wide_ref.npz')
    x = torch.from_numpy(z['planes'].reshape(-1, 7, 7, 7).astype(np.float32)).cuda()
    above = total = 0
    for name in net_wide_ref.NAMES:
        m = ResidualCNN(device='cuda', backend='hip')
        m.load_weights(golden_dir + '/%s.h5' % name)
        assert m.backend == 'hip'
        lg, v = m.predict_batch(x)
        d = np.abs(lg.double().cpu().numpy() - ref['logits_' + name])
        assert d.max() < FP32_CAP_WIDE and (d < LOGIT_TOL).mean() >= FP32_FRACTION_WIDE and d.mean() < 1e-6, (name, d.max(), (d < LOGIT_TOL).mean())
        assert np.abs(v.double().cpu().numpy() - ref['v_' + name]).max() < LOGIT_TOL, name
        above, total = above + int((d >= LOGIT_TOL).sum()), total + d.size
        p, _ = m.evaluate_batch(x)
        assert np.abs(p.cpu().numpy() - net_oracle_softmax(ref['logits_' + name])).max() < 1e-5
        # fp64 mode (PyTorch, float64) meets 1e-9 on the same positions
        if name == 'good_model2':
            m64 = ResidualCNN(device='cuda', precision='fp64', backend='torch')
            m64.load_weights(golden_dir + '/%s.h5' % name)
            l64, _ = m64.predict_batch(x[:1024])
            assert np.abs(l64.cpu().numpy() - ref['logits_' + name][:1024]).max() < 1e-9
    assert above / total < 1e-4                      # >= 99.99 % of all 3 612 672 logits within 1e-5 (99.994 % measured)


def net_oracle_softmax(logits):
    import net_oracle
    return net_oracle.softmax64(logits)
