"""N1: where does a float32 evaluation of the policy/value net lose its distance from the float64 restatement?
CPU experiment (no GPU): the 256 fixture positions through variants of the graph that differ in ONE thing each --
which quantities are rounded to float32 and where sums are accumulated -- each compared with the float64 restatement
(tests/golden/net.npz).  Prints two tables (the second one emulates the float32 FMA chains of the MFMA layer kind by layer kind:
chain_table below); `python tools/n1_floor.py --json out.json` stores them.

    weights : 'exact'  = the file's float32 values, BatchNorm applied in float64 as its own step
              'folded' = BatchNorm folded into the convolution in float64, then rounded to float32 (what the kernel loads)
    acts    : 'f64' | 'f32' (every layer's output rounded to float32 when stored, as any float32 evaluator must)
    accum   : 'f64' (sums exact to double) | 'f32' (float32 accumulation, NumPy's order)
    dense   : accumulation of the 400 -> 294 policy dense layer alone, 'f64' | 'f32'
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from chinesecheckersagent_amd.model import read_keras_weights   # noqa: E402

EPS = 1e-3


def run(w, planes, weights='folded', acts='f32', accum='f32', dense='f32', trunk_from=0, dump=None):
    f32, f64 = np.float32, np.float64

    def g(layer, name):
        return np.asarray(w['%s/%s/%s:0' % (layer, layer, name)])

    def store(x):
        return x.astype(f32).astype(f64) if acts == 'f32' else x

    def conv(x, idx, padding, layer_no):
        k = g('conv2d_%d' % idx, 'kernel').astype(f64)
        b = g('conv2d_%d' % idx, 'bias').astype(f64)
        bn = 'batch_normalization_%d' % idx
        ga, be, mu, var = (g(bn, n).astype(f64) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
        s = ga / np.sqrt(var + EPS)
        if weights == 'folded':
            k = (k * s).astype(f32).astype(f64)
            b = ((b - mu) * s + be).astype(f32).astype(f64)
        kh, kw, _, f = k.shape
        if padding == 'same':
            x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
        n, h, ww, c = x.shape
        oh, ow = h - kh + 1, ww - kw + 1
        at = f32 if (accum == 'f32' and layer_no >= trunk_from) else f64
        out = np.zeros((n, oh, ow, f), dtype=at)
        for i in range(kh):
            for j in range(kw):
                out += np.tensordot(x[:, i:i + oh, j:j + ow, :].astype(at), k[i, j].astype(at), axes=([3], [0]))
        out = (out + b.astype(at)).astype(f64)
        if weights == 'exact':
            out = (out - mu) * s + be
        return out

    x = np.asarray(planes, dtype=f64)
    ln = 0
    x = store(np.maximum(conv(x, 1, 'valid', ln), 0)); ln += 1
    acts_out = [x]
    idx = 2
    for _ in range(9):
        y = store(np.maximum(conv(x, idx, 'valid', ln), 0)); ln += 1
        y = store(np.maximum(conv(y, idx + 1, 'same', ln), 0)); ln += 1
        y = conv(y, idx + 2, 'valid', ln); ln += 1
        x = store(np.maximum(y + x, 0))
        acts_out.append(x)
        idx += 3
    p = store(np.maximum(conv(x, 29, 'valid', ln), 0))
    p = p.reshape(len(p), -1)
    pk, pb = g('policy_head', 'kernel'), g('policy_head', 'bias')
    if dense == 'f32':
        logits = (p.astype(f32) @ pk.astype(f32) + pb.astype(f32)).astype(f64)
    else:
        logits = p @ pk.astype(f64) + pb.astype(f64)
    logits = store(logits)
    if dump is not None:
        dump.extend(acts_out)
    return logits


# ---- second table: the kernel's own accumulation order, layer kind by layer kind -------------------------------------------
# v_mfma_f32_16x16x4_f32 is bit for bit a k-ordered chain of float32 fmaf (cdna_hip_programming.md, "FP32-input MFMA"): one
# rounding per product.  chain_gemm() reproduces that: every output is a sequential float32 FMA chain over k, per segment of the
# k-range, the segments added up in order -- net_forward_kernel's structure (ccsp_net.hip): stem one chain of K = 63, first 1x1 one
# chain of 64, 3x3 four segments of 72, last 1x1 one chain of 32 then + bias + residual, policy conv one chain of 64, policy dense
# four quarters of 100.

def chain_gemm(A, W, nseg, mode):
    f32, f64 = np.float32, np.float64
    if mode == 'f64':
        return A @ W
    M, K = A.shape
    bounds = [(K * c) // nseg for c in range(nseg + 1)]
    total = None
    for c in range(nseg):
        acc = np.zeros((M, W.shape[1]), dtype=f64)
        for k in range(bounds[c], bounds[c + 1]):
            acc = (acc + A[:, k:k + 1] * W[k:k + 1, :]).astype(f32).astype(f64)     # fmaf: the product is exact in float64
        total = acc if total is None else (total + acc).astype(f32).astype(f64)
    return total


def run_chains(w, planes, cfg):
    """folded float32 weights, float32 layer outputs; cfg[layer kind] = ('f64', 1) exact sums | ('chain', segments)"""
    f32, f64 = np.float32, np.float64

    def g(layer, name):
        return np.asarray(w['%s/%s/%s:0' % (layer, layer, name)])

    def r32(x):
        return x.astype(f32).astype(f64)

    def folded(idx):
        k, b = g('conv2d_%d' % idx, 'kernel').astype(f64), g('conv2d_%d' % idx, 'bias').astype(f64)
        bn = 'batch_normalization_%d' % idx
        ga, be, mu, var = (g(bn, n).astype(f64) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
        sc = ga / np.sqrt(var + EPS)
        return r32(k * sc), r32((b - mu) * sc + be)

    def conv(x, idx, kh, pad, kind):
        k, b = folded(idx)
        if pad:
            x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
        n, h, ww, c = x.shape
        oh, ow = h - kh + 1, ww - kh + 1
        A = np.concatenate([x[:, i:i + oh, j:j + ow, :] for i in range(kh) for j in range(kh)], axis=3).reshape(n * oh * ow, kh * kh * c)
        mode, nseg = cfg[kind]
        return chain_gemm(A, k.reshape(-1, k.shape[3]), nseg, mode).reshape(n, oh, ow, -1), b
    exact_epi = cfg.get('l3', ('chain', 1))[0] == 'f64'
    x = np.asarray(planes, dtype=f64)
    o, b = conv(x, 1, 3, False, 'stem')
    x = r32(np.maximum(r32(o + b), 0))
    idx = 2
    for _ in range(9):
        o, b = conv(x, idx, 1, False, 'l1'); y = r32(np.maximum(r32(o + b), 0))
        o, b = conv(y, idx + 1, 3, True, 'l2'); y = r32(np.maximum(r32(o + b), 0))
        o, b = conv(y, idx + 2, 1, False, 'l3')
        x = r32(np.maximum(o + b + x, 0)) if exact_epi else r32(np.maximum(r32(r32(o + b) + x), 0))
        idx += 3
    o, b = conv(x, 29, 1, False, 'pc')
    p = r32(np.maximum(r32(o + b), 0)).reshape(len(x), -1)
    pk, pb = g('policy_head', 'kernel').astype(f64), g('policy_head', 'bias').astype(f64)
    mode, nseg = cfg['pf']
    lg = chain_gemm(p, pk, nseg, mode)
    return r32(lg + pb) if mode == 'f64' else r32(r32(lg) + pb)


KERNEL = dict(stem=('chain', 1), l1=('chain', 1), l2=('chain', 4), l3=('chain', 1), pc=('chain', 1), pf=('chain', 4))
EXACT = {k: ('f64', 1) for k in KERNEL}
KINDS = [('stem', 'stem 3x3 valid, K = 63'), ('l1', 'nine 1x1 64 -> 32, K = 64'), ('l2', 'nine 3x3 32 -> 32, K = 288 in four segments'),
         ('l3', 'nine 1x1 32 -> 64 + bias + residual, K = 32'), ('pc', 'policy 1x1 64 -> 16, K = 64'), ('pf', 'policy dense 400 -> 294 in four quarters')]


def _chain_case(args):
    name, cfg = args
    net = np.load(os.path.join(ROOT, 'tests', 'golden', 'net.npz'))
    w = read_keras_weights(os.path.join(ROOT, 'tests', 'golden', 'good_model.h5'))
    d = np.abs(run_chains(w, net['planes'][:256], cfg) - net['logits_good_model'][:256])
    return dict(variant=name, max=float(d.max()), mean=float(d.mean()), n_above_1e5=int((d >= 1e-5).sum()))


def chain_table():
    from multiprocessing import Pool
    cases = [('exact sums everywhere (the floor of float32 storage)', EXACT), ('float32 FMA chains everywhere, as the kernel accumulates', KERNEL)]
    for k, label in KINDS:
        c = dict(EXACT); c[k] = KERNEL[k]
        cases.append(('float32 FMA chains ONLY in: ' + label, c))
    for k, label in KINDS:
        c = dict(KERNEL); c[k] = ('f64', 1)
        cases.append(('as the kernel, but exact sums in: ' + label, c))
    segs = dict(KERNEL, stem=('chain', 5), l1=('chain', 4), l2=('chain', 9), l3=('chain', 2), pc=('chain', 4))
    cases.append(('as the kernel, every chain cut into segments of <= 16-32 k', segs))
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = pool.map(_chain_case, cases)
    for r in rows:
        print('%-90s max %.2e mean %.2e  >=1e-5: %3d' % (r['variant'], r['max'], r['mean'], r['n_above_1e5']))
    return rows


def main():
    net = np.load(os.path.join(ROOT, 'tests', 'golden', 'net.npz'))
    w = read_keras_weights(os.path.join(ROOT, 'tests', 'golden', 'good_model.h5'))
    planes = net['planes'][:256]
    want = net['logits_good_model'][:256]
    ref_acts = []
    base = run(w, planes, 'exact', 'f64', 'f64', 'f64', dump=ref_acts)
    assert np.abs(base - want).max() < 1e-10
    rows = []
    for name, kw in [
        ('float64 everything (the restatement)', dict(weights='exact', acts='f64', accum='f64', dense='f64')),
        ('only: BatchNorm folded + rounded to float32', dict(weights='folded', acts='f64', accum='f64', dense='f64')),
        ('only: layer outputs stored as float32', dict(weights='exact', acts='f32', accum='f64', dense='f64')),
        ('folded weights + float32 outputs, exact sums (floor of ANY float32-storage evaluator of folded weights)',
         dict(weights='folded', acts='f32', accum='f64', dense='f64')),
        ('the same + float32 accumulation in the policy dense layer only', dict(weights='folded', acts='f32', accum='f64', dense='f32')),
        ('the same + float32 accumulation in the trunk only', dict(weights='folded', acts='f32', accum='f32', dense='f64')),
        ('float32 everywhere (NumPy order)', dict(weights='folded', acts='f32', accum='f32', dense='f32')),
    ]:
        acts = []
        out = run(w, planes, dump=acts, **kw)
        d = np.abs(out - want)
        per_block = [float(np.abs(a - r).max()) for a, r in zip(acts, ref_acts)]
        rows.append(dict(variant=name, max=float(d.max()), mean=float(d.mean()), n_above_1e5=int((d >= 1e-5).sum()),
                         trunk_max_abs_err_after_stem_and_blocks=per_block))
        print('%-100s max %.2e mean %.2e  >=1e-5: %3d   trunk err stem..block9: %s' %
              (name, d.max(), d.mean(), (d >= 1e-5).sum(), ' '.join('%.1e' % e for e in per_block)))
    chains = chain_table() if '--no-chains' not in sys.argv else None
    if '--json' in sys.argv:
        json.dump(dict(rounding_sources=rows, mfma_chain_emulation=chains, positions=256, logits=256 * 294, tolerance=1e-5), open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
