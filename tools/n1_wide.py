"""N1 on the widened fixture (tests/golden/net_wide.npz: 4096 self-play positions x the reference's three weight files): the fused
HIP evaluator against the float64 restatement on all 4096 x 294 x 3 logits, for one or several builds of the library, with the
kernel's time per 2048 positions beside each.

    python tools/n1_wide.py [--json out.json] [name=path/to/libccsp_variant.so ...]      (default: the product's libccsp.so)

Each build runs in its own process (CCSP_LIB).  The float64 logits are computed here by oracle/net_oracle.py and must land on the
fixture's digests (made in the build container by oracle/harness/gen_net_wide_golden.py) before they are used."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
NAMES = ('good_model', 'good_model2', 'version0016-weights')
GOLD = os.path.join(ROOT, 'tests', 'golden')


sys.path.insert(0, os.path.join(ROOT, 'tests'))
from net_wide_ref import reference_logits as _reference_logits  # noqa: E402


def reference_logits(cache='/tmp/n1_wide_ref.npz'):
    return _reference_logits(cache=cache)


def child():
    import torch
    from chinesecheckersagent_amd.model import ResidualCNN
    ref = reference_logits()
    z = np.load(os.path.join(GOLD, 'net_wide.npz'))
    x = torch.from_numpy(z['planes'].reshape(-1, 7, 7, 7).astype(np.float32)).cuda()
    doc = {}
    worst = 0.0
    for name in NAMES:
        m = ResidualCNN(device='cuda', backend='hip')
        m.load_weights(os.path.join(GOLD, name + '.h5'))
        lg, v = m.predict_batch(x)
        d = np.abs(lg.double().cpu().numpy() - ref['logits_' + name])
        dv = np.abs(v.double().cpu().numpy() - ref['v_' + name])
        doc[name] = dict(max=float(d.max()), mean=float(d.mean()), n_above_1e5=int((d >= 1e-5).sum()), logits=int(d.size),
                         p9999=float(np.quantile(d, 0.9999)), v_max=float(dv.max()))
        worst = max(worst, float(d.max()))
    doc['max_all'] = worst
    doc['n_above_1e5_all'] = sum(doc[n]['n_above_1e5'] for n in NAMES)
    m = ResidualCNN(device='cuda', backend='hip')
    m.load_weights(os.path.join(GOLD, 'good_model.h5'))
    xb = x[:2048].contiguous()
    for _ in range(20):
        m.evaluate_batch(xb)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.time()
        for _ in range(200):
            m.evaluate_batch(xb)
        torch.cuda.synchronize()
        best = min(best, (time.time() - t0) / 200 * 1e3)
    doc['ms_per_2048'] = best
    doc['frac_fp32_mfma_peak'] = 2048 * 6483264 / (best * 1e-3) / 157.3e12
    print('RESULT ' + json.dumps(doc))


def main():
    if os.environ.get('N1_WIDE_CHILD'):
        return child()
    args = [a for a in sys.argv[1:] if '=' in a]
    variants = [a.split('=', 1) for a in args] or [['product', os.path.join(ROOT, 'chinesecheckersagent_amd', 'libccsp.so')]]
    reference_logits()
    out = {}
    for name, path in variants:
        env = dict(os.environ, CCSP_LIB=os.path.abspath(path), N1_WIDE_CHILD='1')
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
        if r.returncode or not line:
            out[name] = dict(failed=r.returncode, stderr=r.stderr[-2000:])
        else:
            out[name] = json.loads(line[0][7:])
        print(name, json.dumps(out[name]), flush=True)
    if '--json' in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
