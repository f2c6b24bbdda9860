"""The float64 restatement (oracle/net_oracle.py) on the widened N1 fixture (tests/golden/net_wide.npz: 4096 self-play positions x
the reference's three weight files), recomputed where the tests run and HELD TO THE FIXTURE'S DIGESTS before anything is compared
with it: per-position sum of logits, sum of |logits| and value of all 4096 positions (1e-9), full vectors of the 512-position subset
(1e-11) -- made in the build container by oracle/harness/gen_net_wide_golden.py with weights read by h5py.  Test infrastructure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
GOLD = os.path.join(ROOT, 'tests', 'golden')
NAMES = ('good_model', 'good_model2', 'version0016-weights')


def fixture():
    return np.load(os.path.join(GOLD, 'net_wide.npz'))


def reference_logits(names=NAMES, subset_only=False, cache=None):
    """{'logits_<name>': f64[n, 294], 'v_<name>': f64[n]} of the float64 restatement, n = 4096 (or the 512 of `sub`)"""
    import net_oracle
    from chinesecheckersagent_amd.model import read_keras_weights
    if cache and os.path.exists(cache):
        return dict(np.load(cache))
    z = fixture()
    sub = z['sub']
    planes = z['planes'][sub] if subset_only else z['planes']
    x = planes.reshape(-1, 7, 7, 7).astype(np.float64)
    out = {}
    for name in names:
        lg, v = net_oracle.forward(read_keras_weights(os.path.join(GOLD, name + '.h5')), x)
        if subset_only:
            assert np.abs(lg - z['logits_' + name]).max() < 1e-11 and np.abs(v - z['v_' + name]).max() < 1e-12, name
        else:
            assert np.abs(lg.sum(axis=1) - z['lsum_' + name]).max() < 1e-9, name
            assert np.abs(np.abs(lg).sum(axis=1) - z['labs_' + name]).max() < 1e-9, name
            assert np.abs(v - z['v_all_' + name]).max() < 1e-12 and np.abs(lg[sub] - z['logits_' + name]).max() < 1e-11, name
        out['logits_' + name], out['v_' + name] = lg, v
    if cache:
        np.savez(cache, **out)
    return out
