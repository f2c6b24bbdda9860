#!/usr/bin/env python
"""The ONE way to pin row N1 (Model.predict, model.py:21-24) at the Keras boundary.

Run it once on a machine that has the reference's own stack (README.md:14-19: Python 3, Keras >= 2.1.6 on TensorFlow
>= 1.6, h5py) -- it is NOT runnable in the build container or on the GPU box, which have neither:

    python oracle/harness/gen_keras_net_golden.py /path/to/ChineseCheckersAgent

It imports the reference's model.py from that checkout, builds ResidualCNN(), loads each weight file with the
reference's own load_weights, runs `model.model.predict` (the raw Keras call inside Model.predict, model.py:22: logits
before utils.softmax, and the value) on the 256 positions of tests/golden/net.npz, and writes the outputs -- data only --
to tests/golden/net_keras.npz.  tests/test_model.py::test_against_keras_vectors_when_present then holds the float64
restatement (oracle/net_oracle.py), the PyTorch module and the fused HIP kernel to north_star's 1e-5 against them;
until that file exists the row stays "parity unpinned" (DESIGN.md section 5)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, '..', '..', 'tests', 'golden')


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    ref = os.path.abspath(sys.argv[1])
    sys.path.insert(0, ref)
    os.chdir(ref)
    import keras                                   # fails here if the stack is missing: that is the point
    from model import ResidualCNN                  # the reference's class, imported where it lies
    net = np.load(os.path.join(GOLD, 'net.npz'))
    planes = net['planes'].astype('float64')       # what utils.to_model_input hands to predict (float64 7x7x7 arrays)
    out = dict(planes=net['planes'], keras_version=np.array(keras.__version__), backend=np.array(keras.backend.backend()))
    for name in ('good_model', 'good_model2', 'version0016-weights'):
        path = os.path.join(ref, name + '.h5')
        if not os.path.exists(path):
            continue
        m = ResidualCNN()
        m.load_weights(path)
        logits, v = m.model.predict(planes)        # the call inside Model.predict (model.py:22), whole batch
        one_l, one_v = m.model.predict(planes[:1])  # and batch 1, as MCTS.py:93 calls it
        out['logits_' + name] = np.asarray(logits, dtype=np.float32)
        out['v_' + name] = np.asarray(v, dtype=np.float32).reshape(-1)
        out['logits1_' + name] = np.asarray(one_l, dtype=np.float32)
        p, v0 = m.predict(planes[0])               # the public contract: float64 softmax, 0-d float32 value
        out['p0_' + name] = np.asarray(p, dtype=np.float64)
        out['v0_' + name] = np.asarray(v0, dtype=np.float32)
        print(name, 'logits', logits.min(), logits.max(), 'v', v.min(), v.max())
    np.savez_compressed(os.path.join(GOLD, 'net_keras.npz'), **out)
    print('wrote', os.path.join(GOLD, 'net_keras.npz'))


if __name__ == '__main__':
    main()
