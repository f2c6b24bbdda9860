#!/usr/bin/env python
"""The way to pin next-2 (the training step, train.py:109-146) at the Keras boundary -- companion of
gen_keras_net_golden.py.  Run once on a machine with the reference's own stack (Keras >= 2.1.6 on TensorFlow >= 1.6):

    python oracle/harness/gen_keras_train_golden.py /path/to/ChineseCheckersAgent

It imports the reference's model.py from that checkout, builds ResidualCNN() (compiled there with the reference's own
optimiser and losses, model.py:60,82-85), loads good_model.h5, runs ONE `train_on_batch` on a fixed batch of 32 samples
(planes of tests/golden/net.npz, a fixed softmax target and alternating rewards) and writes -- data only -- to
tests/golden/train_keras.npz: the batch, the losses Keras reports, and every weight tensor after the step (BatchNorm moving
statistics included).  tests/test_train.py::test_against_keras_step_when_present then holds Trainer.step to it; until that file
exists the row stays "parity unpinned" (DESIGN.md section 9)."""
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
    from model import ResidualCNN
    net = np.load(os.path.join(GOLD, 'net.npz'))
    x = net['planes'][:32].astype('float64')
    rng = np.random.RandomState(20261003)
    t = rng.randn(32, 294)
    pi = np.exp(t - t.max(1, keepdims=True))
    pi /= pi.sum(1, keepdims=True)
    z = np.array([1.0, -1.0] * 16)
    m = ResidualCNN()
    m.load_weights(os.path.join(ref, 'good_model.h5'))
    names = list(m.model.metrics_names)
    losses = m.model.train_on_batch(x, [pi, z])    # outputs in the order build_model gives them (policy_head, value_head)
    out = dict(x=net['planes'][:32], pi=pi, z=z, losses=np.asarray(losses, dtype=np.float64), loss_names=np.array(names),
               keras_version=np.array(keras.__version__))
    for layer in m.model.layers:
        for w, v in zip(layer.weights, layer.get_weights()):
            out['after/' + w.name] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLD, 'train_keras.npz'), **out)
    print('losses', dict(zip(names, losses)), '->', os.path.join(GOLD, 'train_keras.npz'))


if __name__ == '__main__':
    main()
