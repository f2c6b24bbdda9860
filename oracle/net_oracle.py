"""float64 NumPy restatement of the reference's policy/value network (row N1 of SURVEY.md §8a):
graph of model.py:58-145, Keras 2.1.6 / TensorFlow layer semantics.

TEST INFRASTRUCTURE ONLY.  Parity at the Keras boundary is UNPINNED: Keras/TensorFlow are not
installed in the build container and the reference holds no test vector for its logits, so this
restatement cannot be checked against the reference's own arithmetic.  What supports it: the shapes
chain with all 186 datasets of the .h5 files, and the policy mass lands on legal moves (0.98 at the
start position).  The product's evaluators are compared with THIS (tests/golden/net.npz,
tolerance 1e-5 on logits, BASELINE.json north_star).  The way to pin it: oracle/harness/
gen_keras_net_golden.py, run once where Keras 2.1.6 / TensorFlow 1.x exist, writes
tests/golden/net_keras.npz, which tests/test_model.py picks up when present.

forward(..., dtype=np.float32) is the same graph in float32 with BatchNormalization as its own step
in TensorFlow's inference form -- the arithmetic type Keras itself runs (floatx float32): it shows how
far ANY float32 evaluation of this 30-layer net sits from the float64 one (its distance is the same
as the product's), so that a float32-vs-float64 difference is not mistaken for a graph difference.

weights: dict 'layer/layer/name:0' -> ndarray exactly as stored by keras save_weights.
"""
import numpy as np

BN_EPS = 1e-3            # keras BatchNormalization default epsilon (model.py:63 gives none)


_DT = [np.float64]        # arithmetic type of the restatement (forward() sets it for the duration of a call)


def _w(weights, layer, name):
    return np.asarray(weights['%s/%s/%s:0' % (layer, layer, name)], dtype=_DT[0])


def conv2d(x, kernel, bias, padding):
    """x [N,H,W,C] ; kernel [kh,kw,C,F] (HWIO, cross-correlation as TensorFlow does)"""
    kh, kw, _, f = kernel.shape
    if padding == 'same':
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        x = np.pad(x, ((0, 0), (ph, kh - 1 - ph), (pw, kw - 1 - pw), (0, 0)))
    n, h, w, c = x.shape
    oh, ow = h - kh + 1, w - kw + 1
    out = np.zeros((n, oh, ow, f), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            out += np.tensordot(x[:, i:i + oh, j:j + ow, :], kernel[i, j], axes=([3], [0]))
    return out + bias


def batchnorm(x, weights, layer):
    g, b = _w(weights, layer, 'gamma'), _w(weights, layer, 'beta')
    m, v = _w(weights, layer, 'moving_mean'), _w(weights, layer, 'moving_variance')
    if _DT[0] == np.float32:
        # tf.nn.batch_normalization: inv = rsqrt(var + eps) * gamma;  x * inv + (beta - mean * inv), all float32
        inv = (np.float32(1.0) / np.sqrt(v + np.float32(BN_EPS))) * g
        return x * inv + (b - m * inv)
    return g * (x - m) / np.sqrt(v + BN_EPS) + b


def conv_bn(x, weights, idx, padding='valid'):
    x = conv2d(x, _w(weights, 'conv2d_%d' % idx, 'kernel'), _w(weights, 'conv2d_%d' % idx, 'bias'), padding)
    return batchnorm(x, weights, 'batch_normalization_%d' % idx)


def forward(weights, planes, dtype=np.float64):
    """planes [N,7,7,7] (row, col, channel) -> (logits [N,294], v [N]) in `dtype` arithmetic"""
    _DT[0] = dtype
    try:
        return _forward(weights, np.asarray(planes, dtype=dtype))
    finally:
        _DT[0] = np.float64


def _forward(weights, x):
    x = np.maximum(conv_bn(x, weights, 1, 'valid'), 0.0)                    # model.py:62-64
    idx = 2
    for _ in range(9):                                                       # model.py:66-76, 120-145
        y = np.maximum(conv_bn(x, weights, idx, 'valid'), 0.0)               # 1x1 -> 32
        y = np.maximum(conv_bn(y, weights, idx + 1, 'same'), 0.0)            # 3x3 same -> 32
        y = conv_bn(y, weights, idx + 2, 'valid')                            # 1x1 -> 64
        x = np.maximum(y + x, 0.0)
        idx += 3
    p = np.maximum(conv_bn(x, weights, 29, 'valid'), 0.0)                    # policy head, model.py:107-117
    p = p.reshape(p.shape[0], -1)                                            # Flatten: NHWC order
    logits = p @ _w(weights, 'policy_head', 'kernel') + _w(weights, 'policy_head', 'bias')
    v = np.maximum(conv_bn(x, weights, 30, 'valid'), 0.0)                    # value head, model.py:90-104
    v = v.reshape(v.shape[0], -1)
    v = np.maximum(v @ _w(weights, 'dense_1', 'kernel') + _w(weights, 'dense_1', 'bias'), 0.0)
    v = np.tanh(v @ _w(weights, 'value_head', 'kernel') + _w(weights, 'value_head', 'bias'))
    return logits, v[:, 0]


def softmax64(logits):
    """utils.softmax (utils.py:187-192)"""
    x = np.copy(logits).astype('float64')
    x -= np.max(x, axis=-1, keepdims=True)
    e = np.exp(x)
    return e / np.sum(e, axis=-1, keepdims=True)
