"""float64 NumPy restatement of ONE optimisation step of the reference's training (row next-2 of SURVEY.md §8f):
the graph of model.py:58-145 in TRAINING mode, the losses of model.py:80-87 / loss.py:3-4, hand-derived gradients and the
optimizer of model.py:83 -- no autograd, no torch.

TEST INFRASTRUCTURE ONLY.  Parity at the Keras boundary is UNPINNED (Keras / TensorFlow are not installed and the reference
holds no training vectors): this is a second, independent derivation of the documented Keras 2.1.6 / TensorFlow 1.x semantics
which the product's PyTorch step (chinesecheckersagent_amd/train.py) is compared with in tests/test_train.py.  The way to pin
it: oracle/harness/gen_keras_train_golden.py, run where Keras exists.

Semantics restated:
  Conv2D / Dense                        cross-correlation, HWIO kernels, bias (model.py:60, 96-103, 113)
  BatchNormalization, training mode     y = gamma * (x - mean_B) / sqrt(var_B + 1e-3) + beta with the BIASED batch variance over
                                        (N, H, W); moving_mean <- 0.99 moving_mean + 0.01 mean_B; moving_variance likewise with the
                                        UNBIASED batch variance (tf.nn.fused_batch_norm, which Keras 2.1.6 uses for a 4-D input
                                        normalised over its last axis)
  loss                                  mean_N softmax_cross_entropy_with_logits(pi, logits)  (loss.py:4, weight 1)
                                        + mean_N (z - v)^2                                    ('mean_squared_error', weight 1)
                                        + 6e-3 * sum(kernel^2) over all 33 kernels            (regularizers.l2, model.py:60)
  SGD(lr 1e-4, momentum 0.9, nesterov)  v <- 0.9 v - lr g ; w <- w + 0.9 v - lr g             (keras/optimizers.py SGD.get_updates)

weights: dict 'conv2d_1/kernel:0' -> float64 ndarray in Keras layouts (as Trainer.state_as_keras gives them).
"""
import numpy as np

BN_EPS, BN_MOMENTUM = 1e-3, 0.99
REG_CONST, LR, MOMENTUM = 6e-3, 1e-4, 0.9
KERNELS = ['conv2d_%d/kernel:0' % i for i in range(1, 31)] + ['policy_head/kernel:0', 'dense_1/kernel:0', 'value_head/kernel:0']


# ---- layers with their backward passes ------------------------------------------------------------------------------

def conv_fwd(x, k, b, same):
    kh, kw = k.shape[:2]
    if same:
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        x = np.pad(x, ((0, 0), (ph, kh - 1 - ph), (pw, kw - 1 - pw), (0, 0)))
    oh, ow = x.shape[1] - kh + 1, x.shape[2] - kw + 1
    out = np.zeros((x.shape[0], oh, ow, k.shape[3]))
    for i in range(kh):
        for j in range(kw):
            out += x[:, i:i + oh, j:j + ow, :] @ k[i, j]
    return out + b, x                       # x = the (padded) input, kept for the backward pass


def conv_bwd(dout, xpad, k, same):
    kh, kw = k.shape[:2]
    oh, ow = dout.shape[1], dout.shape[2]
    dk = np.zeros_like(k)
    dxp = np.zeros_like(xpad)
    for i in range(kh):
        for j in range(kw):
            dk[i, j] = np.tensordot(xpad[:, i:i + oh, j:j + ow, :], dout, axes=([0, 1, 2], [0, 1, 2]))
            dxp[:, i:i + oh, j:j + ow, :] += dout @ k[i, j].T
    if same:
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        dxp = dxp[:, ph:ph + oh, pw:pw + ow, :]
    return dxp, dk, dout.sum(axis=(0, 1, 2))


def bn_fwd(x, gamma, beta):
    mean = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))                                   # biased
    inv = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (x - mean) * inv
    return gamma * xhat + beta, (xhat, inv, mean, var)


def bn_bwd(dy, gamma, cache):
    xhat, inv, _, _ = cache
    m = dy.shape[0] * dy.shape[1] * dy.shape[2]
    dgamma = (dy * xhat).sum(axis=(0, 1, 2))
    dbeta = dy.sum(axis=(0, 1, 2))
    dx = (gamma * inv / m) * (m * dy - dbeta - xhat * dgamma)
    return dx, dgamma, dbeta


# ---- one step ----------------------------------------------------------------------------------------------------------

def step(weights, velocity, x, pi, z):
    """-> (total, policy, value, reg), new weights, new velocity, gradients of the total loss.  `velocity`: {} at the first step."""
    w = weights
    tape = []                                                        # (kind, ...) in forward order
    grads = {}

    def cb(h, i, same=False):
        c = 'conv2d_%d' % i
        b = 'batch_normalization_%d' % i
        out, xpad = conv_fwd(h, w[c + '/kernel:0'], w[c + '/bias:0'], same)
        y, cache = bn_fwd(out, w[b + '/gamma:0'], w[b + '/beta:0'])
        return y, (i, same, xpad, cache)

    relu = lambda a: np.maximum(a, 0.0)
    n = x.shape[0]
    h0, t1 = cb(x, 1)
    a = relu(h0)
    blocks = []
    idx = 2
    for _ in range(9):
        y1, ta = cb(a, idx); r1 = relu(y1)
        y2, tb = cb(r1, idx + 1, True); r2 = relu(y2)
        y3, tc = cb(r2, idx + 2)
        s = y3 + a
        blocks.append((ta, tb, tc, y1, y2, s))
        a = relu(s)
        idx += 3
    trunk = a
    yp, tp = cb(trunk, 29); rp = relu(yp)
    flat_p = rp.reshape(n, -1)
    logits = flat_p @ w['policy_head/kernel:0'] + w['policy_head/bias:0']
    yv, tv = cb(trunk, 30); rv = relu(yv)
    flat_v = rv.reshape(n, -1)
    d1 = flat_v @ w['dense_1/kernel:0'] + w['dense_1/bias:0']
    r_d1 = relu(d1)
    v = np.tanh(r_d1 @ w['value_head/kernel:0'] + w['value_head/bias:0'])[:, 0]

    # losses
    sh = logits - logits.max(axis=1, keepdims=True)
    lsm = sh - np.log(np.exp(sh).sum(axis=1, keepdims=True))
    policy = float(-(pi * lsm).sum(axis=1).mean())
    value = float(((z - v) ** 2).mean())
    reg = float(REG_CONST * sum((w[k] ** 2).sum() for k in KERNELS))
    total = policy + value + reg

    # backward
    dlogits = (np.exp(lsm) * pi.sum(axis=1, keepdims=True) - pi) / n
    dv = 2.0 * (v - z) / n

    def back_cb(dy, tape_entry):
        i, same, xpad, cache = tape_entry
        c, b = 'conv2d_%d' % i, 'batch_normalization_%d' % i
        dconv, dg, db = bn_bwd(dy, w[b + '/gamma:0'], cache)
        grads[b + '/gamma:0'], grads[b + '/beta:0'] = dg, db
        dx, dk, dbias = conv_bwd(dconv, xpad, w[c + '/kernel:0'], same)
        grads[c + '/kernel:0'], grads[c + '/bias:0'] = dk, dbias
        return dx

    # value head
    dpre = (dv * (1.0 - v * v))[:, None]
    grads['value_head/kernel:0'] = r_d1.T @ dpre
    grads['value_head/bias:0'] = dpre.sum(axis=0)
    dd1 = (dpre @ w['value_head/kernel:0'].T) * (d1 > 0)
    grads['dense_1/kernel:0'] = flat_v.T @ dd1
    grads['dense_1/bias:0'] = dd1.sum(axis=0)
    dyv = ((dd1 @ w['dense_1/kernel:0'].T).reshape(rv.shape)) * (yv > 0)
    dtrunk = back_cb(dyv, tv)
    # policy head
    grads['policy_head/kernel:0'] = flat_p.T @ dlogits
    grads['policy_head/bias:0'] = dlogits.sum(axis=0)
    dyp = ((dlogits @ w['policy_head/kernel:0'].T).reshape(rp.shape)) * (yp > 0)
    dtrunk = dtrunk + back_cb(dyp, tp)
    # residual blocks, last to first
    da = dtrunk
    for ta, tb, tc, y1, y2, s in reversed(blocks):
        ds = da * (s > 0)
        dr2 = back_cb(ds, tc)
        dr1 = back_cb(dr2 * (y2 > 0), tb)
        da = ds + back_cb(dr1 * (y1 > 0), ta)
    back_cb(da * (h0 > 0), t1)
    for k in KERNELS:
        grads[k] = grads[k] + 2.0 * REG_CONST * w[k]

    # optimizer + moving statistics
    new_w, new_v = {}, {}
    for k, g in grads.items():
        vel = MOMENTUM * velocity.get(k, 0.0) - LR * g
        new_v[k] = vel
        new_w[k] = w[k] + MOMENTUM * vel - LR * g
    for tape_entry in [t1, tp, tv] + [t for blk in blocks for t in blk[:3]]:
        i, _, _, (xhat, inv, mean, var) = tape_entry
        b = 'batch_normalization_%d' % i
        m = xhat.shape[0] * xhat.shape[1] * xhat.shape[2]
        new_w[b + '/moving_mean:0'] = BN_MOMENTUM * w[b + '/moving_mean:0'] + (1.0 - BN_MOMENTUM) * mean
        new_w[b + '/moving_variance:0'] = BN_MOMENTUM * w[b + '/moving_variance:0'] + (1.0 - BN_MOMENTUM) * var * m / (m - 1.0)
    return (total, policy, value, reg), new_w, new_v, grads
