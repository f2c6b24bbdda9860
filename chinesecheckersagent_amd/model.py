"""The policy/value evaluator of the self-play path, re-implemented on
PyTorch-ROCm behind the reference's interface:

    model = ResidualCNN(); model.load_weights('good_model.h5')       # model.py:46-48, 52-56
    p, v = model.predict(x)          # x [7,7,7] -> (p float64[294] softmaxed, v 0-d float32), model.py:21-24

plus the batched entry points the GPU tree uses (predict_batch / evaluate_batch).  The Keras 2.1.6
weights file is read by h5lite (no h5py / Keras needed); float32 values are used unchanged except
that each BatchNormalization (inference form, eps = 1e-3) is folded into the convolution in front of
it, in float64, at load time.  Graph: model.py:58-145 -- 3x3 valid stem 7->64, nine bottleneck
residual blocks [1x1->32, 3x3 same->32, 1x1->64], policy head (1x1->16, flatten NHWC 400, dense
294 logits), value head (1x1->1, flatten 25, dense 32 relu, dense 1 tanh).
"""
import numpy as np

from .config import INPUT_DIM, NUM_ACTIONS, NUM_FILTERS
from .h5lite import H5File

BN_EPS = 1e-3


def _torch():
    import torch
    return torch


def _build_net(filters):
    torch = _torch()
    nn = torch.nn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            f = filters
            self.stem = nn.Conv2d(INPUT_DIM[2], f, 3, padding=0)
            self.blocks = nn.ModuleList()
            for _ in range(9):
                self.blocks.append(nn.ModuleList([nn.Conv2d(f, 32, 1), nn.Conv2d(32, 32, 3, padding=1), nn.Conv2d(32, f, 1)]))
            self.policy_conv = nn.Conv2d(f, 16, 1)
            self.policy_fc = nn.Linear(400, NUM_ACTIONS)
            self.value_conv = nn.Conv2d(f, 1, 1)
            self.value_fc1 = nn.Linear(25, 32)
            self.value_fc2 = nn.Linear(32, 1)

        def forward(self, x):                      # x [N,7,7,7] (row, col, channel) float32
            x = x.permute(0, 3, 1, 2)
            x = torch.relu(self.stem(x))
            for a, b, c in self.blocks:
                y = torch.relu(a(x))
                y = torch.relu(b(y))
                x = torch.relu(c(y) + x)
            p = torch.relu(self.policy_conv(x)).flatten(1)      # (c, h, w) order; policy_fc rows are permuted to match
            logits = self.policy_fc(p)
            v = torch.relu(self.value_conv(x)).flatten(1)
            v = torch.tanh(self.value_fc2(torch.relu(self.value_fc1(v))))
            return logits, v[:, 0]
    return Net()


def _fold(kernel_hwio, bias, gamma, beta, mean, var):
    """conv (HWIO) followed by inference BatchNorm -> (OIHW weight, bias), folded in float64"""
    k = kernel_hwio.astype(np.float64)
    s = gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + BN_EPS)
    w = np.transpose(k * s, (3, 2, 0, 1))
    b = (bias.astype(np.float64) - mean.astype(np.float64)) * s + beta.astype(np.float64)
    return w, b                                   # float64; cast to the module's dtype when stored


class Model(object):
    def __init__(self, input_dim, filters, version=0):
        self.input_dim = input_dim
        self.filters = filters
        self.version = version


def read_keras_weights(filepath):
    """{'conv2d_1/conv2d_1/kernel:0': ndarray, ...} of a Keras weights file with CANONICAL layer names.
    Keras also reads whole-model files (weights under 'model_weights/', model.py:28-31 / get_model_weights.py) and
    matches layers by ORDER, not by name: the auto-numbered layers of a model built later in a session are called
    conv2d_31.., batch_normalization_31.., dense_2.  Here the i-th conv / batch-norm / dense layer by number becomes
    conv2d_i / batch_normalization_i / dense_i."""
    w = dict(H5File(filepath).walk())
    if not any(k.startswith('conv2d_') for k in w) and any(k.startswith('model_weights/') for k in w):
        w = {k[len('model_weights/'):]: v for k, v in w.items() if k.startswith('model_weights/')}
    rename = {}
    for prefix in ('conv2d_', 'batch_normalization_', 'dense_'):
        names = sorted({k.split('/')[0] for k in w if k.startswith(prefix) and k.split('/')[0][len(prefix):].isdigit()},
                       key=lambda n: int(n[len(prefix):]))
        for i, n in enumerate(names):
            rename[n] = '%s%d' % (prefix, i + 1)
    out = {}
    for k, v in w.items():
        parts = k.split('/')
        out['/'.join(rename.get(p, p) for p in parts)] = v
    return out


class ResidualCNN(Model):
    """Same constructor and methods as the reference class (model.py:52-56); `device` picks where the
    module lives ('cuda' when a GPU is visible)."""

    def __init__(self, input_dim=INPUT_DIM, filters=NUM_FILTERS, device=None, precision='fp32', backend='auto'):
        """precision: 'fp32' (the reference's own arithmetic type: Keras floatx) or 'fp64' (the float32
        weights evaluated in float64: agrees with the float64 restatement to ~1e-12; slower).
        backend: 'hip' = the fused fp32-MFMA kernel of libccsp.so (ccsp_net_forward; GPU + fp32 only),
        'torch' = the PyTorch module (MIOpen/rocBLAS), 'auto' = 'hip' on a GPU in fp32, else 'torch'."""
        Model.__init__(self, input_dim, filters)
        torch = _torch()
        if device is None:
            device = 'cuda' if torch.cuda.is_available() else 'cpu'
        assert precision in ('fp32', 'fp64')
        self.precision = precision
        self.dtype = torch.float32 if precision == 'fp32' else torch.float64
        self.device = torch.device(device)
        torch.backends.cudnn.allow_tf32 = False                 # plain fp32 arithmetic (no reduced-precision modes)
        torch.backends.cuda.matmul.allow_tf32 = False
        self.model = _build_net(filters).to(self.device).to(self.dtype).eval()
        self.weights_path = None
        if backend == 'auto':
            backend = 'hip' if (self.device.type == 'cuda' and precision == 'fp32') else 'torch'
        assert backend in ('hip', 'torch')
        assert backend == 'torch' or (self.device.type == 'cuda' and precision == 'fp32'), 'the fused kernel is GPU fp32 only'
        self.backend = backend
        self._packed = None                                     # device copy of the kernel's weight blob

    # ---- model.py:46-48
    def load_weights(self, filepath):
        torch = _torch()
        w = read_keras_weights(filepath)

        def g(layer, name):
            return w['%s/%s/%s:0' % (layer, layer, name)]

        def conv_bn(i):
            c, b = 'conv2d_%d' % i, 'batch_normalization_%d' % i
            return _fold(g(c, 'kernel'), g(c, 'bias'), g(b, 'gamma'), g(b, 'beta'), g(b, 'moving_mean'), g(b, 'moving_variance'))

        def put(mod, wb):
            with torch.no_grad():
                mod.weight.copy_(torch.from_numpy(np.ascontiguousarray(wb[0], dtype=np.float64)))
                mod.bias.copy_(torch.from_numpy(np.ascontiguousarray(wb[1], dtype=np.float64)))
        net = self.model
        put(net.stem, conv_bn(1))
        i = 2
        for a, b, c in net.blocks:
            put(a, conv_bn(i)); put(b, conv_bn(i + 1)); put(c, conv_bn(i + 2))
            i += 3
        put(net.policy_conv, conv_bn(29))
        put(net.value_conv, conv_bn(30))
        # Dense layers: Keras kernels are [in, out]; Flatten ran over (h, w, c) while
        # torch flattens (c, h, w): permute the 400 input rows of the policy dense layer accordingly
        pk = g('policy_head', 'kernel').reshape(5, 5, 16, NUM_ACTIONS).transpose(2, 0, 1, 3).reshape(400, NUM_ACTIONS)
        put(net.policy_fc, (pk.T, g('policy_head', 'bias')))
        put(net.value_fc1, (g('dense_1', 'kernel').T, g('dense_1', 'bias')))     # 1 channel: (h, w, c) == (c, h, w)
        put(net.value_fc2, (g('value_head', 'kernel').T, g('value_head', 'bias')))
        self.weights_path = filepath
        self._packed = None
        return self.model

    def _plain_parameters(self):
        """the module's (BatchNorm-folded) parameters in Keras order: the input of ccsp_net_pack"""
        net = self.model

        def conv(c):          # OIHW -> HWIO, flattened
            return c.weight.detach().permute(2, 3, 1, 0).reshape(-1).float().cpu().numpy(), c.bias.detach().float().cpu().numpy()
        parts = list(conv(net.stem))
        for a, b, c in net.blocks:
            parts += list(conv(a)) + list(conv(b)) + list(conv(c))
        parts += list(conv(net.policy_conv))
        # torch's policy_fc rows are in (c, h, w) order (see load_weights): back to Keras' (h, w, c)
        pk = net.policy_fc.weight.detach().t().reshape(16, 5, 5, NUM_ACTIONS).permute(1, 2, 0, 3).reshape(400, NUM_ACTIONS)
        parts += [pk.reshape(-1).float().cpu().numpy(), net.policy_fc.bias.detach().float().cpu().numpy()]
        parts += list(conv(net.value_conv))
        parts += [net.value_fc1.weight.detach().t().reshape(-1).float().cpu().numpy(), net.value_fc1.bias.detach().float().cpu().numpy()]
        parts += [net.value_fc2.weight.detach().t().reshape(-1).float().cpu().numpy(), net.value_fc2.bias.detach().float().cpu().numpy()]
        return np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float32).reshape(-1) for x in parts]))

    def _ensure_packed(self):
        if self._packed is None:
            from . import _lib
            torch = _torch()
            L = _lib.lib()
            plain = self._plain_parameters()
            assert plain.size == L.ccsp_net_plain_size(), (plain.size, L.ccsp_net_plain_size())
            packed = np.zeros(L.ccsp_net_packed_size(), dtype=np.float32)
            _lib.check(L.ccsp_net_pack(plain.ctypes.data, packed.ctypes.data), 'ccsp_net_pack')
            self._packed = torch.from_numpy(packed).to(self.device)
        return self._packed

    def _hip_forward(self, x, want_logits, want_p):
        from . import _lib
        from .engine import _stream_ptr
        torch = _torch()
        x = x.reshape(-1, 343)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        n = x.shape[0]
        logits = torch.empty((n, NUM_ACTIONS), dtype=torch.float32, device=self.device) if want_logits else None
        p = torch.empty((n, NUM_ACTIONS), dtype=torch.float64, device=self.device) if want_p else None
        v = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().ccsp_net_forward(self._ensure_packed().data_ptr(), x.data_ptr(), n,
                                               logits.data_ptr() if want_logits else None,
                                               p.data_ptr() if want_p else None, v.data_ptr(), _stream_ptr()), 'ccsp_net_forward')
        return logits, p, v

    # ---- batched entry points
    def predict_batch(self, x):
        """x [N,7,7,7] float32 tensor on self.device -> (logits [N,294] f32, v [N] f32)"""
        torch = _torch()
        if self.backend == 'hip':
            logits, _, v = self._hip_forward(x, True, False)
            return logits, v
        with torch.no_grad():
            logits, v = self.model(x.reshape(-1, *self.input_dim).to(self.dtype))
        return logits, v

    def evaluate_batch(self, x):
        """-> (p float64 [N,294] = utils.softmax(logits) (utils.py:187-192), v float32 [N])"""
        torch = _torch()
        if self.backend == 'hip':
            _, p, v = self._hip_forward(x, False, True)
            return p, v
        logits, v = self.predict_batch(x)
        return torch.softmax(logits.double(), dim=-1).contiguous(), v.float().contiguous()

    def evaluate_requests(self, req, moves, pk=None, v=None):
        """The free-running path's evaluator call (ccsp_advance / ccsp_boundary of include/ccsp.h): req [n, 64] uint8 = ccsp_request
        records, moves [n, 128] int16 = their legal moves -> (pk [n, 128] float64: the softmax entries of those moves, v [n] float32).
        backend 'hip': ONE launch of the fused kernel, which builds the planes from the 32-byte positions itself and writes the compact
        answer (ccsp_net_forward_requests); otherwise planes out (ccsp_encode_requests), the module, and a gather."""
        torch = _torch()
        n = req.shape[0]
        if pk is None:
            pk = torch.empty((n, moves.shape[1]), dtype=torch.float64, device=req.device)
        if v is None:
            v = torch.empty(n, dtype=torch.float32, device=req.device)
        if self.backend == 'hip':
            from . import _lib
            from .engine import _stream_ptr
            _lib.check(_lib.lib().ccsp_net_forward_requests(self._ensure_packed().data_ptr(), req.data_ptr(), moves.data_ptr(), n,
                                                            pk.data_ptr(), v.data_ptr(), _stream_ptr()), 'ccsp_net_forward_requests')
            return pk, v
        return evaluate_requests_with(self.evaluate_batch, req, moves, pk, v)

    # ---- model.py:21-24
    def predict(self, input_board):
        torch = _torch()
        x = torch.from_numpy(np.asarray(input_board, dtype=np.float32)[None]).to(self.device)
        p, v = self.evaluate_batch(x)
        return p[0].cpu().numpy(), np.asarray(v[0].cpu().numpy(), dtype=np.float32)


def evaluate_requests_with(evaluate_batch, req, moves, pk=None, v=None):
    """requests -> compact answers through ANY planes evaluator (`evaluate_batch(x [n,7,7,7] f32 cuda) -> (p f64 [n,294], v f32 [n])`):
    ccsp_encode_requests (utils.to_model_input of every request, zeros where nothing is asked), the evaluator, ccsp_gather_priors
    (pk[i][j] = p[i][action index of move j]).  What the PyTorch-module backend and reference-style `.predict` objects run on."""
    from . import _lib
    from .engine import _stream_ptr
    torch = _torch()
    n = req.shape[0]
    planes = torch.empty((n, 7, 7, 7), dtype=torch.float32, device=req.device)
    L = _lib.lib()
    _lib.check(L.ccsp_encode_requests(req.data_ptr(), n, planes.data_ptr(), _stream_ptr()), 'ccsp_encode_requests')
    p, vv = evaluate_batch(planes)
    p = p.contiguous()
    if pk is None:
        pk = torch.empty((n, moves.shape[1]), dtype=torch.float64, device=req.device)
    _lib.check(L.ccsp_gather_priors(req.data_ptr(), moves.data_ptr(), p.data_ptr(), n, pk.data_ptr(), _stream_ptr()), 'ccsp_gather_priors')
    if v is None:
        return pk, vv.float().contiguous()
    v.copy_(vv)
    return pk, v
