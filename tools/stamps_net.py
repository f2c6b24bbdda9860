"""Diagnostic: where workgroup 0 of net_forward_kernel spends its time (s_memtime at the layer boundaries; a
SEPARATE build, never timed for throughput).  Usage on the GPU box: python tools/stamps_net.py"""
import os, subprocess, sys, ctypes as C
sys.path.insert(0, '.')
here = 'chinesecheckersagent_amd'
so = os.path.join(here, 'libccsp_stamps.so')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-std=c++17', '-shared', '-DCCSP_STAMPS'] + sys.argv[1:] +
                      ['-o', so] + [os.path.join(here, 'csrc', f) for f in ('ccsp_rules_kernels.hip', 'ccsp_engine.hip', 'ccsp_net.hip', 'ccsp_host.hip')])
import torch, numpy as np
from chinesecheckersagent_amd import _lib
_lib.LIB_PATH = so
from chinesecheckersagent_amd.model import ResidualCNN
net = np.load('tests/golden/net.npz')
x = torch.from_numpy(np.tile(net['planes'][:256].astype(np.float32), (16, 1, 1, 1))).cuda()
if os.environ.get('NET_SHAPE'):
    print('workgroup shape', _lib.lib().ccsp_debug_net_shape(int(os.environ['NET_SHAPE'])))
m = ResidualCNN(backend='hip'); m.load_weights('tests/golden/good_model.h5')
for _ in range(3):
    m.evaluate_batch(x)
torch.cuda.synchronize()
out = (C.c_ulonglong * 128)()
out[63] = 0x5eed
L = _lib.lib()
L.ccsp_debug_net_stamps.restype = C.c_int
assert L.ccsp_debug_net_stamps(out) == 0
t = [int(v) for v in out[:32]]
names = ['stem'] + ['b%d.%s' % (b, n) for b in range(9) for n in ('1x1a', '3x3', '1x1b')] + ['pol conv+val1', 'pol dense+val2', 'softmax']
tot = t[31] - t[0]
agg = {}
for i, n in enumerate(names):
    d = t[i + 1] - t[i]
    key = n.split('.')[-1] if n.startswith('b') else n
    agg[key] = agg.get(key, 0) + d
print('total ticks (after input load) %d; input planes -> LDS before that: %d ticks' % (tot, t[0] - int(out[62])))
for k, v in agg.items():
    print('  %-16s %7d  %5.1f%%' % (k, v, 100.0 * v / tot))

t64 = [int(v) for v in out]
g = [t64[32 + b] - t64[2 + 3 * b] for b in range(9)]           # 3x3: from the barrier after 1x1a (+ its reduce) to wave 0's end of GEMM
w = [t64[3 + 3 * b] - t64[32 + b] for b in range(9)]           # wave 0 waiting at the layer barrier
print('3x3 per block: wave-0 reduce+gemm+epilogue %s' % g)
print('3x3 per block: wave-0 barrier wait          %s' % w)

print('block 4, 3x3: end of GEMM per wave (ticks after the layer started):', [t64[44 + w] - t64[2 + 3 * 4] for w in range(8)])
print('HW_ID simd per wave:', [(t64[52 + w] >> 4) & 3 for w in range(8)], 'wave slot:', [t64[52 + w] & 15 for w in range(8)])

rt = t64[61] - t64[60]
print('in-kernel clock of workgroup 0: %d shader ticks / %d realtime ticks (100 MHz) = %.3f GHz' % (t64[31] - t64[0], rt, (t64[31] - t64[0]) / max(rt, 1) * 0.1))

# round 5: inside block 4's last 1x1 layer, per wave: [0] layer start, [1] partial sums of the k-split tile summed (the waves that own it),
# [2] last MFMA issued, [3] epilogue done (stores issued), [4] past the barrier -- ticks after wave 0's layer start
s2 = [[int(out[64 + w * 8 + k]) for k in range(5)] for w in range(8)]
t00 = min(r[0] for r in s2)
print('last 1x1 of block 4, per wave (ticks after the first wave entered the layer): start / sums / MFMAs issued / epilogue / past barrier')
for w in range(8):
    print('  wave %d: %s' % (w, ' '.join('%6d' % (x - t00) for x in s2[w])))
