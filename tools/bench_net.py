"""Development aid: time the fused evaluator alone on 4096 positions and check it against the float64 fixture."""
import sys, json, time
sys.path.insert(0, '.')
import numpy as np, torch
import os, subprocess
if len(sys.argv) > 1:            # extra hipcc flags (e.g. -DCCSP_NET_LDX=72 -DCCSP_NET_LDY=40): an experimental build beside the product's
    from chinesecheckersagent_amd import _lib, build as B
    so = os.path.join('chinesecheckersagent_amd', 'libccsp_exp.so')
    subprocess.check_call(['hipcc'] + B.FLAGS + ['-shared'] + sys.argv[1:] + ['-o', so] + [os.path.join(B.CSRC, f) for f in B.SOURCES])
    _lib.LIB_PATH = so
    print('built', so, sys.argv[1:])
from chinesecheckersagent_amd.model import ResidualCNN

def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3

G = 2048
from chinesecheckersagent_amd import _lib as _L
if os.environ.get('NET_SHAPE'):
    print('workgroup shape', _L.lib().ccsp_debug_net_shape(int(os.environ['NET_SHAPE'])))
net = np.load('tests/golden/net.npz')
x = torch.from_numpy(np.tile(net['planes'][:256].astype(np.float32), (G // 256, 1, 1, 1))).cuda()
mh = ResidualCNN(backend='hip'); mh.load_weights('tests/golden/good_model.h5')
lh, vh = mh.predict_batch(x)
ref = torch.from_numpy(net['logits_good_model']).cuda()
print('hip vs f64 restatement max abs', float((lh[:256].double() - ref).abs().max()))
ms = t(lambda: mh.evaluate_batch(x), 100)
print('hip fused ms', ms, 'TFLOP/s', G * 6483264 / (ms * 1e-3) / 1e12)
