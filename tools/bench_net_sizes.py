"""Development aid: launch time of the fused evaluator for small and large batches in both workgroup shapes (HIP events, 200 launches
back to back on one stream): which shape should carry which batch size.  usage (GPU box): python tools/bench_net_sizes.py"""
import sys
sys.path.insert(0, '.')
import torch
from chinesecheckersagent_amd import _lib
from chinesecheckersagent_amd.engine import _stream_ptr
from chinesecheckersagent_amd.model import ResidualCNN

L = _lib.lib()
m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
packed = m._ensure_packed()
st = _stream_ptr()
print('%8s %12s %12s %12s %12s' % ('n', '<8,8> us', '<4,4> us', '<2,8> us', '<1,8> us'))
import os
SIZES = (1, 64, 256, 512) if os.environ.get('SMALL_ONLY') else (1, 256, 512, 1024, 2048) if os.environ.get('FEW') else (1, 4, 8, 24, 64, 128, 256, 384, 512, 768, 1024, 1280, 1536, 2048, 4096)
for n in SIZES:
    x = torch.rand((n, 343), device='cuda')
    p = torch.empty((n, 294), dtype=torch.float64, device='cuda'); v = torch.empty(n, dtype=torch.float32, device='cuda')
    row = []
    for shape in (8, 4, 2, 1):
        L.ccsp_debug_net_shape(shape)
        for _ in range(50):
            L.ccsp_net_forward(packed.data_ptr(), x.data_ptr(), n, None, p.data_ptr(), v.data_ptr(), st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            L.ccsp_net_forward(packed.data_ptr(), x.data_ptr(), n, None, p.data_ptr(), v.data_ptr(), st)
        b.record(); torch.cuda.synchronize()
        row.append(a.elapsed_time(b) / 200 * 1e3)
    print('%8d %12.1f %12.1f %12.1f %12.1f' % (n, row[0], row[1], row[2], row[3]))
L.ccsp_debug_net_shape(0)
