"""Development aid: how much slower does a one-wave workgroup of a given kind run BESIDE net_forward_kernel than alone?
2048 workgroups each (a half-batch's tree kernel): dependent global loads over a 1-GB ring (one round trip per step), a dependent
f64 multiply-add chain, dependent LDS reads, a scalar chain.  Each kind alone, and launched on a second stream while the evaluator runs
back to back on the first.   python tools/contention_probe.py   (GPU box; compiles tools/probe/contention_probe.hip with hipcc)"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, '.')
import torch
from chinesecheckersagent_amd.model import ResidualCNN
so = '/tmp/libcontention.so'
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, 'tools/probe/contention_probe.hip'])
L = C.CDLL(so)
m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
x = torch.rand((2048, 343), device='cuda')
n_ring = 1 << 28                                   # 1 GB of uint32: no cache holds it
ring = torch.randperm(n_ring, device='cuda', dtype=torch.int32)
out = torch.zeros(2048 * 64, dtype=torch.float64, device='cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
B = 2048
kinds = {
    'global chase, 24 dependent loads': lambda st: L.probe_chase(C.c_void_p(ring.data_ptr()), n_ring, 24, C.c_void_p(out.data_ptr()), B, C.c_void_p(st.cuda_stream)),
    'f64 chain, 4000 dependent fma': lambda st: L.probe_valu(1000, C.c_void_p(out.data_ptr()), B, C.c_void_p(st.cuda_stream)),
    'LDS chain, 2000 dependent reads': lambda st: L.probe_lds(2000, C.c_void_p(out.data_ptr()), B, C.c_void_p(st.cuda_stream)),
    'scalar chain, 8000 steps': lambda st: L.probe_salu(8000, C.c_void_p(out.data_ptr()), B, C.c_void_p(st.cuda_stream)),
}
def timed(fn, beside):
    best = []
    for _ in range(7):
        torch.cuda.synchronize()
        if beside:
            with torch.cuda.stream(s1):
                for _ in range(12):
                    m.evaluate_batch(x)           # ~1.4 ms of evaluator launches on stream 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s2):
            if beside:
                torch.cuda._sleep(200000)         # let the evaluator launches get going first
            e0.record(); fn(s2); e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3)
    best.sort()
    return best[len(best) // 2]
for name, fn in kinds.items():
    a, b = timed(fn, False), timed(fn, True)
    print('%-36s alone %7.1f us   beside net_forward_kernel %7.1f us   x %.2f' % (name, a, b, b / a), flush=True)
# and the evaluator itself beside each kind
def net_time(fn):
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        if fn is not None:
            with torch.cuda.stream(s2):
                for _ in range(6):
                    fn(s2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s1):
            e0.record()
            for _ in range(4):
                m.evaluate_batch(x)
            e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 4)
    ts.sort()
    return ts[len(ts) // 2]
print('net_forward_kernel alone %.1f us' % net_time(None))
for name, fn in kinds.items():
    print('net_forward_kernel beside [%s] %.1f us' % (name, net_time(fn)), flush=True)
