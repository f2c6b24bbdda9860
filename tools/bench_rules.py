"""Development aid: the batched rules kernels alone on the SURVEY 8d position distribution (bench.py reports the same
figures under variants).  usage: python tools/bench_rules.py [stack cap]"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from chinesecheckersagent_amd import _lib, engine, rules

if len(sys.argv) > 1:            # extra hipcc flags (e.g. -DMG_LINES_STRIDE=36 -DMG_LIST_PAD=4): an experimental build beside the product's
    import os, subprocess
    from chinesecheckersagent_amd import build as B
    so = os.path.join('chinesecheckersagent_amd', 'libccsp_exp.so')
    subprocess.check_call(['hipcc'] + B.FLAGS + ['-shared'] + sys.argv[1:] + ['-o', so] + [os.path.join(B.CSRC, f) for f in B.SOURCES])
    _lib.LIB_PATH = so
    print('built', so, sys.argv[1:])
L = _lib.lib()
n = 1 << 23
sd0, pl0 = bench.s1_positions(1 << 16, torch, rules, _lib)
sd, player = sd0.repeat(n >> 16, 1).contiguous(), pl0.repeat(n >> 16).contiguous()
moves, count, masks = rules.movegen(sd, player)
st = engine._stream_ptr()


def t(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


k = float(count.float().mean())
dt = t(lambda: L.ccsp_movegen(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), st))
print('movegen        %.3f G states/s  %.0f GB/s (%.1f%% of HBM)  K=%.1f' % (n / dt / 1e9, n * (80 + 2 * k) / dt / 1e9, n * (80 + 2 * k) / dt / 8e10, k))
dt = t(lambda: L.ccsp_movegen(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), None, st))
print('movegen nomask %.3f G states/s' % (n / dt / 1e9))
dt = t(lambda: L.ccsp_movegen_packed(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), st))
print('movegen packed %.3f G states/s  %.0f GB/s (%.1f%% of HBM)' % (n / dt / 1e9, n * (80 + 2 * k) / dt / 1e9, n * (80 + 2 * k) / dt / 8e10))
ng = 1 << 21
best = torch.zeros((ng, _lib.GREEDY_MAX, 2), dtype=torch.uint8, device='cuda'); cnt = torch.zeros(ng, dtype=torch.uint8, device='cuda')
dt = t(lambda: L.ccsp_greedy_best(sd.data_ptr(), player.data_ptr(), ng, best.data_ptr(), cnt.data_ptr(), st))
print('greedy_best    %.3f G states/s' % (ng / dt / 1e9))
