"""Development aid: movegen_kernel alone, as bench.py times it (2^23 states of SURVEY 8d's distribution), rows and packed layout.
usage (GPU box): [CCSP_LIB=...] python tools/bench_movegen.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chinesecheckersagent_amd import _lib, engine, rules

n = 1 << 23
sd0, pl0 = bench.s1_positions(1 << 16, torch, rules, _lib)
sd = sd0.repeat(n >> 16, 1).contiguous()
player = pl0.repeat(n >> 16).contiguous()
moves, count, masks = rules.movegen(sd, player)
kmean = float(count.float().mean())
L = _lib.lib()
sp_ = engine._stream_ptr()


def t(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


for name, fn in (('rows', L.ccsp_movegen), ('packed', L.ccsp_movegen_packed)):
    dt = t(lambda: fn(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), sp_))
    print('%-7s %.3f G states/s  %.0f GB/s algorithmic = %.3f of 8 TB/s' % (name, n / dt / 1e9, n * (80 + 2 * kmean) / dt / 1e9, n * (80 + 2 * kmean) / dt / 8e12))
