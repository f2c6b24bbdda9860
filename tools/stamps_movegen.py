"""Diagnostic: where a workgroup of movegen_kernel spends its cycles, phase by phase (a -DMG_STAMPS build of the library, never timed for
throughput: tools/build_variant.sh mgstamps ccsp_rules_kernels.hip -DMG_STAMPS).  usage (GPU box): python tools/stamps_movegen.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['CCSP_LIB'] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'chinesecheckersagent_amd', 'libccsp_exp_mgstamps.so')
import torch
import bench
from chinesecheckersagent_amd import _lib, engine, rules

n = 1 << 22
sd0, pl0 = bench.s1_positions(1 << 16, torch, rules, _lib)
sd, player = sd0.repeat(n >> 16, 1).contiguous(), pl0.repeat(n >> 16).contiguous()
moves, count, masks = rules.movegen(sd, player)
L = C.CDLL(os.environ['CCSP_LIB'])
out = (C.c_ulonglong * 8)()
L.ccsp_debug_movegen_stamps(out, 1)
st = engine._stream_ptr()
lib = _lib.lib()
for name, fn in (('rows', lib.ccsp_movegen), ('packed', lib.ccsp_movegen_packed)):
    for _ in range(3):
        fn(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), st)
    L.ccsp_debug_movegen_stamps(out, 1)
    v = [int(x) for x in out]
    tot = sum(v[:7])
    names = ['table load', 'line patterns', 'walks + origin hops + worklist', 'search loop', 'big-stack redo', 'destination masks', 'write-out']
    print(name + ': share of a workgroup\'s lifetime (s_memtime ticks of thread 0, all workgroups, 3 launches): ' +
          ', '.join('%s %.1f %%' % (nm, 100.0 * x / tot) for nm, x in zip(names, v)) + '; %.0f ticks per workgroup of 128 positions' % (tot / (3 * n / 128)))
