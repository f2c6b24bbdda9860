"""What does one instruction of a tree wave cost the evaluator launch it runs beside?  (tools/probe/interference_probe.hip)
For each instruction kind: 2048 one-wave workgroups looping over it at issue priority `--prio`, launched back to back on one stream, while
net_forward_kernel (2048 positions) runs back to back on another; the evaluator's time per launch against its time alone, divided by the
instructions an aggressor wave executes during one evaluator launch.   python tools/interference_probe.py [--prio 2]"""
import argparse, ctypes as C, subprocess, sys
sys.path.insert(0, '.')
import torch
from chinesecheckersagent_amd.model import ResidualCNN
ap = argparse.ArgumentParser(); ap.add_argument('--prio', type=int, default=2); a = ap.parse_args()
so = '/tmp/libinterference.so'
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, 'tools/probe/interference_probe.hip'])
L = C.CDLL(so)
m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
x = (torch.rand((2048, 343), device='cuda') < 0.1).float()
n_ring = 1 << 28
ring = torch.randint(0, n_ring, (n_ring,), device='cuda', dtype=torch.int32)
out = torch.zeros(2048 * 64, dtype=torch.int32, device='cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def launch(kind, steps):
    rc = L.probe_launch(kind, C.c_void_p(ring.data_ptr()), n_ring - 1, steps, a.prio, C.c_void_p(out.data_ptr()), 2048, C.c_void_p(s2.cuda_stream))
    assert rc == 0, rc
def agg_alone_us(kind, steps):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s2):
            e0.record(); launch(kind, steps); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[2]
def net_us(kind=None, steps=0, n_agg=0):
    """median time per evaluator launch (12 back to back), with n_agg aggressor launches queued on the other stream first"""
    ts, tagg = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if kind is not None:
            with torch.cuda.stream(s2):
                a0.record()
                for _ in range(n_agg):
                    launch(kind, steps)
                a1.record()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s1):
            if kind is not None:
                torch.cuda._sleep(100000)          # the aggressors get going first
            e0.record()
            for _ in range(12):
                m.evaluate_batch(x)
            e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 12)
        if kind is not None:
            tagg.append(a0.elapsed_time(a1) * 1e3 / n_agg)
    return sorted(ts)[2], (sorted(tagg)[2] if tagg else 0.0)
for _ in range(3):
    net_us()
base, _ = net_us()
print('net_forward_kernel alone: %.1f us per launch; aggressors: 2048 one-wave workgroups at priority %d' % (base, a.prio), flush=True)
kinds = [(0, 'v_add_u32 (vector integer)', 16), (1, 'v_fma_f64 (vector f64)', 16), (2, 's_add_u32 (scalar)', 16), (3, 'dependent LDS read', 16),
         (4, 'gather load, 64 lines per instruction', 1), (5, 'uniform load, one line per instruction', 1), (6, '256-byte store', 1), (7, 's_sleep 16 (resident, idle)', 1)]
for kind, name, per_step in kinds:
    steps = 8
    while agg_alone_us(kind, steps) < 60.0 and steps < (1 << 20):      # a launch of ~100 us alone
        steps *= 2
    t_net, t_agg = net_us(kind, steps, 24)            # 24 aggressor launches cover the 12 evaluator launches
    insts_per_net_launch = steps * per_step * t_net / max(t_agg, 1e-9)
    slow = t_net / base - 1.0
    print('%-40s %7d per wave and launch of %6.1f us beside: evaluator %.1f us (+%.1f %%) = %.3f %% per 100 instructions per wave and evaluator launch'
          % (name, steps * per_step, t_agg, t_net, 100 * slow, 100 * slow / max(insts_per_net_launch / 100.0, 1e-9)), flush=True)

# resident, idle waves: does the evaluator pay per wave, per workgroup or per launch?
def sleep_case(blocks, threads, steps, n_agg, prio):
    global launch
    def l2(kind, st):
        rc = L.probe_sleep(blocks, threads, st, prio, C.c_void_p(out.data_ptr()), C.c_void_p(s2.cuda_stream)); assert rc == 0
    old = launch
    launch = l2
    try:
        t_alone = agg_alone_us(7, steps)
        t_net, t_agg = net_us(7, steps, n_agg)
    finally:
        launch = old
    print('idle waves: %5d workgroups x %4d threads, priority %d, %3d launches of %6.1f us (%6.1f us beside): evaluator %.1f us (+%.1f %%)'
          % (blocks, threads, prio, n_agg, t_alone, t_agg, t_net, 100 * (t_net / base - 1)), flush=True)
for blocks, threads, steps, n_agg, prio in ((2048, 64, 256, 24, 2), (2048, 64, 256, 24, 0), (512, 256, 256, 24, 2), (256, 512, 256, 24, 2), (1024, 64, 256, 24, 2),
                                            (256, 64, 256, 24, 2), (2048, 64, 1024, 6, 2), (2048, 64, 32, 192, 2)):
    sleep_case(blocks, threads, steps, n_agg, prio)
