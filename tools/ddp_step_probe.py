"""How long does ONE optimisation step of the N-rank fit (train.Trainer(ddp=True): DistributedDataParallel + one all-reduce per BatchNorm
layer and pass) take when N rank processes share ONE device and gloo carries the collectives -- the only way a 1-GPU box can rehearse
BASELINE config 5's N-rank loop?  (On N GPUs the collectives are RCCL kernels on the device; here every one of the ~62 per step is staged
through the host.)   parent: python tools/ddp_step_probe.py <ranks> [steps]      (starts the ranks; never touches the GPU itself)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if os.environ.get('DDP_PROBE_RANK') == '1':
    import numpy as np, torch
    from chinesecheckersagent_amd.launch import init_rank
    from chinesecheckersagent_amd import train as T
    rank, world, local, dist = init_rank()
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    t = T.Trainer(device='cuda:%d' % local, ddp=world > 1)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 7, (32, 7, 7, 7), generator=g).float().cuda()
    pi = torch.softmax(torch.randn(32, 294, generator=g), dim=1).cuda()
    z = (torch.randint(0, 2, (32,), generator=g) * 2 - 1).float().cuda()
    idx = torch.arange(32)[rank::world]
    ts = []
    for i in range(steps):
        torch.cuda.synchronize(); t0 = time.time()
        t.step(x[idx], pi[idx], z[idx], global_rows=32 if world > 1 else None)
        torch.cuda.synchronize(); ts.append(time.time() - t0)
        if rank == 0:
            sys.stderr.write('step %d: %.1f ms\n' % (i, ts[-1] * 1e3)); sys.stderr.flush()
    if rank == 0:
        print('%d ranks on one device (backend %s): first step %.0f ms, median of the rest %.1f ms per step' % (world, dist.get_backend() if dist else None, ts[0] * 1e3, sorted(ts[1:])[len(ts[1:]) // 2] * 1e3), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
else:
    from chinesecheckersagent_amd import launch
    n = int(sys.argv[1])
    extra = {'DDP_PROBE_RANK': '1', 'PYTHONPATH': os.path.dirname(os.path.dirname(os.path.abspath(__file__)))}
    if n > 1:
        extra['CCSP_ONE_DEVICE'] = '1'
    rc = launch.run_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n, devices=[0] * n, extra_env=extra, timeout=300)
    sys.exit(rc)
