"""One process per GPU, started from a process that never touches the GPU itself.

The reference's multi-worker entry is a multiprocessing.Pool of whole-game workers (train.py:71-105,
evaluate_models.py:57-102); here a "worker" is a rank process that owns one MI355X and plays its share of the game
ids.  The parent only spawns, waits and reads the files the ranks leave behind -- it makes no torch.cuda / libccsp call
(never re-exec or fork a process that has initialised the GPU), passes MASTER_ADDR=127.0.0.1 and a free port, and
terminates the other ranks when one fails (they would wait in a collective for ever).
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, device=None, extra=None):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank if device is None else device), WORLD_SIZE=str(world),
               LOCAL_WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    if extra:
        env.update(extra)
    return env


KILL_GRACE_S = 15.0      # seconds between SIGTERM and SIGKILL for ranks that do not leave (wedged in a kernel or a collective)


def run_ranks(argv, world, devices=None, extra_env=None, poll=0.2, timeout=None, kill_grace=None):
    """start `world` copies of `argv` (a full command line) as ranks 0..world-1; devices[r] = the HIP device of rank r
    (default r; [0, 0] puts two ranks on one device for functional tests).  Returns 0, or the first failing rank's exit
    code after terminating the rest (124 when `timeout` seconds passed first); ranks that ignore the SIGTERM are killed
    `kill_grace` seconds later, so the caller always gets an answer."""
    port = free_port()
    procs = []
    for r in range(world):
        dev = r if devices is None else devices[r]
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port, dev, extra_env)))
    rc = 0
    alive = list(procs)
    t0 = time.time()
    t_kill = None
    grace = KILL_GRACE_S if kill_grace is None else float(kill_grace)
    while alive:
        time.sleep(poll)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:
                    q.terminate()
                t_kill = time.time() + grace
        if timeout is not None and alive and rc == 0 and time.time() - t0 > timeout:
            rc = 124
            for q in alive:
                q.terminate()
            t_kill = time.time() + grace
        if t_kill is not None and alive and time.time() > t_kill:
            for q in alive:
                q.kill()
            t_kill = None
    if rc:
        sys.stderr.write('chinesecheckersagent_amd.launch: a rank process failed (exit code %d)\n' % rc)
    return rc


def init_rank(backend=None):
    """inside a rank process: (rank, world, local device, dist module or None).  backend None = 'nccl' (RCCL); 'gloo' when
    several ranks share one device (CCSP_ONE_DEVICE=1: functional tests on a 1-GPU box -- RCCL refuses two ranks per device)."""
    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    from ._lib import prefer_blocking_sync
    from .selfplay import tune_host_allocator
    tune_host_allocator()                                 # a rank process is this library's own: its heap keeps its pages (selfplay.py)
    prefer_blocking_sync(local)                           # before the first GPU call of the rank: waits sleep instead of spinning
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend is None:
            backend = 'gloo' if os.environ.get('CCSP_ONE_DEVICE') == '1' else 'nccl'
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return rank, world, local, dist


def coll_device(dist):
    """where a collective's tensors live: the GPU under RCCL, the host under gloo"""
    return 'cuda' if (dist is not None and dist.get_backend() == 'nccl') else 'cpu'
