#!/usr/bin/env python
"""bench.py -- the self-play hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W        (N > 1: this process starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Headline workload (BASELINE.json configs[2], SURVEY.md §8d config 3 -- the configuration the metric "self-play games/s and
node-expansions/s at 4096 games x 400 sims" is quoted on): 4096 concurrent games per GPU, 400 simulations per move,
good_model.h5 through the fused fp32-MFMA evaluator kernel, played THROUGH THE DELIVERED API (selfplay.SelfPlayRun, what
selfplay_batch / generate_self_play / generate_train_data run on): restarting slots in steady state, two half-batches on two
HIP streams, 25 rounds per hipGraph, the sample log harvested every 2 steps and converted to (board_x, pi_y, v_y)
by a worker thread while the GPU plays on.  One STEP = one ply of every slot = 4096 x (1 root expansion + 400 simulations).
The timed region holds K steps, the conversion of every game that ended in it and the training file they are streamed into
(utils.save_train_data's datasets, chunked: selfplay.TrainDataSink(path)).  `value` = whole-job node-expansions/s (one expansion = one evaluator call, MCTS.py:93); games/s,
won games/s and sample rows/s sit beside it.  `roofline` = the dominant kernel, net_forward_kernel, against the fp32 MFMA peak.

`variants` (one GPU; 2a on every rank): config 2a (table evaluator, fused kernel -- the parity-pinned no-net configuration)
with its real bound, config 2b, the kernel micro-benchmarks.  `config5`: BASELINE configs[4] in miniature through
train.evolve -- self-play at 800 simulations, one fit, the 24-game arena -- with wall seconds per phase (every rank takes
part when N > 1).  Rank 0 finally times the CPU baselines on the host cores actually available to the process: the C oracle
with the net as evaluator (`cpu_baseline`), with the table evaluator, and the reference-shaped pure-Python mirror with both.

Games shard over ranks by game id (rank r plays ids r, r+N, ...); the only collective of the path is the summary all-reduce
after each timed region (RCCL; gloo when CCSP_BENCH_ONE_DEVICE=1 puts all ranks on one device for functional tests).

Prints ONE JSON line on rank 0.
"""
import argparse
import resource
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, 64 FLOP/clk/SIMD
NET_FLOP_PER_EVAL = 6483264   # 3,241,632 MAC
SEED = 20261003


def alg_bytes_per_expansion(D, K):
    """select D*K*20 + leaf state 32 + edges K*24 + child states K*32 + planes
    1372 written + 1372 read + policy row 1176 + v 4 + backup D*24 = 20DK + 56K + 24D + 3956"""
    return 20.0 * D * K + 56.0 * K + 24.0 * D + 3956.0


def structure_bytes_per_expansion(D, K):
    """what the FUSED kernel's own data structures move per expansion: selection reads P, W, N of
    every edge of the D nodes on the path (20 D K) and the chosen edge's child word and move (6 D); the leaf's
    position is rebuilt from the last path node's record (32); the new node block is written once (40 + 26 K);
    backup is a read-modify-write of N and W of the D path edges (24 D).  No planes, no policy row, no child
    positions: config 2a never materialises them."""
    return 20.0 * D * K + 26.0 * K + 30.0 * D + 72.0


def net_roofline(n_pos, launch_ms, isolated_ms, wall_ms_per_launch, step_ms_per_launch, asked_rows_per_s_per_gpu, samples=0,
                 tree_ms=None, launches=None, asked_rows_in_isolated_batch=None):
    """`roofline` of the dominant kernel, net_forward_kernel, against the dense fp32 MFMA peak -- arithmetic a reader can redo from the
    committed rocprof summary (profiles/*_bench_kernel_stats.csv):

        FLOP per launch = n_pos x 6 483 264 (the ALGORITHMIC count, zero-padded taps included)
        achieved = FLOP per launch / launch_ms       launch_ms = the launch's OWN duration as the product runs it: median of the launches of
        frac     = achieved / 157.3 TFLOP/s          the timed region's uncaptured rounds, between HIP events on the launching stream, the
                                                     other half-batch's graphs running beside it (rocprofv3's in-pipeline average is the
                                                     same quantity); no sample -> the wall time per launch
    Beside it, each under its own key: the same by the WALL time of the timed rounds per launch (`frac_by_wall`: gaps count against the
    kernel, overlap between the two half-batches' launches counts for it), by the driver's clock over the whole timed region
    (`frac_by_step`), for the rows that were ASKED only (`useful_frac_by_step`: idle rows carry no useful FLOP), and of the kernel alone
    (`frac_isolated`: a back-to-back burst with nothing else on the device)."""
    flop = float(n_pos) * NET_FLOP_PER_EVAL
    own = launch_ms if launch_ms else wall_ms_per_launch

    def tflops(ms):
        return flop / (ms * 1e-3) / 1e12

    def frac(ms):
        return tflops(ms) / MFMA_F32_PEAK_TFLOPS
    return {
        'bound': 'mfma', 'achieved': tflops(own), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': frac(own), 'traffic': None,
        'useful_frac_by_step': asked_rows_per_s_per_gpu * NET_FLOP_PER_EVAL / 1e12 / MFMA_F32_PEAK_TFLOPS,
        'frac_by_step': frac(step_ms_per_launch), 'frac_by_wall': frac(wall_ms_per_launch), 'frac_isolated': frac(isolated_ms),
        'dtype': 'fp32', 'kernel': 'net_forward_kernel', 'flop_per_position': NET_FLOP_PER_EVAL, 'positions_per_launch': int(n_pos),
        'avg_launch_ms': own, 'avg_launch_ms_isolated': isolated_ms, 'achieved_isolated': tflops(isolated_ms),
        'wall_ms_per_launch': wall_ms_per_launch, 'step_ms_per_launch': step_ms_per_launch,
        'launches_in_flight': own / wall_ms_per_launch,       # > 1: consecutive launches of the two half-batches overlap
        'how': ('frac = FLOP per launch / avg_launch_ms / peak; avg_launch_ms: median of %d launches inside the timed region, HIP events on the '
                'launching stream (net of the event pair\'s own cost), the other half-batch running beside them' % samples) if launch_ms else
               'frac = FLOP per launch / (wall time of the timed rounds / evaluator launches in them) / peak (no uncaptured round to put events around)',
        'launches_in_timed_region_per_gpu': launches, 'tree_kernels_ms_in_the_same_rounds': tree_ms,
        'asked_rows_in_isolated_batch': asked_rows_in_isolated_batch,
    }


# ---- multi-GPU launcher ---------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """`python bench.py --gpus N` without a torchrun around it: start the N rank processes from THIS process, which
    never touches the GPU (no torch.cuda / libccsp call before or after), wait for them, pass rank 0's JSON line
    through (the children inherit stdout) and exit non-zero if any rank fails."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    t_kill = 0.0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:                          # one rank failed: the others would wait in a collective for ever
                    q.terminate()
                t_kill = time.time() + 15.0
        if rc and alive and time.time() > t_kill:        # a rank that ignores SIGTERM (wedged in a kernel or a collective)
            for q in alive:
                q.kill()
    if rc:
        sys.stderr.write('bench.py: a rank process failed (exit code %d)\n' % rc)
    return rc


# ---- CPU baselines (rank 0, after the GPU work) -----------------------------------------------------------------------

def usable_cores():
    """host cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    1-GPU job 16 of its 256 logical CPUs; os.cpu_count() would say 256)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())))
            break
        except Exception:
            continue
    return max(1, n)


def _run_cpu_workers(kind, workers, seconds, sims, plies):
    """`workers` processes of oracle/cpu_worker.py (one per core, games sharded by id -- the reference's own
    multiprocessing.Pool scheme, train.py:73-86), all timing the same window; -> (expansions/s, games, seconds)"""
    start_at = time.time() + 1.5 + 0.02 * workers        # interpreters start and import before the window opens
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'cpu_worker.py'), '--kind', kind, '--seconds', str(seconds),
           '--sims', str(sims), '--plies', str(plies), '--stride', str(workers), '--start-at', repr(start_at)]
    procs = [subprocess.Popen(cmd + ['--first', str(i)], stdout=subprocess.PIPE, text=True) for i in range(workers)]
    done, games, t_end = 0, 0, start_at
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('cpu worker failed')
        r = json.loads(out.strip().splitlines()[-1])
        done += r['expansions']
        games += r['games']
        t_end = max(t_end, r['t_end'])
    return done / (t_end - start_at), games, t_end - start_at


def cpu_baseline(seconds=8.0, sims=400, cores=None):
    """The oracle timed on the host cores, one process per usable core, games sharded by id (the reference's
    own multiprocessing.Pool scheme).  Top level: the C restatement (`port`) WITH THE NET as evaluator -- the product's PyTorch
    module on the CPU, float32, one position per call as MCTS.py:93 calls model.predict -- on config 3 scaled down (same 400
    simulations per move, 16 searched plies per game).  Beside it: the same with the table evaluator (config 2a's baseline), the
    reference-shaped pure-Python mirror (oracle/pymirror.py) with its calibration against the reference, and BASELINE config 1
    (one game, 50 simulations per move, good_model.h5 through the NumPy float32 net, one core)."""
    cores = usable_cores() if not cores else min(int(cores), usable_cores())
    out = {}
    w = weights_path()
    if w:
        net, g0, dt0 = _run_cpu_workers('c_net', cores, seconds, sims, 16)
        out = dict(value=net, unit='node-expansions/s', cores=cores, kind='port',
                   sample='config 3 scaled down: %d processes (one per usable core; os.cpu_count() = %d) x C oracle searches with %s '
                          'through the PyTorch CPU module (float32, 1 thread, batch of one per expansion as MCTS.py:93), %d sims per '
                          'move, 16 MCTS plies per game, %d games, %.1f s' % (cores, os.cpu_count() or 0, os.path.basename(w), sims, g0, dt0),
                   per_core=net / cores)
    one, _, _ = _run_cpu_workers('c', 1, max(2.0, seconds / 3), sims, 16)
    allc, games, dt = _run_cpu_workers('c', cores, seconds, sims, 16)
    table = dict(value=allc, unit='node-expansions/s', cores=cores, kind='port',
                 sample='%d processes x C oracle searches: %d sims per move, table evaluator p=1/294 v=0 (config 2a), 16 MCTS plies '
                        'per game, %d games, %.1f s' % (cores, sims, games, dt),
                 single_thread=one, scaling_factor=allc / one if one else None)
    if not out:
        out = dict(table)
    out['table_evaluator'] = table
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import pymirror
    cal = pymirror.CALIBRATION
    pm, g2, dt2 = _run_cpu_workers('py', cores, seconds, sims, 1)
    out['reference_shaped_python'] = dict(
        value=pm, unit='node-expansions/s', cores=cores, per_core=pm / cores, kind='port (reference-shaped mirror, oracle/pymirror.py)',
        sample='%d processes x 1 searched ply (%d sims, uniform table evaluator) per game, %d games, %.1f s' % (cores, sims, g2, dt2),
        calibration=cal, reference_estimate=pm / cal['mirror_py310_over_reference_py39'])
    if w:
        c1, g3, dt3 = _run_cpu_workers('py_net', 1, seconds, 50, 200)
        out['config1_reference_shaped_python_numpy_net'] = dict(
            value=c1, unit='node-expansions/s', cores=1, kind='port (oracle/pymirror.py + oracle/net_oracle.py float32)',
            sample='BASELINE config 1: one game at a time, 50 sims per move, %s through the NumPy float32 net, one core, %.1f s '
                   '(%d game(s) begun)' % (os.path.basename(w), dt3, g3),
            games_per_s_estimate=c1 / (51.0 * 75.0), reference_estimate=c1 / cal['mirror_py310_over_reference_py39'],
            note='a game is about 75 searched plies x 51 evaluator calls')
    return out


# ---- timed regions -------------------------------------------------------------------------------------------------

def timed_plies(eng, evaluator, steps, torch):
    """`steps` plies of the fused path in ONE ccsp_play_plies call (the library carries every game through up to
    ccsp_debug_plies_per_launch plies per launch of fused_plies_kernel); returns (wall seconds, avg ms per ply by HIP events
    recorded on the stream the kernels are launched on, number of launches)"""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from chinesecheckersagent_amd import _lib
    ppl = _lib.lib().ccsp_debug_plies_per_launch(0)
    torch.cuda.synchronize()
    t0 = time.time()
    a.record()
    eng.play_plies(evaluator, steps)
    b.record()
    torch.cuda.synchronize()
    wall = time.time() - t0
    launches = (steps + ppl - 1) // ppl if ppl > 1 and steps > 1 else 3 * steps
    return wall, a.elapsed_time(b) / steps, launches


def weights_path():
    p = os.path.join(ROOT, 'tests', 'golden', 'good_model.h5')
    return p if os.path.exists(p) else None


def net_kernel_alone(model, req, moves, torch):
    """the dominant kernel by itself, as the delivered path calls it (ccsp_net_forward_requests on one half-batch's request records --
    the run's own, as they stood at the end of the timed region): 400 back-to-back launches with NOTHING else on the device, timed
    with HIP events recorded on the stream the kernel is launched on"""
    from chinesecheckersagent_amd import _lib
    from chinesecheckersagent_amd.engine import _stream_ptr
    L = _lib.lib()
    n_pos = req.shape[0]
    packed = model._ensure_packed()
    pk = torch.empty((n_pos, _lib.REQUEST_MOVES), dtype=torch.float64, device=req.device)
    v_out = torch.empty(n_pos, dtype=torch.float32, device=req.device)
    st = _stream_ptr()
    torch.cuda.synchronize()
    for _ in range(100):                    # (a short burst right after an idle gap is timed at whatever clock the part is ramping through)
        L.ccsp_net_forward_requests(packed.data_ptr(), req.data_ptr(), moves.data_ptr(), n_pos, pk.data_ptr(), v_out.data_ptr(), st)
    iters = 400
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        L.ccsp_net_forward_requests(packed.data_ptr(), req.data_ptr(), moves.data_ptr(), n_pos, pk.data_ptr(), v_out.data_ptr(), st)
    e.record()
    torch.cuda.synchronize()
    burst_ms = a.elapsed_time(e) / iters
    # what a PAIR OF EVENTS around one launch adds to its duration (the markers are packets of their own): the same launches, still alone,
    # each between its own two events -- the in-pipeline figure is taken the same way and is corrected by the difference
    pairs = []
    for _ in range(48):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record()
        L.ccsp_net_forward_requests(packed.data_ptr(), req.data_ptr(), moves.data_ptr(), n_pos, pk.data_ptr(), v_out.data_ptr(), st)
        y.record()
        pairs.append((x, y))
    torch.cuda.synchronize()
    br = sorted(x.elapsed_time(y) for x, y in pairs)
    return burst_ms, max(0.0, br[len(br) // 2] - burst_ms)


def config3(args, torch, rank, world, local, barrier):
    """The headline on this rank: config 3 through the delivered API.  `games` slots x `sims` simulations with the net, restarting
    slots, steady state; K timed steps + conversion of the games that ended + the HDF5 write, all inside the timed region."""
    import shutil
    import tempfile
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    G, S, K, W = args.games, args.sims, args.steps, args.warmup
    w = weights_path()
    model = ResidualCNN(device='cuda:%d' % local)
    if w:
        model.load_weights(w)
    assert model.backend == 'hip', 'the evaluator must be the fused HIP kernel'
    sink = sp.TrainDataSink()
    sink.discard = True
    # the id budget is never the limit: every slot can restart 64 times
    # the slots' first games begin spread over the untimed plies (all but the last twelve of them: every slot is playing well before the
    # warm-up) instead of together: the timed window then sees games END at the long-run rate wherever it sits, not the swell of a first
    # cohort that began together and ends together one mean game length later
    span = max(0, args.spread_plies - 12) * (S + 1)
    run = sp.SelfPlayRun(model, n_games=G * 64, sims=S, seed=SEED, first_game=rank, game_stride=world, device=local, max_slots=G,
                         harvest_every=args.harvest_every, keep_records=False, sink=sink, stagger_span=span or None)
    out_dir = tempfile.mkdtemp(prefix='ccsp-bench-')
    try:
        for _ in range(args.spread_plies + W):          # untimed: the first cohort of games spreads out, then the W warm-up steps
            run.play_ply()
        run.drain()
        sink.discard = False                            # rows of games that ended before the timed region are not its output
        sink.open(os.path.join(out_dir, 'data-for-iter-%d.h5' % rank))
        barrier()
        parts_ = run.b.parts if hasattr(run.b, 'parts') else [run.b]
        for b_ in parts_:                               # the evaluator launches of the uncaptured steps (one per part and timed ply), each
            if getattr(b_, 'free_running', False):      # between HIP events on its own stream while the other half-batch's graphs run
                b_.net_events = []
        c0 = run.counters()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)   # this rank's host side (main thread + converter thread) over the timed region
        t0 = time.time()
        steps = 0
        while True:
            for _ in range(K):
                run.play_ply()
            steps += K
            torch.cuda.synchronize()
            if time.time() - t0 >= args.min_seconds or K == 0:    # a timed region is never shorter than --min-seconds
                break
        t_play = time.time() - t0
        run.drain()                                     # the last harvest + the conversion of every game that ended
        t_drain = time.time() - t0 - t_play
        rows = sink.rows
        path = sink.close()                             # the last chunk, the chunk B-trees and the headers of the streamed file
        t_write = time.time() - t0 - t_play - t_drain
        barrier()
        dt = time.time() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        c1 = run.counters()
        net_in_pipeline_ms = [ev[0].elapsed_time(ev[1]) for b_ in parts_ for ev in (getattr(b_, 'net_events', None) or [])]
        tree_in_pipeline_ms = [ev[1].elapsed_time(ev[2]) for b_ in parts_ for ev in (getattr(b_, 'net_events', None) or [])]
        size = os.path.getsize(path)
        parts = len(run.b.parts) if hasattr(run.b, 'parts') else 1
        graphs = [b._graph is not None for b in (run.b.parts if hasattr(run.b, 'parts') else [run.b])]
        if not all(graphs):
            raise RuntimeError('config 3 did not run on captured hipGraphs')
        if not getattr(run, 'free_running', False) and G >= 1024:
            raise RuntimeError('config 3 did not run on the free-running path')
        torch.cuda.synchronize()
        p0 = parts_[0]
        if getattr(p0, 'free_running', False):          # one half-batch's requests as they stand: the isolated burst below evaluates these
            req_alone, moves_alone = p0._req.clone(), p0._moves.clone()
            asked_now = int((req_alone.view(torch.int32)[:, 8] != 0).sum())
        else:
            req_alone = moves_alone = None
            asked_now = 0
    finally:
        run.close()
        shutil.rmtree(out_dir, ignore_errors=True)
    d = {k: c1[k] - c0[k] for k in c1}
    n_pos = run.n_slots // parts
    if req_alone is None:                               # (a lock-step run of fewer than 1024 slots: requests made up from the log's last positions)
        from chinesecheckersagent_amd import _lib as _l
        r_ = np.zeros(n_pos, dtype=_l.REQUEST_DTYPE)
        r_['state'] = _l.pack_states(np.tile(np.array([42, 35, 43, 28, 36, 44, 6, 13, 5, 20, 12, 4], dtype=np.uint8), (n_pos, 1)))
        r_['kind'], r_['player'], r_['k'] = 1, 1, 14
        req_alone = torch.from_numpy(r_.view(np.uint8).reshape(n_pos, 64)).to('cuda:%d' % local)
        moves_alone = torch.zeros((n_pos, _l.REQUEST_MOVES), dtype=torch.int16, device=req_alone.device)
        asked_now = n_pos
    k_ms, ev_over_ms = net_kernel_alone(model, req_alone, moves_alone, torch)
    return d, dt, dict(n_pos=n_pos, k_ms=k_ms, ev_over_ms=ev_over_ms, parts=parts, weights=os.path.basename(w) if w else 'random-init', backend=model.backend,
                       steps=steps, rows_written=int(rows), file_bytes=size, t_play=t_play, t_drain=t_drain, t_write=t_write,
                       n_slots=run.n_slots, free_running=bool(getattr(run, 'free_running', False)), asked_in_isolated_batch=asked_now,
                       net_in_pipeline_ms=(sorted(net_in_pipeline_ms)[len(net_in_pipeline_ms) // 2] if net_in_pipeline_ms else None),
                       net_in_pipeline_samples=len(net_in_pipeline_ms),
                       tree_in_pipeline_ms=(sorted(tree_in_pipeline_ms)[len(tree_in_pipeline_ms) // 2] if tree_in_pipeline_ms else None),
                       host_cpu_s=(ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime), peak_rss_mb=ru1.ru_maxrss / 1024.0)


def config2a(args, torch, rank, world, local, barrier, engine, _lib):
    """config 2a on this rank: the fused kernel with the table evaluator p = 1/294, v = 0 (what the reference computes with a stub
    model; parity-pinned), `plies` timed plies in steady state"""
    G, S = args.games, args.sims
    plies = args.fused_plies
    eng = engine.SelfPlayEngine(n_slots=G, sims=S, seed=SEED, first_game=rank, game_stride=world,
                                max_games=G * 64, log_capacity=G * (plies + 24), auto_restart=True, device=local)
    EV = _lib.EVAL_UNIFORM
    eng.play_plies(EV, 6)                   # the six random opening plies (selfplay.py:32-33), untimed
    for _ in range(8):
        eng.play_plies(EV, 1)
    barrier()
    c0 = eng.counters()
    t0 = time.time()
    wall, kernel_ms, launches = timed_plies(eng, EV, plies, torch)
    barrier()
    elapsed = time.time() - t0
    c1 = eng.counters()
    d = {k: c1[k] - c0[k] for k in c1}
    return eng, d, elapsed, kernel_ms, launches, plies


def config1(args, torch, local):
    """BASELINE configs[0] on the GPU: ONE game at a time at 50 simulations per move with good_model.h5 through the delivered
    selfplay() -- a batch of one position per evaluator launch, so everything here is latency: seconds per whole game (engine set-up
    and graph capture of every call included) beside the CPU's figure for the same configuration (cpu_baseline.config1_...)"""
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    model = ResidualCNN(device='cuda:%d' % local)
    model.load_weights(weights_path())
    sp.selfplay(model, sims=50, seed=SEED, game_id=10 ** 6)          # (first call: library and allocator warm-up)
    torch.cuda.synchronize()
    games, plies, won = 6, 0, 0
    t0 = time.time()
    for g in range(games):
        hist, reward = sp.selfplay(model, sims=50, seed=SEED, game_id=g)
        if hist is not None:
            won += 1
            plies += len(hist)
    torch.cuda.synchronize()
    dt = time.time() - t0
    return dict(workload='config 1: one game at a time (a single slot: one position per evaluator launch), 50 sims/move, %s, '
                         'selfplay() called %d times in a row' % (os.path.basename(weights_path()), games),
                games=games, games_won=won, seconds_per_game=dt / games, games_per_s=games / dt,
                searched_plies_of_won_games=plies)


def config5(args, torch, rank, world, local, dist):
    """BASELINE configs[4] in miniature through train.evolve: self-play at 800 simulations per move -> convert / augment / save
    -> fit -> 24-game arena; wall seconds per phase.  With N ranks: sharded self-play and arena, DDP fit (train.evolve(dist=...))."""
    import shutil
    import tempfile
    from chinesecheckersagent_amd import train
    w = weights_path()
    work = None
    if rank == 0:
        work = tempfile.mkdtemp(prefix='ccsp-config5-')
    if dist is not None:
        box = [work]
        dist.broadcast_object_list(box, src=0)
        work = box[0]
    timings = []
    if os.environ.get('CCSP_BENCH_TEST_FAIL_RANK') == str(rank):     # test hook: this rank's config 5 raises before its first collective
        raise RuntimeError('config 5 made to fail on rank %d (test hook)' % rank)
    if os.environ.get('CCSP_BENCH_TEST_STALL_RANK') == str(rank):    # test hook: this rank never joins the loop's collectives
        time.sleep(3600)
    # the training step's lazy initialisation (MIOpen kernel choice / build: ~5 s on a fresh box, where the driver runs this; well under a
    # second with a warm cache) is taken OUT of the iteration and reported on its own, so that train_s is what every later iteration pays
    t_w = time.time()

    def progress(*a):                       # the loop's phase messages, with the time since config 5 began (stderr: the line on stdout stays alone)
        sys.stderr.write('bench.py config 5 [%6.1f s] %s\n' % (time.time() - t_w, ' '.join(str(x) for x in a)))
        sys.stderr.flush()
    warm = train.warm_up(device='cuda:%d' % local)
    if rank == 0:
        progress('training step warmed up on every rank')
    try:
        t0 = time.time()
        cur, best, it = train.evolve(w, best_model=w, iterations=1, num_self_play=args.config5_games, eval_games=24,
                                     sims=args.config5_sims, seed=SEED, data_dir=os.path.join(work, 'data'),
                                     weights_dir=os.path.join(work, 'weights'), log=progress, dist=dist, device=local,
                                     timings=timings)
        wall = time.time() - t0
    finally:
        if dist is not None:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(work, ignore_errors=True)
    tm = timings[0]
    tm.pop('iteration', None)
    return dict(workload='config 5 in miniature: train.evolve, one iteration: %d self-play games (sharded over the %d GPU(s) by id) at %d '
                         'sims/move with %s, augment + save + fit (5 epochs of batch 32%s), arena of 24 games at %d sims with the '
                         '100-move limit' % (args.config5_games, world, args.config5_sims, os.path.basename(w),
                            ', DistributedDataParallel' if dist is not None else '', args.config5_sims),
                wall_s=wall, train_warm_up_s=warm, wall_with_warm_up_s=wall + (t0 - t_w),
                selfplay_expansions_per_s=tm['selfplay_expansions'] / tm['selfplay_s'], **tm)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=48)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--games', type=int, default=4096, help='concurrent games per GPU')
    ap.add_argument('--sims', type=int, default=400)
    # 160: the slots' first games begin spread over 148 steps, so that by the timed window games of EVERY length end at their long-run rates --
    # a game discarded for no progress lasts 110-250 plies (80-190 steps), and with the 72 of round 5 the first cohort's were still in flight:
    # discard rate of a 20-step window 0.131 / 0.156 / 0.177 / 0.177 at 72 / 120 / 160 / 200 against 0.19 of whole runs (profiles/r6_spread_sweep.txt);
    # node-expansions/s do not depend on it (+- 0.5 %)
    ap.add_argument('--spread-plies', type=int, default=160, help='untimed plies before the warm-up: the first cohort of games spreads out')
    ap.add_argument('--harvest-every', type=int, default=2)
    ap.add_argument('--min-seconds', type=float, default=1.0, help='shortest timed region: K more steps are added until it is reached')
    ap.add_argument('--fused-plies', type=int, default=192, help='variant 2a: timed plies of the fused kernel')
    ap.add_argument('--no-extras', action='store_true', help='headline region only: no variants, config 5, cpu baseline')
    ap.add_argument('--no-config5', action='store_true')
    ap.add_argument('--config5-games', type=int, default=256, help='config 5: self-play games of the iteration (all GPUs together, as NUM_SELF_PLAY is)')
    ap.add_argument('--config5-timeout', type=float, default=300.0, help='N > 1: seconds after which a config 5 that has not come back is given up')
    ap.add_argument('--config5-sims', type=int, default=800)
    ap.add_argument('--cpu-seconds', type=float, default=8.0, help='seconds per CPU-baseline leg; 0 = no CPU baseline')
    ap.add_argument('--cpu-cores', type=int, default=0, help='cap on the worker processes of the CPU baseline (0 = every usable core)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))               # before anything touches the GPU

    os.environ['CCSP_STRICT'] = '1'                     # a performance path that cannot be taken is an error here
    import torch
    from chinesecheckersagent_amd import _lib, engine, summary
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    from chinesecheckersagent_amd.selfplay import tune_host_allocator
    tune_host_allocator()                   # this process is a rank and nothing else: its heap keeps its pages (host side)
    _lib.prefer_blocking_sync(0 if os.environ.get('CCSP_BENCH_ONE_DEVICE') == '1' else local)      # a rank's waits sleep instead of spinning (a host core per rank: DESIGN.md section 8)
    _lib.require_gpu()                      # no CPU fallback: fail loudly
    # functional test hook for 1-GPU boxes: CCSP_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo for
    # the summary (RCCL refuses two ranks on one device).  Never set by the driver.
    one_dev = os.environ.get('CCSP_BENCH_ONE_DEVICE') == '1'
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    coll_dev = 'cuda'
    # CCSP_BENCH_FORCE_DIST=1 (test hook for 1-GPU boxes): a world of ONE rank still goes through RCCL -- process group on the
    # device, the MAX / SUM all-reduces and the all-gather of the summary -- so that the collective code the N-GPU runs use has
    # executed on the hardware at least once
    if world > 1 or os.environ.get('CCSP_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
        if one_dev:
            dist.init_process_group('gloo')
            coll_dev = 'cpu'
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_run(d, hist, elapsed):
        """max elapsed over ranks; summary all-reduce (the path's only collective); every rank's
        expansions through an all-gather of one int64"""
        if dist is None:
            return d, hist, elapsed, [d['expansions']]
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tot, hist = summary.allreduce_summary(d, hist, dist, device=coll_dev)
        mine = torch.tensor([d['expansions']], dtype=torch.int64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return tot, hist, float(t.item()), [int(x.item()) for x in every]

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    G, S, K, W = args.games, args.sims, args.steps, args.warmup
    extras_wanted = not args.no_extras

    # ---- the headline: config 3 through the delivered API (an exception here is fatal: no result line) ----------------
    d3, dt3, info = config3(args, torch, rank, world, local, barrier)
    tot3, _, dt3, per3 = reduce_run(d3, np.zeros(_lib.NUM_ACTIONS, dtype=np.uint64), dt3)
    k_ms = max_over_ranks(info['k_ms'])
    rows_all, bytes_all = info['rows_written'], info['file_bytes']
    if dist is not None:
        t = torch.tensor([rows_all, bytes_all], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t)
        rows_all, bytes_all = int(t[0]), int(t[1])
    host = [[info['host_cpu_s'], info['peak_rss_mb'], info['t_play']]]
    if dist is not None:
        t = torch.tensor(host[0], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        host = [[float(x) for x in e.tolist()] for e in every]
    # (the in-pipeline launches are timed between a pair of events each: the pair's own cost, measured on the same launches alone, is taken off)
    k_in_local = (info['net_in_pipeline_ms'] - info['ev_over_ms']) if info.get('net_in_pipeline_ms') else (info['t_play'] / (info['steps'] * (S + 1) * info['parts']) * 1e3)
    k_in_all = max_over_ranks(k_in_local)

    out = None
    if rank == 0:
        ex = tot3['expansions']
        steps = info['steps']
        tf = info['n_pos'] * NET_FLOP_PER_EVAL / (k_ms * 1e-3) / 1e12
        done = tot3['games_won'] + tot3['games_discarded']
        launches = steps * (S + 1) * info['parts']                      # evaluator launches of one rank in the timed region
        hits = tot3.get('cache_hits', 0)
        rows_carried = steps * (S + 1) * info['n_slots'] * world        # rows of all evaluator launches of the timed region
        out = {
            'metric': 'mcts_node_expansions_per_s (self-play with the policy/value net, %d games x %d sims/move per GPU); BASELINE games/s = '
                      'config.games_per_s: selfplay() calls that RETURNED per second, discarded ones included (train.py:61-64 counts them); '
                      'config.games_won_per_s: those that returned a history' % (G, S),
            'value': ex / dt3, 'unit': 'node-expansions/s', 'n_gpus': world, 'steps': steps, 'steps_requested': K, 'warmup': W,
            'ms_per_step': dt3 / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'degraded': False,
            # `config`: SCALAR keys only (the driver's record keeps those and cuts strings at 120 characters): what names the workload first,
            # then BASELINE.json's first-named metric -- games/s -- and the figures `value` must be read beside
            'config': {'workload': 'config 3: %d games/GPU x %d sims/move, %s, SelfPlayRun free-running + tree reuse' % (G, S, info['weights']),
                       'games_per_gpu': G, 'sims': S, 'sharding': 'game id mod n_gpus',
                       'games_per_s': done / dt3, 'games_won_per_s': tot3['games_won'] / dt3,
                       'node_expansions_per_s': ex / dt3,
                       'net_evals_per_s': rows_carried / dt3,                              # rows the evaluator launches carried
                       'net_evals_asked_per_s': (ex - hits) / dt3,                         # rows that answered a request
                       'tree_reuse_hit_rate': hits / max(ex, 1),                           # expansions answered by the previous ply's tree
                       'idle_row_share': 1.0 - (ex - hits) / max(rows_carried, 1),         # evaluator rows no slot asked for
                       'timed_region_s': dt3, 'discard_rate': tot3['games_discarded'] / max(done, 1),
                       'plies_per_game': tot3['plies'] / max(done, 1), 'train_rows_per_s': rows_all / dt3,
                       'host_cores_per_rank': max(h[0] for h in host) / dt3, 'host_peak_rss_mb_per_rank': max(h[1] for h in host),
                       'free_running': info['free_running'], 'half_batches': info['parts'], 'harvest_every_steps': args.harvest_every,
                       'untimed_steps_before': args.spread_plies + W,
                       'step': 'sims + 1 = %d evaluator launches per half-batch (a slot plays %.2f searched plies in it)'
                               % (S + 1, tot3['mcts_plies'] / max(steps * info['n_slots'] * world, 1))},
            'description': 'config 3: %d concurrent games/GPU x %d sims/move, %s through the fused fp32-MFMA evaluator kernel (float64 PUCT tree), played '
                           'through selfplay.SelfPlayRun (the API behind selfplay_batch / generate_self_play): restarting slots in steady state, '
                           'free-running stepped path (every slot at its own simulation of its own ply: [evaluator on the request records -> advance '
                           'kernel -> boundary kernel every sixth round]; positions of the previous tree reused; 25 rounds per hipGraph, %d half-batches '
                           'on their own streams), log harvested every %d steps; timed: %d steps + conversion to (board_x, pi_y, v_y) + streaming them '
                           'into the HDF5 training file; untimed before: %d steps in which the first cohort of games spreads out + %d warm-up'
                           % (G, S, info['weights'], info['parts'], args.harvest_every, steps, args.spread_plies, W),
            'measured': {'terminal_sim_share': tot3['terminal_sims'] / max(tot3['sims'], 1),
                         'expansions_from_previous_tree_per_s': hits / dt3,
                         'searched_plies_per_slot_per_step': tot3['mcts_plies'] / max(steps * info['n_slots'] * world, 1),
                         # free-running slots do not share a ply: the time in which every slot searches ONE ply on average
                         # (what `ms_per_step` was in the lock-step form, where a step is a ply of every slot)
                         'ms_per_searched_ply_of_every_slot': dt3 * 1e3 * info['n_slots'] * world / max(tot3['mcts_plies'], 1),
                         'host_cpu_s_per_rank': [h[0] for h in host], 'host_cpu_cores_busy_per_rank': [h[0] / dt3 for h in host],
                         'host_peak_rss_mb_per_rank': [h[1] for h in host], 'usable_cores': usable_cores()},
            'tree_reuse_hit_rate': hits / max(ex, 1), 'idle_row_share': 1.0 - (ex - hits) / max(rows_carried, 1),
            'net_evals_per_s': rows_carried / dt3, 'net_evals_asked_per_s': (ex - hits) / dt3,
            'games_per_s': done / dt3, 'games_won_per_s': tot3['games_won'] / dt3, 'games_finished': done, 'games_won': tot3['games_won'],
            'samples_per_s': tot3['samples'] / dt3, 'plies_per_s': tot3['plies'] / dt3,
            'plies_per_game': tot3['plies'] / max(done, 1), 'discard_rate': tot3['games_discarded'] / max(done, 1),
            'train_rows_written': rows_all, 'train_file_bytes': bytes_all,
            'timed_region_s': {'total': dt3, 'plies': info['t_play'], 'final_harvest_and_conversion': info['t_drain'], 'hdf5_close': info['t_write']},
            'ms_per_sim_step': info['t_play'] / steps / (S + 1) * 1e3,
            # FLOP of the evaluator rows that were ASKED for (expansions answered by the previous ply's tree ran no evaluator), per GPU
            'net_tflops_end_to_end': (ex - hits) * NET_FLOP_PER_EVAL / dt3 / 1e12 / world,
            'mean_depth': tot3['sum_depth'] / max(tot3['sims'], 1), 'mean_children': tot3['sum_children'] / max(ex, 1),
            'errors': tot3['errors'], 'per_rank_expansions': per3, 'precision': 'fp32', 'backend': info['backend'],
            'target_node_expansions_per_s_per_gpu': 1e6,
            # the dominant kernel against the fp32 MFMA peak: see net_roofline (frac = by the launch's own duration in the pipeline)
            'roofline': net_roofline(info['n_pos'], info.get('net_in_pipeline_ms') and k_in_all, k_ms, max(h[2] for h in host) / launches * 1e3,
                                     dt3 / launches * 1e3, (ex - hits) / dt3 / world, samples=info['net_in_pipeline_samples'],
                                     tree_ms=info.get('tree_in_pipeline_ms'), launches=launches,
                                     asked_rows_in_isolated_batch=info.get('asked_in_isolated_batch')),
        }
        out['roofline']['event_pair_overhead_ms'] = info['ev_over_ms']        # taken off avg_launch_ms (raw median: + this)
        try:                                 # HBM bytes of one launch by the counters (static: a --pmc pass cannot run inside this process)
            prof = json.load(open(os.path.join(ROOT, 'profiles', 'counters.json')))['net_forward_kernel']
            if info['n_pos'] == prof['n']:
                out['roofline']['traffic'] = (2.0 * prof['fetch_size_kb'] + prof['write_size_kb']) * 1024.0
                out['roofline']['traffic_source'] = 'static: ' + prof['source']
                kbar = tot3['sum_children'] / max(ex, 1)
                # request records in (64 B), their move rows in (2 K), compact priors and v out (8 K + 4), the weights once
                out['roofline']['algorithmic_bytes_per_launch'] = info['n_pos'] * (64 + 2 * kbar + 8 * kbar + 4) + 4 * 250880
        except (OSError, KeyError, ValueError) as err:
            out['roofline']['traffic_source'] = 'profiles/counters.json not usable: %r' % (err,)
        # the long-run share of discarded games, from WHOLE runs (every game played to its end: 32 768 games, profiles/r3_config4_rehearsal.txt;
        # 8 192: 0.190) -- what `discard_rate` of a timed window tends to once the slots' games are spread over their whole length
        out['discard_rate_of_whole_runs'] = {'value': 6292 / 32768.0, 'source': 'profiles/r3_config4_rehearsal.txt: 32 768 whole games, 6 292 discarded'}
        if tot3['errors']:
            raise SystemExit('config 3 counted %d engine errors' % tot3['errors'])

    # ---- variant 2a on every rank (the no-net configuration: the fused kernel alone) ---------------------------------------
    if extras_wanted:
        eng, d, elapsed, kernel_ms, launches, plies = config2a(args, torch, rank, world, local, barrier, engine, _lib)
        tot, hist, elapsed, per_rank = reduce_run(d, eng.visit_histogram(), elapsed)
        kernel_ms = max_over_ranks(kernel_ms)
        if rank == 0:
            D = tot['sum_depth'] / max(tot['sims'], 1)
            Kc = tot['sum_children'] / max(tot['expansions'], 1)
            exp_per_ply = tot['expansions'] / plies / world             # per GPU
            ppl = plies / launches                                      # plies one launch of fused_plies_kernel carried every game through
            alg = alg_bytes_per_expansion(D, Kc)
            own = structure_bytes_per_expansion(D, Kc)
            v2a = {'workload': 'config 2a: %d concurrent games/GPU x %d sims/move, table evaluator p=1/294 v=0 (no net), fused HIP '
                               'select/movegen/expand/backup kernel, games auto-restart; %d plies timed' % (G, S, plies),
                   'node_expansions_per_s': tot['expansions'] / elapsed, 'ms_per_ply': elapsed / plies * 1e3,
                   'games_per_s': (tot['games_won'] + tot['games_discarded']) / elapsed, 'plies_per_s': tot['plies'] / elapsed,
                   'samples_logged': tot['samples'], 'mean_depth': D, 'mean_children': Kc, 'errors': tot['errors'],
                   'per_rank_expansions': per_rank, 'visit_histogram_sum': int(np.asarray(hist, dtype=np.uint64).sum())}
            # The kernel is bound by latency and instruction issue, not by bandwidth: its `frac` is what the counters
            # say it moves against the HBM peak; SURVEY §8d's byte model (which charges planes, policy rows and child positions this
            # configuration never materialises) is kept beside it as algorithmic_model_frac.
            roof = {'bound': 'latency/issue', 'kernel': 'fused_plies_kernel (one wave per game: root expansion, %d simulations and the move, ply after ply)' % S,
                    'achieved': None, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': None, 'traffic': None,
                    'avg_launch_ms': kernel_ms * ppl, 'launches': launches, 'plies_per_launch': ppl, 'ms_per_ply': kernel_ms,
                    'expansions_per_launch': exp_per_ply * ppl,
                    'algorithmic_model': 'SURVEY.md 8d: 20DK + 56K + 24D + 3956 bytes per expansion', 'bytes_per_expansion': alg,
                    'algorithmic_model_frac': exp_per_ply * alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    'kernel_structure_bytes_per_expansion': own,
                    'hbm_frac_kernel_structures': exp_per_ply * own / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            try:                             # PMC figures cannot be taken inside this process: static, from the committed passes
                prof = json.load(open(os.path.join(ROOT, 'profiles', 'counters.json')))['fused_plies_kernel']
                if G == 4096 and S == 400:
                    traffic = 2.0 * prof['fetch_size_kb'] * 1024.0 + prof['write_size_kb'] * 1024.0      # per ply
                    roof['traffic'] = traffic * ppl                                                      # per launch
                    roof['traffic_source'] = 'static: ' + prof['source']
                    roof['achieved'] = traffic / (kernel_ms * 1e-3) / 1e9                                # counter bytes over THIS run's time
                    roof['frac'] = roof['achieved'] / HBM_PEAK_GBPS
                    if 'insts_valu' in prof:
                        # issue model: a wave64 VALU instruction holds its SIMD-32 for 2 cycles (4 for f64), the CU's one
                        # scalar unit issues one SALU instruction per cycle for all four SIMDs
                        clk = prof.get('clock_ghz', 2.1) * 1e9
                        t_valu = prof['insts_valu'] * 2.0 / (256 * 4) / clk
                        t_salu = prof['insts_salu'] / 256.0 / clk
                        roof['issue_model'] = {'insts_valu': prof['insts_valu'], 'insts_salu': prof['insts_salu'],
                                               'valu_ms_at_2_cycles': t_valu * 1e3, 'salu_ms_at_1_per_cu_cycle': t_salu * 1e3,
                                               'launch_ms': prof['launch_ms'], 'frac_of_issue_limit': max(t_valu, t_salu) * 1e3 / prof['launch_ms'],
                                               'source': 'static: ' + prof['source']}
            except (OSError, KeyError, ValueError) as ex:
                roof['traffic_source'] = 'profiles/counters.json not usable: %r' % (ex,)
            v2a['roofline'] = roof
            out['variants'] = {'2a_fused_table_evaluator': v2a}
            if tot['errors']:
                raise SystemExit('config 2a counted %d engine errors' % tot['errors'])
        if world == 1:
            out['variants'].update(extras(eng, G, S, torch, _lib, engine))
            if weights_path():
                out['config1'] = config1(args, torch, local)
        eng.close()

    # ---- config 5 in miniature: every rank --------------------------------------------------------------------------------
    if extras_wanted and not args.no_config5 and weights_path():
        watchdog = None
        stop_watch = None
        # At N > 1 the loop is full of collectives: a rank that fails alone leaves the others waiting in one, and the headline above must
        # not be lost to that.  Rule (round-3 verdict): the headline exists by now, so such a run is DEGRADED, not failed -- rank 0 prints
        # the line it has with "degraded": true and config5: {failed: ...}, every rank says so on stderr, and the exit status stays 0 (a
        # missing headline is what exits non-zero: everything above this block raises).  The rank that fails drops a flag file; the other
        # ranks' watchdogs see it within a second (or give up after --config5-timeout) instead of sitting in a collective.
        flag = os.path.join(tempfile.gettempdir(), 'ccsp-bench-config5-%s.failed' % os.environ.get('MASTER_PORT', str(os.getpid())))
        out_lock = threading.Lock()

        def leave_degraded(reason):
            with out_lock:                               # (the main thread may be writing to `out`)
                line = None
                if rank == 0:
                    snap = dict(out)
                    snap['config5'] = {'failed': reason}
                    snap['cpu_baseline'] = None
                    snap['degraded'] = True
                    snap['degraded_reason'] = 'config 5: ' + reason
                    line = json.dumps(snap)
                sys.stderr.write('bench.py: rank %d leaves config 5 (%s): the run is DEGRADED\n' % (rank, reason))
                if line is not None:
                    print(line, flush=True)
                sys.stderr.flush()
                os._exit(0)
        if world > 1:
            if rank == 0 and os.path.exists(flag):
                os.remove(flag)
            dist.barrier()
            stop_watch = threading.Event()

            def watch():
                t_end = time.time() + args.config5_timeout
                while not stop_watch.wait(1.0):
                    if os.path.exists(flag):
                        leave_degraded('another rank failed')
                    if time.time() > t_end:
                        leave_degraded('not back after %.0f s (watchdog)' % args.config5_timeout)
            watchdog = threading.Thread(target=watch, daemon=True)
            watchdog.start()
        try:
            c5 = config5(args, torch, rank, world, local, dist)
            if rank == 0:
                with out_lock:
                    out['config5'] = c5
        except Exception as ex:
            if world == 1:
                raise                                                    # one GPU: tested, a failure is a failure
            try:
                open(flag, 'w').write('rank %d: %r\n' % (rank, ex))
            except OSError:
                pass
            leave_degraded('failed on rank %d: %r' % (rank, ex))
        if stop_watch is not None:
            stop_watch.set()
            watchdog.join()
    if rank == 0:
        out['cpu_baseline'] = cpu_baseline(args.cpu_seconds, S, args.cpu_cores) if (extras_wanted and args.cpu_seconds > 0) else None
        if out['cpu_baseline'] and out['cpu_baseline'].get('kind') == 'port' and out.get('plies_per_game'):
            # games/s of the CPU legs that play config 3's workload: their node-expansions/s over the evaluator calls of a game as the GPU
            # run measured it ((sims + 1) per searched ply; plies_per_game counts the six random opening plies too)
            per_game = (S + 1) * max(out['plies_per_game'] - 6.0, 1.0)
            out['cpu_baseline']['games_per_s_estimate'] = out['cpu_baseline']['value'] / per_game
            out['cpu_baseline']['games_per_s_estimate_per_core'] = out['cpu_baseline']['value'] / per_game / out['cpu_baseline']['cores']
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def s1_positions(n_distinct, torch, rules, _lib):
    """SURVEY.md §8d micro-benchmark inputs, built with the product's own kernels: state i = Board() advanced by
    i mod 64 plies of rule S1 (selfplay.py:83-104: a checker that can move, uniformly; then one of its destinations,
    uniformly), every third state a B8 randomised board instead (board.py:61-85)."""
    dev = 'cuda'
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED)
    start = np.tile(np.array([42, 35, 43, 28, 36, 44, 6, 13, 5, 20, 12, 4], dtype=np.uint8), (n_distinct, 1))
    st = rules.to_device_states(_lib.pack_states(start))
    idx = torch.arange(n_distinct, device=dev)
    want = idx % 64
    player = torch.ones(n_distinct, dtype=torch.uint8, device=dev)
    for ply in range(63):
        moves, count, masks = rules.movegen(st, player)
        can = masks != 0                                                     # [n, 6]: checkers with at least one move
        r = torch.rand(n_distinct, generator=gen, device=dev)
        which = (r * can.sum(1)).long().clamp(max=5)                         # the which-th checker that can move
        cid = ((can.cumsum(1) - 1 == which[:, None]) & can).float().argmax(1)
        mine = (moves[:, :, 0].long() == cid[:, None]) & (torch.arange(126, device=dev)[None, :] < count[:, None].long())
        r2 = torch.rand(n_distinct, generator=gen, device=dev)
        nth = (r2 * mine.sum(1)).long()
        j = ((mine.cumsum(1) - 1 == nth[:, None]) & mine).float().argmax(1)
        mv = moves[idx, j].contiguous()
        nxt, winner, _ = rules.step(st, player, mv)
        go = (want > ply) & (winner == 0) & (count > 0)
        st = torch.where(go[:, None], nxt, st).contiguous()
        player = torch.where(go, 3 - player, player).contiguous()
    # every third state: a randomised board
    rng = np.random.RandomState(1)
    cells = np.argsort(rng.rand(n_distinct, 49), axis=1)[:, :12].astype(np.uint8)
    rnd = rules.to_device_states(_lib.pack_states(cells))
    st = torch.where((idx % 3 == 2)[:, None], rnd, st).contiguous()
    return st, player


def extras(eng, G, S, torch, _lib, engine):
    """the other configurations of SURVEY.md §8d and the kernel micro-benchmarks, same run (one GPU)"""
    from chinesecheckersagent_amd import rules
    v = {}
    # config 2b: random-rollout value (BASELINE.json configs[1] wording; no reference counterpart)
    e = engine.SelfPlayEngine(n_slots=G, sims=S, seed=SEED, max_games=G * 8, log_capacity=G * 16, auto_restart=True)
    e.play_plies(_lib.EVAL_ROLLOUT, 6)
    e.play_plies(_lib.EVAL_ROLLOUT, 1)
    torch.cuda.synchronize()
    c0 = e.counters()
    wall, kms, _ = timed_plies(e, _lib.EVAL_ROLLOUT, 2, torch)
    c1 = e.counters()
    v['2b_rollout'] = {'node_expansions_per_s': (c1['expansions'] - c0['expansions']) / wall, 'ms_per_ply': kms,
                       'workload': '%d games x %d sims, v = random playout <= 64 plies, p = 1/294' % (G, S)}
    e.close()
    # config 3 in LOCK-STEP (round 3's delivered mode, still what small batches and the arena use): every slot at the same simulation of
    # the same ply, no tree reuse -- so that the record shows what the free-running path is measured against, and the evaluator's
    # in-pipeline figure when one tree launch per round hides under the other half-batch's evaluator launch
    if weights_path() and G >= 2048:
        from chinesecheckersagent_amd import selfplay as sp
        from chinesecheckersagent_amd.model import ResidualCNN
        m = ResidualCNN(device='cuda:%d' % torch.cuda.current_device())
        m.load_weights(weights_path())
        sink = sp.TrainDataSink(); sink.discard = True
        run = sp.SelfPlayRun(m, n_games=G * 64, sims=S, seed=SEED, max_slots=G, keep_records=False, sink=sink, free_running=False, off_path_ok=True)
        try:
            for _ in range(80):                            # (untimed plies: the lock-step cohort plays in step, its expansions/s do not depend on the spread)
                run.play_ply()
            run.drain()
            torch.cuda.synchronize()
            c0 = run.counters(); t0 = time.time()
            for _ in range(8):
                run.play_ply()
            torch.cuda.synchronize()
            wall = time.time() - t0
            c1 = run.counters()
            parts = len(run.b.parts) if hasattr(run.b, 'parts') else 1
            launches = 8 * (S + 1) * parts
            v['3_lock_step'] = {'node_expansions_per_s': (c1['expansions'] - c0['expansions']) / wall, 'ms_per_step': wall / 8 * 1e3,
                                'games_per_s': (c1['games_won'] + c1['games_discarded'] - c0['games_won'] - c0['games_discarded']) / wall,
                                'evaluator_wall_ms_per_launch': wall / launches * 1e3,
                                'evaluator_wall_frac': (run.n_slots // parts) * NET_FLOP_PER_EVAL / (wall / launches) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                'workload': 'config 3 with lock-step slots (SelfPlayRun(free_running=False)): 80 untimed + 8 timed plies, no conversion sink'}
        finally:
            run.close()
    # games/s with a TABLE evaluator under which games END in wins (the uniform one of 2a never wins: every game is discarded
    # by the no-progress rule): spec.forward_eval, parity-pinned like 2a; steady state with restarts
    e = engine.SelfPlayEngine(n_slots=G, sims=S, seed=SEED, max_games=G * 64, log_capacity=G * 160, auto_restart=True)
    e.play_plies(_lib.EVAL_FORWARD, 70)                  # past the first wave of finishing games
    torch.cuda.synchronize()
    c0 = e.counters()
    t0 = time.time()
    e.play_plies(_lib.EVAL_FORWARD, 70)
    torch.cuda.synchronize()
    wall = time.time() - t0
    c1 = e.counters()
    e.close()
    won = c1['games_won'] - c0['games_won']
    v['2a_forward_games'] = {'games_per_s': (won + c1['games_discarded'] - c0['games_discarded']) / wall, 'games_won_per_s': won / wall,
                             'node_expansions_per_s': (c1['expansions'] - c0['expansions']) / wall,
                             'samples_per_s': (c1['samples'] - c0['samples']) / wall,
                             'workload': '%d games x %d sims, table evaluator spec.forward_eval (games end in wins), restarts, 70 plies timed' % (G, S)}
    # kernel micro-benchmarks: 2^23 states so the working set exceeds the 256 MiB Infinity Cache;
    # 2^16 distinct positions of the prescribed distribution, tiled
    n = 1 << 23
    sd0, pl0 = s1_positions(1 << 16, torch, rules, _lib)
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
    dt = t(lambda: L.ccsp_movegen(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), sp_))
    v['movegen_kernel'] = {'states_per_s': n / dt, 'mean_moves': kmean, 'alg_bytes_per_state': 80 + 2 * kmean,
                           'achieved_GBps': n * (80 + 2 * kmean) / dt / 1e9, 'frac_of_hbm_peak': n * (80 + 2 * kmean) / dt / 1e9 / HBM_PEAK_GBPS,
                           'inputs': 'SURVEY 8d: Board() advanced by i mod 64 S1 plies, every third a randomised board; 2^16 distinct, tiled to 2^23'}
    dt = t(lambda: L.ccsp_movegen_packed(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), sp_))
    v['movegen_kernel_packed'] = {'states_per_s': n / dt, 'alg_bytes_per_state': 80 + 2 * kmean, 'achieved_GBps': n * (80 + 2 * kmean) / dt / 1e9,
                                  'frac_of_hbm_peak': n * (80 + 2 * kmean) / dt / 1e9 / HBM_PEAK_GBPS,
                                  'layout': 'ccsp_movegen_packed: the lists of each chunk of 32 positions back to back (whole sectors written)'}
    moves, count, masks = rules.movegen(sd, player)              # (back to the row layout for what follows)
    mv = moves[:, 0, :].contiguous()
    nxt = torch.empty_like(sd); w = torch.zeros(n, dtype=torch.uint8, device='cuda'); pr = torch.zeros((n, 2), dtype=torch.uint8, device='cuda')
    dt = t(lambda: L.ccsp_step(sd.data_ptr(), player.data_ptr(), mv.data_ptr(), n, nxt.data_ptr(), w.data_ptr(), pr.data_ptr(), sp_))
    v['step_kernel'] = {'states_per_s': n / dt, 'achieved_GBps': n * 70 / dt / 1e9, 'frac_of_hbm_peak': n * 70 / dt / 1e9 / HBM_PEAK_GBPS}
    del moves, masks, nxt
    ne = 1 << 21
    planes = torch.empty((ne, 343), dtype=torch.float32, device='cuda')
    dt = t(lambda: L.ccsp_encode(sd.data_ptr(), player.data_ptr(), ne, planes.data_ptr(), sp_))
    v['encode_kernel'] = {'states_per_s': ne / dt, 'achieved_GBps': ne * 1405 / dt / 1e9, 'frac_of_hbm_peak': ne * 1405 / dt / 1e9 / HBM_PEAK_GBPS}
    del planes
    # next-4: the greedy policy over the same positions (32 B + 1 B read, 64 B + 1 B written per position) ...
    ng = 1 << 21
    best = torch.zeros((ng, _lib.GREEDY_MAX, 2), dtype=torch.uint8, device='cuda')
    cnt = torch.zeros(ng, dtype=torch.uint8, device='cuda')
    dt = t(lambda: L.ccsp_greedy_best(sd.data_ptr(), player.data_ptr(), ng, best.data_ptr(), cnt.data_ptr(), sp_))
    v['greedy_best_kernel'] = {'states_per_s': ng / dt, 'achieved_GBps': ng * 98 / dt / 1e9, 'frac_of_hbm_peak': ng * 98 / dt / 1e9 / HBM_PEAK_GBPS}
    del best, cnt, sd
    # ... and the greedy data generator: whole games, one sample row (32 B state + 16 B meta + 2352 B pi) per ply
    gs = 16384
    e = engine.SelfPlayEngine(n_slots=gs, sims=1, seed=SEED, max_games=gs * 64, log_capacity=gs * 300, auto_restart=True,
                              greedy_data=True)
    e.play_plies(0, 32)
    torch.cuda.synchronize()
    c0 = e.counters()
    t0 = time.time()
    for _ in range(4):
        e.play_plies(0, 64)
    torch.cuda.synchronize()
    wall = time.time() - t0
    c1 = e.counters()
    e.close()
    plies = c1['plies'] - c0['plies']
    try:                                          # HBM bytes per state from the committed rocprofv3 counter passes (static)
        prof = json.load(open(os.path.join(ROOT, 'profiles', 'counters.json')))
        for name, key in (('movegen_kernel', 'movegen_kernel<false>'), ('movegen_kernel_packed', 'movegen_kernel<false, true>'),
                          ('step_kernel', 'step_kernel'), ('encode_kernel', 'encode_kernel'), ('greedy_best_kernel', 'movegen_kernel<true>')):
            if key not in prof:
                continue
            v[name]['hbm_bytes_per_state_counters'] = prof[key]['hbm_bytes_per_state']
            v[name]['counters_source'] = 'static: profiles/counters.json (2 x FETCH_SIZE + WRITE_SIZE, n = %d)' % prof[key]['n']
    except Exception:
        pass
    v['greedy_data_generator'] = {'games_per_s': (c1['games_won'] + c1['games_discarded'] - c0['games_won'] - c0['games_discarded']) / wall,
                                  'plies_per_s': plies / wall, 'samples_per_s': (c1['samples'] - c0['samples']) / wall,
                                  'achieved_GBps': plies * 2400 / wall / 1e9, 'frac_of_hbm_peak': plies * 2400 / wall / 1e9 / HBM_PEAK_GBPS,
                                  'errors': c1['errors'],
                                  'workload': '%d concurrent greedy-vs-greedy generator games, 256 plies per slot in 4 launches' % gs}
    return v


if __name__ == '__main__':
    main()
