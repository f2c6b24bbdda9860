"""BASELINE configs 4 and 5 at their per-rank SHAPE on the one device of a GPU box (the record behind profiles/r6_config4_one_device.txt).
This process never touches the GPU; it starts
  (1) selfplay.generate_self_play_in_parallel with `ranks` rank processes on device 0 (gloo summary), each 4096 free-running slots x 400
      simulations with good_model.h5, for `steps` steps (a bounded rehearsal: the full shape plays for minutes on a shared device), rows
      streamed into the ranks' files;
  (2) `CCSP_BENCH_ONE_DEVICE=1 python bench.py --gpus <ranks> --games 512` with the driver's other defaults: wall time against its timeout.
The pool admits SIX processes on a card at once (process guard): `ranks` <= 6, BASELINE's eight cannot be started on a 1-GPU box.
    python tools/config4_one_device.py [ranks=6] [steps=40] [bench: 1|0]"""
import json, os, resource, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
from chinesecheckersagent_amd import selfplay as sp
import bench

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
with_bench = (sys.argv[3] if len(sys.argv) > 3 else '1') == '1'
assert ranks <= 6, 'process guard of the GPU boxes: at most 6 processes on the card'
G, S = 4096, 400
cores = bench.usable_cores()
out_dir = tempfile.mkdtemp(prefix='ccsp-config4-')
t0 = time.time()
(bx, py, vy, gid), summ = sp.generate_self_play_in_parallel('tests/golden/good_model.h5', ranks * G * 2, ranks, sims=S, seed=20261003, first_game=0,
                                                            devices=[0] * ranks, as_arrays=True, return_summary=True, out_dir=out_dir, max_slots=G,
                                                            timeout=1000, max_steps=steps, with_games=True)
dt = time.time() - t0
c = summ['counters']
print('config 4 shape: %d rank processes on device 0 x %d free-running slots x %d sims, %d steps each: %.1f s wall (process start, model load, capture and '
      'the merge included); usable host cores %d' % (ranks, G, S, steps, dt, cores))
print('all ranks together: node expansions %d (%.2f M/s over the call), reused positions %d (%.3f), games ended %d (won %d, discarded %d), errors %d; '
      'visit histogram sum %d = mcts plies %d x %d: %s; backend %s'
      % (c['expansions'], c['expansions'] / dt / 1e6, c.get('cache_hits', 0), c.get('cache_hits', 0) / max(c['expansions'], 1),
         c['games_won'] + c['games_discarded'], c['games_won'], c['games_discarded'], c['errors'], sum(summ['visit_histogram']), c['mcts_plies'], S,
         sum(summ['visit_histogram']) == c['mcts_plies'] * S, summ['backend']))
for r in range(ranks):
    h = json.load(open(os.path.join(out_dir, 'host-rank%d.json' % r)))
    p = h['path']
    print('rank %d: wall %.1f s, host CPU %.1f s = %.2f cores busy, %d rows streamed, peak RSS %.0f MB; free-running %s, reuse %s, graphs %s, %d slots in %d half-batches, '
          '%d steps, %d games ended' % (r, h['wall_s'], h['host_cpu_s'], h['host_cpu_s'] / h['wall_s'], h['rows'] or 0, h['peak_rss_mb'], p['free_running'],
                                        p['reuse'], p['graphs'], p['n_slots'], p['half_batches'], p['steps'], p['finished']))
print('merged by the parent: %d training rows of %d games, parent peak RSS %.1f GB' % (len(vy), len(set(gid.tolist())), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
print('host budget: %d usable cores / %d ranks = %.2f per rank (a rank needs ~1.1: main thread + converter thread; eight ranks on the box\'s 16: 2.0 each)'
      % (cores, ranks, cores / ranks))
if with_bench:
    env = dict({k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}, CCSP_BENCH_ONE_DEVICE='1')
    t0 = time.time()
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', str(ranks), '--games', '512'], env=env, capture_output=True, text=True, timeout=1500)
    dt = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    print('bench.py --gpus %d --games 512 (all ranks on device 0, gloo; every other flag the driver\'s default): exit %d, %d JSON line(s), wall %.1f s'
          % (ranks, r.returncode, len(lines), dt))
    if lines:
        d = json.loads(lines[0])
        c5 = d.get('config5') or {}
        print('  n_gpus %d, per_rank_expansions %s, degraded %s, value %.2f M node-expansions/s (shared device), steps %d, host cores per rank %s'
              % (d['n_gpus'], d['per_rank_expansions'], d['degraded'], d['value'] / 1e6, d['steps'], ['%.2f' % x for x in d['measured']['host_cpu_cores_busy_per_rank']]))
        print('  config 5 (N-rank loop: sharded self-play, DDP fit, sharded arena): %s' % ({k: (round(v, 2) if isinstance(v, float) else v) for k, v in c5.items() if k != 'workload'}))
    else:
        print(r.stderr[-2000:])
