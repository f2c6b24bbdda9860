"""Rehearsal of the HOST side of BASELINE config 4 on a 1-GPU box: `workers` rank processes of selfplay.generate_self_play_in_parallel on
device 0 (gloo summary), `slots` game slots each, 400 simulations, good_model.h5, every rank streaming its finished games' rows into its
training file (--arrays).  The GPU is shared, so a rank produces rows at 1 / workers of a GPU's rate: what this shows is the per-rank FIXED
host cost (main thread, harvests, the converter thread's bookkeeping) and the host cost PER ROW, from which the load of N ranks with a GPU
each follows:  cores busy per rank at full rate = fixed cores + (host CPU-seconds per row) x (rows/s of one GPU).
(The GPU boxes allow at most 6 processes on the card: `workers` <= 6.)
    python tools/host_rehearsal.py [workers=6] [slots=512] [games per rank=1024]"""
import json, os, resource, sys, tempfile, time
sys.path.insert(0, '.')
from chinesecheckersagent_amd import selfplay as sp
import bench

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 512
per_rank = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
assert workers <= 6, 'process guard of the GPU boxes: at most 6 processes on the card'
out_dir = tempfile.mkdtemp(prefix='ccsp-host-rehearsal-')
t0 = time.time()
(bx, py, vy), summ = sp.generate_self_play_in_parallel('tests/golden/good_model.h5', per_rank * workers, workers, sims=400, seed=20261003, first_game=0,
                                                       devices=[0] * workers, as_arrays=True, return_summary=True, out_dir=out_dir, max_slots=slots,
                                                       timeout=1000)
dt = time.time() - t0
c = summ['counters']
print('%d games over %d rank processes on device 0, %d slots each: %.1f s; usable cores %d' % (per_rank * workers, workers, slots, dt, bench.usable_cores()))
print('won %d, discarded %d, errors %d; node expansions %d (%.2f M/s all ranks together); reused positions %d'
      % (c['games_won'], c['games_discarded'], c['errors'], c['expansions'], c['expansions'] / dt / 1e6, c.get('cache_hits', 0)))
rows_total = 0
for r in range(workers):
    h = json.load(open(os.path.join(out_dir, 'host-rank%d.json' % r)))
    rows_total += h['rows'] or 0
    print('rank %d: wall %.1f s, host CPU %.1f s = %.2f cores busy, %d rows = %.0f rows/s, %.1f ms host CPU per 1000 rows (ALL host work charged to rows), peak RSS %.0f MB'
          % (r, h['wall_s'], h['host_cpu_s'], h['host_cpu_s'] / h['wall_s'], h['rows'] or 0, (h['rows'] or 0) / h['wall_s'],
             1e6 * h['host_cpu_s'] / max(h['rows'] or 1, 1), h['peak_rss_mb']))
print('merged by the parent: %d training rows (%s), parent peak RSS %.1f GB' % (len(vy), bx.dtype, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
