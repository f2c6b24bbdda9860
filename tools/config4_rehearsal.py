"""Rehearsal of BASELINE config 4's CODE PATH on whatever GPUs are there: selfplay.generate_self_play_in_parallel with `workers` rank
processes (on a 1-GPU box all of them on device 0, gloo carrying the summary instead of RCCL) playing `games` games at 400 simulations
with good_model.h5, the parent merging the ranks' rows into (board_x, pi_y, v_y).  Prints wall time, games, rows, the all-reduced
summary and host memory.  usage: python tools/config4_rehearsal.py [games] [workers]"""
import os, resource, sys, time
sys.path.insert(0, '.')
from chinesecheckersagent_amd import selfplay as sp

games = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
import torch
ndev = torch.cuda.device_count()                       # (counting devices does not initialise the GPU)
devices = None if ndev >= workers else [0] * workers
t0 = time.time()
(bx, py, vy), summ = sp.generate_self_play_in_parallel('tests/golden/good_model.h5', games, workers, sims=400, seed=20261003, first_game=0,
                                                       devices=devices, as_arrays=True, return_summary=True)
dt = time.time() - t0
c = summ['counters']
print('%d games over %d rank processes (%s) in %.1f s = %.1f games/s incl. process start, tails and the merge; backend %s'
      % (games, workers, 'one per GPU' if devices is None else 'all on device 0', dt, games / dt, summ['backend']))
print('won %d, discarded %d, errors %d; node expansions %d; visit histogram sum %d' %
      (c['games_won'], c['games_discarded'], c['errors'], c['expansions'], sum(summ['visit_histogram'])))
print('training rows %d: board_x %s %s, pi_y %s, v_y %s (sum %d)' % (len(vy), bx.shape, bx.dtype, py.shape, vy.dtype, int(vy.sum())))
print('parent peak RSS %.1f GB' % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
