"""Soak run of the delivered generator: `n` whole games (default 12 000 = three id cohorts of 4096 slots) at 400 simulations with
good_model.h5 through selfplay.generate_train_data(out_path=...), i.e. SelfPlayRun to the END of its id budget (tail included) with the
rows streamed into the training file.  Prints wall time, games/s over the whole call, rows, file size, the device memory in use while it
runs (hipMemGetInfo through torch) and the host's peak RSS.  usage (GPU box): python tools/soak_generate.py [n_games]"""
import os, resource, sys, threading, time
sys.path.insert(0, '.')
import torch
from chinesecheckersagent_amd import selfplay as sp
from chinesecheckersagent_amd.model import ResidualCNN

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
free0, total = torch.cuda.mem_get_info()
peak = [0]
stop = [False]
def watch():
    while not stop[0]:
        f, _ = torch.cuda.mem_get_info()
        peak[0] = max(peak[0], free0 - f)
        time.sleep(0.5)
t = threading.Thread(target=watch, daemon=True); t.start()
path = '/tmp/soak-data-for-iter-0.h5'
t0 = time.time()
out, summary = sp.generate_train_data(m, n_games=n, sims=400, seed=20261003, out_path=path)
dt = time.time() - t0
stop[0] = True
c = summary['counters']
print('games %d (won %d, discarded %d, errors %d) in %.1f s = %.1f games/s over the whole call (tail included); %d plies of the slowest slot'
      % (n, summary['won'], summary['discarded'], summary['errors'], dt, n / dt, summary['plies']))
print('node expansions %d = %.2f M/s over the whole call' % (c['expansions'], c['expansions'] / dt / 1e6))
print('training rows %d, file %.2f GB (%s)' % (summary['rows'], os.path.getsize(path) / 1e9, path))
print('device memory in use during the run: %.2f GB of %.0f GB; host peak RSS %.2f GB'
      % (peak[0] / 1e9, total / 1e9, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
os.remove(path)
