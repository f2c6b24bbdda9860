"""How many host cores does a rank burn while the GPU plays?  (development aid)  python tools/host_spin_probe.py [blocking]"""
import ctypes, os, resource, sys, time
sys.path.insert(0, '.')
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else 'default'
if mode == 'blocking':
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    print('hipSetDeviceFlags(hipDeviceScheduleBlockingSync) ->', hip.hipSetDeviceFlags(4))
from chinesecheckersagent_amd import selfplay as sp
from chinesecheckersagent_amd.model import ResidualCNN
m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
sink = sp.TrainDataSink(); sink.discard = True
run = sp.SelfPlayRun(m, n_games=4096 * 64, sims=400, seed=1, max_slots=4096, keep_records=False, sink=sink)
for _ in range(8):
    run.play_ply()
run.drain()
import threading
ru0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.time()
for _ in range(24):
    run.play_ply()
run.drain()
ru1 = resource.getrusage(resource.RUSAGE_SELF); dt = time.time() - t0
print(mode, 'wall %.2f s, user %.2f s, sys %.2f s -> %.2f cores busy; threads %d' % (dt, ru1.ru_utime - ru0.ru_utime, ru1.ru_stime - ru0.ru_stime,
      ((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)) / dt, threading.active_count()))
# per-thread CPU times (Linux)
import glob
for t in sorted(glob.glob('/proc/self/task/*/stat')):
    f = open(t).read().split()
    print(' tid', f[0], f[1], 'utime', int(f[13]) / 100.0, 'stime', int(f[14]) / 100.0)
run.close()
