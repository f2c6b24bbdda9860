"""One CPU-baseline worker process of bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE: never imported by the
product).  Plays whole searches with the oracle -- `--kind c`: oracle/ccsp_oracle.c through tests/oracle_ffi.py;
`--kind py`: the reference-shaped pure-Python mirror oracle/pymirror.py -- on game ids first, first + stride, ...
from --start-at until --seconds later (the call in flight is finished), then prints one JSON line.  One process per
core, games sharded by id: the reference's own worker scheme (train.py:73-86)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--kind', choices=['c', 'py'], required=True)
ap.add_argument('--seconds', type=float, required=True)
ap.add_argument('--sims', type=int, default=400)
ap.add_argument('--plies', type=int, default=16)
ap.add_argument('--first', type=int, default=0)
ap.add_argument('--stride', type=int, default=1)
ap.add_argument('--start-at', type=float, default=0.0)
a = ap.parse_args()
SEED = 20261003
if a.kind == 'c':
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_ffi
    L = oracle_ffi.lib()

    def play(g):
        return L.orc_bench_plies(SEED, g, a.sims, 0, a.plies)
else:
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import pymirror

    def play(g):
        return pymirror.bench_plies(SEED, g, a.sims, a.plies)
while time.time() < a.start_at:
    time.sleep(0.005)
t_stop = max(a.start_at, time.time()) + a.seconds
done, games, g = 0, 0, a.first
while time.time() < t_stop:
    done += play(g)
    games += 1
    g += a.stride
print(json.dumps({'expansions': int(done), 'games': games, 't_end': time.time()}))
