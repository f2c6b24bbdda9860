"""One CPU-baseline worker process of bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE: never imported by the
product).  Plays whole searches with the oracle -- `--kind c`: oracle/ccsp_oracle.c through tests/oracle_ffi.py;
`--kind py`: the reference-shaped pure-Python mirror oracle/pymirror.py; `--kind c_net` / `py_net`: the same two with the
policy/value net as evaluator, ONE position per call as MCTS.py:93 calls model.predict (c_net: the product's PyTorch module on
the CPU, float32, one thread; py_net: the NumPy float32 restatement oracle/net_oracle.py -- BASELINE config 1's "CPU NumPy") --
on game ids first, first + stride, ...
from --start-at until --seconds later (the call in flight is finished), then prints one JSON line.  One process per
core, games sharded by id: the reference's own worker scheme (train.py:73-86)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--kind', choices=['c', 'py', 'c_net', 'py_net'], required=True)
ap.add_argument('--weights', default=os.path.join(ROOT, 'tests', 'golden', 'good_model.h5'))
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
elif a.kind == 'c_net':
    import ctypes as C
    import numpy as np
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, ROOT)
    import oracle_ffi
    from chinesecheckersagent_amd.model import ResidualCNN
    L = oracle_ffi.lib()
    L.orc_bench_plies_fn.restype = C.c_long
    L.orc_bench_plies_fn.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    net = ResidualCNN(device='cpu', backend='torch')
    net.load_weights(a.weights)

    def _cb(planes_p, pos12_p, player, p_out, v_out, user):
        x = torch.from_numpy(np.ctypeslib.as_array(planes_p, shape=(343,)).astype(np.float32).reshape(1, 7, 7, 7))
        p, v = net.evaluate_batch(x)
        np.ctypeslib.as_array(p_out, shape=(294,))[:] = p[0].numpy()
        v_out[0] = float(v[0])
    _fn = oracle_ffi.EVAL_FN(_cb)

    def play(g):
        return L.orc_bench_plies_fn(SEED, g, a.sims, a.plies, C.cast(_fn, C.c_void_p), None)
else:
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import pymirror
    model = None
    if a.kind == 'py_net':
        sys.path.insert(0, ROOT)
        from chinesecheckersagent_amd.model import read_keras_weights
        model = pymirror.NumpyNetEvaluator(read_keras_weights(a.weights))

    def play(g):
        return pymirror.bench_plies(SEED, g, a.sims, a.plies, model=model, t_stop=t_stop if a.kind == 'py_net' else None)
while time.time() < a.start_at:
    time.sleep(0.005)
t_stop = max(a.start_at, time.time()) + a.seconds
done, games, g = 0, 0, a.first
while time.time() < t_stop:
    done += play(g)
    games += 1
    g += a.stride
print(json.dumps({'expansions': int(done), 'games': games, 't_end': time.time()}))
