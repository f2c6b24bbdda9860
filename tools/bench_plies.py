"""Development aid: config 2a with one launch per ply (three kernels) against `--ppl` plies per launch (one kernel).
usage (GPU box): python3 tools/bench_plies.py [--games 4096] [--sims 400] [--plies 16] [--ppl 16]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chinesecheckersagent_amd import _lib, engine

ap = argparse.ArgumentParser()
ap.add_argument('--games', type=int, default=4096); ap.add_argument('--sims', type=int, default=400)
ap.add_argument('--plies', type=int, default=16); ap.add_argument('--ppl', type=int, nargs='+', default=[1, 4, 16])
ap.add_argument('--evaluator', default='UNIFORM')
a = ap.parse_args()
EV = getattr(_lib, 'EVAL_' + a.evaluator)
L = _lib.lib()
ref = None
for ppl in a.ppl:
    L.ccsp_debug_plies_per_launch(ppl)
    e = engine.SelfPlayEngine(n_slots=a.games, sims=a.sims, seed=2024, max_games=a.games * 64, log_capacity=a.games * (a.plies + 16), auto_restart=True)
    e.play_plies(EV, 6); e.play_plies(EV, 2)
    torch.cuda.synchronize(); c0 = e.counters(); t0 = time.time()
    e.play_plies(EV, a.plies)
    torch.cuda.synchronize(); dt = time.time() - t0; c1 = e.counters()
    d = {k: c1[k] - c0[k] for k in c1}
    dig = [e.tree_digest(s) for s in (0, 1, a.games - 1)]
    print('plies/launch %3d: %.3f ms/ply  %.1f M node-expansions/s  expansions %d  games finished %d' %
          (ppl, dt / a.plies * 1e3, d['expansions'] / dt / 1e6, d['expansions'], d['games_won'] + d['games_discarded']), flush=True)
    key = (d['expansions'], d['sum_depth'], d['sum_children'], tuple(map(tuple, dig)) if dig and isinstance(dig[0], (tuple, list)) else tuple(dig), e.visit_histogram().tobytes())
    if ref is None: ref = key
    else: print('    same totals, digests and visit histogram as the first run:', key == ref)
    del e
L.ccsp_debug_plies_per_launch(1)
