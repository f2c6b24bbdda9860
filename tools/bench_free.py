"""Development aid: config 3 through SelfPlayRun (free-running slots, tree reuse) for several budgets of evaluator-free simulations
per call, against the lock-step form: node-expansions/s, cache hit rate, plies per `play_ply`.
    python tools/bench_free.py [--plies 12] [--spread 40] [--budgets 2,4,8,16] [--lockstep]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesecheckersagent_amd import _lib, selfplay as sp
from chinesecheckersagent_amd.model import ResidualCNN

ap = argparse.ArgumentParser()
ap.add_argument('--plies', type=int, default=12)
ap.add_argument('--spread', type=int, default=40)
ap.add_argument('--budgets', default='2,4,8,16')
ap.add_argument('--games', type=int, default=4096)
ap.add_argument('--sims', type=int, default=400)
ap.add_argument('--lockstep', action='store_true')
ap.add_argument('--no-reuse', action='store_true')
ap.add_argument('--side', action='store_true')
ap.add_argument('--debug', action='store_true')
ap.add_argument('--boundary-every', type=int, default=0, help='0 = the class default')
ap.add_argument('--harvest', type=int, default=4)
ap.add_argument('--unroll', type=int, default=25)
a = ap.parse_args()
os.environ['CCSP_STRICT'] = '1'
sp.BatchSelfPlay.SIDE_STREAM = a.side
sp.BatchSelfPlay.DEBUG = a.debug
if a.boundary_every:
    sp.BatchSelfPlay.BOUNDARY_EVERY = a.boundary_every
sp.BatchSelfPlay.FREE_UNROLL = a.unroll
m = ResidualCNN()
m.load_weights('tests/golden/good_model.h5')
L = _lib.lib()
cases = [('lockstep', None)] if a.lockstep else []
cases += [('free', int(b)) for b in a.budgets.split(',') if b]
for kind, budget in cases:
    if budget is not None:
        L.ccsp_debug_advance_budget(budget)
    sink = sp.TrainDataSink(); sink.discard = True
    run = sp.SelfPlayRun(m, n_games=a.games * 64, sims=a.sims, seed=20261003, max_slots=a.games, keep_records=False, sink=sink,
                         free_running=(kind == 'free'), reuse=(False if a.no_reuse else None), harvest_every=a.harvest)
    for _ in range(a.spread):
        run.play_ply()
    run.drain()
    torch.cuda.synchronize()
    raw0 = [sum(x) for x in zip(*[b.eng.raw_counters() for b in run.b.parts])] if a.debug else None
    if a.debug:
        [b.eng.debug_read(True) for b in run.b.parts]
    for b_ in run.b.parts:
        b_.net_events = []
    c0 = run.counters(); t0 = time.time()
    for _ in range(a.plies):
        run.play_ply()
    torch.cuda.synchronize()
    dt = time.time() - t0
    c1 = run.counters()
    run.drain()
    evs = [ev for b_ in run.b.parts for ev in (b_.net_events or [])]
    if evs:
        nets = sorted(e[0].elapsed_time(e[1]) for e in evs); trees = sorted(e[1].elapsed_time(e[2]) for e in evs)
        print('in situ (uncaptured rounds, %d samples): net median %.1f us (min %.1f max %.1f), tree kernels median %.1f us (min %.1f max %.1f)' % (len(evs), nets[len(nets)//2]*1e3, nets[0]*1e3, nets[-1]*1e3, trees[len(trees)//2]*1e3, trees[0]*1e3, trees[-1]*1e3), flush=True)
    d = {k: c1[k] - c0[k] for k in c1}
    evals = a.plies * (a.sims + 1) * a.games
    print(json.dumps(dict(kind=kind, budget=budget, exp_per_s=d['expansions'] / dt, ms_per_play_ply=dt / a.plies * 1e3,
                          hit_rate=d['cache_hits'] / max(d['expansions'], 1), terminal_share=d['terminal_sims'] / max(d['sims'], 1),
                          plies_per_slot_per_play_ply=d['mcts_plies'] / a.plies / a.games, useful_eval_share=(d['expansions'] - d['cache_hits']) / evals,
                          games_per_s=(d['games_won'] + d['games_discarded']) / dt, errors=c1['errors'])), flush=True)
    if a.debug:
        raw1 = [sum(x) for x in zip(*[b.eng.raw_counters() for b in run.b.parts])]
        calls = a.plies * (a.sims + 1) * a.games
        dg = [sum(x) for x in zip(*[b.eng.debug_read(True) for b in run.b.parts])]
        if dg[6]:
            tick = 1e-2      # s_memtime ticks at 100 MHz: 10 ns
            print('advance_kernel per call (us, s_memtime at 100 MHz): setup %.1f  expansion %.1f  backup %.1f  selection %.1f  encode %.1f  total %.1f (max %.1f); per call %.2f expansions, %.2f selections' % tuple([dg[i] / dg[6] * tick for i in (0, 1, 2, 3, 4, 5)] + [dg[9] * tick, dg[7] / dg[6], dg[8] / dg[6]]), flush=True)
        print(json.dumps(dict(requests_share=(raw1[12] - raw0[12]) / calls, log_guard_waits=(raw1[13] - raw0[13]), budget_idle_share=(raw1[14] - raw0[14]) / calls)), flush=True)
    run.close()
