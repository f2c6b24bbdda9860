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
ap.add_argument('--parts', type=int, default=0, help='half-batches (0 = SelfPlayRun decides)')
ap.add_argument('--deadlines', default='', help='comma-separated deadlines in microseconds (0 = none)')
ap.add_argument('--time-caps', default='', help='comma-separated caps in microseconds (0 = none): every budget is run with every cap')
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
caps = [float(c) for c in a.time_caps.split(',') if c] or [None]
dls = [float(c) for c in a.deadlines.split(',') if c] or [None]
cases += [('free', int(b), c, d) for b in a.budgets.split(',') if b for c in caps for d in dls]
for kind, budget, *rest in cases:
    cap = rest[0] if rest else None
    dl = rest[1] if rest else None
    if dl is not None:
        L.ccsp_debug_advance_deadline(int(dl * 100))
    if budget is not None:
        L.ccsp_debug_advance_budget(budget)
    if cap is not None:
        L.ccsp_debug_advance_time_cap(int(cap * 100))
    sink = sp.TrainDataSink(); sink.discard = True
    run = sp.SelfPlayRun(m, n_games=a.games * 64, sims=a.sims, seed=20261003, max_slots=a.games, keep_records=False, sink=sink,
                         free_running=(kind == 'free'), reuse=(False if a.no_reuse else None), harvest_every=a.harvest, n_parts=(a.parts or None))
    for _ in range(a.spread):
        run.play_ply()
    run.drain()
    torch.cuda.synchronize()
    raw0 = [sum(x) for x in zip(*[b.eng.raw_counters() for b in getattr(run.b, 'parts', [run.b])])] if a.debug else None
    if a.debug:
        [b.eng.debug_read(True) for b in getattr(run.b, 'parts', [run.b])]
    for b_ in getattr(run.b, 'parts', [run.b]):
        b_.net_events = []
    c0 = run.counters(); t0 = time.time()
    for _ in range(a.plies):
        run.play_ply()
    torch.cuda.synchronize()
    dt = time.time() - t0
    c1 = run.counters()
    run.drain()
    evs = [ev for b_ in getattr(run.b, 'parts', [run.b]) for ev in (b_.net_events or [])]
    if evs:
        nets = sorted(e[0].elapsed_time(e[1]) for e in evs); trees = sorted(e[1].elapsed_time(e[2]) for e in evs)
        print('in situ (uncaptured rounds, %d samples): net median %.1f us (min %.1f max %.1f), tree kernels median %.1f us (min %.1f max %.1f)' % (len(evs), nets[len(nets)//2]*1e3, nets[0]*1e3, nets[-1]*1e3, trees[len(trees)//2]*1e3, trees[0]*1e3, trees[-1]*1e3), flush=True)
    d = {k: c1[k] - c0[k] for k in c1}
    evals = a.plies * (a.sims + 1) * a.games
    print(json.dumps(dict(kind=kind, budget=budget, cap_us=cap, deadline_us=dl, exp_per_s=d['expansions'] / dt, ms_per_play_ply=dt / a.plies * 1e3,
                          hit_rate=d['cache_hits'] / max(d['expansions'], 1), terminal_share=d['terminal_sims'] / max(d['sims'], 1),
                          plies_per_slot_per_play_ply=d['mcts_plies'] / a.plies / a.games, useful_eval_share=(d['expansions'] - d['cache_hits']) / evals,
                          games_per_s=(d['games_won'] + d['games_discarded']) / dt, errors=c1['errors'])), flush=True)
    if a.debug:
        raw1 = [sum(x) for x in zip(*[b.eng.raw_counters() for b in getattr(run.b, 'parts', [run.b])])]
        calls = a.plies * (a.sims + 1) * a.games
        import numpy as np
        for pi_, b_ in enumerate(getattr(run.b, 'parts', [run.b])):          # the LAST advance launch of each half-batch: when its waves began and ended (100 MHz clock)
            sl_ = b_.eng.debug_read_slots()
            ok = sl_[:, 17] > 0
            if ok.sum() < 16:
                continue
            b64, e64 = sl_[ok, 16].astype(np.int64), sl_[ok, 17].astype(np.int64)
            near = np.abs(b64 - int(np.median(b64))) < 100000          # within a millisecond of the median: this launch (other slots' last call was an earlier one)
            b64, e64 = b64[near], e64[near]
            beg, end = (b64 - b64.min()) * 1e-2, (e64 - b64.min()) * 1e-2
            sp_ = (sl_[ok, 18][near] & 0xFFFFFFFF).astype(np.int64)
            for k_ in range(4):
                m_ = sp_ == k_
                if m_.sum():
                    print('   %d free simulations: %d waves, own time mean %.1f max %.1f us, end max %.1f us' % (k_, m_.sum(), (end - beg)[m_].mean(), (end - beg)[m_].max(), end[m_].max()))
            q = lambda a, f: float(np.quantile(a, f))
            print('half %d, last launch, %d waves: begin p50 %.1f p90 %.1f p99 %.1f max %.1f us | end p50 %.1f p90 %.1f p99 %.1f max %.1f us | own time p50 %.1f p90 %.1f p99 %.1f max %.1f us'
                  % (pi_, len(beg), q(beg, .5), q(beg, .9), q(beg, .99), beg.max(), q(end, .5), q(end, .9), q(end, .99), end.max(),
                     q(end - beg, .5), q(end - beg, .9), q(end - beg, .99), (end - beg).max()), flush=True)
        dg = [sum(x) for x in zip(*[b.eng.debug_read(True) for b in getattr(run.b, 'parts', [run.b])])]
        if dg[6]:
            tick = dg[10] * 1e-2 / max(dg[5], 1)      # microseconds per s_memtime tick, from s_memrealtime (100 MHz) over the same calls
            print('advance_kernel per call and wave (us): setup %.1f  expansion %.1f  backup %.1f  selection %.1f  encode %.1f  total %.1f (max %.1f); per call %.2f expansions, %.2f selections' % tuple([dg[i] / dg[6] * tick for i in (0, 1, 2, 3, 4, 5)] + [dg[9] * tick, dg[7] / dg[6], dg[8] / dg[6]]), flush=True)
        if dg[6]:
            print('calls by evaluator-free simulations completed (0 / 1 / 2 / 3+): share %s, mean time %s us' % (
                ' / '.join('%.3f' % (dg[16 + k] / dg[6]) for k in range(4)), ' / '.join('%.1f' % (dg[11 + k] / max(dg[16 + k], 1) * tick) for k in range(4))), flush=True)
        print(json.dumps(dict(requests_share=(raw1[12] - raw0[12]) / calls, log_guard_waits=(raw1[13] - raw0[13]), budget_idle_share=(raw1[14] - raw0[14]) / calls)), flush=True)
    run.close()
