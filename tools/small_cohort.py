"""Config 5's REAL cohort is small (config.py:57-58: 180 self-play games per iteration; 23 per GPU on eight): is it faster to give every rank
its 23 games, or to play all 180 on one GPU?  A small batch is latency-bound (throughput ~ slots), so eight ranks x 23 slots finish in
the time ONE rank needs for its 23 -- measured here on the one GPU of the box as (a) 180 games in 180 slots and (b) 23 games in 23 slots
(one of eight ranks' share), both at 800 simulations with good_model.h5 through generate_train_data (whole games, tail included).
    python tools/small_cohort.py [games=180] [ranks=8] [sims=800]"""
import sys, time
sys.path.insert(0, '.')
import torch
from chinesecheckersagent_amd import selfplay as sp
from chinesecheckersagent_amd.model import ResidualCNN

games = int(sys.argv[1]) if len(sys.argv) > 1 else 180
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 800
m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
sp.generate_train_data(m, n_games=2, sims=32, seed=1)            # (warm-up: library, allocator, graph capture paths)
rows = []
for label, n, first, stride in (('all %d games on ONE GPU (%d slots)' % (games, games), games, 0, 1),
                                ('one of %d ranks: its share of %d games (ids r, r + %d, ...)' % (ranks, (games + ranks - 1) // ranks, ranks), (games + ranks - 1) // ranks, 0, ranks),
                                ('half the ranks: %d games per rank' % ((games + ranks // 2 - 1) // (ranks // 2)), (games + ranks // 2 - 1) // (ranks // 2), 0, ranks // 2)):
    torch.cuda.synchronize()
    t0 = time.time()
    bx, py, vy, summ = sp.generate_train_data(m, n_games=n, sims=sims, seed=20261003, first_game=first, game_stride=stride)
    torch.cuda.synchronize()
    dt = time.time() - t0
    c = summ['counters']
    rows.append((label, n, dt, c['expansions'], summ['plies']))
    print('%-62s %4d slots: %6.2f s wall, %9d node expansions (%.2f M/s), %d steps of the slowest slot, won %d discarded %d'
          % (label, n, dt, c['expansions'], c['expansions'] / dt / 1e6, summ['plies'], summ['won'], summ['discarded']), flush=True)
one, share = rows[0][2], rows[1][2]
print('=> %d ranks x %d slots finish the cohort in %.2f s (every rank at once) against %.2f s on one GPU: %.2f x faster in wall time, at %d x the GPUs'
      % (ranks, rows[1][1], share, one, one / share, ranks))
