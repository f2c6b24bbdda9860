#!/opt/conda/bin/python3.9
"""Time the REFERENCE itself (imported read-only from /root/reference through refenv.py) on the benchmark's
workload shape: plies of MCTS search at `sims` simulations with the uniform table evaluator (config 2a), one
process, one core.  Build container only; prints node-expansions/s (= evaluator calls/s, MCTS.py:93).

    cd oracle/harness && /opt/conda/bin/python3.9 time_reference.py [sims] [plies] [--mirror]

--mirror: the same work (same seed, game, plies: identical trees) through oracle/pymirror.py as well, and the ratio
mirror / reference that bench.py's cpu_baseline quotes (pymirror.CALIBRATION).
"""
import sys
import time

import refenv
import spec
from refenv import ctx, ref_board, ref_mcts, ref_selfplay
from gen_golden import SEED, quiet, board_after_random_plies

args = [a for a in sys.argv[1:] if not a.startswith('--')]
sims = int(args[0]) if len(args) > 0 else 400
plies = int(args[1]) if len(args) > 1 else 3
refenv.set_sims(sims)
model = refenv.TableModel(spec.EVAL_UNIFORM)


def time_reference():
    root = board_after_random_plies(123, 6)
    ctx.seed, ctx.game = SEED, 123
    t0 = time.time()
    calls0 = model.calls
    for ply in range(6, 6 + plies):
        ctx.ply = ply
        with quiet():
            root = ref_selfplay.make_move(root, model, 1, [])
    return model.calls - calls0, time.time() - t0


if '--mirror' not in sys.argv:
    n, dt = time_reference()
    print('reference (CPython %d.%d, one core): %d evaluator calls in %.1f s = %.0f node-expansions/s at %d sims/move'
          % (sys.version_info[0], sys.version_info[1], n, dt, n / dt, sims))
else:
    import os
    import statistics
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import pymirror
    ref, mir = [], []
    for _ in range(5):                                   # alternate, so that both see the same machine noise
        n, dt = time_reference()
        ref.append(n / dt)
        t0 = time.time()
        m = pymirror.bench_plies(SEED, 123, sims, plies)
        mir.append(m / (time.time() - t0))
        assert m == n, 'not the same work: %d vs %d evaluator calls' % (m, n)
    r, q = statistics.median(ref), statistics.median(mir)
    print('reference (CPython %d.%d, one core): median %.0f node-expansions/s over 5 runs of %d calls at %d sims/move  %s'
          % (sys.version_info[0], sys.version_info[1], r, n, sims, [round(x) for x in ref]))
    print('mirror    (same interpreter, same core, identical trees): median %.0f  %s' % (q, [round(x) for x in mir]))
    print('mirror / reference = %.3f' % (q / r))
