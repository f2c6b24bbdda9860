"""A rank process of the N-GPU generator: one MI355X, its share of the game ids (train.generate_self_play's role,
train.py:27-67, with a GPU instead of a CPU core).

    python -m chinesecheckersagent_amd.worker selfplay --model M.h5 --games N --sims S --seed Z --first-game F --out DIR

started by launch.run_ranks with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment.  Rank r of R plays ids
F + j for j = r, r + R, ... < N -- a game's record depends on its id alone, so the union over the ranks is what ONE GPU
would have played.  Every rank leaves DIR/selfplay-rank{r}.npz (the sample rows and results of its games); the ranks'
counters and root visit-count histograms meet in ONE all-reduce (RCCL over xGMI) and rank 0 writes
DIR/summary.json.
"""
import argparse
import json
import os
import sys

import numpy as np


def selfplay_rank(a):
    import torch
    from . import _lib, summary
    from . import selfplay as sp
    from .launch import coll_device, init_rank
    rank, world, local, dist = init_rank()
    _lib.require_gpu()
    model, model2 = sp._load_models(a.model, a.model2, device='cuda:%d' % local)
    n_mine = len(range(rank, a.games, world))
    out = dict(rank=rank, world=world, games=n_mine)
    hist = np.zeros(_lib.NUM_ACTIONS, dtype=np.uint64)
    counters = {}
    import resource
    import time
    t0 = time.time()
    if n_mine > 0:
        # --arrays: the finished games' rows are converted and STREAMED into this rank's file while the GPU plays on (host memory
        # bounded by a chunk, whatever the number of games); without it the rows are kept for the parent's object path (play
        # histories as Python objects: for small counts)
        sink = sp.TrainDataSink(path=os.path.join(a.out, 'selfplay-rank%d.h5' % rank), with_games=True) if a.arrays else None
        run = sp.SelfPlayRun(model, model2, n_games=n_mine, sims=a.sims, seed=a.seed, randomised=a.randomised,
                             first_game=a.first_game + rank, game_stride=world, device=local, max_slots=a.max_slots,
                             keep_records=not a.arrays, sink=sink)
        try:
            run.run(max_plies=a.max_steps if a.max_steps > 0 else None)      # (--max-steps: a bounded rehearsal leaves games unfinished)
            counters = run.counters()
            parts = run.b.parts if hasattr(run.b, 'parts') else [run.b]
            # what this rank really ran on (the N-rank tests assert it: not a fallback)
            out_path = dict(free_running=bool(run.free_running), n_slots=int(run.n_slots), half_batches=len(parts), steps=int(run.plies),
                            reuse=all(bool(getattr(b, 'reuse', False)) for b in parts),
                            graphs=all(getattr(b, '_graph', None) is not None for b in parts),
                            backend=getattr(model, 'backend', None), finished=int(run.store.n_finished()))
            for b in parts:
                hist += b.eng.visit_histogram()
            store = run.store
            if a.arrays:
                out['rows'] = int(sink.rows)
                sink.close()
            else:
                store.take_finished()
                rows = list(store._records)
                if rows:
                    st, meta, pi = (np.concatenate([x[i] for x in rows]) for i in range(3))
                else:
                    st, meta, pi = np.zeros(0, dtype=_lib.STATE_DTYPE), np.zeros(0, dtype=_lib.META_DTYPE), np.zeros((0, _lib.NUM_ACTIONS))
                np.savez(os.path.join(a.out, 'selfplay-rank%d.npz' % rank), state=st, meta=meta, pi=pi, results=store.results,
                         first_game=a.first_game + rank, game_stride=world)
        finally:
            if sink is not None and sink.writer is not None:
                sink.writer.abort()
            run.close()
    ru = resource.getrusage(resource.RUSAGE_SELF)
    host = dict(rank=rank, wall_s=time.time() - t0, host_cpu_s=ru.ru_utime + ru.ru_stime, peak_rss_mb=ru.ru_maxrss / 1024.0,
                rows=out.get('rows'), games=n_mine, counters={k: int(v) for k, v in counters.items()},
                visit_histogram_sum=int(hist.sum()), path=out_path if n_mine > 0 else None)
    with open(os.path.join(a.out, 'host-rank%d.json' % rank), 'w') as f:          # (what the N = 8 host-load rehearsal reads)
        json.dump(host, f)
    tot, hist_all = counters, hist
    if dist is not None:
        tot, hist_all = summary.allreduce_summary(counters, hist, dist, device=coll_device(dist))
        dist.barrier()
    if rank == 0:
        out.update(world=world, counters=tot, visit_histogram=[int(x) for x in hist_all],
                   backend=dist.get_backend() if dist is not None else None)
        with open(os.path.join(a.out, 'summary.json'), 'w') as f:
            json.dump(out, f)
    if dist is not None:
        dist.destroy_process_group()
    torch.cuda.synchronize()


def main(argv=None):
    ap = argparse.ArgumentParser(prog='python -m chinesecheckersagent_amd.worker')
    sub = ap.add_subparsers(dest='task', required=True)
    s = sub.add_parser('selfplay')
    s.add_argument('--model', default=None)
    s.add_argument('--model2', default=None)
    s.add_argument('--games', type=int, required=True)
    s.add_argument('--sims', type=int, required=True)
    s.add_argument('--seed', type=int, required=True)
    s.add_argument('--first-game', type=int, default=0)
    s.add_argument('--max-slots', type=int, default=4096)
    s.add_argument('--max-steps', type=int, default=0, help='stop after this many steps of sims + 1 rounds (0 = play every game to its end)')
    s.add_argument('--randomised', action='store_true')
    s.add_argument('--arrays', action='store_true', help='stream (board_x, pi_y, v_y, game) into DIR/selfplay-rank{r}.h5 instead of keeping the rows')
    s.add_argument('--out', required=True)
    e = sub.add_parser('evolve')
    e.add_argument('--config', required=True, help='JSON file with the arguments of train.evolve')
    a = ap.parse_args(argv)
    if a.task == 'selfplay':
        selfplay_rank(a)
    else:
        from . import train
        train.evolve_rank(a.config)


if __name__ == '__main__':
    main()
    sys.exit(0)
