"""How many evaluator calls of a ply's search fall on positions the PREVIOUS ply's search already evaluated inside the subtree
of the move that was then played?  (selfplay.make_move returns the chosen child as a fresh root, selfplay.py:130-133: the tree is
discarded and the next ply re-evaluates those positions from scratch.)  CPU measurement with the C oracle + the product's
PyTorch module on the CPU as evaluator, one position per call; a hit = the net input (the 343 planes) was evaluated during the previous
ply's search.  Two figures per ply: `hit_any` (the planes were evaluated anywhere in the previous search: subtree reuse + transpositions)
and the number of calls.

    python tools/cache_hit_rate.py [--games 4] [--sims 400] [--plies 40]
"""
import argparse
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def play(job):
    game, sims, plies, seed = job
    import torch
    torch.set_num_threads(1)
    import oracle_ffi as orc
    from chinesecheckersagent_amd.model import ResidualCNN
    net = ResidualCNN(device='cpu', backend='torch')
    net.load_weights(os.path.join(ROOT, 'tests', 'golden', 'good_model.h5'))
    seen_now = []

    def cb(planes_p, pos12_p, player, p_out, v_out, user):
        pl = np.ctypeslib.as_array(planes_p, shape=(343,))
        seen_now.append(pl.tobytes())
        x = torch.from_numpy(pl.astype(np.float32).reshape(1, 7, 7, 7))
        p, v = net.evaluate_batch(x)
        np.ctypeslib.as_array(p_out, shape=(294,))[:] = p[0].numpy()
        v_out[0] = float(v[0])
    fn = orc.EVAL_FN(cb)
    pos, last, player = orc.initial_pos12(), orc.NO_LAST.copy(), 1
    ply = 0
    for _ in range(6):
        cid, dest = orc.random_move(pos, player, seed, game, ply)
        pos, last, w = orc.step(pos, last, player, cid, dest)
        player, ply = 3 - player, ply + 1
    prev = None
    prev2 = None
    rows = []
    for _ in range(plies):
        del seen_now[:]
        o = orc.search(pos, last, player, seed, game, ply, sims, ply >= 16, 4, fn=fn)
        cur = list(seen_now)
        root_N = sorted([o.N[j] for j in range(o.n_root)], reverse=True)
        chosen = [j for j in range(o.n_root) if o.id[j] == o.chosen_id and o.dest[j] == o.chosen_dest][0]
        if prev is not None:
            hits = sum(1 for k in cur if k in prev)
            hits2 = sum(1 for k in cur if (k not in prev) and prev2 is not None and (k in prev2))
            rows.append(dict(ply=ply, calls=len(cur), hit_any=hits, hit_two_plies_ago_only=hits2, distinct=len(set(cur))))
        rows[-1:] and rows[-1].update(chosen_N=int(o.N[chosen]), top_N=root_N[:3])
        prev2 = prev
        prev = set(cur)
        pos, last, w = orc.step(pos, last, player, o.chosen_id, o.chosen_dest)
        player, ply = 3 - player, ply + 1
        if w:
            break
    return dict(game=game, rows=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--games', type=int, default=4)
    ap.add_argument('--sims', type=int, default=400)
    ap.add_argument('--plies', type=int, default=40)
    ap.add_argument('--seed', type=int, default=20261003)
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    with Pool(min(a.games, os.cpu_count() or 1)) as pool:
        games = pool.map(play, [(g, a.sims, a.plies, a.seed) for g in range(a.games)], chunksize=1)
    calls = sum(r['calls'] for g in games for r in g['rows'])
    hits = sum(r['hit_any'] for g in games for r in g['rows'])
    early = [r for g in games for r in g['rows'] if r['ply'] < 16]
    late = [r for g in games for r in g['rows'] if r['ply'] >= 16]
    hits2 = sum(r.get('hit_two_plies_ago_only', 0) for g in games for r in g['rows'])
    doc = dict(sims=a.sims, games=a.games, calls=calls, hits=hits, hit_rate=hits / max(calls, 1), extra_hit_rate_with_two_plies=hits2 / max(calls, 1),
               hit_rate_tau1=sum(r['hit_any'] for r in early) / max(1, sum(r['calls'] for r in early)),
               hit_rate_tau001=sum(r['hit_any'] for r in late) / max(1, sum(r['calls'] for r in late)),
               per_game=[sum(r['hit_any'] for r in g['rows']) / max(1, sum(r['calls'] for r in g['rows'])) for g in games])
    print(json.dumps(doc))
    if a.json:
        doc['games_rows'] = games
        json.dump(doc, open(a.json, 'w'))


if __name__ == '__main__':
    main()
