#!/opt/conda/bin/python3.9
"""G7 widened (round 4): 4096 positions taken from REAL self-play logs -- whole games played by the C oracle
(oracle/ccsp_oracle.c, `orc_selfplay`) with the float64 restatement of the net (oracle/net_oracle.py, weights read by
h5py) as evaluator, one position per call as MCTS.py:93 calls model.predict: opening (random opening plies included in the
history of the planes), middle game, near-win and final positions, both players to move, normal and `randomised` starts,
one-model and two-model games -- and the float64 restatement's logits / values on them for the reference's three weight files.

Writes DATA ONLY to tests/golden/net_wide.npz:
    pos12 u8[4096,12], last u8[4096,4], player u8[4096], planes u8[4096,343]           the positions
    game i32[4096], ply i32[4096], kind u8[4096] (0 normal, 1 randomised, 2 two-model)  where each came from
    sub i32[512]                                  indices of a stratified subset whose full vectors are kept:
    logits_<name> f64[512,294], v_<name> f64[512]                                       full float64 vectors on `sub`
    lsum_<name> f64[4096], labs_<name> f64[4096], v_all_<name> f64[4096]                per-position digests of ALL 4096
        (sum of the 294 logits, sum of |logit|, value): tests recompute the restatement on the GPU box and must land
        on these to 1e-9 before they use it as the checker of the HIP kernel on all 4096 x 294 x 3 logits.
    legal_mass_<name> f64[4096]                   softmax mass on the legal moves of the position (sanity of the graph reading)

Run in the build container (h5py lives in /opt/conda; needs oracle/libccsp_oracle.so: `make -C oracle`):
    cd oracle/harness && /opt/conda/bin/python3.9 gen_net_wide_golden.py [n_workers]
"""
import ctypes as C
import os
import sys
from multiprocessing import Pool

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, '..', '..')
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import net_oracle  # noqa: E402
import oracle_ffi as orc  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
NAMES = ('good_model', 'good_model2', 'version0016-weights')
SEED = 20261004
N_WANT, N_SUB = 4096, 512


def load(name):
    w = {}
    with h5py.File('/root/reference/%s.h5' % name, 'r') as f:
        f.visititems(lambda n, o: w.__setitem__(n, o[...]) if isinstance(o, h5py.Dataset) else None)
    return w


def evaluator(w):
    def cb(planes_p, pos12_p, player, p_out, v_out, user):
        x = np.ctypeslib.as_array(planes_p, shape=(343,)).astype(np.float64).reshape(1, 7, 7, 7)
        lg, v = net_oracle.forward(w, x)
        np.ctypeslib.as_array(p_out, shape=(294,))[:] = net_oracle.softmax64(lg)[0]
        v_out[0] = np.float32(v[0])
    return cb


def play(job):
    """one whole self-play game; returns its searched positions (the rows of play_history)"""
    game, kind, sims = job
    w1 = load(NAMES[game % 3])
    fn1 = orc.EVAL_FN(evaluator(w1))
    L = orc.lib()
    if kind == 2:           # two-model game (selfplay.py:30,59): the oracle's second evaluator slot takes the same callback type
        # orc_selfplay has ONE callback; a two-model game alternates weights by the player to move inside it
        w2 = load(NAMES[(game + 1) % 3])
        e1, e2 = evaluator(w1), evaluator(w2)

        def cb(planes_p, pos12_p, player, p_out, v_out, user):
            (e1 if player == 1 else e2)(planes_p, pos12_p, player, p_out, v_out, user)
        fn1 = orc.EVAL_FN(cb)
    g = orc.selfplay(SEED, game, sims, 4, randomised=(kind == 1), fn=fn1)
    n = len(g['hist_player'])
    return dict(game=game, kind=kind, status=g['status'], pos12=g['hist_pos12'], last=g['hist_last'], player=g['hist_player'],
                ply=np.arange(n, dtype=np.int32))


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    # ~ 70-90 searched plies per game: 72 games give > 5000 rows; 12 simulations per move keep a game at ~ 1 CPU-minute
    jobs = [(g, (0, 0, 1, 2)[g % 4], 12) for g in range(72)]
    with Pool(workers) as pool:
        games = pool.map(play, jobs, chunksize=1)
    rows = []
    for gm in games:
        for i in range(len(gm['player'])):
            rows.append((gm['game'], int(gm['ply'][i]), gm['kind'], gm['pos12'][i], gm['last'][i], int(gm['player'][i]), gm['status']))
    print('games', len(games), 'rows', len(rows), 'statuses', np.bincount([g['status'] for g in games]))
    # distinct positions, then an even spread over (game, ply): every game's opening, middle and end are kept
    seen, uniq = set(), []
    for r in rows:
        key = (bytes(r[3]), bytes(r[4]), r[5])
        if key not in seen:
            seen.add(key)
            uniq.append(r)
    assert len(uniq) >= N_WANT, len(uniq)
    last_of_game = {}
    for i, r in enumerate(uniq):
        last_of_game[r[0]] = i
    keep = set(last_of_game.values())                                       # final (near-win) positions of every game
    keep |= set(i for i, r in enumerate(uniq) if r[1] < 2)                  # the first searched plies (opening)
    rest = [i for i in range(len(uniq)) if i not in keep]
    step = len(rest) / float(N_WANT - len(keep))
    keep |= set(rest[int(j * step)] for j in range(N_WANT - len(keep)))
    sel = sorted(keep)[:N_WANT]
    assert len(sel) == N_WANT
    uniq = [uniq[i] for i in sel]
    pos12 = np.array([r[3] for r in uniq], dtype=np.uint8)
    last = np.array([r[4] for r in uniq], dtype=np.uint8)
    player = np.array([r[5] for r in uniq], dtype=np.uint8)
    planes = np.array([orc.planes(pos12[i], last[i], int(player[i])) for i in range(N_WANT)], dtype=np.uint8)
    out = dict(pos12=pos12, last=last, player=player, planes=planes, game=np.array([r[0] for r in uniq], dtype=np.int32),
               ply=np.array([r[1] for r in uniq], dtype=np.int32), kind=np.array([r[2] for r in uniq], dtype=np.uint8))
    sub = np.arange(0, N_WANT, N_WANT // N_SUB, dtype=np.int32)[:N_SUB]
    out['sub'] = sub
    legal = []
    for i in range(N_WANT):
        mv = orc.movegen(pos12[i], int(player[i]))
        legal.append(np.array([int(a) * 49 + int(b) for a, b in mv], dtype=np.int64))
    x = planes.reshape(-1, 7, 7, 7).astype(np.float64)
    for name in NAMES:
        logits, v = net_oracle.forward(load(name), x)
        p = net_oracle.softmax64(logits)
        out['logits_' + name] = logits[sub]
        out['v_' + name] = v[sub]
        out['lsum_' + name] = logits.sum(axis=1)
        out['labs_' + name] = np.abs(logits).sum(axis=1)
        out['v_all_' + name] = v
        mass = np.array([p[i, legal[i]].sum() for i in range(N_WANT)])
        out['legal_mass_' + name] = mass
        print(name, 'logits range %.3f %.3f' % (logits.min(), logits.max()), 'v range %.3f %.3f' % (v.min(), v.max()),
              'legal mass: mean %.4f min %.4f, share of positions above 0.5: %.4f' % (mass.mean(), mass.min(), (mass > 0.5).mean()))
    print('players', np.bincount(player), 'plies', out['ply'].min(), out['ply'].max(), 'kinds', np.bincount(out['kind']))
    np.savez_compressed(os.path.join(OUT, 'net_wide.npz'), **out)
    print('wrote', os.path.join(OUT, 'net_wide.npz'), os.path.getsize(os.path.join(OUT, 'net_wide.npz')), 'bytes')


if __name__ == '__main__':
    main()
