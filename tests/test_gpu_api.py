"""The drop-in Python boundary on the GPU: selfplay() / selfplay_batch() /
generate_self_play() and the (state, pi, z) -> board_x / pi_y / v_y output (row O1), checked against
the reference's own games (tests/golden/games.json)."""
import hashlib
import json

import numpy as np
import pytest

import oracle_ffi as orc
from test_gpu_tree import _decode_planes, _table_eval

pytestmark = pytest.mark.gpu


class TableModel(object):
    """reference-style duck-typed evaluator (MCTS.py:93: predict(x[7,7,7]) -> (p f64[294], v 0-d f32))"""

    def __init__(self, kind):
        self.kind = kind

    def predict(self, x):
        pos12, player = _decode_planes(np.asarray(x, dtype=np.float32).reshape(343))
        p, v = _table_eval(self.kind, pos12, player)
        return p, np.float32(v)


def test_selfplay_matches_reference_game(golden_dir, tmp_path):
    from chinesecheckersagent_amd import selfplay as sp, utils
    from chinesecheckersagent_amd import h5lite
    doc = json.load(open(golden_dir + '/games.json'))
    won = [g for g in doc['games'] if g['status'] == 'won' and not g['randomised']]
    g = min(won, key=lambda x: len(x['plies']))
    hist, reward = sp.selfplay(TableModel(g['evaluator']), sims=g['sims'], seed=doc['seed'], game_id=g['game'])
    assert reward == g['reward'] and len(hist) == len(g['pi_sha'])
    assert [hashlib.sha256(np.asarray(pi, dtype='<f8').tobytes()).hexdigest()[:16] for _, pi in hist] == g['pi_sha']
    # O1 through the reference-shaped host helpers (utils.convert_to_train_data on the Board-like views)
    bx, py, vy = utils.convert_to_train_data([(hist, reward)])
    assert hashlib.sha256(np.array(bx).astype('<f8').tobytes()).hexdigest() == g['o1']['board_x_sha']
    assert hashlib.sha256(np.array(py).astype('<f8').tobytes()).hexdigest() == g['o1']['pi_y_sha']
    assert [int(v) for v in vy] == g['o1']['v_y']
    # and the same planes from the GPU encode kernel
    from chinesecheckersagent_amd import _lib, rules
    pos12 = np.array([[b.checkers_pos[1][i][0] * 7 + b.checkers_pos[1][i][1] for i in range(6)] +
                      [b.checkers_pos[2][i][0] * 7 + b.checkers_pos[2][i][1] for i in range(6)] for b, _ in hist], dtype=np.uint8)
    last = np.array([[255] * (4 - 2 * len(b.hist_moves)) if False else
                     sum([[m[0][0] * 7 + m[0][1], m[1][0] * 7 + m[1][1]] for m in reversed(b.hist_moves)], []) +
                     [255] * (4 - 2 * len(b.hist_moves)) for b, _ in hist], dtype=np.uint8)
    players = np.array([1 + i % 2 for i in range(len(hist))], dtype=np.uint8)
    planes = rules.encode(rules.to_device_states(_lib.pack_states(pos12, last)), players).cpu().numpy()
    assert np.array_equal(planes.astype(np.float64), np.array(bx))
    # the training file the reference's train.py consumes (utils.py:48-56)
    path = utils.save_train_data(bx, py, vy, 7, directory=str(tmp_path))
    back = dict(h5lite.H5File(path).walk())
    assert back['board_x'].shape == (len(vy), 7, 7, 7) and back['pi_y'].shape == (len(vy), 294) and back['v_y'].dtype == np.int64
    # discarded games come back as (None, None) (selfplay.py:45-47)
    d = [x for x in doc['games'] if x['status'] == 'repetition'][0]
    assert sp.selfplay(TableModel(d['evaluator']), sims=d['sims'], seed=doc['seed'], game_id=d['game']) == (None, None)


def test_selfplay_with_the_net(golden_dir):
    """good_model.h5 through PyTorch-ROCm: batch of 4 games at 16 sims; deterministic under a fixed seed,
    independent of batch composition, well-formed output"""
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    b = sp.BatchSelfPlay(m, n_slots=4, sims=16, seed=5, max_games=4, log_capacity=4 * 64)
    for _ in range(10):
        b.play_ply()
    st, meta, pi = b.eng.log()
    assert len(meta) == 16 and b.eng.counters()['errors'] == 0
    assert np.allclose(pi.sum(axis=1), 1.0) and (pi >= 0).all()
    b.close()
    b2 = sp.BatchSelfPlay(m, n_slots=2, sims=16, seed=5, first_game=1, max_games=2, log_capacity=2 * 64)   # games 1, 2 only
    for _ in range(10):
        b2.play_ply()
    st2, meta2, pi2 = b2.eng.log()
    b2.close()
    # the opening is a function of (seed, game id) only: the first searched position of games 1 and 2 is the
    # same in both batches (the whole records are, too: test_sharding_independence_with_the_net)
    first = {int(meta['game'][i]): st[i].tobytes() for i in range(len(meta)) if int(meta['ply'][i]) == 6}
    first2 = {int(meta2['game'][i]): st2[i].tobytes() for i in range(len(meta2)) if int(meta2['ply'][i]) == 6}
    assert first2 == {g: first[g] for g in (1, 2)}


def test_sharding_independence_with_the_net(golden_dir):
    """the fused evaluator is a function of the position alone (tests/test_model.py), so with the NET, too, a game's whole
    record -- every searched position and pi, bit for bit -- is the same whichever batch, slot or shard plays it: 11 games
    in one context == the same ids over two strided contexts (a rank of 2 each) == one game alone"""
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    seed, sims, plies = 9, 24, 14

    def run(n_slots, first, stride):
        b = sp.BatchSelfPlay(m, n_slots=n_slots, sims=sims, seed=seed, first_game=first, game_stride=stride, max_games=n_slots,
                             log_capacity=n_slots * 64)
        for _ in range(plies):
            b.play_ply()
        st, meta, pi = b.eng.log()
        assert b.eng.counters()['errors'] == 0
        b.close()
        out = {}
        for i in np.lexsort((meta['ply'], meta['game'])):
            out.setdefault(int(meta['game'][i]), []).append((int(meta['ply'][i]), st[i].tobytes(), pi[i].tobytes()))
        return out
    whole = run(11, 100, 1)
    assert len(whole) == 11 and all(len(v) == plies - 6 for v in whole.values())
    shards = {}
    shards.update(run(6, 100, 2))
    shards.update(run(5, 101, 2))
    assert shards == whole
    assert run(1, 107, 1)[107] == whole[107]


def test_generate_self_play_signature(golden_dir):
    from chinesecheckersagent_amd import selfplay as sp
    sp.set_seed(11, first_game=0)
    games = sp.generate_self_play(1, golden_dir + '/good_model.h5', 2, sims=8)
    assert isinstance(games, list)
    for hist, reward in games:
        assert reward in (1, -1) and hist[0][1].shape == (294,)


def test_two_models_match_reference(golden_dir):
    """selfplay(model1, model2): model1 searches for player one, model2 for player two (selfplay.py:30,36,59).
    The reference played these games with two different table evaluators; ply count, evaluator calls and the
    outcome must agree."""
    from chinesecheckersagent_amd import selfplay as sp
    doc = json.load(open(golden_dir + '/games.json'))
    two = [g for g in doc['games'] if isinstance(g['evaluator'], list)]
    assert len(two) >= 2
    g = min(two, key=lambda x: x['evals'])
    b = sp.BatchSelfPlay(TableModel(g['evaluator'][0]), TableModel(g['evaluator'][1]), n_slots=1, sims=g['sims'],
                         seed=doc['seed'], first_game=g['game'], max_games=1, log_capacity=1024)
    out = b.run_to_completion()
    res = b.eng.results()[0]
    st, meta, pi = b.eng.log()
    b.close()
    status = {1: 'won', 2: 'won', 3: 'repetition', 4: 'no_progress'}[int(res['status'])]
    assert status == g['status'] and int(res['n_plies']) == len(g['plies']) and int(res['expansions']) == g['evals']
    assert out[0] == ((None, None) if status != 'won' else out[0])
    # every logged position is the one the reference's move list leads to
    pos = orc.initial_pos12()
    last = orc.NO_LAST.copy()
    player, row = 1, 0
    for kind, cid, dest in g['plies']:
        if kind != 0:
            assert [int(x) for x in st[row]['pos'].reshape(12)] == [int(x) for x in pos]
            row += 1
        pos, last, _ = orc.step(pos, last, player, cid, dest)
        player = 3 - player
    assert row == len(meta)


def test_search_with_net_priors_matches_oracle(golden_dir):
    """T2-T4 with REAL priors (SURVEY.md §8d config 1 in spirit): plies searched with good_model.h5 on the GPU
    (stepped path: select kernel -> fused evaluator -> expand/backup kernel, hipGraph replay) against the CPU oracle
    calling back into the same evaluator one position at a time.  Arbitrary float64 priors and float32 values
    exercise every rounding of the PUCT arithmetic (the table evaluators of the golden cases are dyadic)."""
    import torch
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    seed, sims, nply = 77, 50, 5
    for game in (3, 10):
        b = sp.BatchSelfPlay(m, n_slots=1, sims=sims, seed=seed, first_game=game, max_games=1, log_capacity=64)
        for _ in range(6 + nply):
            b.play_ply()
        st, meta, pi = b.eng.log()
        b.close()
        order = np.argsort(meta['ply'])
        st, meta, pi = st[order], meta[order], pi[order]
        calls = [0]

        def cb(planes_p, pos12_p, player, p_out, v_out, user):
            x = np.ctypeslib.as_array(planes_p, shape=(343,)).astype(np.float32).reshape(1, 7, 7, 7)
            p, v = m.evaluate_batch(torch.from_numpy(x).cuda())
            pn = p[0].cpu().numpy()
            for i in range(294):
                p_out[i] = pn[i]
            v_out[0] = float(v[0])
            calls[0] += 1
        fn = orc.EVAL_FN(cb)
        assert len(meta) == nply
        for k in range(nply):                                # every searched ply from the position the GPU logged
            o = orc.search(st[k]['pos'].reshape(12), st[k]['last'], int(meta[k]['player']), seed, game, int(meta[k]['ply']), sims,
                           False, 4, fn=fn)
            assert np.array_equal(pi[k], np.array(o.pi[:])), 'pi of searched ply %d of game %d differs from the oracle' % (k, game)
        assert calls[0] >= nply * 40


def test_config3_full_size_ply_matches_oracle(golden_dir):
    """BASELINE.json config 3 at FULL size on the LOCK-STEP kernels (what the arena, small batches and bench.py's `3_lock_step` variant
    run; the delivered free-running mode at this size is test_full_size_restarting_run_plays_the_oracles_games): 4096 slots x 400
    simulations with good_model.h5, PipelinedSelfPlay (two half-batches on two HIP streams, 25 simulation steps per captured hipGraph,
    fused evaluator kernel on 2048 positions per launch).  One searched ply; 16 rows of the sample log spread over both halves are
    re-searched by the CPU oracle calling back into the same evaluator one position at a time: pi must agree bit for
    bit (400 simulations of float64 PUCT on float32-net priors, and the evaluator's independence of its batch)."""
    import torch
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    seed, sims, G = 20261003, 400, 4096
    b = sp.PipelinedSelfPlay(m, n_slots=G, n_parts=2, sims=sims, seed=seed, log_capacity=G * 4)
    assert all(part.use_graph for part in b.parts)
    for _ in range(7):
        b.play_ply()
    torch.cuda.synchronize()
    assert all(part._graph is not None and part._unroll == 25 for part in b.parts)     # the captured-graph path really ran
    c = b.counters()
    assert c['errors'] == 0 and c['samples'] == G and c['sims'] == G * sims and c['expansions'] + c['terminal_sims'] == G * (sims + 1)
    rows = []
    for part in b.parts:
        st, meta, pi = part.eng.log()
        assert len(meta) == G // 2 and (meta['ply'] == 6).all()
        assert np.abs(pi.sum(axis=1) - 1).max() < 1e-12
        pick = np.argsort(meta['game'])[np.linspace(0, G // 2 - 1, 8).astype(int)]
        rows += [(st[r], meta[r], pi[r]) for r in pick]
    b.close()

    def cb(planes_p, pos12_p, player, p_out, v_out, user):
        x = np.ctypeslib.as_array(planes_p, shape=(343,)).astype(np.float32).reshape(1, 7, 7, 7)
        p, v = m.evaluate_batch(torch.from_numpy(x).cuda())
        np.ctypeslib.as_array(p_out, shape=(294,))[:] = p[0].cpu().numpy()
        v_out[0] = float(v[0])
    fn = orc.EVAL_FN(cb)
    for st, meta, pi in rows:
        o = orc.search(st['pos'].reshape(12), st['last'], int(meta['player']), seed, int(meta['game']), 6, sims, False, 4, fn=fn)
        assert np.array_equal(pi, np.array(o.pi[:])), 'pi of game %d differs from the oracle at full size' % int(meta['game'])


def test_pipelined_halves_equal_standalone_batches(golden_dir):
    """PipelinedSelfPlay (two half-batches on two HIP streams sharing one evaluator) returns, in game-id order, exactly
    what the two halves return when run alone one after the other"""
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    n, sims, seed = 16, 8, 31
    p = sp.PipelinedSelfPlay(m, n_slots=n, n_parts=2, sims=sims, seed=seed, first_game=40, log_capacity=n * 300)
    got = p.run_to_completion(max_plies=400)
    p.close()
    want = [None] * n
    for i in range(2):
        b = sp.BatchSelfPlay(m, n_slots=n // 2, sims=sims, seed=seed, first_game=40 + i, game_stride=2, max_games=n // 2, log_capacity=n * 150)
        out = b.run_to_completion(max_plies=400)
        b.close()
        for j, g in enumerate(out):
            want[2 * j + i] = g
    assert len(got) == n
    kinds = set()
    for a, b_ in zip(got, want):
        assert (a[0] is None) == (b_[0] is None) and a[1] == b_[1]
        kinds.add(a[0] is None)
        if a[0] is not None and not isinstance(a[0], str):
            assert len(a[0]) == len(b_[0])
            for (s1, p1), (s2, p2) in zip(a[0], b_[0]):
                assert s1.pos12 == s2.pos12 and np.array_equal(p1, p2)


def test_collect_train_data_equals_object_path(golden_dir):
    """BatchSelfPlay.collect_train_data() (arrays straight from the sample log) == convert_to_train_data(collect())"""
    from chinesecheckersagent_amd import selfplay as sp, utils
    doc = json.load(open(golden_dir + '/games.json'))
    won = [g for g in doc['games'] if g['status'] == 'won' and not isinstance(g['evaluator'], list)]
    for randomised in (False, True):
        gs = [g for g in won if g['randomised'] == randomised][:1]
        if not gs:
            continue
        g = gs[0]
        b = sp.BatchSelfPlay(TableModel(g['evaluator']), n_slots=3, sims=g['sims'], seed=doc['seed'], first_game=g['game'],
                             max_games=3, randomised=randomised, log_capacity=3 * 400)
        games = b.run_to_completion(max_plies=600)
        bx, py, vy = b.collect_train_data()
        b.close()
        kept = [(h, r) for h, r in games if h is not None and not isinstance(h, str)]
        wx, wp, wv = utils.convert_to_train_data(kept)
        assert len(wx) == len(bx) > 0
        assert (np.array(wx) == bx).all() and (np.array(wp) == py).all() and list(vy) == wv


def test_collect_refuses_a_run_with_dropped_rows(golden_dir):
    """a sample log that is too small: the games that lost a row end in ERROR -- their slots stay out of play, the rest of the id
    budget is NOT turned into ERROR games -- and collect() / collect_train_data() raise instead of handing back histories with
    missing plies (their labels alternate from the first row); allow_errors=True hands the whole games back beside them"""
    from chinesecheckersagent_amd import _lib, selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    b = sp.BatchSelfPlay(m, n_slots=4, sims=8, seed=3, max_games=64, auto_restart=True, log_capacity=6)
    for _ in range(12):
        b.play_ply()
    c = b.eng.counters()
    assert 0 < c['errors'] <= 4                                  # at most one ERROR per slot: no cascade through the id budget
    assert (b.eng.slots()['status'] == _lib.ST_ERROR).sum() == c['errors']
    with pytest.raises(_lib.CcspError, match='log'):
        b.collect()
    with pytest.raises(_lib.CcspError, match='log'):
        b.collect_train_data()
    kinds = [g[0] for g in b.collect(allow_errors=True)]
    assert kinds.count('error') == c['errors']
    b.close()


def _records(games):
    """comparable form of selfplay_batch's list"""
    out = []
    for h, r in games:
        if h is None or isinstance(h, str):
            out.append((h, r))
        else:
            out.append(([(b.pos12, b.last4, pi.tobytes()) for b, pi in h], r))
    return out


def test_restarting_slots_return_the_records_of_one_slot_per_game(golden_dir, tmp_path):
    """The delivered mode (SelfPlayRun: a few restarting slots, the log harvested every few plies and cleared, optionally two
    half-batches, conversion in a worker thread) returns per game id EXACTLY what one slot per game returns: status, reward,
    every searched position and pi.  Also: generate_train_data == convert_to_train_data of that list, row for row."""
    from chinesecheckersagent_amd import _lib, selfplay as sp, utils
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    n, sims, seed, first = 22, 8, 17, 300
    b = sp.BatchSelfPlay(m, n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, log_capacity=n * 600)
    want = _records(b.run_to_completion(max_plies=1100))
    b.close()
    assert all(h != 'unfinished' for h, _ in want)
    assert any(h is None for h, _ in want) and any(isinstance(h, list) for h, _ in want)     # both kinds of ending occur
    for kw in (dict(max_slots=5, harvest_every=3), dict(max_slots=8, harvest_every=16, n_parts=2), dict(max_slots=64, harvest_every=7),
               dict(max_slots=5, harvest_every=3, free_running=True), dict(max_slots=8, harvest_every=2, n_parts=2, free_running=True)):
        run = sp.SelfPlayRun(m, n_games=n, sims=sims, seed=seed, first_game=first, **kw)
        got = _records(run.run().games())
        c = run.counters()
        run.close()
        assert got == want, kw
        assert c['errors'] == 0 and c['games_won'] + c['games_discarded'] == n
    assert _records(sp.selfplay_batch(m, n_games=n, sims=sims, seed=seed, first_game=first, max_slots=6)) == want
    # randomised starts (Board(randomised=True): the first three rows of a won game are dropped, selfplay.py:76-78) and two models
    m2 = ResidualCNN()
    m2.load_weights(golden_dir + '/good_model.h5')
    b = sp.BatchSelfPlay(m, m2, n_slots=9, sims=sims, seed=seed, first_game=first, max_games=9, randomised=True, log_capacity=9 * 600)
    want_r = _records(b.run_to_completion(max_plies=1100))
    b.close()
    assert _records(sp.selfplay_batch(m, m2, n_games=9, sims=sims, seed=seed, first_game=first, randomised=True, max_slots=4, harvest_every=5)) == want_r
    assert any(isinstance(h, list) for h, _ in want_r)
    bx, py, vy, summary = sp.generate_train_data(m, n_games=n, sims=sims, seed=seed, first_game=first, max_slots=6, harvest_every=5)
    kept = sp.selfplay_batch(m, n_games=n, sims=sims, seed=seed, first_game=first)
    kept = [(h, r) for h, r in kept if h is not None]
    wx, wp, wv = utils.convert_to_train_data(kept)
    assert summary['won'] == len(kept) and summary['won'] + summary['discarded'] == n and summary['errors'] == 0
    assert len(vy) == len(wv) > 0 and (np.array(wx) == bx).all() and (np.array(wp) == py).all() and list(vy) == wv
    # streamed into the training file instead: the same rows (games in the order they ended)
    from chinesecheckersagent_amd.h5lite import H5File
    path, summary2 = sp.generate_train_data(m, n_games=n, sims=sims, seed=seed, first_game=first, max_slots=6, harvest_every=5,
                                            out_path=str(tmp_path / 'data-for-iter-0.h5'))
    f = H5File(path)
    fx, fp, fv = f.get('board_x'), f.get('pi_y'), f.get('v_y')
    assert summary2['rows'] == len(fv) == len(vy) and fv.dtype == np.int64
    key = lambda x, p_, v: sorted(zip([a.tobytes() for a in x], [a.tobytes() for a in p_], [int(t) for t in v]))
    assert key(fx, fp, fv) == key(bx, py, vy)


def test_randomised_two_model_games_at_1024_restarting_slots(golden_dir):
    """`randomised=True` two-model games (selfplay.py:11-17, 76-78; board.py:61-85) in the delivered mode at a real batch size:
    1536 game ids through 1024 restarting slots (every third slot plays a second game), harvested every 4 plies, against one slot
    per game -- every game's status, reward, searched positions and pi"""
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m1, m2 = ResidualCNN(), ResidualCNN()
    m1.load_weights(golden_dir + '/good_model.h5')
    m2.load_weights(golden_dir + '/good_model2.h5')
    n, sims, seed, first = 1536, 6, 23, 5000
    b = sp.BatchSelfPlay(m1, m2, n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, randomised=True, log_capacity=n * 700)
    want = _records(b.run_to_completion(max_plies=1100))
    b.close()
    assert all(h != 'unfinished' for h, _ in want) and any(h is None for h, _ in want) and sum(isinstance(h, list) for h, _ in want) > n // 2
    run = sp.SelfPlayRun(m1, m2, n_games=n, sims=sims, seed=seed, first_game=first, randomised=True, max_slots=1024, harvest_every=4)
    got = _records(run.run().games())
    c = run.counters()
    run.close()
    assert c['errors'] == 0 and c['games_won'] + c['games_discarded'] == n
    assert len(got) == len(want) == n
    for k, (a, w) in enumerate(zip(got, want)):
        assert a == w, 'game %d differs' % (first + k)


def test_free_running_slots_play_the_lock_step_games(golden_dir):
    """ccsp_advance (BatchSelfPlay(free_running=True): every slot at its own simulation of its own ply, won leaves backed up without the
    net, positions of the previous ply's tree reused instead of evaluated again) plays EXACTLY the games of the lock-step kernels: status,
    reward, every searched position and pi of 24 games; the same number of expansions, terminal simulations and simulations; whatever the
    budget of evaluator-free simulations per call; with and without tree reuse; on plain launches and on captured graphs; two-model
    games (no reuse: the previous ply was searched with the other model)"""
    from chinesecheckersagent_amd import _lib, selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    L = _lib.lib()
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    n, sims, seed, first = 24, 16, 41, 900
    b = sp.BatchSelfPlay(m, n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, log_capacity=n * 600)
    want = _records(b.run_to_completion(max_plies=1100))
    cw = b.eng.counters()
    b.close()
    assert any(h is None for h, _ in want) and sum(isinstance(h, list) for h, _ in want) > n // 2 and cw['cache_hits'] == 0
    was = L.ccsp_debug_advance_budget(-1), L.ccsp_debug_advance_time_cap(-1), L.ccsp_debug_advance_deadline(-1)
    assert was[0] >= 1 and was[1] > 0 and was[2] > 0          # the defaults: a time cap and a deadline are in force
    try:
        # ... and whatever the time cap on such simulations and the deadline of a call's closing selection (ticks of 10 ns; 1 = passed at
        # every check: no second evaluator-free simulation in a call, every selection that follows other work given up; 0 = none)
        for budget, reuse, graph, cap, deadline in ((4, True, True, 0, 0), (1, True, True, was[1], was[2]), (2, True, False, 0, 1),
                                                    (64, True, True, 1, 1), (4, False, True, 0, 0), (8, True, True, 300, 500),
                                                    (64, True, False, 0, 200)):
            L.ccsp_debug_advance_budget(budget)
            L.ccsp_debug_advance_time_cap(cap)
            L.ccsp_debug_advance_deadline(deadline)
            b = sp.BatchSelfPlay(m, n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, log_capacity=n * 600,
                                 free_running=True, reuse=reuse, use_graph=graph)
            got = _records(b.run_to_completion(max_plies=1100))
            c = b.eng.counters()
            b.close()
            assert got == want, (budget, reuse, graph, cap, deadline)
            for k in ('expansions', 'terminal_sims', 'sims', 'plies', 'mcts_plies', 'games_won', 'games_discarded', 'sum_depth', 'sum_children',
                      'select_edges', 'samples', 'errors'):
                assert c[k] == cw[k], (k, budget, reuse, graph, cap, deadline)
            assert (c['cache_hits'] > 0.1 * c['expansions']) if reuse else c['cache_hits'] == 0, (c['cache_hits'], c['expansions'])
    finally:
        L.ccsp_debug_advance_budget(was[0])
        L.ccsp_debug_advance_time_cap(was[1])
        L.ccsp_debug_advance_deadline(was[2])
    # the slots' first games begin spread over 37 rounds (CCSP_ADVANCE_STAGGER + ccsp_set_stagger_span): the same games
    b = sp.BatchSelfPlay(m, n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, log_capacity=n * 600, free_running=True, reuse=True,
                         stagger=True, stagger_span=37)
    got = _records(b.run_to_completion(max_plies=1100))
    b.close()
    assert got == want
    # ccsp_boundary on a stream of its own beside the next evaluator launch (CCSP_ADVANCE_OVERLAPPED: a root request is then answered by
    # the launch after the next): plain launches and a captured graph with the fork in it -- the same games
    sp.BatchSelfPlay.SIDE_STREAM = True
    try:
        for graph in (False, True):
            b = sp.BatchSelfPlay(m, n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, log_capacity=n * 600, free_running=True, reuse=True,
                                 use_graph=graph)
            got = _records(b.run_to_completion(max_plies=1100))
            b.close()
            assert got == want, ('side stream', graph)
    finally:
        sp.BatchSelfPlay.SIDE_STREAM = False
    # the diagnostic build of the kernel (per-phase stamps, tools/bench_free.py --debug) plays the same games and accounts for every call
    sp.BatchSelfPlay.DEBUG = True
    try:
        b = sp.BatchSelfPlay(m, n_slots=n, sims=sims, seed=seed, first_game=first, max_games=n, log_capacity=n * 600, free_running=True, reuse=True)
        got = _records(b.run_to_completion(max_plies=1100))
        dg, per_slot = b.eng.debug_read(clear=False), b.eng.debug_read_slots()
        b.close()
    finally:
        sp.BatchSelfPlay.DEBUG = False
    assert got == want
    assert dg[6] > 0 and dg[6] == int(per_slot[:, 6].sum()) and dg[16] + dg[17] + dg[18] + dg[19] == dg[6]      # calls = calls by simulations completed
    assert dg[7] >= cw['expansions'] - n * 200 and int(per_slot[:, 17].max()) > int(per_slot[:, 16].min())          # expansions seen; begin < end on the shared clock
    # two models, randomised starts
    m2 = ResidualCNN()
    m2.load_weights(golden_dir + '/good_model2.h5')
    b = sp.BatchSelfPlay(m, m2, n_slots=8, sims=sims, seed=seed, first_game=first, max_games=8, randomised=True, log_capacity=8 * 600)
    want2 = _records(b.run_to_completion(max_plies=1100))
    b.close()
    b = sp.BatchSelfPlay(m, m2, n_slots=8, sims=sims, seed=seed, first_game=first, max_games=8, randomised=True, log_capacity=8 * 600, free_running=True)
    assert b.reuse is False
    got2 = _records(b.run_to_completion(max_plies=1100))
    b.close()
    assert got2 == want2
    with pytest.raises(ValueError):
        sp.BatchSelfPlay(m, m2, n_slots=2, sims=4, free_running=True, reuse=True)
    # Board(randomised=True) starts with ONE model (tree reuse on), more simulations, restarting slots through SelfPlayRun
    kw = dict(sims=50, seed=43, first_game=7000, randomised=True)
    b = sp.BatchSelfPlay(m, n_slots=12, max_games=12, log_capacity=12 * 600, **kw)
    want3 = _records(b.run_to_completion(max_plies=1100))
    b.close()
    run = sp.SelfPlayRun(m, n_games=12, max_slots=5, harvest_every=2, free_running=True, **kw)
    got3 = _records(run.run().games())
    c3 = run.counters()
    run.close()
    assert got3 == want3 and c3['errors'] == 0 and c3['cache_hits'] > 0


def test_config1_one_whole_game_50_sims_matches_oracle(golden_dir):
    """BASELINE config 1 for real (selfplay.py:155-175): ONE whole game at 50 simulations per move with good_model.h5 through
    the delivered selfplay() on the HIP path, against the CPU oracle playing the same game id with a callback into the same
    evaluator one position at a time: status, z, every move, every searched position and every pi bit for bit."""
    import torch
    from chinesecheckersagent_amd import selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    assert m.backend == 'hip'
    calls = [0]

    def cb(planes_p, pos12_p, player, p_out, v_out, user):
        x = np.ctypeslib.as_array(planes_p, shape=(343,)).astype(np.float32).reshape(1, 7, 7, 7)
        p, v = m.evaluate_batch(torch.from_numpy(x).cuda())
        np.ctypeslib.as_array(p_out, shape=(294,))[:] = p[0].cpu().numpy()
        v_out[0] = float(v[0])
        calls[0] += 1
    fn = orc.EVAL_FN(cb)
    seed, sims = 20261003, 50
    seen = set()
    for game in (0, 1, 2, 3):                                     # until a won game has been compared (a discarded one has no history)
        o = orc.selfplay(seed, game, sims, 4, fn=fn)
        hist, reward = sp.selfplay(m, sims=sims, seed=seed, game_id=game)
        if o['status'] in (orc.ST_WON_P1, orc.ST_WON_P2):
            assert reward == o['reward'] and len(hist) == len(o['pi'])
            for k, (b, pi) in enumerate(hist):
                assert list(b.pos12) == [int(x) for x in o['hist_pos12'][k]], 'position of searched ply %d of game %d' % (k, game)
                assert np.array_equal(pi, o['pi'][k]), 'pi of searched ply %d of game %d differs from the oracle' % (k, game)
            seen.add('won')
            break
        assert (hist, reward) == (None, None)
        seen.add('discarded')
    assert 'won' in seen and calls[0] > 20 * 51


def test_full_size_restarting_run_plays_the_oracles_games(golden_dir):
    """BASELINE config 3 at FULL size in the delivered mode: 4096 restarting slots x 400 simulations with good_model.h5 through
    SelfPlayRun exactly as bench.py runs it (two half-batches of free-running slots on two streams, tree reuse, requests evaluated in
    place by the fused kernel, 25 rounds per hipGraph, harvests every 4 plies, worker thread) -- and the test says so: it fails if the
    run fell back to lock-step, ran without reuse, off the HIP evaluator or off its captured graphs.  When enough games have ended,
    TEN of them are replayed by the CPU oracle with a callback into the same evaluator, one position per call: eight won games spread
    over the length distribution (shortest to longest: long games carry the most reused subtrees) -- every searched position, every pi
    and z, bit for bit -- and one game discarded by each rule that occurred (status, ply count, evaluator calls: a discarded game's
    rows are not kept, selfplay.py:45-47, 72-74)."""
    import torch
    from chinesecheckersagent_amd import _lib, selfplay as sp
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    assert m.backend == 'hip'
    seed, sims, G = 20261003, 400, 4096
    run = sp.SelfPlayRun(m, n_games=G * 4, sims=sims, seed=seed, max_slots=G)
    try:
        assert run.free_running and hasattr(run.b, 'parts') and len(run.b.parts) == 2       # the path bench.py times, not a fallback
        assert all(b.free_running and b.reuse and b.log_guard and b.use_graph for b in run.b.parts)
        won = disc = []
        for _ in range(20):                                   # <= 160 steps of 401 rounds (a slot plays ~1.3 plies per step)
            for _ in range(8):                                # (play_ply harvests every sp.HARVEST_EVERY plies by itself)
                run.play_ply()
            st = run.store.results['status']
            won = np.nonzero((st == _lib.ST_WON_P1) | (st == _lib.ST_WON_P2))[0]
            disc = np.nonzero((st == _lib.ST_DISCARD_REPETITION) | (st == _lib.ST_DISCARD_NO_PROGRESS))[0]
            if len(won) >= 64 and len(disc) >= 1:
                break
        c = run.counters()
        assert len(won) >= 8 and len(disc) >= 1 and c['errors'] == 0
        assert c['cache_hits'] > 0.15 * c['expansions'], (c['cache_hits'], c['expansions'])       # tree reuse really answered expansions
        assert all(b._graph is not None and b._unroll == sp.BatchSelfPlay.FREE_UNROLL for b in run.b.parts)     # the captured-graph path really ran
        run.store.take_finished()
        st_, meta_, pi_ = (np.concatenate([r[i] for r in run.store._records]) for i in range(3))
        results = run.store.results.copy()
        by_len = sorted(won, key=lambda j: int(results['n_plies'][j]))
        picked = [by_len[i] for i in sorted(set(np.linspace(0, len(by_len) - 1, 8).astype(int)))]
        for kind in (_lib.ST_DISCARD_REPETITION, _lib.ST_DISCARD_NO_PROGRESS):
            of_kind = [j for j in disc if int(results['status'][j]) == kind]
            if of_kind:
                picked.append(min(of_kind, key=lambda j: int(results['n_plies'][j])))
    finally:
        run.close()

    def cb(planes_p, pos12_p, player, p_out, v_out, user):
        x = np.ctypeslib.as_array(planes_p, shape=(343,)).astype(np.float32).reshape(1, 7, 7, 7)
        p, v = m.evaluate_batch(torch.from_numpy(x).cuda())
        np.ctypeslib.as_array(p_out, shape=(294,))[:] = p[0].cpu().numpy()
        v_out[0] = float(v[0])
    fn = orc.EVAL_FN(cb)
    assert len(picked) >= 9
    for j in picked:
        o = orc.selfplay(seed, int(j), sims, 4, fn=fn)
        res = results[j]
        assert o['status'] == int(res['status']) and len(o['plies']) == int(res['n_plies']) and o['evals'] == int(res['expansions']), j
        if o['status'] not in (orc.ST_WON_P1, orc.ST_WON_P2):
            assert o['n_searched'] == int(res['n_samples'])
            continue
        assert o['reward'] == int(res['reward'])
        rows = np.nonzero(meta_['game'] == j)[0]
        rows = rows[np.argsort(meta_['ply'][rows])]
        assert len(rows) == len(o['pi']) == int(res['n_samples'])
        for k, r in enumerate(rows):
            assert [int(x) for x in st_[r]['pos'].reshape(12)] == [int(x) for x in o['hist_pos12'][k]], (j, k)
            assert np.array_equal(pi_[r], o['pi'][k]), 'pi of searched ply %d of game %d differs from the oracle at full size' % (k, j)


def test_generate_self_play_in_parallel_world2_on_one_device(golden_dir):
    """train.generate_self_play_in_parallel with GPUs for workers: two rank processes (both on cuda:0 here, gloo carrying the
    summary) started from this process play ids j mod 2; the merged list equals generate_self_play of the same ids on one
    rank, the all-reduced summary adds up"""
    from chinesecheckersagent_amd import selfplay as sp
    w = golden_dir + '/good_model.h5'
    n, sims, seed, first = 9, 8, 23, 500
    par, summ = sp.generate_self_play_in_parallel(w, n, 2, sims=sims, seed=seed, first_game=first, devices=[0, 0], return_summary=True,
                                                  min_games_per_rank=1)
    one = sp.generate_self_play(1, w, n, sims=sims, seed=seed, first_game=first)
    assert _records(par) == _records(one) and len(one) > 0
    # the default policy for a cohort this small (latency-bound: selfplay.selfplay_ranks, profiles/r6_small_cohort.txt): ONE rank plays it
    # -- the same records, the summary says how many ranks played
    assert sp.selfplay_ranks(9, 2) == 1 and sp.selfplay_ranks(180, 8) == 1 and sp.selfplay_ranks(257, 8) == 2 and sp.selfplay_ranks(32768, 8) == 8
    par1, summ1 = sp.generate_self_play_in_parallel(w, n, 2, sims=sims, seed=seed, first_game=first, devices=[0, 0], return_summary=True)
    assert _records(par1) == _records(one) and summ1['world'] == 1 and summ1['counters']['expansions'] == summ['counters']['expansions']
    c = summ['counters']
    assert summ['world'] == 2 and summ['backend'] == 'gloo' and c['errors'] == 0
    assert c['games_won'] == len(one) and c['games_won'] + c['games_discarded'] == n
    assert sum(summ['visit_histogram']) == c['mcts_plies'] * sims
    bx, py, vy = sp.generate_self_play_in_parallel(w, n, 2, sims=sims, seed=seed, first_game=first, devices=[0, 0], as_arrays=True,
                                                   min_games_per_rank=1)
    assert len(vy) == sum(len(h) for h, _ in one)
    from chinesecheckersagent_amd import utils
    wx, wp, wv = utils.convert_to_train_data(one)
    assert (np.array(wx) == bx).all() and (np.array(wp) == py).all() and list(vy) == wv
    # at a real size: 2600 games through two ranks of 1024 FREE-RUNNING slots each (every slot restarts), the rows streamed into the
    # ranks' files (worker --arrays) and merged by game id == the arrays one process generates
    n, sims, first = 2600, 5, 9000
    (bx, py, vy), summ = sp.generate_self_play_in_parallel(w, n, 2, sims=sims, seed=seed, first_game=first, devices=[0, 0], as_arrays=True,
                                                           return_summary=True, max_slots=1024, timeout=600)
    ox, op, ov, osum = sp.generate_train_data(ResidualCNN_loaded(w), n_games=n, sims=sims, seed=seed, first_game=first, max_slots=2048)
    c = summ['counters']
    assert c['errors'] == 0 and c['games_won'] + c['games_discarded'] == n and c['cache_hits'] > 0
    assert c['games_won'] == osum['won'] and c['expansions'] == osum['counters']['expansions']
    assert len(vy) == len(ov) > 0 and (bx == ox).all() and (py == op).all() and (vy == ov).all()


def test_config4_shape_five_ranks_of_4096_free_running_slots_on_one_device(golden_dir, tmp_path):
    """BASELINE config 4 AT ITS PER-RANK SHAPE on the one device of the box: train.generate_self_play_in_parallel (train.py:71-105) with FIVE
    rank processes -- the most this process can start beside itself under the pool's limit of six GPU processes per card; the
    stand-alone rehearsal (tools/config4_one_device.py, profiles/r6_config4_one_device.txt) runs six -- each with 4096 free-running slots
    at 400 simulations (two half-batches, tree reuse, 25 rounds per hipGraph: what bench.py times on one GPU), ids sharded j mod 5, the
    summary all-reduced over gloo, for a BOUNDED number of steps (max_steps: the whole shape would play for minutes on a shared device).
    Checked: every rank really ran that path; the all-reduced counters and the 294-bin visit histogram are the sums of the ranks'; and
    the games that ended -- at least 32 of them -- have exactly the rows a single-rank run of the same ids produces."""
    import os
    from chinesecheckersagent_amd import selfplay as sp
    w = golden_dir + '/good_model.h5'
    R, G, sims, seed, steps = 5, 4096, 400, 20261003, 56
    out_dir = str(tmp_path)
    (bx, py, vy, gid), summ = sp.generate_self_play_in_parallel(w, R * G * 2, R, sims=sims, seed=seed, first_game=0, devices=[0] * R, as_arrays=True,
                                                                return_summary=True, out_dir=out_dir, max_slots=G, timeout=900,
                                                                max_steps=steps, with_games=True)
    ranks = [json.load(open(os.path.join(out_dir, 'host-rank%d.json' % r))) for r in range(R)]
    for h in ranks:
        pth = h['path']
        assert pth['free_running'] and pth['reuse'] and pth['graphs'] and pth['backend'] == 'hip', pth      # the measured path, not a fallback
        assert pth['n_slots'] == G and pth['half_batches'] == 2 and pth['steps'] == steps
        assert h['counters']['errors'] == 0 and h['counters']['cache_hits'] > 0.15 * h['counters']['expansions']
    c = summ['counters']
    assert summ['world'] == R and summ['backend'] == 'gloo' and c['errors'] == 0
    for k in c:                                                  # the one collective of the path: all-reduce(sum) of counters + histogram
        assert c[k] == sum(h['counters'].get(k, 0) for h in ranks), k
    assert sum(summ['visit_histogram']) == sum(h['visit_histogram_sum'] for h in ranks) == c['mcts_plies'] * sims
    assert c['expansions'] > R * G * steps * 0.5 * (sims + 1) * 0.2                  # (every rank's slots were really searching)
    ended = np.unique(gid)
    assert c['games_won'] == len(ended) >= 32 and len(vy) == len(gid) > 0
    # the same ids on ONE rank in this process: slot s plays ids s, s + 4096 -- the first of which five-rank slot (s - r) / 5 of rank s mod 5 played
    m = ResidualCNN_loaded(w)
    sink = sp.TrainDataSink()
    run = sp.SelfPlayRun(m, n_games=G * 2, sims=sims, seed=seed, max_slots=G, keep_records=False, sink=sink)
    try:
        run.run(max_plies=steps)
        assert run.free_running and run.counters()['errors'] == 0
    finally:
        run.close()
    ox, op, ov, og = sink.arrays(with_games=True)
    common = np.intersect1d(ended, np.unique(og))
    assert len(common) >= 32, len(common)
    a, b = np.isin(gid, common), np.isin(og, common)
    assert (gid[a] == og[b]).all() and (bx[a] == ox[b]).all() and (py[a] == op[b]).all() and (vy[a] == ov[b]).all()
    print('config 4 shape on one device: %d ranks x %d slots x %d sims, %d steps: %d games ended, %d compared with one rank; wall per rank %s s, '
          'host cores per rank %s, peak RSS %s MB' % (R, G, sims, steps, len(ended), len(common), [round(h['wall_s'], 1) for h in ranks],
                                                      [round(h['host_cpu_s'] / h['wall_s'], 2) for h in ranks], [int(h['peak_rss_mb']) for h in ranks]))


def ResidualCNN_loaded(path):
    from chinesecheckersagent_amd.model import ResidualCNN
    m = ResidualCNN()
    m.load_weights(path)
    return m


def _bench(extra, common=None, **envx):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    if common is None:
        common = ['--sims', '24', '--steps', '3', '--warmup', '1', '--spread-plies', '6', '--min-seconds', '0', '--fused-plies', '3',
                  '--cpu-cores', '2', '--config5-games', '4', '--config5-sims', '8']
    if '--cpu-seconds' not in extra + common:
        common = common + ['--cpu-seconds', '0.4']
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + extra + common, env=dict(env, **envx),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_launcher_world2_on_one_device(tmp_path):
    """`python bench.py --gpus 2` as the driver starts it: the launcher spawns both ranks before anything touches the GPU;
    CCSP_BENCH_ONE_DEVICE=1 puts both on cuda:0 with a gloo summary (RCCL refuses two ranks on one device).  Games shard
    by id, so two ranks of 32 slots do exactly the work of one run of 64 slots while no game ends -- in the headline (config 3
    through SelfPlayRun) and in variant 2a; config 5 runs its N-rank loop (sharded self-play, DDP fit, sharded arena)."""
    two = _bench(['--gpus', '2', '--games', '32'], CCSP_BENCH_ONE_DEVICE='1')
    one = _bench(['--gpus', '1', '--games', '64', '--no-config5', '--cpu-seconds', '0'])
    assert two['n_gpus'] == 2 and one['n_gpus'] == 1 and one['cpu_baseline'] is None and 'config5' not in one
    for doc in (one, two):
        assert doc['steps'] == 3 and doc['errors'] == 0 and doc['backend'] == 'hip' and doc['value'] > 0 and doc['degraded'] is False
        # what the driver's record keeps: BASELINE's games/s and the run's description inside `config`, the isolated figure beside the
        # delivered one in `roofline`, host CPU seconds and peak RSS of every rank
        cfg, ms = doc['config'], doc['measured']
        # what the driver's record keeps of `config`: scalars only, strings of at most 120 characters -- BASELINE's games/s among them
        assert all(isinstance(x, (int, float, bool)) or (isinstance(x, str) and len(x) <= 120) for x in cfg.values()), cfg
        assert cfg['workload'].startswith('config 3') and cfg['games_per_s'] >= 0 and 0 <= cfg['discard_rate'] <= 1 and cfg['timed_region_s'] > 0
        assert cfg['node_expansions_per_s'] == doc['value'] and cfg['net_evals_per_s'] > 0 and 0 <= cfg['tree_reuse_hit_rate'] < 1
        assert 0 <= cfg['idle_row_share'] < 1 and cfg['host_cores_per_rank'] > 0 and cfg['host_peak_rss_mb_per_rank'] > 100
        assert len(ms['host_cpu_s_per_rank']) == len(ms['host_peak_rss_mb_per_rank']) == doc['n_gpus']
        assert doc['roofline']['frac_by_step'] > 0 and doc['roofline']['step_ms_per_launch'] > 0
        # (32 positions per launch: 45-us launches whose medians wander by several per cent from burst to burst)
        assert 0 < doc['roofline']['frac'] <= doc['roofline']['frac_isolated'] * 1.25 and doc['roofline']['avg_launch_ms_isolated'] > 0
        assert doc['roofline']['bound'] == 'mfma' and 0 < doc['roofline']['frac'] < 1
        v2a = doc['variants']['2a_fused_table_evaluator']
        assert v2a['errors'] == 0 and v2a['roofline']['bound'] == 'latency/issue'
    assert one['config1']['games'] == 6 and one['config1']['seconds_per_game'] > 0 and 'config1' not in two
    cb = two['cpu_baseline']
    assert cb['value'] > 0 and cb['kind'] == 'port' and cb['cores'] <= 2 and 'PyTorch CPU module' in cb['sample']
    assert cb['table_evaluator']['value'] > 0 and cb['reference_shaped_python']['value'] > 0
    assert cb['config1_reference_shaped_python_numpy_net']['value'] > 0
    c5 = two['config5']
    assert 'failed' not in c5 and c5['selfplay_games'] == 4 and c5['train_s'] > 0 and 'arena_wins' in c5
    assert c5['train_warm_up_s'] > 0 and c5['train_epochs_s'] >= 0 and c5['train_first_steps_s'] >= 0
    assert len(two['per_rank_expansions']) == 2
    assert sum(two['per_rank_expansions']) == one['per_rank_expansions'][0]           # id-sharding: same games, same work
    a, b = two['variants']['2a_fused_table_evaluator'], one['variants']['2a_fused_table_evaluator']
    assert sum(a['per_rank_expansions']) == b['per_rank_expansions'][0] and a['visit_histogram_sum'] == b['visit_histogram_sum']
    assert 'movegen_kernel' in one['variants'] and 'movegen_kernel' not in two['variants']


def test_bench_five_ranks_on_one_device_at_400_sims():
    """`python bench.py --gpus N` END TO END with as many ranks as a 1-GPU box lets this process start beside itself (five; the stand-alone
    rehearsal runs six): every rank a free-running SelfPlayRun of 1024 slots at 400 simulations (two half-batches on captured
    graphs -- CCSP_STRICT refuses anything else), variant 2a on every rank, and BASELINE config 5's N-rank loop (sharded self-play, DDP
    fit with global-batch statistics, sharded arena, one gating decision) -- so that the driver's first real N-GPU run meets nothing for
    the first time except RCCL itself (gloo carries the collectives here: RCCL refuses two ranks on one device).
    Config 5 is kept tiny (4 games at 8 simulations, ~50 optimisation steps): with three or more ranks on ONE device every step of the
    N-rank fit costs about a second over gloo (62 host-staged collectives: tools/ddp_step_probe.py, profiles/r6_config4_one_device.txt:
    8 / 39 / 742 / 1115 ms per step at 1 / 2 / 3 / 5 ranks) -- an artefact of the rehearsal, not of the loop."""
    import time
    t0 = time.time()
    doc = _bench(['--gpus', '5', '--games', '1024'],
                 common=['--sims', '400', '--steps', '4', '--warmup', '1', '--spread-plies', '14', '--min-seconds', '0', '--fused-plies', '4',
                         '--cpu-seconds', '0', '--config5-games', '4', '--config5-sims', '8', '--config5-timeout', '500'],
                 CCSP_BENCH_ONE_DEVICE='1')
    wall = time.time() - t0
    assert doc['n_gpus'] == 5 and doc['degraded'] is False and doc['errors'] == 0 and doc['backend'] == 'hip'
    assert len(doc['per_rank_expansions']) == 5 and min(doc['per_rank_expansions']) > 0
    assert doc['config']['free_running'] is True and doc['config']['half_batches'] == 2 and doc['config']['tree_reuse_hit_rate'] > 0.1
    assert len(doc['measured']['host_cpu_s_per_rank']) == 5
    assert sum(doc['variants']['2a_fused_table_evaluator']['per_rank_expansions']) > 0
    c5 = doc['config5']
    assert 'failed' not in c5 and c5['selfplay_games'] == 4 and c5['train_s'] > 0 and 'arena_wins' in c5
    assert c5['selfplay_ranks'] == 1                       # four games: ONE rank plays them (selfplay.selfplay_ranks), all five fit and play the arena
    print('bench.py --gpus 5 on one device (1024 slots x 400 sims per rank, config 5 with 4 games x 8 sims): wall %.1f s; headline %.2f M '
          'node-expansions/s over all ranks; config 5 %.1f s' % (wall, doc['value'] / 1e6, c5['wall_s']))


def test_bench_survives_a_rank_that_never_joins_config5():
    """N > 1: config 5 is a loop of collectives, and a rank that stalls alone would leave the others waiting for ever.  The bench must
    not lose its measurements to that: with rank 1 held back (test hook) and a 6-second --config5-timeout, every rank gives config 5
    up, rank 0 prints its line with config5 marked failed, and the job exits with status 0"""
    two = _bench(['--gpus', '2', '--games', '32', '--config5-timeout', '6', '--cpu-seconds', '0'], CCSP_BENCH_ONE_DEVICE='1',
                 CCSP_BENCH_TEST_STALL_RANK='1')
    assert two['n_gpus'] == 2 and two['value'] > 0 and two['errors'] == 0 and 'watchdog' in two['config5']['failed']
    assert two['variants']['2a_fused_table_evaluator']['node_expansions_per_s'] > 0
    assert two['degraded'] is True and 'config 5' in two['degraded_reason']      # the record is not silently clean
    # a rank whose config 5 RAISES: the others leave within a second or two of its flag file, not after the timeout
    import time
    t0 = time.time()
    two = _bench(['--gpus', '2', '--games', '32', '--config5-timeout', '600', '--cpu-seconds', '0'], CCSP_BENCH_ONE_DEVICE='1',
                 CCSP_BENCH_TEST_FAIL_RANK='1')
    assert time.time() - t0 < 300
    assert two['degraded'] is True and two['value'] > 0 and 'failed' in two['config5'] and two['cpu_baseline'] is None


def test_bench_summary_collectives_through_rccl_world1():
    """§8e: the summary collectives of the N-GPU bench (process group on the device, MAX / SUM all-reduce, all-gather, barrier)
    through RCCL itself -- a world of one rank, which is all a 1-GPU box allows (CCSP_BENCH_FORCE_DIST=1); the result must equal
    the plain 1-GPU run"""
    quick = ['--games', '64', '--no-config5', '--cpu-seconds', '0']
    a, b = _bench(quick, CCSP_BENCH_FORCE_DIST='1'), _bench(quick)
    assert a['n_gpus'] == b['n_gpus'] == 1 and a['errors'] == 0
    assert a['per_rank_expansions'] == b['per_rank_expansions']
    va, vb = a['variants']['2a_fused_table_evaluator'], b['variants']['2a_fused_table_evaluator']
    assert va['per_rank_expansions'] == vb['per_rank_expansions'] and va['visit_histogram_sum'] == vb['visit_histogram_sum']
