"""GPU parity of the batched tree search and the self-play driver, through the C ABI, against
the reference's own results on substituted draws (tests/golden/tree.json, games.json) and the
CPU oracle: rows T1-T4, S1-S3, O1 of SURVEY.md §8a."""
import hashlib
import json
import struct

import numpy as np
import pytest

import oracle_ffi as orc

pytestmark = pytest.mark.gpu


def bits(x):
    return struct.unpack('<Q', struct.pack('<d', float(x)))[0]


@pytest.fixture(scope='module')
def eng():
    import torch
    from chinesecheckersagent_amd import _lib, engine
    _lib.require_gpu()
    assert torch.cuda.is_available()
    return engine


def _check_case(c, root, pi_row, digest, new_pos12, tag):
    k = len(c['N'])
    assert len(root['N']) == k, tag
    assert [int(m) // 49 for m in root['mv']] == c['cid'] and [int(m) % 49 for m in root['mv']] == c['dest'], tag
    assert [int(x) for x in root['N']] == c['N'], 'visit counts differ: ' + tag
    assert [bits(x) for x in root['W']] == c['W'], 'W bits differ: ' + tag
    assert [bits(x) for x in root['P']] == c['P'], 'P bits differ: ' + tag
    nz = [int(j) for j in np.nonzero(pi_row)[0]]
    assert nz == c['pi_idx'] and [bits(pi_row[j]) for j in nz] == c['pi_bits'], 'pi bits differ: ' + tag
    d, nodes, edges = digest
    assert nodes == c['nodes'] and edges == c['edges'], tag
    assert d == c['tree_sha'], 'whole-tree digest differs: ' + tag
    who = c['player']
    cid, dest = c['chosen']
    assert int(new_pos12[(who - 1) * 6 + cid]) == dest, 'sampled move differs: ' + tag


def _groups(cases):
    g = {}
    for i, c in enumerate(cases):
        g.setdefault((c['sims'], c['evaluator']), []).append(i)
    return g


def test_fused_search_matches_reference(eng, golden_dir):
    """every golden make_move() case (105: three evaluators, sims 50/175/400, both taus, normal /
    randomised / near-win roots) as one slot of the fused kernel"""
    from chinesecheckersagent_amd import _lib
    doc = json.load(open(golden_dir + '/tree.json'))
    seed, cases = doc['seed'], doc['cases']
    for (sims, ev), idxs in sorted(_groups(cases).items()):
        n = len(idxs)
        e = eng.SelfPlayEngine(n_slots=n, sims=sims, seed=seed, max_games=n, log_capacity=n)
        cs = [cases[i] for i in idxs]
        states = _lib.pack_states([c['pos12'] for c in cs], [c['last'] for c in cs])
        e.set_positions(states, [c['player'] for c in cs], [c['game'] for c in cs], [c['nplies'] for c in cs],
                        [0 if c['tau'] == 1 else 1 for c in cs])
        e.play_plies(ev, 1)
        st, meta, pi = e.log()
        assert len(meta) == n
        row_of = {int(m['game']): r for r, m in enumerate(meta)}
        slots = e.slots()
        cnt = e.counters()
        assert cnt['errors'] == 0
        assert cnt['expansions'] == sum(c['evals'] for c in cs)
        for s, c in enumerate(cs):
            tag = 'case %d (ev=%d sims=%d tau=%s start=%s)' % (idxs[s], ev, sims, c['tau'], c['start'])
            r = row_of[c['game']]
            assert [int(x) for x in st[r]['pos'].reshape(12)] == c['pos12'] and int(meta[r]['player']) == c['player'], tag
            _check_case(c, e.read_root(s), pi[r], e.tree_digest(s), slots['state'][s]['pos'].reshape(12), tag)
        e.close()


def _decode_planes(x):
    """planes [343] -> (pos12, player) (what oracle/harness/refenv.TableModel does)"""
    x = x.reshape(49, 7)
    player = 2 if x[0, 6] == 1 else 1
    cur, opp = {}, {}
    for cell in range(49):
        if x[cell, 0]:
            cur[int(x[cell, 0]) - 1] = cell
        if x[cell, 1]:
            opp[int(x[cell, 1]) - 1] = cell
    p1, p2 = (cur, opp) if player == 1 else (opp, cur)
    return [p1[i] for i in range(6)] + [p2[i] for i in range(6)], player


def _table_eval(ev, pos12, player):
    import ctypes as C
    L = orc.lib()
    L.orc_forward_eval.argtypes = L.orc_hash_eval.argtypes
    a = np.array(pos12, dtype=np.uint8)
    p = (C.c_double * 294)()
    v = C.c_float()
    if ev == 0:
        return np.full(294, 1.0 / 294.0), 0.0
    (L.orc_hash_eval if ev == 1 else L.orc_forward_eval)(a.ctypes.data_as(C.POINTER(C.c_uint8)), player, p, C.byref(v))
    return np.array(p[:]), v.value


def test_stepped_search_matches_reference(eng, golden_dir):
    """the same cases at sims = 50 through the external-evaluator path: ply_begin -> root_expand ->
    50 x (select -> expand_backup) -> ply_end, with (p, v) computed on the host from the PLANES the
    kernels emit (so the encode path is covered as well); and the same with expand_backup and the next select in one launch"""
    import torch
    from chinesecheckersagent_amd import _lib
    doc = json.load(open(golden_dir + '/tree.json'))
    seed, cases = doc['seed'], doc['cases']
    for ((sims, ev), idxs), merged in [(g, m) for g in sorted(_groups(cases).items()) for m in (False, True)]:
        if sims != 50:
            continue
        n = len(idxs)
        e = eng.SelfPlayEngine(n_slots=n, sims=sims, seed=seed, max_games=n, log_capacity=n)
        cs = [cases[i] for i in idxs]
        states = _lib.pack_states([c['pos12'] for c in cs], [c['last'] for c in cs])
        e.set_positions(states, [c['player'] for c in cs], [c['game'] for c in cs], [c['nplies'] for c in cs],
                        [0 if c['tau'] == 1 else 1 for c in cs])
        planes = torch.zeros((n, 343), dtype=torch.float32, device='cuda')
        p = torch.zeros((n, 294), dtype=torch.float64, device='cuda')
        v = torch.zeros(n, dtype=torch.float32, device='cuda')

        def evaluate():
            ph = planes.cpu().numpy()
            pp = np.zeros((n, 294))
            vv = np.zeros(n, dtype=np.float32)
            for s in range(n):
                if ph[s].any():
                    pos12, pl = _decode_planes(ph[s])
                    pp[s], vv[s] = _table_eval(ev, pos12, pl)
            p.copy_(torch.from_numpy(pp))
            v.copy_(torch.from_numpy(vv))

        e.ply_begin(planes)
        evaluate()
        e.root_expand(p, v)
        if merged:                       # expand_backup + the next simulation's select in one launch (ccsp_expand_backup_select)
            e.select(planes)
            for i in range(sims):
                evaluate()
                if i + 1 < sims:
                    e.expand_backup_select(p, v, planes)
                else:
                    e.expand_backup(p, v)
        else:
            for _ in range(sims):
                e.select(planes)
                evaluate()
                e.expand_backup(p, v)
        e.ply_end()
        st, meta, pi = e.log()
        row_of = {int(m['game']): r for r, m in enumerate(meta)}
        slots = e.slots()
        assert e.counters()['errors'] == 0
        for s, c in enumerate(cs):
            tag = 'stepped case %d (ev=%d, merged=%s)' % (idxs[s], ev, merged)
            r = row_of[c['game']]
            _check_case(c, e.read_root(s), pi[r], e.tree_digest(s), slots['state'][s]['pos'].reshape(12), tag)
        e.close()


def test_whole_games_match_reference(eng, golden_dir):
    """selfplay() end to end on the GPU: opening plies, searches, tau switch, discard rules, result;
    compared with the reference's games (status, reward, ply count, evaluator calls, every recorded
    state and pi)"""
    doc = json.load(open(golden_dir + '/games.json'))
    seed = doc['seed']
    groups = {}
    for g in doc['games']:
        if isinstance(g['evaluator'], list):
            continue                                   # two-model games: tests/test_gpu_api.py (stepped path)
        groups.setdefault((g['sims'], g['evaluator'], g['randomised']), []).append(g)
    seen = set()
    for (sims, ev, randomised), gs in sorted(groups.items()):
        for g in gs:                                   # one context per game id (ids are not contiguous)
            e = eng.SelfPlayEngine(n_slots=1, sims=sims, seed=seed, first_game=g['game'], max_games=1,
                                   log_capacity=1024, randomised=randomised)
            for _ in range(64):
                e.play_plies(ev, 16)
                if e.slots()['status'][0] != 0:
                    break
            res = e.results()[0]
            status = {1: 'won', 2: 'won', 3: 'repetition', 4: 'no_progress'}[int(res['status'])]
            tag = 'game %d' % g['game']
            assert status == g['status'], tag
            seen.add(status)
            assert int(res['n_plies']) == len(g['plies']), tag
            assert int(res['expansions']) == g['evals'], tag
            st, meta, pi = e.log()
            assert (meta['game'] == g['game']).all()
            assert [int(x) for x in meta['ply']] == [i for i, p in enumerate(g['plies']) if p[0] != 0], tag
            if status == 'won':
                assert int(res['reward']) == g['reward'], tag
                drop = 3 if randomised else 0              # selfplay.py:76-78
                st, meta, pi = st[drop:], meta[drop:], pi[drop:]
                assert [[int(x) for x in s['pos'].reshape(12)] for s in st] == g['hist_pos12'], tag
                assert [hashlib.sha256(np.asarray(r, dtype='<f8').tobytes()).hexdigest()[:16] for r in pi] == g['pi_sha'], tag
                assert hashlib.sha256(pi.astype('<f8').tobytes()).hexdigest() == g['o1']['pi_y_sha'], tag
            e.close()
    assert seen == {'won', 'repetition', 'no_progress'}


def test_sharding_independence(eng):
    """results are a function of the game id only: 8 games in one context == the same 8 games
    split over two contexts with stride 2 (how ranks shard games)"""
    seed, sims = 99, 16
    a = eng.SelfPlayEngine(n_slots=8, sims=sims, seed=seed, first_game=100, max_games=8, log_capacity=8 * 64)
    a.play_plies(1, 24)
    sa, ma, pa = a.log()
    key = lambda m: (int(m['game']), int(m['ply']))
    rows = {key(m): (sa[i].tobytes(), pa[i].tobytes()) for i, m in enumerate(ma)}
    a.close()
    got = {}
    for r in range(2):
        b = eng.SelfPlayEngine(n_slots=4, sims=sims, seed=seed, first_game=100 + r, game_stride=2, max_games=4, log_capacity=4 * 64)
        b.play_plies(1, 24)
        sb, mb, pb = b.log()
        got.update({key(m): (sb[i].tobytes(), pb[i].tobytes()) for i, m in enumerate(mb)})
        b.close()
    assert rows == got and len(rows) == 8 * 18


def test_plies_per_launch_changes_nothing(eng):
    """one launch per phase of a ply (begin / simulations / end kernels) == one launch carrying every game through many plies
    (fused_plies_kernel, the default): same counters, same game results, same sample log row for row, same visit histogram,
    restarts included (a slot's next game index is slot + k * n_slots: no dependence on which slot finishes first)"""
    from chinesecheckersagent_amd import _lib
    L = _lib.lib()
    was = L.ccsp_debug_plies_per_launch(0)
    assert was > 1, 'the multi-ply kernel is the default path'
    runs = []
    try:
        for ppl in (1, 5, was):
            L.ccsp_debug_plies_per_launch(ppl)
            e = eng.SelfPlayEngine(n_slots=37, sims=12, seed=4242, first_game=3, game_stride=2, max_games=37 * 3, log_capacity=37 * 1200,
                                   auto_restart=True)
            e.play_plies(_lib.EVAL_FORWARD, 130)
            e.play_plies(_lib.EVAL_FORWARD, 1)              # a single ply goes through the three-kernel path whatever the setting
            e.play_plies(_lib.EVAL_FORWARD, 70)
            c = e.counters()
            st, meta, pi = e.log()
            order = np.lexsort((meta['ply'], meta['game']))
            res = e.results()
            sl = e.slots()
            runs.append((c, st[order].tobytes(), meta[order].tobytes(), pi[order].tobytes(), res.tobytes(), e.visit_histogram().tobytes(),
                         sl['game'].tobytes(), sl['ply'].tobytes(), [e.tree_digest(g) for g in range(37)]))
            e.close()
    finally:
        L.ccsp_debug_plies_per_launch(was)
    c = runs[0][0]
    assert c['games_won'] + c['games_discarded'] > 37 and c['errors'] == 0, 'the run must include restarts'
    for r in runs[1:]:
        assert r == runs[0]


def test_rollout_evaluator_matches_cpu_restatement(eng):
    """config 2b (random-playout value) has no reference counterpart (the reference has no
    rollouts); its definition lives in DESIGN.md and is restated on the CPU in oracle/ccsp_oracle.c
    (evaluator 3).  GPU and CPU restatement must agree bit-for-bit."""
    from chinesecheckersagent_amd import _lib
    seed, sims, n = 31337, 64, 24
    e = eng.SelfPlayEngine(n_slots=n, sims=sims, seed=seed, max_games=n, log_capacity=n * 8)
    e.play_plies(_lib.EVAL_ROLLOUT, 6)
    e.play_plies(_lib.EVAL_ROLLOUT, 2)
    st, meta, pi = e.log()
    assert len(meta) == 2 * n and e.counters()['errors'] == 0
    for r in range(len(meta)):
        o = orc.search(st[r]['pos'].reshape(12), st[r]['last'], int(meta[r]['player']), seed, int(meta[r]['game']),
                       int(meta[r]['ply']), sims, False, 3)
        assert np.array_equal(pi[r], np.array(o.pi[:])), 'rollout search differs from the CPU restatement (row %d)' % r
    e.close()


def test_restart_budget_log_overflow_and_odd_sizes(eng):
    """edge cases of the engine: slots restart on fresh game ids until the id budget is spent and then go idle; a
    sample log that is too small is reported through the error counter (nothing is written out of bounds);
    batch sizes that are not multiples of anything"""
    from chinesecheckersagent_amd import _lib
    # 3 slots, 7 games in total, random-ish play at 4 sims: games end by the discard rules
    e = eng.SelfPlayEngine(n_slots=3, sims=4, seed=77, first_game=10, game_stride=3, max_games=7, log_capacity=7 * 400,
                           auto_restart=True)
    for _ in range(200):
        e.play_plies(_lib.EVAL_HASH, 16)
        if (e.slots()['status'] == _lib.ST_IDLE).all():
            break
    s = e.slots()
    assert (s['status'] == _lib.ST_IDLE).all()
    res = e.results()
    assert len(res) == 7 and set(int(x) for x in res['status']) <= {1, 2, 3, 4}
    c = e.counters()
    assert c['games_won'] + c['games_discarded'] == 7 and c['errors'] == 0
    st, meta, pi = e.log()
    assert sorted(set(int(g) for g in meta['game'])) == [10 + 3 * k for k in range(7)]          # ids first + k * stride
    assert c['samples'] == len(meta) == int(res['n_samples'].sum())
    assert c['plies'] == int(res['n_plies'].sum())
    e.close()
    # every game against the oracle playing it alone: restart order and the fast-forwarded openings of restarted games
    # (their six random plies are played in the call that starts them) change no game's record
    by_game = {}
    for i in np.lexsort((meta['ply'], meta['game'])):
        by_game.setdefault(int(meta['game'][i]), []).append(i)
    for k in range(7):
        o = orc.selfplay(77, 10 + 3 * k, 4, 1)
        assert o['status'] == int(res['status'][k]) and len(o['plies']) == int(res['n_plies'][k])
        rows = by_game[10 + 3 * k]
        assert [int(meta['ply'][r]) for r in rows] == list(range(6, 6 + len(rows))) and len(rows) == int(res['n_samples'][k])
        if len(o['pi']):                                   # the oracle hands back the history of won games only
            assert len(rows) == len(o['pi']) and all(np.array_equal(pi[r], o['pi'][j]) for j, r in enumerate(rows))
    # log too small: 2 rows for 5 slots
    e = eng.SelfPlayEngine(n_slots=5, sims=4, seed=1, max_games=5, log_capacity=2)
    e.play_plies(_lib.EVAL_UNIFORM, 8)
    c = e.counters()
    # the three games whose first row did not fit end in ERROR at once, the other two when their second row does not
    assert e.log_size() == 2 and c['errors'] == 5 and c['samples'] == 2
    assert (e.slots()['status'] == _lib.ST_ERROR).all() and (e.results()['status'] == _lib.ST_ERROR).all()
    e.close()
    with pytest.raises(_lib.CcspError):
        eng.SelfPlayEngine(n_slots=0, sims=4, seed=1)


def test_c_abi_error_paths(eng):
    """allocation failure is CCSP_ENOMEM (not a generic HIP error) and leaves the device usable; a stream of another
    device / a context used while another device is current (single-GPU box: the current-device half only)"""
    import ctypes as C
    import torch
    from chinesecheckersagent_amd import _lib
    L = _lib.lib()
    cfg = _lib.Config(n_slots=4, sims=4, randomised=0, auto_restart=0, seed=1, first_game=0, game_stride=1, max_games=4,
                      log_capacity=1 << 42, device=0, max_plies=0, mode=0, arena_det_tau=0, enforce_move_limit=0, greedy=0,
                      stuck_limit=0, pad=0)                      # 2^42 rows x 2.4 KB of pi: cannot be allocated
    err = C.c_int(0)
    assert not L.ccsp_create(C.byref(cfg), C.byref(err)) and err.value == _lib.ENOMEM
    with pytest.raises(_lib.CcspError, match='ENOMEM'):
        eng.SelfPlayEngine(n_slots=4, sims=4, seed=1, max_games=4, log_capacity=1 << 42)
    # ... and the next context works, set_positions included (its staging buffers are freed on every path)
    e = eng.SelfPlayEngine(n_slots=4, sims=4, seed=1, max_games=4, log_capacity=64)
    free0 = torch.cuda.mem_get_info()[0]
    s = e.slots()
    for _ in range(50):
        e.set_positions(s['state'], s['player'], s['game'], np.full(4, 6, dtype=np.uint32), np.zeros(4, dtype=np.uint8))
    assert torch.cuda.mem_get_info()[0] >= free0 - (1 << 20)     # nothing leaked by 50 calls
    e.play_plies(_lib.EVAL_HASH, 1)
    assert e.counters()['errors'] == 0 and e.log_size() == 4
    assert L.ccsp_set_positions(e.ctx, None, None, None, None, None, None) == _lib.EINVAL
    # the stepped entry points refuse to be called out of order (no simulation is selected yet) and with missing arguments
    planes = torch.zeros((4, 343), dtype=torch.float32, device='cuda')
    p = torch.zeros((4, 294), dtype=torch.float64, device='cuda'); v = torch.zeros(4, dtype=torch.float32, device='cuda')
    ESTATE = _lib.ESTATE
    assert L.ccsp_expand_backup(e.ctx, p.data_ptr(), v.data_ptr(), None) == ESTATE
    assert L.ccsp_expand_backup_select(e.ctx, p.data_ptr(), v.data_ptr(), planes.data_ptr(), None) == ESTATE
    assert L.ccsp_expand_backup_select(e.ctx, p.data_ptr(), v.data_ptr(), None, None) == _lib.EINVAL
    # the free-running entry points: unknown flag bits, tree reuse before its pool exists, missing hand-off buffers, a stagger span
    # beyond the slot's 16-bit countdown
    req, moves, pk, v4 = e.request_buffers()
    assert L.ccsp_advance(e.ctx, pk.data_ptr(), v4.data_ptr(), req.data_ptr(), moves.data_ptr(), None, 1 << 9, None) == _lib.EINVAL
    assert L.ccsp_advance(e.ctx, pk.data_ptr(), v4.data_ptr(), req.data_ptr(), moves.data_ptr(), None, _lib.ADVANCE_REUSE, None) == ESTATE
    assert L.ccsp_boundary(e.ctx, pk.data_ptr(), v4.data_ptr(), None, moves.data_ptr(), None, 0, None) == _lib.EINVAL
    assert L.ccsp_boundary(e.ctx, pk.data_ptr(), v4.data_ptr(), req.data_ptr(), None, None, 0, None) == _lib.EINVAL
    assert L.ccsp_set_advance_limits(None, 1, 1, 1) == _lib.EINVAL and L.ccsp_set_advance_limits(e.ctx, 4, 0, 0) == 0
    assert L.ccsp_encode_requests(None, 4, planes.data_ptr(), None) == _lib.EINVAL and L.ccsp_encode_requests(None, 0, None, None) == 0
    assert L.ccsp_gather_priors(req.data_ptr(), moves.data_ptr(), None, 4, pk.data_ptr(), None) == _lib.EINVAL
    assert L.ccsp_net_forward_requests(None, req.data_ptr(), moves.data_ptr(), 4, pk.data_ptr(), v4.data_ptr(), None) == _lib.EINVAL
    assert L.ccsp_debug_table_eval(7, req.data_ptr(), moves.data_ptr(), 4, pk.data_ptr(), v4.data_ptr(), None) == _lib.EINVAL
    # a request buffer that holds STALE records (a caller that did not zero it after ccsp_reset): a record's kind means nothing to a
    # slot that has not asked -- the games start as they should
    stale = np.zeros(4, dtype=_lib.REQUEST_DTYPE); stale['kind'] = 3; stale['k'] = 126
    req.copy_(torch.from_numpy(stale.view(np.uint8).reshape(4, 64)))
    e.reset()
    e.boundary(pk, v4, req, moves)
    torch.cuda.synchronize()
    assert e.counters()['errors'] == 0 and e.counters()['expansions'] == 0 and (e.slots()['status'] == _lib.ST_RUNNING).all()
    assert L.ccsp_set_stagger_span(e.ctx, 65536) == _lib.EINVAL and L.ccsp_set_stagger_span(e.ctx, -1) == _lib.EINVAL
    assert L.ccsp_set_stagger_span(e.ctx, 4010) == 0 and L.ccsp_set_stagger_span(None, 1) == _lib.EINVAL
    assert L.ccsp_debug_read_slots(e.ctx, None) == _lib.EINVAL
    e.close()


def test_full_size_invariants_and_oracle_sample(eng):
    """BASELINE.json's size (4096 concurrent games x 400 simulations): size-independent properties of every
    slot, run-to-run determinism, and 24 slots checked bit-for-bit against the CPU oracle"""
    import hashlib
    from chinesecheckersagent_amd import _lib
    G, S, seed = 4096, 400, 20261003

    def run():
        e = eng.SelfPlayEngine(n_slots=G, sims=S, seed=seed, max_games=G, log_capacity=G * 4)
        e.play_plies(_lib.EVAL_HASH, 6)
        e.play_plies(_lib.EVAL_HASH, 2)
        st, meta, pi = e.log()
        c = e.counters()
        roots = {s: e.read_root(s) for s in range(0, G, 171)}
        e.close()
        order = np.lexsort((meta['ply'], meta['game']))
        return st[order], meta[order], pi[order], c, roots
    st, meta, pi, c, roots = run()
    assert len(meta) == 2 * G and c['errors'] == 0
    assert c['sims'] == 2 * G * S and c['expansions'] + c['terminal_sims'] == 2 * G * (S + 1)
    assert np.abs(pi.sum(axis=1) - 1.0).max() < 1e-12 and (pi >= 0).all()
    # tau = 1 at these plies: pi = N / sum(N) with sum(N) = number of simulations (each one passes one root edge)
    assert np.allclose(pi * S, np.round(pi * S), atol=1e-9)
    for s, r in roots.items():
        assert int(r['N'].sum()) == S and len(set(int(m) for m in r['mv'])) == len(r['mv'])
    # pi lives on legal moves only
    for r in range(0, 2 * G, 509):
        legal = set(int(a) * 49 + int(b) for a, b in orc.movegen(st[r]['pos'].reshape(12), int(meta[r]['player'])))
        assert set(int(i) for i in np.nonzero(pi[r])[0]) <= legal
    digest = hashlib.sha256(st.tobytes() + pi.tobytes()).hexdigest()
    st2, meta2, pi2, c2, _ = run()
    assert hashlib.sha256(st2.tobytes() + pi2.tobytes()).hexdigest() == digest, 'two runs with one seed differ'
    assert c2 == c
    for r in range(0, 2 * G, 341):                       # 24 rows, both plies, against the oracle at 400 simulations
        o = orc.search(st[r]['pos'].reshape(12), st[r]['last'], int(meta[r]['player']), seed, int(meta[r]['game']),
                       int(meta[r]['ply']), S, False, 1)
        assert np.array_equal(pi[r], np.array(o.pi[:])), 'row %d differs from the oracle' % r


def test_config_2b_full_size_rollout_invariants_and_oracle_sample(eng):
    """BASELINE configs[1] in its north-star wording (SURVEY.md 8d "2b": random-rollout value, uniform priors, no net) AT FULL SIZE: 4096
    concurrent games x 400 simulations through the fused kernel with CCSP_EVAL_ROLLOUT for two searched plies -- the run bench.py's
    `variants.2b_rollout` times.  No reference counterpart exists (MCTS.py:93 always calls the model): the definition is restated in
    oracle/ccsp_oracle.c (evaluator 3).  Size-independent properties of every row, run-to-run determinism, and 24 rows -- both plies --
    bit for bit against the CPU restatement at 400 simulations."""
    import hashlib
    from chinesecheckersagent_amd import _lib
    G, S, seed = 4096, 400, 20261003

    def run():
        e = eng.SelfPlayEngine(n_slots=G, sims=S, seed=seed, max_games=G, log_capacity=G * 4)
        e.play_plies(_lib.EVAL_ROLLOUT, 6)
        e.play_plies(_lib.EVAL_ROLLOUT, 2)
        st, meta, pi = e.log()
        c = e.counters()
        roots = {s: e.read_root(s) for s in range(0, G, 171)}
        e.close()
        order = np.lexsort((meta['ply'], meta['game']))
        return st[order], meta[order], pi[order], c, roots
    st, meta, pi, c, roots = run()
    assert len(meta) == 2 * G and c['errors'] == 0
    assert c['sims'] == 2 * G * S and c['expansions'] + c['terminal_sims'] == 2 * G * (S + 1)
    assert np.abs(pi.sum(axis=1) - 1.0).max() < 1e-12 and (pi >= 0).all()
    assert np.allclose(pi * S, np.round(pi * S), atol=1e-9)             # tau = 1: pi = N / 400
    for s, r in roots.items():
        assert int(r['N'].sum()) == S and len(set(int(m) for m in r['mv'])) == len(r['mv'])
        assert np.abs(r['W']).max() <= S                                 # a playout's value is -1, 0 or +1
    for r in range(0, 2 * G, 509):                                       # pi lives on legal moves only
        legal = set(int(a) * 49 + int(b) for a, b in orc.movegen(st[r]['pos'].reshape(12), int(meta[r]['player'])))
        assert set(int(i) for i in np.nonzero(pi[r])[0]) <= legal
    digest = hashlib.sha256(st.tobytes() + pi.tobytes()).hexdigest()
    st2, meta2, pi2, c2, _ = run()
    assert hashlib.sha256(st2.tobytes() + pi2.tobytes()).hexdigest() == digest and c2 == c, 'two runs with one seed differ'
    for r in range(0, 2 * G, 341):                                       # 24 rows, both plies, against the CPU restatement at 400 simulations
        o = orc.search(st[r]['pos'].reshape(12), st[r]['last'], int(meta[r]['player']), seed, int(meta[r]['game']),
                       int(meta[r]['ply']), S, False, 3)
        assert np.array_equal(pi[r], np.array(o.pi[:])), 'row %d differs from the CPU restatement of the rollout evaluator' % r


def test_config5_simulation_count_800(eng):
    """SURVEY.md §8d config 5 searches with 800 simulations per move: above the 510 that the reciprocal table in LDS
    covers, so the fused kernel takes its IEEE-division path and the wider tables -- same results, bit for bit"""
    from chinesecheckersagent_amd import _lib
    G, S, seed = 6, 800, 424242
    for ev in (_lib.EVAL_HASH, _lib.EVAL_FORWARD):
        e = eng.SelfPlayEngine(n_slots=G, sims=S, seed=seed, first_game=50, max_games=G, log_capacity=G * 8)
        e.play_plies(ev, 6)
        e.play_plies(ev, 2)
        st, meta, pi = e.log()
        c = e.counters()
        e.close()
        assert len(meta) == 2 * G and c['errors'] == 0 and c['sims'] == 2 * G * S
        for r in range(len(meta)):
            o = orc.search(st[r]['pos'].reshape(12), st[r]['last'], int(meta[r]['player']), seed, int(meta[r]['game']),
                           int(meta[r]['ply']), S, False, ev)
            assert np.array_equal(pi[r], np.array(o.pi[:])), 'row %d (evaluator %d) differs from the oracle' % (r, ev)


def test_fuzz_every_ply_of_many_games_against_oracle(eng):
    """every searched ply of 2 x 384 whole games (openings, middle games, wins, terminal simulations, both tau
    regimes, discard rules) against the CPU oracle, row by row"""
    from chinesecheckersagent_amd import _lib
    seed, S, G = 99, 32, 384
    for ev in (_lib.EVAL_FORWARD, _lib.EVAL_HASH):
        e = eng.SelfPlayEngine(n_slots=G, sims=S, seed=seed, first_game=1000, max_games=G, log_capacity=G * 260)
        for _ in range(40):
            e.play_plies(ev, 16)
            if (e.slots()['status'] != 0).all():
                break
        st, meta, pi = e.log()
        res = e.results()
        c = e.counters()
        e.close()
        assert c['errors'] == 0 and (res['status'] != 0).all()
        if ev == _lib.EVAL_FORWARD:
            assert ((res['status'] == 1) | (res['status'] == 2)).sum() > G // 2        # most games under this evaluator are won
        assert len(meta) > G * 10
        taus = set()
        for r in range(len(meta)):
            ply = int(meta[r]['ply'])
            det = ply > 16                                                            # selfplay.py:62-65
            taus.add(det)
            o = orc.search(st[r]['pos'].reshape(12), st[r]['last'], int(meta[r]['player']), seed, int(meta[r]['game']), ply, S, det, ev)
            assert np.array_equal(pi[r], np.array(o.pi[:])), 'game %d ply %d (evaluator %d) differs from the oracle' % (
                int(meta[r]['game']), ply, ev)
        assert taus == {False, True}


def test_pow_overflow_at_det_tau_ends_the_game_with_an_error(eng):
    """MCTS.py:132: N**100 leaves float64 at N = 1210 and the reference raises OverflowError.  Twelve near-win roots
    under 2000 simulations at tau = 0.01: a slot whose root holds an edge with >= 1210 visits ends with status ERROR (counted, no
    log row, no NaN pi) exactly where the restatement reports the overflow; the other slots give the restatement's pi bit for bit;
    the same roots at tau = 1 all play"""
    from chinesecheckersagent_amd import _lib
    seed, n, sims = 99, 12, 2000
    pos = [orc.near_win_pos12(seed, g, 1) for g in range(n)]
    states = _lib.pack_states(pos)
    for det in (1, 0):
        e = eng.SelfPlayEngine(n_slots=n, sims=sims, seed=seed, max_games=n, log_capacity=n)
        e.set_positions(states, [1] * n, list(range(n)), [20] * n, [det] * n)
        e.play_plies(_lib.EVAL_UNIFORM, 1)
        st, meta, pi = e.log()
        row_of = {int(m['game']): r for r, m in enumerate(meta)}
        status = e.slots()['status']
        errors = 0
        for g in range(n):
            try:
                o = orc.search(pos[g], orc.NO_LAST, 1, seed, g, 20, sims, bool(det), 0)
            except RuntimeError as ex:
                assert '-3' in str(ex) and det == 1
                errors += 1
                assert int(status[g]) == _lib.ST_ERROR and g not in row_of, g
                continue
            assert int(status[g]) != _lib.ST_ERROR and np.array_equal(pi[row_of[g]], np.array(o.pi[:])), g
        assert np.isfinite(pi).all()
        assert e.counters()['errors'] == errors and (errors >= 2 if det else errors == 0)
        e.close()
    # searches of up to 1209 simulations cannot overflow: ccsp_create still takes them, and more (up to 4000) for tau = 1 callers
