"""The DELIVERED kernels -- advance_kernel / boundary_kernel behind ccsp_advance / ccsp_boundary, the free-running path bench.py times,
with tree reuse (the one place where the path departs from selfplay.py:130-133: the reference throws the tree away) -- pinned DIRECTLY
to the reference's own results, through the C ABI:

  (a) every game of tests/golden/games.json (made by the imported reference: selfplay.py:11-133 on substituted draws): status, reward,
      ply count, evaluator calls, every recorded position, every pi, the convert_to_train_data hashes -- single-model games with
      CCSP_ADVANCE_REUSE, two-model games without; at the default limits of a call and with every limit at one tick;
  (b) the 109 make_move() cases of tests/golden/tree.json (MCTS.py:49-153): root N, W / P / pi bits, the sampled move, the whole-tree
      digest -- with and without reuse;
  (c) 1536 whole games through 1024 free-running RESTARTING slots against the CPU oracle, every searched ply of every game.

The evaluator of (a) and (b) is a table evaluator answered on the host FROM THE PLANES of the requests (ccsp_encode_requests: C1 on the
request records; the same decoding as oracle/harness/refenv.TableModel), handed back as the compact answer the C ABI takes."""
import ctypes as C
import hashlib
import json

import numpy as np
import pytest

import oracle_ffi as orc
from test_gpu_tree import _check_case, _decode_planes, _groups, _table_eval

pytestmark = pytest.mark.gpu


class FreeRunner(object):
    """[evaluate the requests -> ccsp_advance -> ccsp_boundary] round by round through the C ABI, the evaluator on the host"""

    def __init__(self, eng, evaluators, reuse, limits=None):
        import torch
        from chinesecheckersagent_amd import _lib
        self.torch, self._lib, self.L = torch, _lib, _lib.lib()
        self.e, self.reuse = eng, bool(reuse)
        self.ev = list(evaluators) if isinstance(evaluators, (list, tuple)) else [evaluators, evaluators]
        n = eng.n_slots
        self.req, self.moves, self.pk, self.v = eng.request_buffers()           # zero-filled: nothing asked yet
        self.model_sel = torch.zeros(n, dtype=torch.uint8, device='cuda')
        self.planes = torch.zeros((n, 343), dtype=torch.float32, device='cuda')
        if self.reuse:
            eng.enable_tree_reuse()
        if limits is not None:
            eng.set_advance_limits(*limits)
        self.cache = {}
        self.rounds = self.asked = 0

    def answer(self):
        """the requests on the table -> (pk, v): planes of every request (ccsp_encode_requests), decoded and evaluated on the host"""
        _lib, L, torch = self._lib, self.L, self.torch
        n = self.e.n_slots
        _lib.check(L.ccsp_encode_requests(self.req.data_ptr(), n, self.planes.data_ptr(), None), 'ccsp_encode_requests')
        torch.cuda.synchronize()
        req = self.req.cpu().numpy().view(_lib.REQUEST_DTYPE).reshape(n)
        ask = np.nonzero(req['kind'])[0]
        if len(ask) == 0:
            return
        planes = self.planes.cpu().numpy()
        moves = self.moves.cpu().numpy().view(np.uint16)
        sel = self.model_sel.cpu().numpy()
        pk = np.zeros((n, _lib.REQUEST_MOVES))
        v = np.zeros(n, dtype=np.float32)
        for s in ask:
            ev = self.ev[int(sel[s])]
            key = (ev, planes[s].tobytes())
            if key not in self.cache:
                pos12, player = _decode_planes(planes[s])
                assert player == int(req['player'][s]) and pos12 == [int(x) for x in req['state']['pos'][s].reshape(12)]
                self.cache[key] = _table_eval(ev, pos12, player)
            p, val = self.cache[key]
            k = int(req['k'][s])
            pk[s, :k] = p[moves[s, :k] & 0x1FF]
            v[s] = val
        self.asked += len(ask)
        self.pk.copy_(torch.from_numpy(pk))
        self.v.copy_(torch.from_numpy(v))

    def round(self):
        e = self.e
        self.answer()
        e.advance(self.pk, self.v, self.req, self.moves, self.model_sel, reuse=self.reuse)
        e.boundary(self.pk, self.v, self.req, self.moves, self.model_sel, reuse=self.reuse)
        self.rounds += 1


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('limits', [None, (64, 1, 1)], ids=['default-limits', 'one-tick-limits'])
def test_reference_games_through_advance_and_boundary(golden_dir, limits):
    """(a) all 26 reference games, one context per game id (the ids are not contiguous)"""
    from chinesecheckersagent_amd import _lib, engine, utils
    doc = json.load(open(golden_dir + '/games.json'))
    seed = doc['seed']
    seen, hits = set(), 0
    for g in doc['games']:
        two = isinstance(g['evaluator'], list)
        e = engine.SelfPlayEngine(n_slots=1, sims=g['sims'], seed=seed, first_game=g['game'], max_games=1, log_capacity=1024,
                                  randomised=g['randomised'])
        fr = FreeRunner(e, g['evaluator'], reuse=not two, limits=limits)
        tag = 'game %d (%s)' % (g['game'], 'two models' if two else 'reuse')
        for i in range(400000):
            fr.round()
            if i % 64 == 63 and e.slots()['status'][0] != _lib.ST_RUNNING:
                break
        res = e.results()[0]
        status = {1: 'won', 2: 'won', 3: 'repetition', 4: 'no_progress'}[int(res['status'])]
        assert status == g['status'], tag
        seen.add(status)
        assert int(res['n_plies']) == len(g['plies']), tag
        assert int(res['expansions']) == g['evals'], tag              # positions reused from the previous tree are expansions all the same
        c = e.counters()
        assert c['errors'] == 0 and c['expansions'] == g['evals'], tag
        if two:
            assert c['cache_hits'] == 0 and fr.asked == g['evals'], tag               # every expansion asked the evaluator, as the reference does
        else:
            assert fr.asked + c['cache_hits'] == g['evals'], tag       # the evaluator was asked for the rest ...
            hits += c['cache_hits']
        st, meta, pi = e.log()
        order = np.argsort(meta['ply'])
        st, meta, pi = st[order], meta[order], pi[order]
        assert (meta['game'] == g['game']).all()
        assert [int(x) for x in meta['ply']] == [i for i, p in enumerate(g['plies']) if p[0] != 0], tag
        if status == 'won':
            assert int(res['reward']) == g['reward'], tag
            drop = 3 if g['randomised'] else 0                 # selfplay.py:76-78
            assert [[int(x) for x in s['pos'].reshape(12)] for s in st[drop:]] == g['hist_pos12'], tag
            assert [_sha(np.asarray(r, dtype='<f8'))[:16] for r in pi[drop:]] == g['pi_sha'], tag
            # O1: utils.convert_to_train_data's arrays from the log (utils.py:60-73)
            bx, py, vy = utils.log_to_train_data(st, meta, pi, e.results(), first_game=g['game'], game_stride=1, randomised=g['randomised'])
            assert _sha(bx.astype('<f8')) == g['o1']['board_x_sha'] and _sha(py.astype('<f8')) == g['o1']['pi_y_sha'], tag
            assert [int(x) for x in vy] == g['o1']['v_y'], tag
        e.close()
    assert seen == {'won', 'repetition', 'no_progress'}
    assert hits > 10000                                         # ... and the reuse path really ran


def test_request_buffer_is_output_only(golden_dir):
    """The caller's request buffer is an OUTPUT of ccsp_advance / ccsp_boundary: where a new node hangs, the path length, a walk to be
    resumed, even the leaf's position and k live in the context's own record.  After every evaluation this test overwrites each record's
    position, k and reserved words with 0xFF bytes (kind and player stay: they say which rows the NEXT evaluation reads) -- what a caller
    that re-uses, swaps or scribbles on its buffer does -- at one-tick limits, so that resumed walks are exercised too: the reference's
    games come out bit for bit all the same (before round 6 the kernels took link offsets and walk state from these words)."""
    import torch
    from chinesecheckersagent_amd import _lib, engine

    class Scribbler(FreeRunner):
        def round(self):
            e = self.e
            self.answer()
            w = self.req.view(torch.int32)                       # [n, 16]: state 0-7 | kind 8 | reserved 9, 10 | player 11 | k 12 | reserved 13-15
            w[:, 0:8] = -1
            w[:, 9:11] = -1
            w[:, 12:16] = -1
            e.advance(self.pk, self.v, self.req, self.moves, self.model_sel, reuse=self.reuse)
            e.boundary(self.pk, self.v, self.req, self.moves, self.model_sel, reuse=self.reuse)
            self.rounds += 1

        def answer(self):
            # (the records are read BEFORE the scribble of this round and after the engine's writes of the last one: whole again where kind != 0)
            return FreeRunner.answer(self)
    doc = json.load(open(golden_dir + '/games.json'))
    seed = doc['seed']
    games = sorted((g for g in doc['games'] if not isinstance(g['evaluator'], list)), key=lambda g: g['evals'])[:4]
    hits = 0
    for g in games:
        e = engine.SelfPlayEngine(n_slots=1, sims=g['sims'], seed=seed, first_game=g['game'], max_games=1, log_capacity=1024, randomised=g['randomised'])
        fr = Scribbler(e, g['evaluator'], reuse=True, limits=(64, 1, 1))
        for i in range(400000):
            fr.round()
            if i % 64 == 63 and e.slots()['status'][0] != _lib.ST_RUNNING:
                break
        res, c = e.results()[0], e.counters()
        assert {1: 'won', 2: 'won', 3: 'repetition', 4: 'no_progress'}[int(res['status'])] == g['status']
        assert int(res['n_plies']) == len(g['plies']) and int(res['expansions']) == g['evals'] == c['expansions'] and c['errors'] == 0
        assert fr.asked + c['cache_hits'] == g['evals']
        hits += c['cache_hits']
        st, meta, pi = e.log()
        order = np.argsort(meta['ply'])
        if g['status'] == 'won':
            drop = 3 if g['randomised'] else 0
            assert [_sha(np.asarray(r, dtype='<f8'))[:16] for r in pi[order][drop:]] == g['pi_sha']
        e.close()
    assert hits > 0


@pytest.mark.parametrize('reuse', [True, False], ids=['reuse', 'no-reuse'])
def test_reference_make_move_cases_through_advance_and_boundary(golden_dir, reuse):
    """(b) the 109 make_move() cases: ccsp_set_positions -> [answer -> advance -> boundary] until the slot's row is in the log; its root
    statistics and the digest of its whole tree are read at once (the finished tree stays whole while the next ply is searched: with
    reuse in the other pool, without until the next root's answer is taken)"""
    from chinesecheckersagent_amd import _lib, engine
    doc = json.load(open(golden_dir + '/tree.json'))
    seed, cases = doc['seed'], doc['cases']
    done_cases = 0
    for (sims, ev), idxs in sorted(_groups(cases).items()):
        n = len(idxs)
        e = engine.SelfPlayEngine(n_slots=n, sims=sims, seed=seed, max_games=n, log_capacity=4 * n)
        cs = [cases[i] for i in idxs]
        states = _lib.pack_states([c['pos12'] for c in cs], [c['last'] for c in cs])
        e.set_positions(states, [c['player'] for c in cs], [c['game'] for c in cs], [c['nplies'] for c in cs],
                        [0 if c['tau'] == 1 else 1 for c in cs])
        fr = FreeRunner(e, ev, reuse=reuse)
        slot_of = {c['game']: s for s, c in enumerate(cs)}
        assert len(slot_of) == n
        checked = set()
        have = 0
        for _ in range(40 * (sims + 8)):
            fr.round()
            size = e.log_size()
            if size == have:
                continue
            st, meta, pi = e.log(have, size - have)
            have = size
            slots = None
            for r in range(len(meta)):
                game, s = int(meta[r]['game']), slot_of[int(meta[r]['game'])]
                c = cs[s]
                if int(meta[r]['ply']) != c['nplies'] or s in checked:
                    continue                                    # (a slot's SECOND ply: not a case)
                slots = e.slots() if slots is None else slots
                tag = 'free-running case %d (ev=%d sims=%d tau=%s start=%s reuse=%s)' % (idxs[s], ev, sims, c['tau'], c['start'], reuse)
                assert [int(x) for x in st[r]['pos'].reshape(12)] == c['pos12'] and int(meta[r]['player']) == c['player'], tag
                _check_case(c, e.read_root(s), pi[r], e.tree_digest(s), slots['state'][s]['pos'].reshape(12), tag)
                checked.add(s)
            if len(checked) == n:
                break
        assert len(checked) == n and e.counters()['errors'] == 0
        done_cases += n
        e.close()
    assert done_cases == len(cases) == 109


class TableEvaluator(object):
    """the built-in table evaluators answering requests on the device (ccsp_debug_table_eval): an `evaluate_requests` model"""
    backend = 'hip'

    def __init__(self, kind):
        import torch
        self.kind, self.device = kind, torch.device('cuda')
        self.calls = 0

    def evaluate_requests(self, req, moves, pk=None, v=None):
        import torch
        from chinesecheckersagent_amd import _lib
        from chinesecheckersagent_amd.engine import _stream_ptr
        n = req.shape[0]
        pk = torch.zeros((n, moves.shape[1]), dtype=torch.float64, device=req.device) if pk is None else pk
        v = torch.zeros(n, dtype=torch.float32, device=req.device) if v is None else v
        _lib.check(_lib.lib().ccsp_debug_table_eval(self.kind, req.data_ptr(), moves.data_ptr(), n, pk.data_ptr(), v.data_ptr(), _stream_ptr()),
                   'ccsp_debug_table_eval')
        self.calls += 1
        return pk, v

    def evaluate_batch(self, x):                                # (the batched-model interface; never called on the free-running path)
        raise AssertionError('the free-running path asks through evaluate_requests')


@pytest.mark.parametrize('ev', [2, 1], ids=['forward', 'hash'])
def test_restarting_free_running_slots_play_the_oracles_games(ev):
    """(c) 1536 whole games through 1024 free-running restarting slots (every second slot plays a second game; tree reuse, log guard,
    staggered starts, boundary every sixth round: what SelfPlayRun configures at this size) against orc_selfplay: status, reward, ply
    count, evaluator calls and EVERY pi of every game -- the discarded ones included"""
    from chinesecheckersagent_amd import _lib, selfplay as sp
    n_slots, n_games, sims, seed, first = 1024, 1536, 24, 4711, 30000
    m = TableEvaluator(ev)
    b = sp.BatchSelfPlay(m, n_slots=n_slots, sims=sims, seed=seed, first_game=first, max_games=n_games, auto_restart=True,
                         log_capacity=n_games * 400, free_running=True, reuse=True, stagger=True, stagger_span=64, use_graph=False)
    for i in range(4000):
        b.play_steps(64)
        if i % 8 == 7 and (b.eng.slots()['status'] != _lib.ST_RUNNING).all():
            break
    c = b.eng.counters()
    res = b.eng.results()
    st, meta, pi = b.eng.log()
    b.close()
    assert c['errors'] == 0 and c['games_won'] + c['games_discarded'] == n_games and (res['status'] != 0xFF).all()
    assert c['cache_hits'] > 0.1 * c['expansions'] and m.calls > 1000
    rows = {}
    for r in np.lexsort((meta['ply'], meta['game'])):
        rows.setdefault(int(meta['game'][r]), []).append(r)
    kinds = set()
    for k in range(n_games):
        game = first + k
        o = orc.selfplay(seed, game, sims, ev)
        tag = 'game %d' % game
        assert o['status'] == int(res['status'][k]) and len(o['plies']) == int(res['n_plies'][k]) and o['evals'] == int(res['expansions'][k]), tag
        assert o['n_searched'] == int(res['n_samples'][k]) == len(rows.get(game, [])), tag
        kinds.add(o['status'])
        if o['status'] in (orc.ST_WON_P1, orc.ST_WON_P2):
            assert o['reward'] == int(res['reward'][k]), tag
        want_pos, want_pi = o['searched_pos12'], o['searched_pi']
        for j, r in enumerate(rows.get(game, [])):
            assert [int(x) for x in st[r]['pos'].reshape(12)] == [int(x) for x in want_pos[j]], (tag, j)
            assert np.array_equal(pi[r], want_pi[j]), 'pi of searched ply %d of %s differs from the oracle' % (j, tag)
    # (games END in wins under the forward evaluator; under the hash evaluator -- random priors and values -- nearly all are discarded)
    assert orc.ST_DISCARD_NO_PROGRESS in kinds or orc.ST_DISCARD_REPETITION in kinds
    assert ev != 2 or {orc.ST_WON_P1, orc.ST_WON_P2} & kinds


def test_requests_evaluated_in_place_equal_planes_evaluated(golden_dir):
    """ccsp_net_forward_requests (C1 inside the evaluator's input phase, compact answer from its epilogue) == ccsp_encode_requests ->
    ccsp_net_forward -> ccsp_gather_priors, bit for bit, in every workgroup shape and for ragged batches; rows that ask for nothing are
    left alone.  The requests: real positions with their histories (the reference-made rules fixture) and their real move lists."""
    import torch
    from chinesecheckersagent_amd import _lib, rules
    from chinesecheckersagent_amd.model import ResidualCNN, evaluate_requests_with
    L = _lib.lib()
    z = np.load(golden_dir + '/rules.npz')
    pos12, last, player = z['pos12'], z['last'], z['player']
    m = ResidualCNN()
    m.load_weights(golden_dir + '/good_model.h5')
    assert m.backend == 'hip'
    rng = np.random.RandomState(5)
    for n in (1, 2, 3, 8, 19, 256, 515, 1027, 2048):
        pick = rng.choice(len(pos12), n, replace=len(pos12) < n)
        states = _lib.pack_states(pos12[pick], last[pick])
        pl = player[pick].astype(np.uint8)
        mv, cnt, _ = rules.movegen(rules.to_device_states(states), pl)
        mv, cnt = mv.cpu().numpy(), cnt.cpu().numpy()
        req = np.zeros(n, dtype=_lib.REQUEST_DTYPE)
        req['state'], req['player'], req['k'] = states, pl, cnt
        req['kind'] = np.where(np.arange(n) % 7 == 3, 0, 1 + 2 * (np.arange(n) % 2))        # kinds 1 and 3 ask, 0 does not
        moves = np.zeros((n, _lib.REQUEST_MOVES), dtype=np.uint16)
        for i in range(n):
            k = int(cnt[i])
            moves[i, :k] = mv[i, :k, 0].astype(np.uint16) * 49 + mv[i, :k, 1]
            moves[i, :k:5] |= 0x8000                                                       # (the "wins" mark is not part of the index)
        d_req = torch.from_numpy(req.view(np.uint8).reshape(n, 64)).cuda()
        d_mv = torch.from_numpy(moves.view(np.int16)).cuda()
        want_pk, want_v = evaluate_requests_with(m.evaluate_batch, d_req, d_mv)
        asks = torch.from_numpy(req['kind'] != 0).cuda()
        for shape in (0, 8, 4, 2, 1):
            L.ccsp_debug_net_shape(shape)
            try:
                pk = torch.full((n, _lib.REQUEST_MOVES), -7.0, dtype=torch.float64, device='cuda')
                v = torch.full((n,), -7.0, dtype=torch.float32, device='cuda')
                m.evaluate_requests(d_req, d_mv, pk, v)
                torch.cuda.synchronize()
            finally:
                L.ccsp_debug_net_shape(0)
            live = torch.arange(_lib.REQUEST_MOVES, device='cuda')[None, :] < torch.from_numpy(cnt.astype(np.int64)).cuda()[:, None]
            live &= asks[:, None]
            assert torch.equal(pk[live], want_pk[live]), (n, shape)
            assert (pk[~live] == -7.0).all() and torch.equal(v[asks], want_v[asks]) and (v[~asks] == -7.0).all(), (n, shape)
            assert (pk[live] > 0).all() and abs(float(want_v[asks].abs().max())) <= 1.0
