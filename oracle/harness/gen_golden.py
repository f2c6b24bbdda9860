#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (imported
read-only from /root/reference, see refenv.py) on seeded inputs.

Run in the build container only:
    cd oracle/harness && /opt/conda/bin/python3.9 gen_golden.py [rules] [tree] [games] [rng]

Only data leaves this script: inputs and the reference's outputs.  (SURVEY.md §8c G1-G6, G8;
the net vectors G7 come from gen_net_golden.py.)
"""
import contextlib
import copy
import hashlib
import io
import json
import os
import struct
import sys
import time

import numpy as np

import refenv
import spec
from refenv import ctx, ref_board, ref_utils, ref_mcts, ref_selfplay, ref_config

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')
SEED = 20261003


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def f64bits(x):
    return struct.unpack('<Q', struct.pack('<d', float(x)))[0]


def set_position(b, pos12):
    """put the 12 checkers of a fresh Board on the given cells (same bookkeeping as board.py:61-85)."""
    b.board[:, :, 0] = 0
    b.checkers_pos = [None, {}, {}]
    b.checkers_id = [None, {}, {}]
    for pl in (1, 2):
        for cid in range(6):
            cell = pos12[(pl - 1) * 6 + cid]
            rc = (cell // 7, cell % 7)
            b.board[rc][0] = pl
            b.checkers_pos[pl][cid] = rc
            b.checkers_id[pl][rc] = cid


# --------------------------------------------------------------------------------------------
# G1-G4: rules.  Trajectories of seeded random play (rule S1 = selfplay.make_random_move),
# game kinds by g % 4: two normal starts, one randomised board (B8), one crafted near-win.  Each ply yields one record.

def gen_rules(n_games=1700, max_plies=60, n_explicit=3000):
    recs = []
    t0 = time.time()
    for g in range(n_games):
        ctx.seed, ctx.game = SEED, g
        kind = g % 4                      # 0,1 normal start; 2 randomised (B8); 3 crafted near-win
        b = ref_board.Board(randomised=(kind == 2))
        if kind == 3:
            set_position(b, spec.near_win_position(SEED, g, 1 + (g // 4) % 2))
        root = ref_mcts.Node(b, 1)
        for ply in range(max_plies):
            ctx.ply, ctx.draw = ply, 0
            st = root.state
            player = root.currPlayer
            if st.check_win():
                break
            pos12 = refenv.pos12_of(st)
            last = refenv.last_moves_of(st)
            vm = st.get_valid_moves(player)                                   # B4
            moves = []
            for cpos, dests in vm.items():
                cid = st.checkers_id[player][cpos]
                for d in dests:
                    moves.append((cid, d[0] * 7 + d[1]))
            planes = ref_utils.to_model_input(st, player)                     # C1
            assert planes.dtype == np.float64
            progress = (st.player_progress(1), st.player_progress(2))         # B7
            with quiet():
                nxt = ref_selfplay.make_random_move(root)                     # S1 + B5 + B6
            npos12 = refenv.pos12_of(nxt.state)
            nlast = refenv.last_moves_of(nxt.state)
            mover = [i for i in range(12) if pos12[i] != npos12[i]]
            assert len(mover) == 1
            cid = mover[0] % 6
            chosen = (cid, npos12[mover[0]])
            assert chosen in moves
            winner = nxt.state.check_win()
            recs.append(dict(game=g, ply=ply, pos12=pos12, player=player, last=last, moves=moves,
                             planes=planes.astype(np.uint8).reshape(-1), chosen=chosen,
                             npos12=npos12, nlast=nlast, winner=winner, progress=progress,
                             nboard=nxt.state.board.copy()))
            root = nxt
        if g % 100 == 0:
            print('rules: game', g, 'records', len(recs), '%.0fs' % (time.time() - t0), file=sys.stderr)
    n = len(recs)
    # digests over ALL records (canonical little-endian byte strings)
    h_moves, h_step, h_planes = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()
    for r in recs:
        h_moves.update(bytes(r['pos12']) + bytes([r['player'], len(r['moves'])]) +
                       bytes([x for m in r['moves'] for x in m]))
        h_step.update(bytes(r['pos12']) + bytes([r['player']]) + bytes(r['chosen']) +
                      bytes(r['npos12']) + bytes(r['nlast']) +
                      bytes([r['winner'], r['progress'][0], r['progress'][1]]))
        h_planes.update(bytes(r['pos12']) + bytes([r['player']]) + bytes(r['last']) + r['planes'].tobytes())
    ex = recs[:n_explicit]
    # keep explicit records that cover rarely hit cases too: wins, big move lists, randomised boards
    extra = [r for r in recs[n_explicit:] if r['winner'] != 0 or len(r['moves']) >= 60][:400]
    ex = ex + extra
    counts = np.array([len(r['moves']) for r in ex], dtype=np.uint8)
    flat = np.array([x for r in ex for m in r['moves'] for x in m], dtype=np.uint8).reshape(-1, 2)
    np.savez_compressed(
        os.path.join(OUT, 'rules.npz'),
        seed=np.uint64(SEED), n_games=np.int32(n_games), max_plies=np.int32(max_plies),
        n_records=np.int64(n),
        game=np.array([r['game'] for r in ex], dtype=np.int32),
        ply=np.array([r['ply'] for r in ex], dtype=np.int32),
        pos12=np.array([r['pos12'] for r in ex], dtype=np.uint8),
        player=np.array([r['player'] for r in ex], dtype=np.uint8),
        last=np.array([r['last'] for r in ex], dtype=np.uint8),
        move_count=counts, moves=flat,
        planes=np.array([r['planes'] for r in ex], dtype=np.uint8),
        chosen=np.array([r['chosen'] for r in ex], dtype=np.uint8),
        npos12=np.array([r['npos12'] for r in ex], dtype=np.uint8),
        nlast=np.array([r['nlast'] for r in ex], dtype=np.uint8),
        winner=np.array([r['winner'] for r in ex], dtype=np.uint8),
        progress=np.array([r['progress'] for r in ex], dtype=np.uint8),
        nboard=np.array([r['nboard'] for r in ex], dtype=np.uint8),
        # all-record side info so a test can replay the same trajectories with its own rules engine
        all_game=np.array([r['game'] for r in recs], dtype=np.int32),
        all_count=np.array([len(r['moves']) for r in recs], dtype=np.uint8),
        sha_moves=np.frombuffer(h_moves.digest(), dtype=np.uint8),
        sha_step=np.frombuffer(h_step.digest(), dtype=np.uint8),
        sha_planes=np.frombuffer(h_planes.digest(), dtype=np.uint8))
    mc = np.array([len(r['moves']) for r in recs])
    print('rules: %d records, moves mean %.2f max %d, wins %d' %
          (n, mc.mean(), mc.max(), sum(1 for r in recs if r['winner'])), file=sys.stderr)

    gen_wins()

    # G4: index codec, all 294 (utils.py:164-183)
    enc = []
    for cid in range(6):
        for r in range(7):
            for c in range(7):
                i = ref_utils.encode_checker_index(cid, (r, c))
                cid2, dest = ref_utils.decode_checker_index(i)
                assert (cid2, dest) == (cid, (r, c))
                enc.append((cid, r, c, i))
    np.save(os.path.join(OUT, 'codec.npy'), np.array(enc, dtype=np.int32))


def gen_wins(n_pos=600):
    """B6 coverage: crafted one-move-from-winning positions, EVERY legal move of the near-winner
    applied (deepcopy + Board.place) and the returned winner recorded; plus static positions."""
    rows = []
    wins = 0
    for i in range(n_pos):
        who = 1 + i % 2
        pos12 = spec.near_win_position(SEED, 100000 + i, who)
        b = ref_board.Board()
        set_position(b, pos12)
        vm = b.get_valid_moves(who)
        for cpos, dests in vm.items():
            cid = b.checkers_id[who][cpos]
            for d in dests:
                nb = copy.deepcopy(b)
                w = nb.place(who, cpos, d)
                wins += (w != 0)
                rows.append(pos12 + [who, cid, d[0] * 7 + d[1], w])
    static = []
    t1, t2 = spec.TARGET[1], spec.TARGET[2]
    mid = [21, 22, 23, 24, 25, 26]
    for pos12 in (t1 + t2,                 # both complete: player 1 has priority (board.py:111)
                  mid + t2,                # only player 2 complete
                  t1 + mid,                # only player 1 complete
                  t2 + t1,                 # each sits in its OWN start corner: nobody has won
                  t1[:5] + [21] + t2,      # player 1 one short, player 2 complete
                  mid + t2[:5] + [27],     # nobody
                  [42, 35, 43, 28, 36, 44, 6, 13, 5, 20, 12, 4]):   # initial position
        b = ref_board.Board()
        set_position(b, pos12)
        static.append(pos12 + [b.check_win(), b.player_progress(1), b.player_progress(2)])
    np.savez_compressed(os.path.join(OUT, 'wins.npz'), moves=np.array(rows, dtype=np.uint8),
                        static=np.array(static, dtype=np.uint8))
    print('wins: %d move rows, %d winning' % (len(rows), wins), file=sys.stderr)


# --------------------------------------------------------------------------------------------
# G5: one make_move() (root expansion + noise + S simulations + pi + sampling) per case.

def board_after_random_plies(game, nplies, randomised=False):
    ctx.seed, ctx.game = SEED, game
    b = ref_board.Board(randomised=randomised)
    root = ref_mcts.Node(b, 1)
    for ply in range(nplies):
        ctx.ply, ctx.draw = ply, 0
        if root.state.check_win():
            break
        with quiet():
            root = ref_selfplay.make_random_move(root)
    return root


def tree_digest(root):
    """depth-first over nodes, each node's edges in list order: chain of spec.mix64 over
    (N, bits(W), bits(P)) of every edge."""
    h = 0
    nodes = edges = 0
    stack = [root]
    while stack:
        node = stack.pop()
        nodes += 1
        for e in node.edges:
            edges += 1
            h = spec.mix64(h ^ int(e.stats['N']))
            h = spec.mix64(h ^ f64bits(e.stats['W']))
            h = spec.mix64(h ^ f64bits(e.stats['P']))
        for e in reversed(node.edges):
            if e.outNode.edges:
                stack.append(e.outNode)
    return h, nodes, edges


def gen_tree(wide_only=False):
    cases = []
    plan = []
    cid = 0
    for ev in (spec.EVAL_UNIFORM, spec.EVAL_HASH, spec.EVAL_FORWARD):
        for sims, tau, nplies, count in ((50, 1, 6, 6), (50, 0.01, 14, 4), (400, 1, 6, 2), (400, 0.01, 20, 2),
                                         (175, 1, 30, 2), (50, 1, 40, 3)):
            for j in range(count):
                plan.append((ev, sims, tau, nplies, 1000 + cid, False))
                cid += 1
        plan.append((ev, 50, 1, 4, 1000 + cid, True)); cid += 1      # randomised board
        plan.append((ev, 50, 0.01, 9, 1000 + cid, True)); cid += 1
    # crafted near-win roots (terminal leaves inside the search, MCTS.py:81-90): ids whose position
    # has an immediately winning move, and a few that do not
    winners = {1: [], 2: []}
    for i in range(600):
        who = 1 + i % 2
        b = ref_board.Board()
        set_position(b, spec.near_win_position(SEED, 100000 + i, who))
        vm = b.get_valid_moves(who)
        haswin = False
        for cpos, dests in vm.items():
            for d in dests:
                nb = copy.deepcopy(b)
                if nb.place(who, cpos, d):
                    haswin = True
        if haswin:
            winners[who].append(i)
    for ev in (spec.EVAL_UNIFORM, spec.EVAL_HASH, spec.EVAL_FORWARD):
        for who in (1, 2):
            for i in winners[who][:3]:
                plan.append((ev, 50, 1, 31, 100000 + i, ('near', who)))
                plan.append((ev, 175, 0.01, 31, 100000 + i, ('near', who)))
            plan.append((ev, 50, 1, 31, 100000 + who - 1, ('near', who)))
    # roots with MORE THAN 64 legal moves (found in the rules trajectories: randomised games 418, 398, 438, 70 at
    # plies 6, 23, 38, 56 have 72, 69, 68, 67 moves): the second half-wave of every per-edge loop on the GPU
    wide = [(spec.EVAL_HASH, 50, 1, 6, 418, True), (spec.EVAL_UNIFORM, 50, 0.01, 23, 398, True),
            (spec.EVAL_FORWARD, 175, 1, 38, 438, True), (spec.EVAL_HASH, 400, 0.01, 56, 70, True)]
    if wide_only:
        plan = wide
        cases = json.load(open(os.path.join(OUT, 'tree.json')))['cases']
        cases = [c for c in cases if not (c['start'] == 'randomised' and len(c['N']) > 64)]
    else:
        plan = plan + wide
    t0 = time.time()
    for ev, sims, tau, nplies, game, randomised in plan:
        refenv.set_sims(sims)
        if isinstance(randomised, tuple):
            who = randomised[1]
            ctx.seed, ctx.game = SEED, game
            b = ref_board.Board()
            set_position(b, spec.near_win_position(SEED, game, who))
            root = ref_mcts.Node(b, who)
            randomised = 'near'
        else:
            root = board_after_random_plies(game, nplies, randomised)
        if root.state.check_win():
            continue
        ctx.ply = nplies
        model = refenv.TableModel(ev)
        hist = []
        pos12 = refenv.pos12_of(root.state)
        last = refenv.last_moves_of(root.state)
        player = root.currPlayer
        # run make_move but keep the tree: re-implement nothing -- call it, then read the tree we kept
        kept = {}
        orig_search = ref_mcts.MCTS.search

        def spy_search(self):
            out = orig_search(self)
            kept['tree'] = self
            kept['stats'] = [(e.stats['N'], e.stats['W'], e.stats['Q'], e.stats['P'],
                              self.root.state.checkers_id[self.root.currPlayer][e.fromPos],
                              e.toPos[0] * 7 + e.toPos[1]) for e in self.root.edges]
            kept['wtype'] = type(self.root.edges[0].stats['W']).__name__
            kept['digest'] = tree_digest(self.root)
            return out
        ref_mcts.MCTS.search = spy_search
        try:
            with quiet():
                nxt = ref_selfplay.make_move(root, model, tau, hist)
        finally:
            ref_mcts.MCTS.search = orig_search
        pi = hist[0][1]
        npos12 = refenv.pos12_of(nxt.state)
        mover = [i for i in range(12) if pos12[i] != npos12[i]]
        st = kept['stats']
        digest, nodes, edges = kept['digest']
        cases.append(dict(
            evaluator=ev, sims=sims, tau=tau, nplies=nplies, game=game,
            start=('near' if randomised == 'near' else ('randomised' if randomised else 'normal')),
            pos12=pos12, last=last, player=player,
            N=[s[0] for s in st], W=[f64bits(s[1]) for s in st], Q=[f64bits(s[2]) for s in st],
            P=[f64bits(s[3]) for s in st], cid=[s[4] for s in st], dest=[s[5] for s in st],
            pi_idx=[int(i) for i in np.nonzero(pi)[0]], pi_bits=[f64bits(pi[i]) for i in np.nonzero(pi)[0]],
            chosen=[mover[0] % 6, npos12[mover[0]]],
            evals=model.calls, wtype=kept['wtype'], tree_sha=digest, nodes=nodes, edges=edges))
        print('tree: case %d ev=%d sims=%d tau=%s evals=%d nodes=%d edges=%d wtype=%s %.0fs' %
              (len(cases), ev, sims, tau, model.calls, nodes, edges, kept['wtype'], time.time() - t0),
              file=sys.stderr)
    with open(os.path.join(OUT, 'tree.json'), 'w') as f:
        json.dump(dict(seed=SEED, cases=cases), f)


# --------------------------------------------------------------------------------------------
# G6: whole selfplay() games.

def gen_games(incremental=False, cpu_only=False):
    """cpu_only: a second set (games_cpu.json) that only the oracle is checked against (tests/test_oracle_tree.py) --
    more whole games at the reference's default 175 simulations per move without lengthening the GPU suite"""
    games = []
    plan = []
    gid = 5000
    for ev, sims, count in ((spec.EVAL_FORWARD, 24, 6), (spec.EVAL_FORWARD, 50, 4), (spec.EVAL_FORWARD, 8, 2),
                            (spec.EVAL_HASH, 8, 3), (spec.EVAL_UNIFORM, 8, 2), (spec.EVAL_HASH, 24, 1)):
        for j in range(count):
            plan.append((ev, sims, gid, False))
            gid += 1
    plan.append((spec.EVAL_FORWARD, 24, gid, True)); gid += 1
    plan.append((spec.EVAL_FORWARD, 50, gid, True)); gid += 1
    plan.append((spec.EVAL_HASH, 8, gid, True)); gid += 1
    plan.append((spec.EVAL_FORWARD, 8, 6024, False))      # found by scanning with the oracle: ends by the repetition rule
    # two different evaluators: model1 (forward) moves for player one, model2 (hash) for player two (selfplay.py:30,59)
    plan.append(((spec.EVAL_FORWARD, spec.EVAL_HASH), 16, 6100, False))
    plan.append(((spec.EVAL_HASH, spec.EVAL_FORWARD), 24, 6101, False))
    # the reference's default simulation count (config.py:35) and the benchmark's, whole games
    plan.append((spec.EVAL_FORWARD, 175, 6200, False))
    plan.append((spec.EVAL_FORWARD, 400, 6201, False))
    fname = 'games.json'
    if cpu_only:
        fname = 'games_cpu.json'
        plan = [(spec.EVAL_FORWARD, 175, 6300, False), (spec.EVAL_FORWARD, 175, 6301, True), (spec.EVAL_HASH, 175, 6302, False),
                ((spec.EVAL_FORWARD, spec.EVAL_HASH), 175, 6303, False), (spec.EVAL_FORWARD, 100, 6304, False),
                (spec.EVAL_UNIFORM, 175, 6305, False), (spec.EVAL_FORWARD, 175, 6306, False)]
    path = os.path.join(OUT, fname)
    if incremental and os.path.exists(path):               # keep what is there, add what is missing
        games = json.load(open(path))['games']
        have = set((str(x['evaluator']), x['sims'], x['game'], x['randomised']) for x in games)
        plan = [x for x in plan if (str(list(x[0]) if isinstance(x[0], tuple) else x[0]), x[1], x[2], bool(x[3])) not in have]
    t0 = time.time()
    for ev, sims, game, randomised in plan:
        refenv.set_sims(sims)
        ctx.seed, ctx.game = SEED, game
        if isinstance(ev, tuple):
            model, model_b = refenv.TableModel(ev[0]), refenv.TableModel(ev[1])
        else:
            model, model_b = refenv.TableModel(ev), None
        plies = []
        ply_counter = [0]
        orig_rand, orig_move = ref_selfplay.make_random_move, ref_selfplay.make_move

        def rec_move(kind, root, nxt):
            a = refenv.pos12_of(root.state)
            b = refenv.pos12_of(nxt.state)
            mover = [i for i in range(12) if a[i] != b[i]]
            plies.append([kind, mover[0] % 6, b[mover[0]]])

        def w_rand(root):
            ctx.ply, ctx.draw = ply_counter[0], 0
            nxt = orig_rand(root)
            rec_move(0, root, nxt)
            ply_counter[0] += 1
            return nxt

        def w_move(root, m, tau, hist):
            ctx.ply = ply_counter[0]
            nxt = orig_move(root, m, tau, hist)
            rec_move(1 if tau == ref_config.TREE_TAU else 2, root, nxt)
            ply_counter[0] += 1
            return nxt
        ref_selfplay.make_random_move, ref_selfplay.make_move = w_rand, w_move
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                hist, reward = ref_selfplay.selfplay(model, model_b, randomised)
        finally:
            ref_selfplay.make_random_move, ref_selfplay.make_move = orig_rand, orig_move
        text = buf.getvalue()
        if hist is None:
            status = 'repetition' if 'Repetition' in text else 'no_progress'
            pis = []
            states = []
        else:
            status = 'won'
            pis = [hashlib.sha256(np.asarray(pi, dtype='<f8').tobytes()).hexdigest()[:16] for _, pi in hist]
            states = [refenv.pos12_of(b) for b, _ in hist]
        # the (state, pi, z) -> board_x / pi_y / v_y conversion of utils.py:60-73 (O1)
        if hist is not None:
            bx, py, vy = ref_utils.convert_to_train_data([(hist, reward)])
            bx = np.array(bx)
            py = np.array(py)
            vy = np.array(vy)
            o1 = dict(board_x_sha=hashlib.sha256(bx.astype('<f8').tobytes()).hexdigest(),
                      pi_y_sha=hashlib.sha256(py.astype('<f8').tobytes()).hexdigest(),
                      v_y=[int(v) for v in vy], n=int(len(vy)))
        else:
            o1 = None
        games.append(dict(evaluator=list(ev) if isinstance(ev, tuple) else ev, sims=sims, game=game, randomised=bool(randomised), status=status,
                          reward=reward, plies=plies, pi_sha=pis, hist_pos12=states,
                          evals=model.calls + (model_b.calls if model_b else 0), o1=o1))
        print('games: %d ev=%s sims=%d status=%s plies=%d %.0fs' %
              (game, ev, sims, status, len(plies), time.time() - t0), file=sys.stderr)
    with open(path, 'w') as f:
        json.dump(dict(seed=SEED, games=games), f)


# --------------------------------------------------------------------------------------------
# G8: known answers of the stream and samplers (so the C and HIP copies can be pinned to spec.py)

def gen_rng():
    ka = dict(
        mix64=[[x, spec.mix64(x)] for x in (0, 1, 0xFFFFFFFFFFFFFFFF, 0x123456789ABCDEF)],
        rng=[[k, spec.rng(*k)] for k in ((0, 0, 0, 0, 0, 0), (SEED, 5, 7, 11, 3, 1), (1, 2 ** 40, 300, 799, 125, 6),
                                          (SEED, 4095, 6, 0, 0, 4))],
        choice=[[u, n, spec.choice_index(u, n)] for u, n in ((0, 1), (2 ** 64 - 1, 126), (0x8000000000000000, 3),
                                                            (0x6a70141414b9b3f5, 37))],
        det_log=[[f64bits(x), f64bits(spec.det_log(x))] for x in
                 (1.0, 2.0, 0.5, 1.5, 1e-16, 0.97, 3.3e5, spec.uniform_open(12345 << 12), 1.4142135623730951,
                  1.4142135623730954)],
        det_exp=[[f64bits(x), f64bits(spec.det_exp(x))] for x in
                 (0.0, 1.0, -1.0, -708.0, -708.5, 22.0, -33.33333, 0.34657, -0.34657, -700.25)],
        gamma=[[g, ply, e, f64bits(spec.gamma_small(SEED, g, ply, e, 0.03))]
               for g, ply, e in ((0, 6, 0), (0, 6, 1), (17, 9, 35), (4095, 100, 125), (3, 7, 2), (3, 7, 3))],
        dirichlet=[[g, ply, k, [f64bits(v) for v in spec.dirichlet(SEED, g, ply, k, 0.03)]]
                   for g, ply, k in ((0, 6, 10), (9, 7, 36), (77, 30, 1), (5, 8, 74))],
        pick_distinct=[[g, spec.pick_distinct(SEED, g, 49, 12)] for g in (0, 2, 5, 4094)],
        hash_eval=[], forward_eval=[],
    )
    for pos12, player in (([42, 35, 43, 28, 36, 44, 6, 13, 5, 20, 12, 4], 1),
                          ([42, 35, 43, 28, 36, 44, 6, 13, 5, 20, 12, 4], 2),
                          ([10, 13, 43, 20, 0, 8, 27, 1, 42, 31, 23, 4], 2)):
        p, v = spec.hash_eval(pos12, player)
        ka['hash_eval'].append([pos12, player, spec.state_key(pos12, player),
                                [f64bits(x) for x in p[:8]] + [f64bits(p[293])], f64bits(v)])
        p, v = spec.forward_eval(pos12, player)
        ka['forward_eval'].append([pos12, player, [f64bits(x) for x in p[:8]] + [f64bits(p[293])], f64bits(v)])
    # sample_index on a few pi-like vectors
    cases = []
    for j, (g, ply) in enumerate(((1, 6), (2, 7), (3, 20))):
        vec = [0.0] * 294
        for i in range(5 + 7 * j):
            vec[(i * 37 + j) % 294] = float((i * i + 1) % 11 + 1)
        s = sum(vec)
        vec = [v / s for v in vec]
        u = spec.rng(SEED, g, ply, 0, 0, spec.P_SAMPLE)
        cases.append([g, ply, [f64bits(v) for v in vec], spec.sample_index(u, vec)])
    ka['sample_index'] = cases
    with open(os.path.join(OUT, 'rng.json'), 'w') as f:
        json.dump(dict(seed=SEED, known=ka), f)


if __name__ == '__main__':
    what = sys.argv[1:] or ['rng', 'rules', 'tree', 'games']
    os.makedirs(OUT, exist_ok=True)
    if 'rng' in what:
        gen_rng()
    if 'rules' in what:
        gen_rules()
    if 'tree' in what:
        gen_tree()
    if 'tree_wide' in what:
        gen_tree(wide_only=True)
    if 'games' in what:
        gen_games()
    if 'games_extra' in what:
        gen_games(incremental=True)
    if 'games_cpu' in what:
        gen_games(incremental=True, cpu_only=True)


def gen_arena():
    """next-3: Game.start (game.py:58-100) between two AiPlayers (player.py:133-166) with table
    evaluators on the substituted stream; keys: ply = total_moves, sim = MCTS iteration index."""
    from refenv import ref_game, ref_player
    # (evaluator of player 1, of player 2, sims, initial tree_tau, enforce_move_limit, game id)
    plan = [(spec.EVAL_FORWARD, spec.EVAL_FORWARD, 8, 0.01, False, 7000), (spec.EVAL_FORWARD, spec.EVAL_FORWARD, 8, 0.01, False, 7001),
            (spec.EVAL_FORWARD, spec.EVAL_FORWARD, 24, 0.01, True, 7002), (spec.EVAL_FORWARD, spec.EVAL_HASH, 16, 1, True, 7003),
            (spec.EVAL_UNIFORM, spec.EVAL_UNIFORM, 8, 1, True, 7004), (spec.EVAL_HASH, spec.EVAL_FORWARD, 24, 0.01, True, 7005),
            (spec.EVAL_FORWARD, spec.EVAL_FORWARD, 50, 0.01, False, 7006),
            # the default simulation count, tree_tau = 1 until total_moves > 16 (player.py:152-155), mixed evaluators
            (spec.EVAL_FORWARD, spec.EVAL_FORWARD, 175, 1, True, 7007), (spec.EVAL_FORWARD, spec.EVAL_HASH, 100, 0.01, True, 7008),
            (spec.EVAL_HASH, spec.EVAL_HASH, 50, 1, True, 7009)]
    games = []
    orig_decide = ref_player.AiPlayer.decide_move
    t0 = time.time()
    for ev1, ev2, sims, tau, enforce, game in plan:
        refenv.set_sims(sims)
        ctx.seed, ctx.game = SEED, game
        m1, m2 = refenv.TableModel(ev1), refenv.TableModel(ev2)
        moves = []

        def decide(self, board, verbose=False, total_moves=None):
            ctx.ply = total_moves
            frm, to = orig_decide(self, board, verbose, total_moves)
            moves.append([board.checkers_id[self.player_num][frm], to[0] * 7 + to[1]])
            return frm, to
        ref_player.AiPlayer.decide_move = decide
        try:
            with quiet():
                gm = ref_game.Game(p1_type='a', p2_type='a', verbose=False, model1=m1, model2=m2, tree_tau=tau)
                winner = gm.start(enforce_move_limit=enforce)
        finally:
            ref_player.AiPlayer.decide_move = orig_decide
        games.append(dict(ev1=ev1, ev2=ev2, sims=sims, tau=tau, enforce=enforce, game=game, winner=winner,
                          moves=moves, evals=m1.calls + m2.calls))
        print('arena: %d ev=%d/%d sims=%d winner=%s moves=%d %.0fs' % (game, ev1, ev2, sims, winner, len(moves), time.time() - t0),
              file=sys.stderr)
    with open(os.path.join(OUT, 'arena.json'), 'w') as f:
        json.dump(dict(seed=SEED, games=games), f)


def gen_augment():
    """next-1: utils.augment_train_data (utils.py:77-97) on a few reference samples"""
    rules = np.load(os.path.join(OUT, 'rules.npz'))
    bx = [rules['planes'][i].reshape(7, 7, 7).astype(np.float64) for i in (3, 40, 777, 1500, 2999)]
    rng = np.random.RandomState(5)
    py = [rng.rand(294) for _ in bx]
    vy = [1, -1, 1, -1, 1]
    obx, opy, ovy = ref_utils.augment_train_data(list(bx), list(py), list(vy))
    np.savez_compressed(os.path.join(OUT, 'augment.npz'), board_x=np.array(bx), pi_y=np.array(py), v_y=np.array(vy),
                        out_board_x=np.array(obx), out_pi_y=np.array(opy), out_v_y=np.array(ovy))


if __name__ == '__main__' and 'augment' in sys.argv[1:]:
    gen_augment()
if __name__ == '__main__' and 'arena' in sys.argv[1:]:
    gen_arena()


def gen_greedy():
    """next-4: GreedyPlayer.decide_move (player.py:67-129), GreedyDataGenerator.generate_play
    (data_generators.py:25-80) and Game.start with greedy players (ai_vs_greedy.py:26-59, greedy_vs_greedy.py)"""
    from refenv import ref_game, ref_player, ref_datagen, pos12_of, last_moves_of
    t0 = time.time()
    # (1) the policy on positions of seeded random play, both sides: filtered best moves, in order
    policy = []
    for g in range(260):
        root = board_after_random_plies(9000 + g, 4 + (g * 7) % 40, randomised=(g % 5 == 4))
        b = root.state
        if b.check_win():
            continue
        for pl in (1, 2):
            best = ref_player.GreedyPlayer(player_num=pl).decide_move(b, training=True)
            mv = []
            for st, en in best:
                frm = ref_datagen.board_utils.human_coord_to_np_index(st)
                to = ref_datagen.board_utils.human_coord_to_np_index(en)
                mv.append([b.checkers_id[pl][frm], to[0] * 7 + to[1]])
            policy.append(dict(pos12=pos12_of(b), player=pl, best=mv))
    # (2) generator games.  limit = the ply form of STUCK_TIME_LIMIT (refenv.FakeClock)
    LIMIT = 200
    games = []
    orig_place = ref_board.Board.place
    plan = [(9500 + i, False, False) for i in range(8)] + [(9600 + i, False, True) for i in range(6)] + \
           [(9700 + i, True, False) for i in range(6)] + list(GREEDY_EXTRA)
    for game, randomised, random_start in plan:
        ctx.seed, ctx.game, ctx.ply, ctx.draw = SEED, game, 0, 0
        moves = []

        def place(self, player, frm, to):
            moves.append([self.checkers_id[player][frm], to[0] * 7 + to[1]])
            out = orig_place(self, player, frm, to)
            ctx.ply += 1
            ctx.draw = 0
            return out
        gen = ref_datagen.GreedyDataGenerator(randomised=randomised, random_start=random_start)   # Board() draws P_INIT
        start12 = pos12_of(gen.board)
        ref_board.Board.place = place
        clock = refenv.FakeClock(LIMIT)
        ref_datagen.datetime = clock
        try:
            with quiet():
                hist, reward = gen.generate_play()
        finally:
            ref_board.Board.place = orig_place
        rows = []
        for bd, pi in hist:
            nz = [int(i) for i in np.nonzero(pi)[0]]
            rows.append(dict(pos12=pos12_of(bd), last=last_moves_of(bd), idx=nz, p=f64bits(pi[nz[0]])))
        games.append(dict(game=game, randomised=randomised, random_start=random_start, start12=start12, moves=moves,
                          reward=int(reward), rows=rows, stuck=bool(clock.calls > LIMIT + 1)))
        print('greedy gen: game %d rand=%s rs=%s plies=%d rows=%d reward=%d stuck=%s %.0fs' %
              (game, randomised, random_start, len(moves), len(rows), reward, clock.calls > LIMIT + 1, time.time() - t0), file=sys.stderr)
    # (3) Game.start with greedy players: ai (table evaluator) vs greedy both ways, greedy vs greedy
    arena = []
    orig_ai = ref_player.AiPlayer.decide_move
    orig_gr = ref_player.GreedyPlayer.decide_move
    for game, p1, p2, ev, sims, enforce in [(9800, 'a', 'g', spec.EVAL_FORWARD, 8, False), (9801, 'g', 'a', spec.EVAL_FORWARD, 8, False),
                                            (9802, 'a', 'g', spec.EVAL_HASH, 16, True), (9803, 'g', 'a', spec.EVAL_UNIFORM, 8, True),
                                            (9804, 'g', 'g', 0, 0, False), (9805, 'g', 'g', 0, 0, True), (9806, 'g', 'g', 0, 0, False)]:
        refenv.set_sims(max(sims, 1))
        ctx.seed, ctx.game = SEED, game
        m = refenv.TableModel(ev)
        moves = []

        def wrap(orig):
            def decide(self, board, verbose=False, training=False, total_moves=None):
                ctx.ply = total_moves
                frm, to = orig(self, board, verbose=verbose, total_moves=total_moves)
                moves.append([board.checkers_id[self.player_num][frm], to[0] * 7 + to[1]])
                return frm, to
            return decide
        ref_player.AiPlayer.decide_move = wrap(orig_ai)
        ref_player.GreedyPlayer.decide_move = wrap(orig_gr)
        try:
            with quiet():
                gm = ref_game.Game(p1_type=p1, p2_type=p2, verbose=False, model1=m, model2=m)
                winner = gm.start(enforce_move_limit=enforce)
        finally:
            ref_player.AiPlayer.decide_move = orig_ai
            ref_player.GreedyPlayer.decide_move = orig_gr
        arena.append(dict(game=game, p1=p1, p2=p2, ev=ev, sims=sims, enforce=enforce, winner=winner, moves=moves, evals=m.calls))
        print('greedy arena: %d %s/%s winner=%s moves=%d %.0fs' % (game, p1, p2, winner, len(moves), time.time() - t0), file=sys.stderr)
    with open(os.path.join(OUT, 'greedy.json'), 'w') as f:
        json.dump(dict(seed=SEED, limit=LIMIT, policy=policy, games=games, arena=arena), f)


def gen_greedy_stochastic():
    """GreedyPlayer(stochastic=True) (player.py:77-97; what game.py:105-113 plays against the deterministic one): single decisions on
    positions of seeded random play, and whole Game.start games with such seats -> tests/golden/greedy_stochastic.json"""
    from refenv import ref_game, ref_player, ref_datagen, pos12_of
    t0 = time.time()
    to_idx = ref_datagen.board_utils.human_coord_to_np_index
    decisions = []
    for g in range(240):
        root = board_after_random_plies(12000 + g, 2 + (g * 5) % 44, randomised=(g % 4 == 3))
        b = root.state
        if b.check_win():
            continue
        for pl in (1, 2):
            ctx.seed, ctx.game, ctx.ply = SEED, 12000 + g, 3 * g + pl
            with quiet():
                frm, to = ref_player.GreedyPlayer(player_num=pl, stochastic=True).decide_move(b)
            decisions.append(dict(pos12=pos12_of(b), player=pl, game=12000 + g, ply=3 * g + pl,
                                  move=[b.checkers_id[pl][tuple(frm)], int(to[0]) * 7 + int(to[1])]))
    arena = []
    orig_ai = ref_player.AiPlayer.decide_move
    orig_gr = ref_player.GreedyPlayer.decide_move
    plan = [(12500, 's', 'g', 0, 0, False), (12501, 'g', 's', 0, 0, False), (12502, 's', 's', 0, 0, False), (12503, 's', 's', 0, 0, True),
            (12504, 'a', 's', spec.EVAL_FORWARD, 8, False), (12505, 's', 'a', spec.EVAL_HASH, 16, True), (12506, 's', 'g', 0, 0, True),
            (12507, 'g', 's', 0, 0, True)]
    for game, p1, p2, ev, sims, enforce in plan:
        refenv.set_sims(max(sims, 1))
        ctx.seed, ctx.game = SEED, game
        m = refenv.TableModel(ev)
        moves = []

        def wrap(orig):
            def decide(self, board, verbose=False, training=False, total_moves=None):
                ctx.ply = total_moves
                frm, to = orig(self, board, verbose=verbose, total_moves=total_moves)
                moves.append([board.checkers_id[self.player_num][tuple(frm)], int(to[0]) * 7 + int(to[1])])
                return frm, to
            return decide
        ref_player.AiPlayer.decide_move = wrap(orig_ai)
        ref_player.GreedyPlayer.decide_move = wrap(orig_gr)
        try:
            with quiet():
                kind = {'a': 'a', 'g': 'g', 's': 'g'}
                gm = ref_game.Game(p1_type=kind[p1], p2_type=kind[p2], verbose=False, model1=m, model2=m)
                if p1 == 's':
                    gm.player_one = ref_player.GreedyPlayer(player_num=1, stochastic=True)       # as game.py:111 does for player two
                if p2 == 's':
                    gm.player_two = ref_player.GreedyPlayer(player_num=2, stochastic=True)
                gm.cur_player, gm.next_player = gm.player_one, gm.player_two                      # (bound at construction, game.py:31-32)
                winner = gm.start(enforce_move_limit=enforce)
        finally:
            ref_player.AiPlayer.decide_move = orig_ai
            ref_player.GreedyPlayer.decide_move = orig_gr
        arena.append(dict(game=game, p1=p1, p2=p2, ev=ev, sims=sims, enforce=enforce, winner=winner, moves=moves, evals=m.calls))
        print('stochastic greedy arena: %d %s/%s winner=%s moves=%d %.0fs' % (game, p1, p2, winner, len(moves), time.time() - t0), file=sys.stderr)
    with open(os.path.join(OUT, 'greedy_stochastic.json'), 'w') as f:
        json.dump(dict(seed=SEED, decisions=decisions, arena=arena), f)


if __name__ == '__main__' and 'greedy_stochastic' in sys.argv[1:]:
    gen_greedy_stochastic()

# (game, randomised, random_start) found with the oracle (tests/oracle_ffi.py) whose generator game gets stuck:
# none in 6000 normal starts, 2 in 3000 randomised boards
GREEDY_EXTRA = [(20178, True, False), (22641, True, False)]

if __name__ == '__main__' and 'greedy' in sys.argv[1:]:
    gen_greedy()


# ---- round 4: the statistics game.py records (game.py:103-119) -----------------------------------------------------------
# 10 000 Game.start games "greedy against GreedyPlayer(stochastic=True)": Counter({1: 5172, 2: 4675, None: 153}) /
# Counter({1: 5233, 2: 4594, None: 173}).  The reference's CURRENT stochastic policy (player.py:77-97: a forward move drawn with
# probability proportional to its distance) cannot produce those figures -- it loses nine games in ten to the deterministic
# player -- while two deterministic players do.  This generator plays the imported reference itself, every seating, on the
# substituted draws: per game (winner, number of moves), so that the oracle and the GPU must reproduce every game AND the
# frequencies, and the record in game.py can be compared with what the reference's own code does today.
STATS_FIRST, STATS_N = 30000, 10000


def _stats_job(job):
    seating, first, n = job
    from refenv import ref_game, ref_player
    orig_gr = ref_player.GreedyPlayer.decide_move
    out = []
    for game in range(first, first + n):
        ctx.seed, ctx.game = SEED, game
        count = [0]

        def decide(self, board, verbose=False, training=False, total_moves=None):
            ctx.ply = total_moves
            count[0] += 1
            return orig_gr(self, board, verbose=verbose, total_moves=total_moves)
        ref_player.GreedyPlayer.decide_move = decide
        try:
            with quiet():
                gm = ref_game.Game(p1_type='g', p2_type='g', verbose=False)
                if seating[0] == 's':
                    gm.player_one = ref_player.GreedyPlayer(player_num=1, stochastic=True)
                if seating[1] == 's':
                    gm.player_two = ref_player.GreedyPlayer(player_num=2, stochastic=True)       # game.py:111
                gm.cur_player, gm.next_player = gm.player_one, gm.player_two
                winner = gm.start()
        finally:
            ref_player.GreedyPlayer.decide_move = orig_gr
        out.append([winner or 0, count[0]])
    return out


def gen_greedy_stats():
    from multiprocessing import Pool
    t0 = time.time()
    doc = dict(seed=SEED, first_game=STATS_FIRST, n=STATS_N, seatings={},
               record_in_game_py=[{'1': 5172, '2': 4675, 'None': 153}, {'1': 5233, '2': 4594, 'None': 173}])
    chunk = 125
    with Pool(8) as pool:
        for seating in ('gg', 'gs', 'sg'):
            parts = pool.map(_stats_job, [(seating, STATS_FIRST + i, chunk) for i in range(0, STATS_N, chunk)], chunksize=1)
            rows = [r for p in parts for r in p]
            c = {w: sum(1 for r in rows if r[0] == w) for w in (1, 2, 0)}
            doc['seatings'][seating] = dict(winner=[r[0] for r in rows], moves=[r[1] for r in rows], counts={'1': c[1], '2': c[2], 'None': c[0]})
            print('greedy stats %s: %s  %.0fs' % (seating, doc['seatings'][seating]['counts'], time.time() - t0), file=sys.stderr)
    with open(os.path.join(OUT, 'greedy_stats.json'), 'w') as f:
        json.dump(doc, f, separators=(',', ':'))


if __name__ == '__main__' and 'greedy_stats' in sys.argv[1:]:
    gen_greedy_stats()
