"""GPU parity of the batched rules kernels, through the C ABI (libccsp.so), against the golden
vectors produced by the reference and against the CPU oracle."""
import hashlib

import numpy as np
import pytest

import oracle_ffi as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    import torch
    from chinesecheckersagent_amd import _lib, rules
    _lib.require_gpu()
    assert torch.cuda.is_available()
    return rules


def _moves_list(moves, count, i):
    return moves[i, :count[i]]


def test_explicit_golden_records(gpu, golden_dir):
    from chinesecheckersagent_amd import _lib
    g = np.load(golden_dir + '/rules.npz')
    n = len(g['pos12'])
    states = _lib.pack_states(g['pos12'], g['last'])
    sd = gpu.to_device_states(states)
    moves, count, masks = gpu.movegen(sd, g['player'])
    moves, count, masks = moves.cpu().numpy(), count.cpu().numpy(), masks.cpu().numpy().view(np.uint64)
    off = 0
    for i in range(n):
        c = int(g['move_count'][i])
        want = g['moves'][off:off + c]
        off += c
        assert count[i] == c and (moves[i, :c] == want).all(), 'record %d: legal moves / order differ' % i
        for cid in range(6):
            m = 0
            for d in want[want[:, 0] == cid][:, 1]:
                m |= 1 << int(d)
            assert int(masks[i, cid]) == m
    planes = gpu.encode(sd, g['player']).cpu().numpy().reshape(n, 343)
    assert (planes == g['planes'].astype(np.float32)).all()
    nxt, winner, progress = gpu.step(sd, g['player'], g['chosen'])
    nxt = nxt.cpu().numpy().view(_lib.STATE_DTYPE).reshape(n)
    assert (nxt['pos'].reshape(n, 12) == g['npos12']).all()
    assert (nxt['last'] == g['nlast']).all()
    assert (winner.cpu().numpy() == g['winner']).all()
    want_next = _lib.pack_states(g['npos12'], g['nlast'])
    assert (nxt['occ'] == want_next['occ']).all()
    # progress is reported for the NEW state
    pr = progress.cpu().numpy()
    for i in range(0, n, 37):
        assert pr[i, 0] == orc.progress(g['npos12'][i], 1) and pr[i, 1] == orc.progress(g['npos12'][i], 2)


def _all_trajectory_states(rules_npz):
    """the 102 000 states of the fixture's trajectories, rebuilt with the oracle (pinned to the
    reference by tests/test_oracle_rules.py)"""
    seed = int(rules_npz['seed'])
    P, L, PL, CH = [], [], [], []
    for g in range(int(rules_npz['n_games'])):
        kind = g % 4
        pos12 = orc.randomised_pos12(seed, g) if kind == 2 else (
            orc.near_win_pos12(seed, g, 1 + (g // 4) % 2) if kind == 3 else orc.initial_pos12())
        last, player = orc.NO_LAST.copy(), 1
        for ply in range(int(rules_npz['max_plies'])):
            if orc.check_win(pos12):
                break
            cid, dest = orc.random_move(pos12, player, seed, g, ply)
            P.append(pos12); L.append(last); PL.append(player); CH.append((cid, dest))
            pos12, last, _ = orc.step(pos12, last, player, cid, dest)
            player = 3 - player
    return np.array(P, np.uint8), np.array(L, np.uint8), np.array(PL, np.uint8), np.array(CH, np.uint8)


def test_all_records_against_reference_digests(gpu, golden_dir):
    """every one of the fixture's 102 000 positions through the GPU kernels; the results are hashed in
    the fixture's canonical order and compared with the SHA-256 the REFERENCE produced."""
    from chinesecheckersagent_amd import _lib
    g = np.load(golden_dir + '/rules.npz')
    pos12, last, player, chosen = _all_trajectory_states(g)
    n = len(pos12)
    assert n == int(g['n_records'])
    sd = gpu.to_device_states(_lib.pack_states(pos12, last))
    moves, count, _ = gpu.movegen(sd, player, want_masks=False)
    moves, count = moves.cpu().numpy(), count.cpu().numpy()
    planes = gpu.encode(sd, player).cpu().numpy().reshape(n, 343).astype(np.uint8)
    nxt, winner, _ = gpu.step(sd, player, chosen)
    nxt = nxt.cpu().numpy().view(_lib.STATE_DTYPE).reshape(n)
    winner = winner.cpu().numpy()
    # progress of the CURRENT state = what step reports for the previous ply; take it from the oracle-free
    # identity progress(p) = popcount(occ[p] & target)
    T1 = sum(1 << c for c in (4, 5, 6, 12, 13, 20))
    T2 = sum(1 << c for c in (28, 35, 36, 42, 43, 44))
    cur = _lib.pack_states(pos12, last)
    h_moves, h_step, h_planes = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()
    for i in range(n):
        c = int(count[i])
        h_moves.update(bytes(pos12[i]) + bytes([int(player[i]), c]) + moves[i, :c].tobytes())
        h_planes.update(bytes(pos12[i]) + bytes([int(player[i])]) + bytes(last[i]) + planes[i].tobytes())
        p1 = bin(int(cur['occ'][i][0]) & T1).count('1')
        p2 = bin(int(cur['occ'][i][1]) & T2).count('1')
        h_step.update(bytes(pos12[i]) + bytes([int(player[i]), int(chosen[i][0]), int(chosen[i][1])]) +
                      nxt['pos'][i].tobytes() + nxt['last'][i].tobytes() + bytes([int(winner[i]), p1, p2]))
    assert h_moves.digest() == g['sha_moves'].tobytes(), 'legal-move lists differ from the reference'
    assert h_planes.digest() == g['sha_planes'].tobytes(), 'planes differ from the reference'
    assert h_step.digest() == g['sha_step'].tobytes(), 'next states / winners differ from the reference'


def test_hop_search_overflow_path(gpu, golden_dir):
    """the per-lane hop-search stacks hold 20 entries and real positions need at most 8, so the redo path (a search that
    does not fit is run again on the wave's big stack) never runs on real data: force it with stacks of 6 / 7 / 9 entries
    and require the same move lists and destination masks as with the full stacks, on 102 000 reference positions and
    200 000 random ones (ragged size)"""
    import torch
    from chinesecheckersagent_amd import _lib
    L = _lib.lib()
    g = np.load(golden_dir + '/rules.npz')
    pos12, last, player, chosen = _all_trajectory_states(g)
    rng = np.random.RandomState(3)
    cells = np.argsort(rng.rand(200003, 49), axis=1)[:, :12].astype(np.uint8)
    pos12 = np.concatenate([pos12, cells])
    player = np.concatenate([player, (1 + (np.arange(len(cells)) & 1)).astype(np.uint8)])
    sd = gpu.to_device_states(_lib.pack_states(pos12))
    assert L.ccsp_debug_movegen_stack_cap(0) == 20
    want = [t.clone() for t in gpu.movegen(sd, player)]
    want_g = [t.clone() for t in gpu.greedy_best(sd, player)]
    try:
        for cap in (6, 7, 9):
            assert L.ccsp_debug_movegen_stack_cap(cap) == cap
            got = gpu.movegen(sd, player)
            for a, b in zip(got, want):
                assert torch.equal(a, b), 'stack cap %d changes the result' % cap
            for a, b in zip(gpu.greedy_best(sd, player), want_g):
                assert torch.equal(a, b)
    finally:
        assert L.ccsp_debug_movegen_stack_cap(0) == 20


def test_wins(gpu, golden_dir):
    from chinesecheckersagent_amd import _lib
    z = np.load(golden_dir + '/wins.npz')
    rows = z['moves']
    sd = gpu.to_device_states(_lib.pack_states(rows[:, :12]))
    _, winner, _ = gpu.step(sd, rows[:, 12].copy(), rows[:, 13:15].copy())
    assert (winner.cpu().numpy() == rows[:, 15]).all()
    assert (rows[:, 15] != 0).sum() >= 20


def test_edge_cases(gpu):
    import torch
    from chinesecheckersagent_amd import _lib
    # empty batch
    e = torch.zeros((0, 32), dtype=torch.uint8, device='cuda')
    p = torch.zeros(0, dtype=torch.uint8, device='cuda')
    m, c, k = gpu.movegen(e, p)
    assert m.shape[0] == 0 and c.shape[0] == 0
    assert gpu.encode(e, p).shape[0] == 0
    # ragged sizes around the workgroup tiles (32 states per movegen block, 64 per encode block)
    base = _lib.pack_states(np.repeat(orc.initial_pos12()[None], 131, 0))
    for n in (1, 31, 33, 63, 65, 131):
        sd = gpu.to_device_states(base[:n])
        pl = np.ones(n, dtype=np.uint8)
        mv, cnt, _ = gpu.movegen(sd, pl)
        assert (cnt.cpu().numpy() == 10).all()
        want = orc.movegen(orc.initial_pos12(), 1)
        assert (mv.cpu().numpy()[:, :10] == want[None]).all()
        pln = gpu.encode(sd, pl).cpu().numpy().reshape(n, 343)
        assert (pln == orc.planes(orc.initial_pos12(), orc.NO_LAST, 1)[None].astype(np.float32)).all()


def test_large_batch_properties(gpu):
    """2^20 states (the movegen micro-benchmark's size): invariants that need no oracle, plus a
    1-in-257 sample checked against the oracle."""
    import torch
    from chinesecheckersagent_amd import _lib
    n = 1 << 20
    rng = np.random.RandomState(7)
    # random legal-looking positions: 12 distinct cells each
    cells = np.argsort(rng.rand(n, 49), axis=1)[:, :12].astype(np.uint8)
    player = (1 + (np.arange(n) & 1)).astype(np.uint8)
    states = _lib.pack_states(cells)
    sd = gpu.to_device_states(states)
    moves, count, masks = gpu.movegen(sd, player)
    torch.cuda.synchronize()
    count_h = count.cpu().numpy()
    masks_h = masks.cpu().numpy().view(np.uint64)
    occ = states['occ'][:, 0] | states['occ'][:, 1]
    assert ((masks_h & occ[:, None]) == 0).all(), 'a destination is occupied'
    tot = np.zeros(n, dtype=np.int64)
    mm = masks_h.copy()
    for _ in range(49):
        tot += (mm & np.uint64(1)).sum(axis=1).astype(np.int64)
        mm >>= np.uint64(1)
    assert (tot == count_h).all(), 'count != number of mask bits'
    moves_h = moves.cpu().numpy()
    for i in range(0, n, 257 * 16):
        want = orc.movegen(cells[i], int(player[i]))
        assert count_h[i] == len(want) and (moves_h[i, :len(want)] == want).all()
    # step: idempotence of a move and its reverse restores occupancy
    mv = moves_h[:, 0, :].copy()
    has = count_h > 0
    nxt, _, _ = gpu.step(sd, player, mv)
    nxt_h = nxt.cpu().numpy().view(_lib.STATE_DTYPE).reshape(n)
    frm = np.where(player == 1, 0, 1)
    origin = states['pos'][np.arange(n), frm, mv[:, 0]]
    back = np.stack([mv[:, 0], origin], axis=1).astype(np.uint8)
    again, _, _ = gpu.step(nxt, player, back)
    again_h = again.cpu().numpy().view(_lib.STATE_DTYPE).reshape(n)
    assert (again_h['occ'][has] == states['occ'][has]).all() and (again_h['pos'][has] == states['pos'][has]).all()
    assert (nxt_h['last'][has, 0] == origin[has]).all() and (nxt_h['last'][has, 1] == mv[has, 1]).all()


def test_packed_move_lists_equal_the_rows(gpu, golden_dir):
    """ccsp_movegen_packed (the lists of each chunk of 32 positions back to back: whole sectors instead of sparse 252-byte rows)
    holds exactly the lists of ccsp_movegen, at rules.list_starts(count); n not a multiple of 32 and n < 32 included"""
    import torch
    from chinesecheckersagent_amd import _lib, rules
    g = np.load(golden_dir + '/rules.npz')
    pos, player = g['pos12'], g['player']
    for n in (len(pos), 3200, 33, 31, 1):
        sd = rules.to_device_states(_lib.pack_states(pos[:n]))
        pl = torch.from_numpy(player[:n]).cuda()
        mv, cnt, masks = rules.movegen(sd, pl)
        pm, pcnt, pmasks = rules.movegen_packed(sd, pl)
        assert torch.equal(cnt, pcnt) and torch.equal(masks, pmasks)
        starts = rules.list_starts(pcnt)
        mv, pm, cnt, starts = mv.cpu().numpy(), pm.cpu().numpy(), cnt.cpu().numpy(), starts.cpu().numpy()
        assert np.array_equal(starts, rules.list_starts(cnt))
        for i in range(n):
            assert np.array_equal(pm[starts[i]:starts[i] + cnt[i]], mv[i, :cnt[i]]), i
