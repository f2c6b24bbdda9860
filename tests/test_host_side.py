"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/ccsp.h
declares, constants agree between Python and the HIP side, the Board-like view and the reference-shaped
helpers reproduce the oracle, and the multi-rank summary all-reduce works (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_ffi as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from chinesecheckersagent_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, 'include', 'ccsp.h')).read()
    declared = set(re.findall(r'\b(ccsp_[a-z_0-9]+)\s*\(', hdr))
    declared -= {'ccsp_ctx'}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r' T (ccsp_[a-z_0-9]+)', out))
    assert declared <= exported
    assert b'gfx950' in L.ccsp_version()


def test_no_cpu_fallback_without_a_gpu():
    """on a box without a GPU the product must fail loudly, not fall back to anything"""
    from chinesecheckersagent_amd import _lib, engine
    if _lib.lib().ccsp_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_lib.CcspError):
        engine.SelfPlayEngine(n_slots=1, sims=8, seed=1)
    with pytest.raises(_lib.CcspError):
        from chinesecheckersagent_amd import rules
        rules.movegen(np.zeros((1, 32), dtype=np.uint8), np.ones(1, dtype=np.uint8))
    # and nothing in the package imports, loads or links the oracle
    bad = re.compile(r'oracle_ffi|ccsp_oracle|libccsp_oracle|import\s+net_oracle|from\s+oracle|import\s+oracle|orc_[a-z_]+\(')
    pkg = os.path.join(ROOT, 'chinesecheckersagent_amd')
    for base, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.h', '.hip', '.cpp')):
                src = open(os.path.join(base, fn)).read()
                assert not bad.search(src), fn


def test_constants_agree_with_hip_side():
    from chinesecheckersagent_amd import config
    h = open(os.path.join(ROOT, 'chinesecheckersagent_amd', 'csrc', 'ccsp_rules.h')).read()

    def c(name):
        return float(re.search(r'#define %s\s+([0-9.e+-]+)' % name, h).group(1))
    assert c('CCSP_TOTAL_HIST_MOVES') == config.TOTAL_HIST_MOVES
    assert c('CCSP_UNIQUE_DEST_LIMIT') == config.UNIQUE_DEST_LIMIT
    assert c('CCSP_DIRICHLET_ALPHA') == config.DIRICHLET_ALPHA
    assert c('CCSP_DIR_NOISE_FACTOR') == config.DIR_NOISE_FACTOR
    assert c('CCSP_PROGRESS_MOVE_LIMIT') == config.PROGRESS_MOVE_LIMIT
    assert c('CCSP_C_PUCT') == config.C_PUCT
    assert c('CCSP_EPSILON') == config.EPSILON
    assert c('CCSP_TOTAL_MOVES_TILL_TAU0') == config.TOTAL_MOVES_TILL_TAU0
    assert c('CCSP_INITIAL_RANDOM_MOVES') == config.INITIAL_RANDOM_MOVES
    assert config.NUM_ACTIONS == 294 and config.MAX_MOVES == 126 and config.INPUT_DIM == (7, 7, 7)


def test_pack_states_and_board_view(golden_dir):
    from chinesecheckersagent_amd import _lib, utils
    from chinesecheckersagent_amd.board import BoardView
    g = np.load(golden_dir + '/rules.npz')
    n = 400
    st = _lib.pack_states(g['pos12'][:n], g['last'][:n])
    for i in range(n):
        b = BoardView(st[i])
        pl = int(g['player'][i])
        # reference-shaped to_model_input on the view == the reference's planes
        assert np.array_equal(utils.to_model_input(b, pl).reshape(343), g['planes'][i].astype(np.float64))
        assert b.check_win() == orc.check_win(g['pos12'][i])
        assert b.player_progress(1) == int(g['progress'][i][0]) and b.player_progress(2) == int(g['progress'][i][1])
        assert sorted(b.checkers_id[1].values()) == list(range(6))
    # the record after a move: plane 1 of the view is the position before it (board.py:243)
    j = 5
    nb = BoardView(_lib.pack_states(g['npos12'][j:j + 1], g['nlast'][j:j + 1])[0])
    assert np.array_equal(nb.board[:, :, 0], g['nboard'][j][:, :, 0]) and np.array_equal(nb.board[:, :, 1], g['nboard'][j][:, :, 1])
    with pytest.raises(_lib.CcspError):
        _lib.pack_states([[99] * 12])
    for cid, r, c, idx in np.load(golden_dir + '/codec.npy'):
        assert utils.encode_checker_index(int(cid), (int(r), int(c))) == idx
        assert utils.decode_checker_index(int(idx)) == (cid, (r, c))


def _rank_main(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from chinesecheckersagent_amd import summary
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ids = summary.shard_game_ids(10, rank, world)
    counters = {'expansions': 100 * (rank + 1), 'games_won': len(ids), 'errors': 0}
    hist = np.zeros(294, dtype=np.uint64)
    hist[rank] = 7
    hist[293] = 2 ** 40
    tot, h = summary.allreduce_summary(counters, hist, dist, device='cpu')
    q.put((rank, ids, tot, [int(h[0]), int(h[1]), int(h[293])]))
    dist.destroy_process_group()


def test_summary_allreduce_gloo_world2():
    """games shard by id, the only collective is the summary all-reduce(sum)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    ps = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7, 9]          # every id on exactly one rank
    for _, _, tot, h in res:
        assert tot['expansions'] == 300 and tot['games_won'] == 10 and h == [7, 7, 2 ** 41]


def test_summary_allreduce_gloo_world8():
    """BASELINE config 4's world size: EIGHT ranks (on the CPU over gloo: a 1-GPU box admits six GPU processes, the N = 8 hardware run is
    the driver's) -- id sharding j mod 8 covers every id once, the all-reduced counters and histogram are the sums over the eight"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 30600 + os.getpid() % 1000
    ps = [ctx.Process(target=_rank_main, args=(r, 8, port, q)) for r in range(8)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=300) for _ in ps)
    [p.join(60) for p in ps]
    assert sorted(i for r in res for i in r[1]) == list(range(10)) and all(r[1] == list(range(r[0], 10, 8)) for r in res)
    for _, _, tot, h in res:
        assert tot['expansions'] == 100 * 36 and tot['games_won'] == 10 and h == [7, 7, 8 * 2 ** 40]


def test_augment_train_data_matches_reference(golden_dir):
    """utils.augment_train_data incl. its un-mirrored pi (fixture made by the reference)"""
    from chinesecheckersagent_amd import utils
    g = np.load(golden_dir + '/augment.npz')
    bx, py, vy = list(g['board_x']), list(g['pi_y']), [int(v) for v in g['v_y']]
    obx, opy, ovy = utils.augment_train_data(bx, py, vy)
    assert obx is bx and len(obx) == 10
    assert np.array_equal(np.array(obx), g['out_board_x']) and np.array_equal(np.array(opy), g['out_pi_y'])
    assert [int(v) for v in ovy] == [int(v) for v in g['out_v_y']]


def test_vectorised_plane_encoder_and_log_conversion(golden_dir):
    """utils.states_to_model_input against the reference's own planes (3339 records of rules.npz) and
    utils.log_to_train_data against convert_to_train_data over BoardView objects, incl. the randomised-board quirk"""
    from chinesecheckersagent_amd import _lib, utils
    from chinesecheckersagent_amd.board import BoardView
    g = np.load(golden_dir + '/rules.npz')
    states = _lib.pack_states(g['pos12'], g['last'])
    x = utils.states_to_model_input(states, g['player'])
    assert x.shape == (len(states), 7, 7, 7) and x.dtype == np.float64
    assert (x.reshape(len(states), 343) == g['planes'].astype(np.float64)).all()
    for i in range(0, len(states), 97):                      # and the per-object path agrees with itself
        assert (utils.to_model_input(BoardView(states[i]), int(g['player'][i])) == x[i]).all()
    # a synthetic log: 5 games (ids 10, 13, 16, ...), statuses won / discarded, rows shuffled
    rng = np.random.RandomState(4)
    n_games, first, stride = 5, 10, 3
    res = np.zeros(n_games, dtype=_lib.RESULT_DTYPE)
    res['status'] = [_lib.ST_WON_P1, _lib.ST_DISCARD_REPETITION, _lib.ST_WON_P2, _lib.ST_WON_P1, _lib.ST_DISCARD_NO_PROGRESS]
    res['reward'] = [1, 0, -1, 1, 0]
    rows = []
    for k in range(n_games):
        for j in range(4 + k):
            rows.append((first + k * stride, 6 + j, 1 + j % 2, rng.randint(len(states))))
    rows = [rows[i] for i in rng.permutation(len(rows))]
    meta = np.zeros(len(rows), dtype=_lib.META_DTYPE)
    meta['game'], meta['ply'], meta['player'] = [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows]
    st = states[[r[3] for r in rows]]
    pi = rng.rand(len(rows), 294)
    for randomised in (False, True):
        bx, py, vy = utils.log_to_train_data(st, meta, pi, res, first_game=first, game_stride=stride, randomised=randomised)
        games = []                                           # the object path: what BatchSelfPlay.collect() builds
        order = np.lexsort((meta['ply'], meta['game']))
        for k in range(n_games):
            if int(res['status'][k]) not in (_lib.ST_WON_P1, _lib.ST_WON_P2):
                continue
            mine = [r for r in order if meta['game'][r] == first + k * stride]
            if randomised:
                mine = mine[3:]
            games.append(([(BoardView(st[r]), pi[r]) for r in mine], int(res['reward'][k])))
        wx, wp, wv = utils.convert_to_train_data(games)
        assert len(wx) == len(bx) > 0
        assert (np.array(wx) == bx).all() and (np.array(wp) == py).all() and list(vy) == wv and vy.dtype == np.int64


def test_bench_launcher_starts_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` (how the driver starts it: no torchrun around it) must start its two rank processes
    itself; in this container they have no GPU, so each must stop at require_gpu() -- no CPU fallback -- and the
    launcher must hand the failure on as a non-zero exit code, printing no result line"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by the -m gpu launcher test')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--games', '8', '--sims', '4', '--steps', '1',
                        '--warmup', '0', '--no-extras'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ''
    assert 'no HIP device visible' in r.stderr and 'a rank process failed' in r.stderr


def test_usable_cores_and_cpu_baseline_workers():
    """bench.py's CPU-baseline leg: core count from affinity / cgroup quota, and the worker processes of both oracle forms"""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)
    rate, games, dt = bench._run_cpu_workers('c', 2, 0.5, 16, 2)
    assert rate > 0 and games >= 2
    rate, games, dt = bench._run_cpu_workers('py', 1, 0.2, 8, 1)
    assert rate > 0 and games >= 1
    # the net-inclusive forms: C oracle + the PyTorch CPU module, reference-shaped mirror + the NumPy float32 net
    rate, games, dt = bench._run_cpu_workers('c_net', 1, 0.3, 8, 1)
    assert rate > 0 and games >= 1
    rate, games, dt = bench._run_cpu_workers('py_net', 1, 0.3, 8, 1)
    assert rate > 0 and games >= 1


def test_game_store_streams_the_rows_log_to_train_data_gives_at_once():
    """selfplay.GameStore (the host side of a harvested run): rows arrive harvest by harvest from two strided parts, games end at
    different times, some discarded; the chunks take_finished() hands out -- whatever the harvest boundaries -- add up to exactly
    what utils.log_to_train_data makes of the whole log at once, and games() rebuilds every play_history"""
    from chinesecheckersagent_amd import _lib, selfplay as sp, utils
    rng = np.random.RandomState(5)
    n_games, first, stride = 37, 1000, 3
    length = rng.randint(1, 40, size=n_games)
    status = rng.choice([_lib.ST_WON_P1, _lib.ST_WON_P2, _lib.ST_DISCARD_REPETITION, _lib.ST_DISCARD_NO_PROGRESS], size=n_games)
    start = rng.randint(0, 30, size=n_games)                          # the "ply clock" at which a game starts
    rows = []                                                         # (time, game index, ply)
    for j in range(n_games):
        rows += [(start[j] + k, j, 6 + k) for k in range(length[j])]
    rows.sort()
    end_time = start + length
    n = len(rows)
    st = np.zeros(n, dtype=_lib.STATE_DTYPE)
    st['pos'] = np.argsort(rng.rand(n, 49), axis=1)[:, :12].astype(np.uint8).reshape(n, 2, 6)
    st['last'] = 255
    meta = np.zeros(n, dtype=_lib.META_DTYPE)
    meta['game'] = [first + j * stride for _, j, _ in rows]
    meta['ply'] = [p for _, _, p in rows]
    meta['player'] = 1 + (meta['ply'] % 2)
    pi = rng.rand(n, 294)
    t_of = np.array([t for t, _, _ in rows])
    full = np.zeros(n_games, dtype=_lib.RESULT_DTYPE)
    full['status'] = status
    full['reward'] = np.where(status == _lib.ST_WON_P1, 1, np.where(status == _lib.ST_WON_P2, -1, 0))
    want = utils.log_to_train_data(st, meta, pi, full, first_game=first, game_stride=stride, return_games=True)
    for every in (1, 4, 9, 100):
        store = sp.GameStore(n_games, first, stride, False, keep_records=True)
        got = []
        for t0 in range(0, int(end_time.max()) + every, every):
            sel = (t_of >= t0) & (t_of < t0 + every)
            parts = []
            for part in range(2):                                     # two parts: game indices part, part + 2, ... (ids strided accordingly)
                jj = np.arange(part, n_games, 2)
                res = np.zeros(len(jj), dtype=_lib.RESULT_DTYPE)
                res['status'] = 0xFF
                done = end_time[jj] <= t0 + every
                res[done] = full[jj[done]]
                mine = sel & (((meta['game'] - first) // stride) % 2 == part)
                parts.append((st[mine], meta[mine], pi[mine], res, first + part * stride, stride * 2))
            store.add(parts)
            c = store.take_finished()
            if c is not None:
                got.append(c)
        assert store.finished() and not store._batches
        cat = [np.concatenate([c[i] for c in got]) for i in range(4)]
        o = np.argsort(cat[3], kind='stable')
        for a, b in zip(cat, want):
            assert np.array_equal(a[o], b)
        games = store.games()
        assert len(games) == n_games
        for j, (h, r) in enumerate(games):
            if status[j] in (_lib.ST_WON_P1, _lib.ST_WON_P2):
                assert r == int(full['reward'][j]) and len(h) == length[j]
            else:
                assert (h, r) == (None, None)


def test_parallel_generator_fails_loudly_without_a_gpu(tmp_path):
    """selfplay.generate_self_play_in_parallel / train.evolve_in_parallel start their rank processes from a parent that makes no GPU
    call; in this container the ranks have no GPU and must stop at require_gpu() -- no CPU fallback -- and the parent must turn that
    into an exception (the surviving rank is terminated, nothing is returned)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by the -m gpu tests')
    from chinesecheckersagent_amd import _lib, selfplay as sp, train as tr
    w = os.path.join(ROOT, 'tests', 'golden', 'good_model.h5')
    with pytest.raises(_lib.CcspError, match='rank process failed'):
        sp.generate_self_play_in_parallel(w, 4, 2, sims=4, seed=1, first_game=0, out_dir=str(tmp_path))
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith('selfplay-rank')]
    with pytest.raises(RuntimeError, match='rank process failed'):
        tr.evolve_in_parallel(2, w, iterations=1, num_self_play=2, eval_games=2, sims=4, seed=1, work_dir=str(tmp_path),
                              data_dir=str(tmp_path / 'd'), weights_dir=str(tmp_path / 'w'))


def test_run_ranks_times_out_and_kills_a_rank_that_ignores_sigterm():
    """launch.run_ranks: a rank that stalls without exiting (a blocked collective, a wedged kernel) must not hang the caller for ever:
    after `timeout` the ranks get SIGTERM, after `kill_grace` more seconds SIGKILL, and the caller gets 124; a rank that fails makes
    the others leave the same way"""
    import time
    from chinesecheckersagent_amd import launch
    stubborn = [sys.executable, '-c',
                'import signal, time, os\n'
                'signal.signal(signal.SIGTERM, signal.SIG_IGN)\n'
                'time.sleep(0.3 if os.environ["RANK"] == "1" and os.environ.get("FAIL_ONE") else 600)\n'
                'raise SystemExit(7 if os.environ.get("FAIL_ONE") else 0)\n']
    t0 = time.time()
    assert launch.run_ranks(stubborn, 2, timeout=1.0, kill_grace=1.0, poll=0.05) == 124
    assert time.time() - t0 < 20
    t0 = time.time()
    assert launch.run_ranks(stubborn, 2, extra_env={'FAIL_ONE': '1'}, kill_grace=1.0, poll=0.05) == 7
    assert time.time() - t0 < 20


def test_the_hot_path_entry_is_guarded(monkeypatch, capsys):
    """SelfPlayRun's entry check (no GPU needed): a CUDA model that is not on the fused HIP kernel, a reference-style .predict object at a
    batch size where it matters, or lock-step forced at >= 1024 slots leaves the delivered path -- reported once on stderr, an error
    under CCSP_STRICT=1; the delivered configuration, small plumbing runs and deliberate comparisons pass silently"""
    from chinesecheckersagent_amd import _lib, selfplay as sp

    class Dev(object):
        def __init__(self, t):
            self.type = t

    class Fake(object):
        def __init__(self, backend, dev='cuda', precision='fp32'):
            self.backend, self.device, self.precision = backend, Dev(dev), precision

        def evaluate_batch(self, x):
            raise AssertionError

    class Duck(object):
        def predict(self, x):
            raise AssertionError
    monkeypatch.delenv('CCSP_STRICT', raising=False)
    sp._warned.clear()
    sp._check_hot_path(Fake('hip'), None, 4096, None)                  # the delivered evaluator, any size
    sp._check_hot_path(Fake('torch', dev='cpu'), None, 4096, None)     # a CPU module: not this check's business (the engine refuses it)
    sp._check_hot_path(Duck(), None, 8, None)                          # plumbing sizes with a .predict object
    sp._check_hot_path(Fake('hip'), Fake('hip'), 1024, True)
    assert capsys.readouterr().err == ''
    for args, word in (((Fake('torch'), None, 64, None), 'PyTorch modules'), ((Fake('torch', precision='fp64'), None, 64, None), 'fp64'),
                       ((Fake('hip'), Fake('torch'), 64, None), 'model2'), ((Duck(), None, 2048, None), 'one position at a time'),
                       ((Fake('hip'), None, 1024, False), 'lock-step')):
        sp._warned.clear()
        sp._check_hot_path(*args)
        sp._check_hot_path(*args)                                       # ... once
        err = capsys.readouterr().err
        assert err.count('leaves the delivered hot path') == 1 and word in err, (word, err)
        monkeypatch.setenv('CCSP_STRICT', '1')
        with pytest.raises(_lib.CcspError, match='hot path'):
            sp._check_hot_path(*args)
        monkeypatch.delenv('CCSP_STRICT')


def test_bench_roofline_arithmetic_on_canned_numbers():
    """bench.net_roofline (the `roofline` object of the bench line) on round 5's own figures: `frac` is FLOP per launch over the launch's OWN
    duration -- the number a reader recomputes from profiles/*_bench_kernel_stats.csv -- and every other fraction sits under its own key"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.net_roofline(2048, 0.11845, 0.10429, 0.11196, 0.11365, 17.11e6, samples=40, tree_ms=0.0839, launches=16040,
                           asked_rows_in_isolated_batch=1950)
    flop = 2048 * 6483264
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 157.3 and r['traffic'] is None
    assert abs(r['achieved'] - flop / 118.45e-6 / 1e12) < 1e-9 and abs(r['frac'] - 0.7126) < 5e-4          # rocprof's 118.45 us -> 0.713
    assert abs(r['frac_isolated'] - 0.8094) < 5e-4 and abs(r['frac_by_wall'] - 0.7539) < 5e-4 and abs(r['frac_by_step'] - 0.7427) < 5e-4
    assert abs(r['useful_frac_by_step'] - 17.11e6 * 6483264 / 1e12 / 157.3) < 1e-12 and abs(r['useful_frac_by_step'] - 0.7052) < 5e-4
    assert list(r)[:8] == ['bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'useful_frac_by_step', 'frac_by_step']
    assert r['launches_in_flight'] > 1 and r['avg_launch_ms'] == 0.11845 and 'median of 40 launches' in r['how']
    # no uncaptured round to put events around: the wall time per launch stands in, and the record says so
    r2 = bench.net_roofline(2048, None, 0.10429, 0.11196, 0.11365, 17.11e6)
    assert r2['frac'] == r2['frac_by_wall'] and r2['avg_launch_ms'] == 0.11196 and 'wall time' in r2['how']


def test_small_cohort_rank_policy():
    """selfplay.selfplay_ranks: a cohort of fewer than 256 games per rank is latency-bound (profiles/r6_small_cohort.txt: 180 games x 800
    simulations take 6.52 s in 180 slots of one GPU, 5.91 s as one of eight ranks' 23), so min(R, ceil(games / 256)) ranks play it"""
    from chinesecheckersagent_amd import selfplay as sp
    assert sp.MIN_GAMES_PER_RANK == 256
    assert [sp.selfplay_ranks(n, 8) for n in (1, 180, 256, 257, 512, 1024, 2047, 2048, 32768)] == [1, 1, 1, 2, 2, 4, 8, 8, 8]
    assert sp.selfplay_ranks(180, 1) == 1 and sp.selfplay_ranks(0, 8) == 1
    assert sp.selfplay_ranks(9, 2, min_games_per_rank=1) == 2 and sp.selfplay_ranks(40, 5, min_games_per_rank=16) == 3
