"""Development aid for the counter passes (tools/pmc_round.sh): a few launches of every kernel of the library on the
benchmark's shapes -- move generation / step / encode / greedy on 2^22 positions of the SURVEY 8d distribution, the
evaluator kernel on 2048 positions, three launches of four plies of config 2a (fused path), one ply of config 3 (stepped path)."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from chinesecheckersagent_amd import _lib, engine, rules, selfplay as sp
from chinesecheckersagent_amd.model import ResidualCNN

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
L = _lib.lib()
st = engine._stream_ptr()
if what in ('all', 'rules'):
    n = 1 << 22
    sd0, pl0 = bench.s1_positions(1 << 16, torch, rules, _lib)
    sd, player = sd0.repeat(n >> 16, 1).contiguous(), pl0.repeat(n >> 16).contiguous()
    moves, count, masks = rules.movegen(sd, player)
    for _ in range(3):
        L.ccsp_movegen(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), st)
    mv = moves[:, 0, :].contiguous()
    for _ in range(3):
        L.ccsp_movegen_packed(sd.data_ptr(), player.data_ptr(), n, moves.data_ptr(), count.data_ptr(), masks.data_ptr(), st)
    nxt = torch.empty_like(sd); w = torch.zeros(n, dtype=torch.uint8, device='cuda'); pr = torch.zeros((n, 2), dtype=torch.uint8, device='cuda')
    for _ in range(3):
        L.ccsp_step(sd.data_ptr(), player.data_ptr(), mv.data_ptr(), n, nxt.data_ptr(), w.data_ptr(), pr.data_ptr(), st)
    ne = 1 << 20
    planes = torch.empty((ne, 343), dtype=torch.float32, device='cuda')
    for _ in range(3):
        L.ccsp_encode(sd.data_ptr(), player.data_ptr(), ne, planes.data_ptr(), st)
    best = torch.zeros((ne, _lib.GREEDY_MAX, 2), dtype=torch.uint8, device='cuda'); cnt = torch.zeros(ne, dtype=torch.uint8, device='cuda')
    for _ in range(3):
        L.ccsp_greedy_best(sd.data_ptr(), player.data_ptr(), ne, best.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    print('rules kernels: n = %d, mean moves %.2f' % (n, float(count.float().mean())))
if what in ('all', 'net'):
    m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
    x = torch.rand((2048, 343), device='cuda')
    for _ in range(8):
        m.evaluate_batch(x)
    torch.cuda.synchronize()
if what in ('all', 'fused'):
    e = engine.SelfPlayEngine(n_slots=4096, sims=400, seed=bench.SEED, max_games=4096 * 8, log_capacity=4096 * 16, auto_restart=True)
    e.play_plies(_lib.EVAL_UNIFORM, 6)          # the opening plies: a launch of its own (dropped by pmc_aggregate.py as the cold one)
    for _ in range(3):
        e.play_plies(_lib.EVAL_UNIFORM, 4)      # fused_plies_kernel, four searched plies per launch
    torch.cuda.synchronize()
    e.close()
if what in ('all', 'free'):         # the free-running stepped path on PLAIN launches (no hipGraph under --pmc): net_forward_kernel beside
    m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')     # advance_kernel / boundary_kernel at 4096 slots x 400 simulations
    # (every slot starts at once -- stagger span of one boundary call -- so that the launches the aggregator keeps, the LAST THIRD of
    # each kernel's, are the steady state: all 2048 slots of a half-batch searching, reused positions, ply ends and roots included)
    run = sp.SelfPlayRun(m, n_games=4096 * 8, sims=400, seed=bench.SEED, max_slots=4096, keep_records=False, use_graph=False, stagger_span=6)
    for b in (run.b.parts if hasattr(run.b, 'parts') else [run.b]):
        b.play_steps(int(sys.argv[2]) if len(sys.argv) > 2 else 1300)     # (1300 rounds: three plies of every slot)
    torch.cuda.synchronize()
    print('free-running: counters', run.counters())
    run.close()
if what == 'pipe':           # tools/pmc_pipeline.sh: the evaluator inside the running free-running pipeline, then the same launches alone
    m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
    run = sp.SelfPlayRun(m, n_games=4096 * 8, sims=400, seed=bench.SEED, max_slots=4096, keep_records=False, use_graph=False, stagger_span=6)
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 900
    parts = run.b.parts
    done = 0
    while done < rounds:                 # the two half-batches' rounds handed to their streams alternately, 25 at a time (PipelinedSelfPlay.play_ply)
        for b, s_ in zip(parts, run.b.streams):
            with torch.cuda.stream(s_):
                b.play_steps(25)
        done += 25
    torch.cuda.synchronize()
    p0 = parts[0]
    req, moves = p0._req.clone(), p0._moves.clone()
    packed = m._ensure_packed()
    pk = torch.empty((req.shape[0], _lib.REQUEST_MOVES), dtype=torch.float64, device='cuda'); v = torch.empty(req.shape[0], dtype=torch.float32, device='cuda')
    print('pipeline: counters', run.counters(), 'asked rows in the burst batch', int((req.view(torch.int32)[:, 8] != 0).sum()))
    run.close()
    torch.cuda.synchronize()
    for _ in range(60):
        L.ccsp_net_forward_requests(packed.data_ptr(), req.data_ptr(), moves.data_ptr(), req.shape[0], pk.data_ptr(), v.data_ptr(), st)
    torch.cuda.synchronize()
if what == 'stepped':            # hipGraph replays: NOT under --pmc (the counter passes never finished with it)
    print(sp.bench_net_plies(4096, 400, plies=1))
