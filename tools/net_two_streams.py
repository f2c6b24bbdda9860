"""What does an evaluator launch cost when two half-batches' launches follow each other on two streams with NOTHING else on the device?
(the pipeline's 0.128 ms per launch against 0.1186 ms for a back-to-back burst on one stream: how much of the difference is the tree
kernels beside the launches, how much the hand-over between the two streams' graphs)   python tools/net_two_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chinesecheckersagent_amd.model import ResidualCNN

m = ResidualCNN(); m.load_weights('tests/golden/good_model.h5')
x = [(torch.rand(2048, 343, device='cuda') < 0.1).float() for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s, xi in zip(streams, x):                      # warm-up (allocator) on each stream
    with torch.cuda.stream(s):
        for _ in range(3):
            m.evaluate_batch(xi)
torch.cuda.synchronize()
graphs = []
for s, xi in zip(streams, x):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(25):
                m.evaluate_batch(xi)
    graphs.append(g)
torch.cuda.synchronize()
for mode in ('one stream', 'two streams, graphs handed over alternately'):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        n = 0
        for _ in range(40):
            if mode == 'one stream':
                with torch.cuda.stream(streams[0]):
                    graphs[0].replay(); graphs[0].replay()
            else:
                for s, g in zip(streams, graphs):
                    with torch.cuda.stream(s):
                        g.replay()
            n += 50
        torch.cuda.synchronize(); dt = time.time() - t0
    print('%-45s %.4f ms of wall per evaluator launch of 2048 positions' % (mode, dt / n * 1e3), flush=True)

# the same launches with a stream of EMPTY-ish kernels beside them (each kernel boundary is an acquire / release of the device's caches):
# does the evaluator pay for other kernels' boundaries?
tiny = torch.zeros(64, device='cuda')
gt = torch.cuda.CUDAGraph()
with torch.cuda.stream(streams[1]):
    for _ in range(3):
        tiny.add_(1.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(gt, stream=streams[1]):
        for _ in range(100):
            tiny.add_(1.0)
torch.cuda.synchronize()
for per_replay in (0, 1, 4):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time(); n = 0
        for _ in range(40):
            with torch.cuda.stream(streams[0]):
                graphs[0].replay()
            with torch.cuda.stream(streams[1]):
                for _ in range(per_replay):
                    gt.replay()
            n += 25
        torch.cuda.synchronize(); dt = time.time() - t0
    print('one stream of evaluator launches + %3d tiny kernels per 25 launches on another stream: %.4f ms per evaluator launch' % (100 * per_replay, dt / n * 1e3), flush=True)
