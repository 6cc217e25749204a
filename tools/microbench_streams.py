import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops
dev = torch.device('cuda:0')
w = torch.ones(256, device=dev)
def make_graph(rows, nk):
    x = torch.randn(rows, 256, device=dev); y = torch.empty_like(x)
    def body():
        for _ in range(nk // 2):
            ops.row_ln(x, ln=(w, w), out=y); ops.row_ln(y, ln=(w, w), out=x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return g
for rows in (8, 300):
    for ns in (1, 2, 4, 8):
        graphs = [make_graph(rows, 100) for _ in range(ns)]
        streams = [torch.cuda.Stream() for _ in range(ns)]
        def run(reps):
            for _ in range(reps):
                for g, s in zip(graphs, streams):
                    with torch.cuda.stream(s):
                        g.replay()
        run(3); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); t1 = time.perf_counter()
        nk = 20 * ns * 100
        print(f'rows={rows} streams={ns}: {1e6*(t1-t0)/nk:.2f} us per kernel aggregate ({nk/(t1-t0)/1e3:.0f} k kernels/s)')
