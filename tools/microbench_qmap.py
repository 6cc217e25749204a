"""xattn_qmap / xattn_ctxmap alone at R rows (graph-replayed, HIP events): python tools/microbench_qmap.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops, synthetic
from mv2d_amd.engine import HeadEngine
dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4800
eng = HeadEngine(synthetic.make_head_state(seed=0), 'S', dev, num_views=6)
W = eng.w
q = torch.randn(R, 256, device=dev)
Qt = torch.empty(R * 512 * 8, device=dev, dtype=torch.bfloat16)
z = torch.randn(R, 8, 256, device=dev)
ctx = torch.empty(R, 256, device=dev)
rp = torch.arange(R + 1, device=dev, dtype=torch.int32)
def graph_time(fn, n=20, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
print(f'R={R}: qmap {graph_time(lambda: ops.xattn_qmap(q, W["ca_mapA0"], Qt, R=R)):.1f} us, '
      f'ctxmap {graph_time(lambda: ops.xattn_ctxmap(z, W["ca_mapB0"], W["ca_v_b0"], rp, ctx, R, empty_nan=True)):.1f} us')
