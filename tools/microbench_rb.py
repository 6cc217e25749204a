import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops
dev = torch.device('cuda:0')
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
M = 300
r = lambda *s: torch.randn(*s, device=dev)
ctx, res, qpos = r(M, 256), r(M, 256), r(M, 256)
Wo, bo, Wq, bq, lw, lb = r(256, 256), r(256), r(256, 256), r(256), r(256), r(256)
x1, q, o = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
print('attn_out_fused 2-stage:', graph_time(lambda: ops.attn_out_fused(ctx, res, Wo, bo, (lw, lb), x1, qpos=qpos, Wq=Wq, bq=bq, qscale=0.2, q_out=q)))
print('attn_out_fused 1-stage:', graph_time(lambda: ops.attn_out_fused(ctx, res, Wo, bo, (lw, lb), x1)))
def sep():
    ops.gemm_f32(ctx, Wo, bo, out=o); ops.row_ln(o, residual=res, ln=(lw, lb), out=x1, addvec=qpos, out_plus=o); ops.gemm_f32(o, Wq, bq, scale=0.2, out=q)
print('separate gemm+ln+gemm (3 kernels):', graph_time(sep) * 1)
def sep1():
    ops.gemm_f32(ctx, Wo, bo, out=o); ops.row_ln(o, residual=res, ln=(lw, lb), out=x1)
print('separate gemm+ln (2 kernels):', graph_time(sep1))
