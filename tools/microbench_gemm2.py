import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
def graph_time(fn, n=20, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
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
def bench(M, N, K, out_f32=False, tag=''):
    A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) * 0.05).to(BF); b = torch.randn(N, device=dev)
    out = torch.empty((M, N), device=dev, dtype=torch.float32 if out_f32 else BF)
    us = graph_time(lambda: ops.gemm_bf16(A, W, b, out=out))
    print(f'{tag:10s} M={M:6d} N={N:5d} K={K:5d} {"f32" if out_f32 else "bf16"}: {us:8.2f} us {2.0*M*N*K/us/1e6:8.1f} TF/s')
for K in (64, 128, 256, 512, 1024):
    bench(14700, 3072, K, tag='kv-like')
for K in (64, 256, 1024):
    bench(8832, 256, K, tag='pe_out')
for K in (64, 192, 384):
    bench(8832, 1024, K, tag='pe_in')
bench(14700, 256, 2304, out_f32=True, tag='conv-like')
for M in (1024, 4096, 16384):
    bench(M, 3072, 256, tag='kv-M')
x = torch.randn(300, 256, device=dev); w = torch.ones(256, device=dev); y = torch.empty_like(x)
print('row_ln 300 graph:', graph_time(lambda: ops.row_ln(x, ln=(w, w), out=y)))
A = torch.randn(300, 256, device=dev); W = torch.randn(256, 256, device=dev); o = torch.empty(300, 256, device=dev)
print('gemm_f32 300x256x256 graph:', graph_time(lambda: ops.gemm_f32(A, W, None, out=o)))
W2 = torch.randn(2048, 256, device=dev); o2 = torch.empty(300, 2048, device=dev)
print('gemm_f32 300x2048x256 graph:', graph_time(lambda: ops.gemm_f32(A, W2, None, out=o2)))
qkv = torch.randn(300, 768, device=dev); c = torch.empty(300, 256, device=dev)
print('self_attn 300 graph:', graph_time(lambda: ops.self_attn(qkv, c)))
