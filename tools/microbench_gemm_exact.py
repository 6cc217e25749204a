"""The four K-concatenated products of the index-exact PE block alone (M key positions): python tools/microbench_gemm_exact.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 140000
def graph_time(fn, n=5, reps=5):
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
tot = 0.0
for tag, N, K, split3 in (('frustum L1', 1024, 192, True), ('frustum L2', 256, 1024, False), ('gate L1', 256, 256, True), ('gate L2', 256, 256, False)):
    A = torch.randn(M, 3 * K, device=dev).to(BF); W = (torch.randn(N, 3 * K, device=dev) * 0.05).to(BF); b = torch.randn(N, device=dev)
    out = torch.empty((M, 3 * N), device=dev, dtype=BF) if split3 else torch.empty((M, N), device=dev)
    us = graph_time(lambda: ops.gemm_bf16(A, W, b, act=1 if split3 else 0, out=out, split3=split3, M=M))
    tot += us
    print(f'{tag:10s} M={M} N={N:5d} K\'={3 * K:5d}: {us:8.1f} us {2.0 * M * N * 3 * K / us / 1e6:7.1f} TFLOP/s  (A {M * 3 * K * 2 / 1e6:.0f} MB in, C {out.numel() * out.element_size() / 1e6:.0f} MB out)')
print(f'sum {tot:.0f} us')
