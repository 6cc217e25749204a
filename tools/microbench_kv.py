"""K/V projection kernel in isolation (HIP events, graph replay of 20 launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 14700
L = 6
A = torch.randn(M, 256, device=dev).bfloat16(); A2 = torch.randn(M, 256, device=dev).bfloat16()
W = (torch.randn(2 * L * 256, 256, device=dev) * 0.06).bfloat16(); b = torch.randn(2 * L * 256, device=dev)
out = torch.empty((2 * L, M, 256), device=dev, dtype=torch.bfloat16)
def run_new(): ops.kv_proj(A, W, b, out, A2=A2, n_split=L * 256, ldc=256, c_blk_stride=M * 256, c_blk_cols=256)
def run_old(): ops.gemm_bf16(A, W, b, A2=A2, n_split=L * 256, out=out, ldc=256, c_blk_stride=M * 256, c_blk_cols=256)
for name, fn in (('kv_proj', run_new), ('gemm_bf16', run_old)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * M * 3072 * 256
    print(f'{name}: M={M} {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s  out {M*3072*2/us/1e6:.2f} TB/s')
