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
for (M, N, K, sk) in [(300, 256, 256, 1), (300, 768, 256, 1), (300, 1024, 256, 1), (300, 2048, 256, 1), (300, 256, 2048, 8), (300, 256, 2048, 4), (900, 2048, 256, 1), (300, 512, 1056, 1), (300, 512, 1056, 3)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); o = torch.empty((sk, M, N) if sk > 1 else (M, N), device=dev)
    print(f'NT={os.environ.get("MV2D_F32_NT","auto")} gemm_f32 {M}x{N}x{K} split{sk}: {graph_time(lambda: ops.gemm_f32(A, W, None, split_k=sk, out=o)):.2f} us')

for (M, N, K, sk) in [(300, 256, 256, 1), (300, 768, 256, 1), (300, 2048, 256, 1), (300, 256, 2048, 8), (900, 2048, 256, 1), (300, 512, 1056, 1)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); o = torch.empty((sk, M, N) if sk > 1 else (M, N), device=dev)
    hl = ops.split_bf16x2(W)
    t = graph_time(lambda: ops.gemm_x3(A, hl, None, split_k=sk, out=o))
    ref = A.double() @ W.double().T
    got = o.sum(0) if sk > 1 else o
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    print(f'gemm_x3 {M}x{N}x{K} split{sk}: {t:.2f} us  relerr {err:.2e}')
