"""sparse cross attention forward / backward in isolation on the S-path key pattern (every query reads its own RoI's 49 cells + one
correlated RoI) and on a T-path-like pattern (150 random keys per query out of S).  usage: python tools/microbench_xattn_bwd.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def bench(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name in ('S-path', 'T-path'):
    if name == 'S-path':
        S = R * 49
        cols = torch.cat([torch.cat([torch.arange(r * 49, r * 49 + 49), torch.arange(((r + 3) % R) * 49, ((r + 3) % R) * 49 + 49)]) for r in range(R)])
        per = 98
    else:
        S = 95000
        per = 150
        cols = torch.stack([torch.randperm(S, generator=g)[:per].sort().values for _ in range(R)]).reshape(-1)
    row_ptr = (torch.arange(R + 1) * per).to(torch.int32).to(dev)
    col = cols.to(torch.int32).to(dev)
    q = torch.randn(R, 256, generator=g).to(dev); K = torch.randn(S, 256, generator=g).to(dev).bfloat16(); V = torch.randn(S, 256, generator=g).to(dev).bfloat16()
    dout = torch.randn(R, 256, generator=g).to(dev)
    out = ops.sparse_xattn(q, K, V, row_ptr, col)
    tf = bench(lambda: ops.sparse_xattn(q, K, V, row_ptr, col, out=out))
    tr = ops.csr_transpose(row_ptr, col, S)
    tb = bench(lambda: ops.sparse_xattn_bwd(q, K, V, row_ptr, col, out, dout, transposed=tr))
    tt = bench(lambda: ops.csr_transpose(row_ptr, col, S))
    nnz = R * per
    print(f'{name}: R={R} S={S} nnz={nnz}: forward {tf:.1f} us ({nnz * 1024 / tf / 1e6:.2f} TB/s of K/V rows), backward {tb:.1f} us (+ {tt:.1f} us to group the pairs by key, torch ops)')
