"""extra_enc.0 (K = 1056) as one GEMM vs 11 split-K slices + slab sum: kernel times in a hipGraph (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops
dev = torch.device('cuda:0')
R = 300
A = torch.randn(R, 1056, device=dev); W = torch.randn(512, 1056, device=dev) * 0.03; b = torch.randn(512, device=dev)
out = torch.empty(R, 512, device=dev); parts = torch.empty(11, R, 512, device=dev); out2 = torch.empty(R, 512, device=dev)
def one(): ops.gemm_f32(A, W, b, act=1, out=out)
def split():
    ops.gemm_f32(A, W, b, split_k=11, out=parts)
    ops.row_ln(parts.view(11, 2 * R, 256), relu=True, out=out2.view(2 * R, 256), M=2 * R)
def split3():
    ops.gemm_f32(A, W, b, split_k=3, out=parts[:3])
    ops.row_ln(parts[:3].view(3, 2 * R, 256), relu=True, out=out2.view(2 * R, 256), M=2 * R)
for name, fn in (('one', one), ('split11', split), ('split3', split3)):
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
    print(f'{name}: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us', float((out - out2).abs().max()) if name != 'one' else '')
