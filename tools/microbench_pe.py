"""pe_fused alone at the cfg2_s size: back to back (weights warm in L2) and after an L2 flush.
usage: python tools/microbench_pe.py [M]   (MV2D_HIP_LIB=<variant .so from tools/build_variant.sh> for experiments)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops, synthetic
from mv2d_amd.engine import HeadEngine

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8794
dev = torch.device('cuda:0')
eng = HeadEngine(synthetic.make_head_state(seed=0), 'S', dev, num_views=6)
wp = eng.w['pe_pack']
g = torch.Generator(device='cpu').manual_seed(0)
A1 = torch.randn(M, 192, generator=g).to(dev).bfloat16(); A2 = torch.randn(M, 384, generator=g).to(dev).bfloat16()
Xf = torch.randn(M, 256, generator=g).to(dev); Xfb = Xf.bfloat16()
pe = torch.empty(M, 256, device=dev); xk = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def run(mode, n=30):
    ts = []
    for _ in range(n):
        if mode != 'hot':
            junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.pe_fused(A1, A2, Xfb, Xf, None, wp, pe, xk); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for mode in os.environ.get('MV2D_PE_BENCH_MODES', 'hot,flushed,hot').split(','):
    print(f'M={M} {mode:8s}: {run(mode):7.1f} us (event-timed, includes ~launch overhead)')
