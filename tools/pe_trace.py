"""per-phase s_memtime stamps of block 0 / thread 0 of the fused PE kernel (variant built with -DMV2D_PE_TRACE):
   MV2D_HIP_LIB=mv2d_amd/lib/variants/libpetrace.so python tools/pe_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops, synthetic, _lib
from mv2d_amd.engine import HeadEngine
M = 8794
dev = torch.device('cuda:0')
eng = HeadEngine(synthetic.make_head_state(seed=0), 'S', dev, num_views=6)
wp = eng.w['pe_pack']
g = torch.Generator().manual_seed(0)
A1 = torch.randn(M, 192, generator=g).to(dev).bfloat16(); A2 = torch.randn(M, 384, generator=g).to(dev).bfloat16()
Xf = torch.randn(M, 256, generator=g).to(dev); Xfb = Xf.bfloat16()
pe = torch.empty(M, 256, device=dev); xk = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    ops.pe_fused(A1, A2, Xfb, Xf, None, wp, pe, xk)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * 64)()
ctypes.CDLL(os.environ['MV2D_HIP_LIB']).mv2d_pe_trace_read(buf, 64)
t = list(buf)
names = ['start'] + [f'{m}{h}:{ph}' for m, h in (('G', 0), ('A', 0), ('A', 1), ('B', 0), ('B', 1)) for ph in ('L1 end', 'after barrier', 'L2 end', 'after barrier')] + ['kernel end']
prev = t[0]
for n, v in zip(names, t[:len(names)]):
    print(f'{n:22s} +{(v - prev) / 100.0:8.2f} us   (at {(v - t[0]) / 100.0:7.2f})')
    prev = v
