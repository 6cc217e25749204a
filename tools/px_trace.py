"""per-phase s_memtime stamps of one block of the split-precision PE kernel (csrc/pe_x3.hip); variant library:
   tools/build_variant.sh pxtrace pe_x3.hip -DMV2D_PX_TRACE=300 && MV2D_HIP_LIB=mv2d_amd/lib/variants/libpxtrace.so python tools/px_trace.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
M = 140000
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)      # noqa: E731
A1, X = r(M, 192, sc=3.0), r(M, 256)
W = dict(w1a=r(1024, 192, sc=0.08), w1b=r(256, 1024, sc=0.04), wr=r(256, 256, sc=0.07), we=r(256, 256, sc=0.07))
wx = {k: ops.pack_x3(v) for k, v in W.items()}
wx.update(b1a=r(1024), b1b=r(256), br=r(256), be=r(256))
tab = r(4096, 256)
ri = torch.randperm(M, generator=g).to(torch.int32).to(dev)
pe = torch.empty(M, 256, device=dev)
for _ in range(3):
    ops.pe_fused_x3(A1, X, None, wx, tab, 4096, pe=pe, row_index=ri)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.pe_fused_x3(A1, X, None, wx, tab, 4096, pe=pe, row_index=ri)
e1.record()
torch.cuda.synchronize()
print(f'pe_x3 {M} rows: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us; output checksum {int(pe.view(torch.int32).to(torch.int64).sum().item())} (equal across bitwise-equal variants)')
buf = (ctypes.c_longlong * 32)()
lib = ctypes.CDLL(os.environ['MV2D_HIP_LIB'])
if hasattr(lib, 'mv2d_px_trace_read'):
    lib.mv2d_px_trace_read(buf, 32)
    t = list(buf)[:17]
    names = ['prologue+A', 'L1 p0', 'L2 p0', 'L1 p1', 'L2 p1', 'L1 p2', 'L2 p2', 'L1 p3', 'feat stage', 'L2 p3', 'barrier', 'gate L1', 'gate L2', 'sigmoid+barrier', 'out jp0', 'out jp1']
    for n, a, b in zip(names, t[:-1], t[1:]):
        print(f'{n:16s} {b - a:8d}')
    print('total', t[16] - t[0])
