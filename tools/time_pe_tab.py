"""Time mv2d_pe_fused_tab (both kernels) on headline-sized input: python tools/time_pe_tab.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 70349
dev = torch.device('cuda:0'); bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
A1 = r(M, 192).to(bf); Xf32 = r(M, 256); Xfb = Xf32.to(bf)
wp = {k: ops.pack_wfrag(v.to(bf)) for k, v in dict(w1a=r(1024, 192, sc=.08), w1b=r(256, 1024, sc=.04), wr=r(256, 256, sc=.07), we=r(256, 256, sc=.07)).items()}
wp.update(dict(b1a=r(1024), b1b=r(256), br=r(256), be=r(256)))
tab = r(8800, 256)
pe = torch.empty((M, 256), device=dev); xk = torch.empty((M, 256), device=dev, dtype=bf)
VARIANTS = (('64', '0'), ('96', '0'), ('96', '3'), ('2', '0'), ('2', '3'))
if os.environ.get('MV2D_PE_ONLY'):
    VARIANTS = ((os.environ['MV2D_PE_ONLY'], '0'),)
for sel, ex in VARIANTS:
    os.environ['MV2D_PE_TAB_KERNEL'] = sel; os.environ['MV2D_PE_EXP'] = ex
    for _ in range(5):
        ops.pe_fused_tab(A1, Xfb, Xf32, None, wp, tab, 8800, pe, xk, M=M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.pe_fused_tab(A1, Xfb, Xf32, None, wp, tab, 8800, pe, xk, M=M)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * M * (192 * 1024 + 1024 * 256 + 2 * 256 * 256)
    print('kernel %s exp %s: %.1f us  %.1f TFLOP/s (%.3f of 2500)' % (sel, ex, us, fl / us / 1e6, fl / us / 1e6 / 2500))
