"""per-phase s_memtime stamps (100 MHz) of one block of pe_tab96_kernel; standalone variant:
   hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -DMV2D_PE_TRACE=300 mv2d_amd/csrc/pe_tab96.hip -o gpurun_out/libpe96trace.so"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import ops
M = 70349
dev = torch.device('cuda:0'); bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
A1 = r(M, 192).to(bf); Xf32 = r(M, 256); Xfb = Xf32.to(bf)
wp = {k: ops.pack_wfrag(v.to(bf)) for k, v in dict(w1a=r(1024, 192, sc=.08), w1b=r(256, 1024, sc=.04), wr=r(256, 256, sc=.07), we=r(256, 256, sc=.07)).items()}
wp.update(dict(b1a=r(1024), b1b=r(256), br=r(256), be=r(256)))
tab = r(8800, 256)
pe = torch.empty((M, 256), device=dev); xk = torch.empty((M, 256), device=dev, dtype=bf)
lib = ctypes.CDLL(sys.argv[1])
P = ctypes.c_void_p
p = lambda t: P(t.data_ptr())
for ex, shape, with_xk in (('0', 1, True), ('0', 1, False)):
    os.environ['MV2D_PE_EXP'] = ex
    for _ in range(3):
        rc = lib.mv2d_pe_fused_tab2(p(A1), p(Xfb), p(Xf32), None, None, M, p(wp['w1a']), p(wp['b1a']), p(wp['w1b']), p(wp['b1b']), p(wp['wr']), p(wp['br']),
                                     p(wp['we']), p(wp['be']), p(tab), 8800, p(pe), p(xk) if with_xk else None, shape, None)
        assert rc == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.mv2d_pe96_trace_read(buf, 64)
    t = list(buf)
    names = ['start', 'prologue', 'A0 L1+bar', 'A0 L2', 'A1 L1+bar', 'A1 L2', 'A2 L1+bar', 'A2 L2', 'A3 L1+bar', 'A3 L2 + stage', 'barrier', 'G L1+bar', 'G L2',
             'ri + gate math + barrier', 'out cols 0', 'out cols 1']
    print('exp', ex, 'shape', shape, 'Xk written' if with_xk else 'pe only (S path)')
    prev = t[0]
    for n, v in zip(names, t):
        print(f'  {n:18s} +{(v - prev) / 100.0:7.2f} us   (at {(v - t[0]) / 100.0:7.2f})')
        prev = v
