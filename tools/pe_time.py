"""Times of the two shapes of the split-precision PE kernel (csrc/pe_x3.hip, csrc/pe_x3b.hip) on M rows; variant libraries through MV2D_HIP_LIB
(tools/build_variant.sh pb1 pe_x3b.hip -DMV2D_PB_DBG=1 ...):   python tools/pe_time.py [M] [rows16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
rows16 = len(sys.argv) > 2 and sys.argv[2] == '1'
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)      # noqa: E731
A1, X = r(M, 192, sc=3.0), r(M, 256)
W = dict(w1a=r(1024, 192, sc=0.08), w1b=r(256, 1024, sc=0.04), wr=r(256, 256, sc=0.07), we=r(256, 256, sc=0.07))
wx = {k: ops.pack_x3(v) for k, v in W.items()}
wx['w1a_p'], wx['wr_p'] = ops.pack_x3_rowperm(W['w1a']), ops.pack_x3_rowperm(W['wr'])
wx.update(b1a=r(1024), b1b=r(256), br=r(256), be=r(256))
tab = r(4096, 256)
ri = torch.randperm(M, generator=g).to(torch.int32).to(dev)
k16 = ops.key16_dtype()
pe = None if rows16 else torch.empty(M, 256, device=dev)
pairs = [tuple(torch.empty((M, 256), device=dev, dtype=k16) for _ in range(2)) for _ in range(2)] if rows16 else [None, None]
for name, fn in (('pe_x3  (round 5)', ops.pe_fused_x3), ('pe_x3b (round 6)', ops.pe_fused_x3b)):
    for _ in range(3):
        fn(A1, X, None, wx, tab, 4096, pe=pe, Xk=pairs[0], Xv=pairs[1], row_index=ri)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(A1, X, None, wx, tab, 4096, pe=pe, Xk=pairs[0], Xv=pairs[1], row_index=ri)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f'{name} {M} rows ({"key16 rows" if rows16 else "pe"}): {us:7.1f} us = {M * 589824 * 2 * 3 / us / 1e6:6.0f} TFLOP/s of issued MFMA work ({os.environ.get("MV2D_HIP_LIB", "built library")})')
