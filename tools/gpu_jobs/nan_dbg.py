import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
from mv2d_amd import ops
from oracle import mv2d_oracle as O
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
R, S = 40, 400
allowed = torch.rand((R, S), generator=g) < 0.2
q = (torch.randn((R, 256), generator=g) * 0.3).to(dev)
xk, xv = torch.randn((S, 256), generator=g).to(dev), torch.randn((S, 256), generator=g).to(dev)
Wk, Wv = (torch.randn((256, 256), generator=g) * 0.06).to(dev), (torch.randn((256, 256), generator=g) * 0.06).to(dev)
bv = torch.randn(256, generator=g).to(dev)
rp, col = O.csr_from_allowed(allowed); rp, col = rp.to(dev), col.to(dev)
WA, WB = ops.pack_xattn_maps(Wk, Wv)
for what in ('key row NaN', 'value row NaN', 'query NaN'):
    xk2, xv2, q2 = xk.clone(), xv.clone(), q.clone()
    if what.startswith('key'): xk2[7] = float('nan')
    if what.startswith('value'): xv2[7] = float('nan')
    if what.startswith('query'): q2[3] = float('nan')
    Xk, Xk_lo = ops.f32_to_key16(xk2, with_lo=True); Xv, Xv_lo = ops.f32_to_key16(xv2, with_lo=True)
    out = ops.xattn_fused(q2, WA, WB, bv, Xk, Xv, rp, col, Xk_lo=ops.lo8_encode(Xk_lo), Xv_lo=ops.lo8_encode(Xv_lo))
    rows = allowed[:, 7].nonzero().flatten().tolist() if not what.startswith('query') else [3]
    print(what, ': rows that should be NaN', len(rows), '-> NaN rows', int(torch.isnan(out).any(1).sum()), 'all-NaN rows', int(torch.isnan(out).all(1).sum()), '(lib', os.environ.get('MV2D_HIP_LIB', 'built'), ')')
