#!/usr/bin/env python
"""Tile attention with hi + lo key / value rows (index-exact route): separate lo arrays vs hi | lo interleaved in one 1 KB row.
    python tools/microbench_xlo.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
R, S, nk = 2400, 2400 * 49, 56
g = torch.Generator(device=dev).manual_seed(1)
row_ptr = (torch.arange(R + 1, device=dev, dtype=torch.int32) * nk)
col = torch.stack([torch.randperm(S, device=dev, generator=g)[:nk] for _ in range(64)]).repeat(R // 64 + 1, 1)[:R].reshape(-1).to(torch.int32)
col = (torch.arange(R, device=dev).repeat_interleave(nk) * 49 + torch.arange(nk, device=dev).repeat(R) % 49).to(torch.int32)     # S-path pattern: own RoI rows
Qt = torch.randn(R, 4096, device=dev, generator=g).to(torch.bfloat16)
il_k = torch.randn(S, 2, 256, device=dev, generator=g).to(torch.bfloat16)
il_v = torch.randn(S, 2, 256, device=dev, generator=g).to(torch.bfloat16)
sep = [il_k[:, 0].contiguous(), il_v[:, 0].contiguous(), il_k[:, 1].contiguous(), il_v[:, 1].contiguous()]
z = torch.empty(R, 8, 256, device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for w in (2, 4):
    t_sep = timed(lambda: ops.xattn_tile(Qt, sep[0], sep[1], row_ptr, col, z, R, waves=w, Xk_lo=sep[2], Xv_lo=sep[3]))
    z1 = z.clone()
    t_il = timed(lambda: ops.xattn_tile(Qt, il_k[:, 0], il_v[:, 0], row_ptr, col, z, R, waves=w, Xk_lo=il_k[:, 1], Xv_lo=il_v[:, 1], row_bytes=1024))
    print(f'{w} waves: separate lo arrays {t_sep:.1f} us, hi | lo interleaved rows {t_il:.1f} us, equal {bool(torch.equal(z, z1))}')
t0 = timed(lambda: ops.xattn_tile(Qt, sep[0], sep[1], row_ptr, col, z, R, waves=2))
print(f'hi only (default route): {t0:.1f} us')
