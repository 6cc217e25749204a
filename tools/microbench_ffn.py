#!/usr/bin/env python
"""ffn_x3 alone (2400 rows unless given): kernel time per slices-per-block setting; MV2D_HIP_LIB selects a variant library (row tiles per block).
    python tools/microbench_ffn.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, 256, device=dev, generator=g)
W1 = torch.randn(2048, 256, device=dev, generator=g) * 0.05
W2 = torch.randn(256, 2048, device=dev, generator=g) * 0.05
b1 = torch.randn(2048, device=dev, generator=g) * 0.1
w1x, w2x = ops.pack_x3(W1), ops.pack_x3(W2)
ref = torch.relu(x.double() @ W1.double().T + b1.double()) @ W2.double().T


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for G in (2, 4, 8):
    slabs = torch.empty((32 // G, M, 256), device=dev)
    t = timed(lambda: ops.ffn_fused_x3(x, w1x, b1, w2x, slabs, M, groups=G))
    err = float((slabs.sum(0).double() - ref).abs().max() / ref.abs().max())
    print(f'rows {M}, {G} slices per block ({32 // G} slabs): {t:.1f} us = {M * 2 * 2 * 256 * 2048 * 3 / t / 1e6:.0f} TFLOP/s of bf16 MFMA work; rel err {err:.1e}')
