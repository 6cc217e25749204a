"""GPU time of the dense block of the denoising rows alone (csrc/dense_attn.hip): forward, backward query side, backward key side.

    python tools/train_dense_time.py [--n 400] [--nk 15800] [--p 0.1]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=400)
ap.add_argument('--nk', type=int, default=15800)
ap.add_argument('--p', type=float, default=0.1)
a = ap.parse_args()
lib = _lib.load()
dev = 'cuda'
q, k, v, g = (torch.randn(r, 256, device=dev) for r in (a.n, a.nk, a.nk, a.n))
q *= 0.3
ctx, lse = torch.empty_like(q), torch.empty(8, a.n, device=dev)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
wf = torch.empty(int(lib.mv2d_dense_attn_ws_bytes(a.n, a.nk, 0)), device=dev, dtype=torch.uint8)
wb = torch.empty(int(lib.mv2d_dense_attn_ws_bytes(a.n, a.nk, 1)), device=dev, dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()  # noqa: E731


def fwd():
    _lib.check(lib.mv2d_dense_attn_fwd(P(q), P(k), P(v), a.n, a.nk, a.p, 7, P(ctx), P(lse), P(wf), st), 'fwd')


def bwd(parts):
    _lib.check(lib.mv2d_dense_attn_bwd_parts(P(q), P(k), P(v), P(ctx), P(g), P(lse), a.n, a.nk, a.p, 7, 1.0, P(dq), P(dk), P(dv), P(wb), parts, st), 'bwd')


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


fwd()
print(json.dumps(dict(n=a.n, nk=a.nk, p=a.p, lib=os.environ.get('MV2D_HIP_LIB', 'default'), forward_us=timed(fwd), backward_q_us=timed(lambda: bwd(1)),
                      backward_kv_us=timed(lambda: bwd(2)))))
