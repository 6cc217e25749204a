#!/usr/bin/env python
"""Time the T-path cross-attention kernels alone on the engine's own operands: query tiles (8 / 16 per tile) vs one block per query.
    python tools/microbench_qtile.py --workload cfg3_t --batch 8"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import ops, synthetic  # noqa: E402
from mv2d_amd.engine import HeadEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='cfg3_t')
ap.add_argument('--batch', type=int, default=8)
a = ap.parse_args()
dev = torch.device('cuda:0')
probs = [synthetic.make_problem(a.workload, seed=s) for s in range(a.batch)]
feats = torch.cat([torch.from_numpy(p['feat']) for p in probs]).to(dev)
props = [[torch.from_numpy(x) for x in p['proposals']] for p in probs]
metas = [p['img_metas'] for p in probs]
sd = synthetic.make_head_state(seed=0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for qpt in (2, 4, 8, 16):
    eng = HeadEngine(sd, 'T', dev, num_views=probs[0]['views_per_frame'])
    eng.qtile, eng.qtile_queries = True, qpt          # (opt-in since the per-query kernel in smallest-key order shipped)
    out = eng.run_batch(feats, props, metas) if a.batch > 1 else eng.run(feats, props[0], metas[0])
    torch.cuda.synchronize()
    ws, qt = out['ws'], out['ws']['qt']
    R = ws['x'].shape[0]
    nt = int(qt['nt'].item())
    tot = int(qt['ucnt'][:nt].sum().item())
    S, nnz = int(ws['S_dev'].item()), int(ws['row_ptr'][R].item())
    t_q = timed(lambda: ops.xattn_qtile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], qt, ws['zh'], R))
    print(f'{a.workload} x{a.batch}, {qpt} queries per tile: {nt} tiles, union lists {tot} keys = {tot / S:.2f} x distinct ({S}), per-query lists {nnz / S:.2f} x; '
          f'query-tile kernel {t_q:.1f} us = {tot * 1024 / t_q / 1e6:.2f} TB/s of union rows, {S * 1024 / t_q / 1e6:.2f} TB/s of distinct rows')
t_t = timed(lambda: ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], ws['zh'], R, waves=2))
print(f'per-query kernel {t_t:.1f} us = {nnz * 1024 / t_t / 1e6:.2f} TB/s gathered, {S * 1024 / t_t / 1e6:.2f} TB/s of distinct rows')
t_o = timed(lambda: ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], ws['zh'], R, waves=2, order=qt['perm']))
print(f'per-query kernel, blocks in smallest-key order {t_o:.1f} us = {nnz * 1024 / t_o / 1e6:.2f} TB/s gathered')
for nw_ in (1, 4):
    t_o = timed(lambda: ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], ws['zh'], R, waves=nw_, order=qt['perm']))
    print(f'   ... {nw_} waves per query: {t_o:.1f} us')
