#!/usr/bin/env python
"""Launch order of the per-query blocks of the tile cross attention (T path, index-exact route): the engine's smallest-key order against
length-aware orders computed on the host (experiment for LOG.md; the kernel results are bitwise the same for every order).
    python tools/microbench_tile_order.py [cfg3_t|cfg5_t] [samples per launch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mv2d_amd import ops, synthetic  # noqa: E402
from mv2d_amd.engine import HeadEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3_t'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if name == 'cfg3_t' else 4)
dev = torch.device('cuda:0')
probs = [synthetic.make_problem(name, seed=s) for s in range(B)]
eng = HeadEngine(synthetic.make_head_state(seed=0), probs[0]['kind'], dev, num_views=probs[0]['views_per_frame'])
eng.fork_qg = False
feats = torch.cat([torch.from_numpy(p['feat']) for p in probs]).to(dev)
out = eng.run_batch(feats, [[torch.from_numpy(x) for x in p['proposals']] for p in probs], [p['img_metas'] for p in probs])
torch.cuda.synchronize()
ws = out['ws']
R = ws['x'].shape[0]
rp = ws['row_ptr'][:R + 1].cpu().numpy()
ln = np.diff(rp)
base = ws['q_order'].cpu().numpy().copy()
print(f'{name} x {B}: R = {R}, keys per query mean {ln.mean():.0f}, max {ln.max()}, nnz {rp[-1]}')


def timed(order):
    o = None if order is None else torch.from_numpy(order.astype(np.int32)).to(dev)
    kw = dict(Xk_lo=ws['xk_lo'], Xv_lo=ws['xv_lo'])
    for _ in range(3):
        ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], ws['zh'], R, waves=eng.xattn_waves, order=o, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], ws['zh'], R, waves=eng.xattn_waves, order=o, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


def xcd_chunks(n):                      # the kernel's block -> logical index map: XCD x gets one contiguous range of the order
    q, rem = n // 8, n % 8
    b = [0]
    for x in range(8):
        b.append(b[-1] + q + (1 if x < rem else 0))
    return b


res = {'natural (no order)': timed(None), 'smallest key (engine)': timed(base)}
res['longest first, global'] = timed(np.argsort(-ln, kind='stable'))
b = xcd_chunks(R)
o = base.copy()
for x in range(8):
    seg = o[b[x]:b[x + 1]]
    o[b[x]:b[x + 1]] = seg[np.argsort(-ln[seg], kind='stable')]
res['smallest-key chunks per XCD, longest first inside'] = timed(o)
for win in (32, 128):
    o = base.copy()
    for x in range(8):
        for s in range(b[x], b[x + 1], win):
            seg = o[s:min(s + win, b[x + 1])]
            o[s:s + len(seg)] = seg[np.argsort(-ln[seg], kind='stable')]
    res[f'smallest key, longest first inside windows of {win}'] = timed(o)
# the longest rows of every XCD chunk first (they set the makespan), the rest in smallest-key order
o = base.copy()
for x in range(8):
    seg = o[b[x]:b[x + 1]]
    big = ln[seg] > 1.5 * ln.mean()
    o[b[x]:b[x + 1]] = np.concatenate([seg[big][np.argsort(-ln[seg][big], kind='stable')], seg[~big]])
res['rows > 1.5 x mean first, rest in smallest-key order'] = timed(o)
for k, v in res.items():
    print(f'  {k:60s} {v:7.1f} us')
