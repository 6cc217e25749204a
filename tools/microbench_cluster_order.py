#!/usr/bin/env python
"""Launch order of the cross-attention blocks: the engine's smallest-key order against a CLUSTER order computed on the host (queries that share
keys become neighbours: greedy sum-linkage on the exact pairwise overlaps, groups of 8) -- experiment for the round-6 key-sharing work.
    python tools/microbench_cluster_order.py [cfg3_t|cfg5_t|cfg2_s_nc6] [samples per launch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

from mv2d_amd import ops, synthetic  # noqa: E402
from mv2d_amd.engine import HeadEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3_t'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (4 if name == 'cfg5_t' else 16)
dev = torch.device('cuda:0')
probs = [synthetic.make_problem(name, seed=s) for s in range(B)]
eng = HeadEngine(synthetic.make_head_state(seed=0), probs[0]['kind'], dev, num_views=probs[0]['views_per_frame'])
eng.fork_qg = False
eng.group_xattn = False
feats = torch.cat([torch.from_numpy(p['feat']) for p in probs]).to(dev)
out = eng.run_batch(feats, [[torch.from_numpy(x) for x in p['proposals']] for p in probs], [p['img_metas'] for p in probs])
torch.cuda.synchronize()
ws = out['ws']
R = ws['x'].shape[0]
rp = ws['row_ptr'][:R + 1].cpu().numpy()
ci = ws['col_idx'][:rp[-1]].cpu().numpy()
grp = ws['grp_start'].cpu().numpy().tolist() + [R]
base = ws['q_order'].cpu().numpy().copy()
ln = np.diff(rp)
S = int(ci.max()) + 1
A = sp.csr_matrix((np.ones(len(ci), np.float32), ci, rp), shape=(R, S))
print(f'{name} x {B}: R = {R}, keys per query {ln.mean():.0f}, nnz {rp[-1]}, distinct keys {len(np.unique(ci))} (x {rp[-1] / len(np.unique(ci)):.2f})')


def union_factor(order, G=8):
    tot = 0
    for b in range(len(grp) - 1):
        for s0 in range(grp[b], grp[b + 1], G):
            m = order[s0:min(s0 + G, grp[b + 1])]
            tot += len(np.unique(np.concatenate([ci[rp[r]:rp[r + 1]] for r in m]))) if len(m) else 0
    return tot / len(np.unique(ci))


def cluster(lo, hi, G=8):
    """greedy sum-linkage inside one sample's slot range; seeds in smallest-key order"""
    rows = base[lo:hi]
    sub = A[rows]
    ov = (sub @ sub.T).toarray()
    sz = np.maximum(np.asarray(sub.sum(1)).ravel(), 1)
    n = hi - lo
    left = np.ones(n, bool)
    outl = []
    for s in range(n):
        if not left[s]:
            continue
        left[s] = False
        g = [s]
        score = ov[s].copy()
        while len(g) < G and left.any():
            sc = score / sz
            sc[~left] = -1
            c = int(np.argmax(sc))
            if sc[c] <= 0:
                c = int(np.argmax(left))
            left[c] = False
            g.append(c)
            score += ov[c]
        outl += g
    return rows[np.array(outl, dtype=np.int64)]


clu = base.copy()
for b in range(len(grp) - 1):
    if grp[b + 1] > grp[b]:
        clu[grp[b]:grp[b + 1]] = cluster(grp[b], grp[b + 1])
print(f'union of 8 consecutive slots / distinct keys: smallest-key order {union_factor(base):.3f}, cluster order {union_factor(clu):.3f}')


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


kw = dict(Xk_lo=ws['xk_lo'], Xv_lo=ws['xv_lo'])
W_ = eng.w
# launch orders that put rows of similar LENGTH side by side (the one-launch kernel waits for the longest of a block's 8 rows): by length inside every
# sample, and by length inside windows of 64 / 256 slots of the smallest-key order
bylen, win64, win256 = base.copy(), base.copy(), base.copy()
for b in range(len(grp) - 1):
    seg = base[grp[b]:grp[b + 1]]
    bylen[grp[b]:grp[b + 1]] = seg[np.argsort(-ln[seg], kind='stable')]
    for o_, w_ in ((win64, 64), (win256, 256)):
        for s0 in range(grp[b], grp[b + 1], w_):
            sg = o_[s0:min(s0 + w_, grp[b + 1])]
            o_[s0:s0 + len(sg)] = sg[np.argsort(-ln[sg], kind='stable')]
for label, o in (('natural', None), ('smallest key (engine)', base), ('cluster order', clu), ('by row length', bylen), ('smallest key, length in 64s', win64), ('smallest key, length in 256s', win256)):
    od = None if o is None else torch.from_numpy(o.astype(np.int32)).to(dev)
    t_tile = timed(lambda: ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], ws['zh'], R, waves=eng.xattn_waves, order=od, **kw))
    t_fused = timed(lambda: ops.xattn_fused(ws['q'], W_['ca_mapA0'], W_['ca_mapB0'], W_['ca_v_b0'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'],
                                             out=ws['ctx'], R=R, order=od, **kw))
    print(f'  {label:28s} tile kernel {t_tile:7.1f} us   one-launch kernel (xattn_fused) {t_fused:7.1f} us')
# the shared-tile kernel on both orders
for label, o in (('smallest key (engine)', base), ('cluster order', clu)):
    od = torch.from_numpy(o.astype(np.int32)).to(dev)
    tab = ops.xattn_group_alloc(R, B, int(ws['col_idx'].numel()), dev)
    ops.xattn_group_tables(ws['row_ptr'], ws['col_idx'], ws['grp_start'], R, tab, order=od)
    t_g = timed(lambda: ops.xattn_group(ws['q'], W_['ca_mapA0'], W_['ca_mapB0'], W_['ca_v_b0'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], tab, out=ws['ctx'], R=R,
                                         order=od, **kw))
    print(f'  {label:28s} shared-tile kernel (xattn_group) {t_g:7.1f} us, union {int(tab["ctl"][0])} entries')
