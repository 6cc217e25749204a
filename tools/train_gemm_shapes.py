"""Which dense products does one training step of the head issue, and what does each cost alone?  (shape census + stand-alone timings)

    python tools/train_gemm_shapes.py [--problem cfg2_s]
"""
import argparse
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import autograd_ops as ao, configs, registry, synthetic  # noqa: E402
import mv2d_amd.plugin  # noqa: F401,E402

ap = argparse.ArgumentParser()
ap.add_argument('--problem', default='cfg2_s')
a = ap.parse_args()
dev = 'cuda'
prob = synthetic.make_problem(a.problem, seed=0)
kind = prob['kind']
cfg = configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t()
if kind == 'T':
    cfg['num_views'] = prob['views_per_frame']
head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
head = head.to(dev)
gtc = synthetic.make_train_gt(40, 3)
gt, labels = [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])]
feat = torch.from_numpy(prob['feat']).to(dev).requires_grad_(True)
props = [torch.from_numpy(p) for p in prob['proposals']]
metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]


def step():
    losses = head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=True)
    for p in head.parameters():
        p.grad = None
    feat.grad = None
    sum(losses.values()).backward()


for _ in range(2):
    step()
fwd, bwd = collections.Counter(), collections.Counter()
mm, lb = ao.matmul_nt, ao._linear_bwd


def mm_log(A, B, bias=None, act=0, trans_a=False, trans_b=False):
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    N = B.shape[1] if trans_b else B.shape[0]
    fwd[(M, N, K, int(trans_a), int(trans_b))] += 1
    return mm(A, B, bias, act, trans_a, trans_b)


def lb_log(x2, W, y, g, need_x, need_w, need_b, dW_out=None, db_out=None):
    bwd[(g.shape[0], W.shape[0], W.shape[1], int(need_x), int(need_w), int(need_b), int(y is not None))] += 1
    return lb(x2, W, y, g, need_x, need_w, need_b, dW_out, db_out)


ao.matmul_nt, ao._linear_bwd = mm_log, lb_log
step()
ao.matmul_nt, ao._linear_bwd = mm, lb
torch.cuda.synchronize()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows, tot = [], 0.0
for (M, N, K, ta, tb), c in sorted(fwd.items()):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((K, N) if tb else (N, K), device=dev)
    us = timed(lambda: mm(A, B, None, 0, bool(ta), bool(tb)))
    rows.append(dict(kind='product', M=M, N=N, K=K, ta=ta, tb=tb, calls=c, us=round(us, 1), us_step=round(us * c, 1)))
    tot += us * c
for (M, N, K, nx, nw, nb, relu), c in sorted(bwd.items()):
    x, W, g = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
    us_all = timed(lambda: lb(x, W, None, g, bool(nx), bool(nw), bool(nb)))
    us_x = timed(lambda: mm(g, W, trans_b=True)) if nx else 0.0
    us_w = timed(lambda: mm(g, x, trans_a=True, trans_b=True)) if nw else 0.0
    us_b = timed(lambda: ao.colsum(g)) if nb else 0.0
    rows.append(dict(kind='linear_bwd', M=M, N=N, K=K, dx=nx, dW=nw, db=nb, relu=relu, calls=c, us=round(us_all, 1), us_dx=round(us_x, 1),
                     us_dW=round(us_w, 1), us_db=round(us_b, 1), us_step=round(us_all * c, 1)))
    tot += us_all * c
for r in rows:
    print(json.dumps(r))
print(json.dumps(dict(total_us_per_step=round(tot, 1))))
