"""A/B of the training step: decoder issued from C (csrc/train_decoder.hip) vs the per-operator autograd graph -- losses, gradients, time.

    python tools/train_fused_ab.py [--problem cfg2_s] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import configs, registry, synthetic  # noqa: E402
import mv2d_amd.plugin  # noqa: F401,E402

ap = argparse.ArgumentParser()
ap.add_argument('--problem', default='cfg2_s')
ap.add_argument('--iters', type=int, default=20)
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


def step(fused):
    if getattr(head, '_train_decoder', None) is not None:
        head._train_decoder.fused = fused
    else:
        os.environ['MV2D_TRAIN_FUSED'] = '1' if fused else '0'
    losses = head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=True)
    for p in head.parameters():
        p.grad = None
    feat.grad = None
    sum(losses.values()).backward()
    return losses


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


head.eval()                                     # no dropout: the two routes compute the same function
res = {}
for fused in (False, True):
    l = step(fused)
    torch.cuda.synchronize()
    res[fused] = ({k: float(v) for k, v in l.items()}, {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}, feat.grad.clone())
dl = max(abs(res[True][0][k] - res[False][0][k]) / max(abs(res[False][0][k]), 1e-6) for k in res[False][0])
worst, missing = (0.0, ''), [n for n in res[False][1] if n not in res[True][1]]
for n, g in res[False][1].items():
    if n in res[True][1]:
        e = float((res[True][1][n] - g).abs().max() / g.abs().max().clamp(min=1e-12))
        worst = max(worst, (e, n))
ef = float((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max())
out = dict(problem=a.problem, loss_rel=dl, grad_rel_worst=worst[0], grad_worst_name=worst[1], feat_grad_rel=ef, missing=missing,
           n_grads=len(res[False][1]))
head.train()
out['ms_operator_graph'] = round(timed(lambda: step(False), a.iters), 3)
out['ms_fused_decoder'] = round(timed(lambda: step(True), a.iters), 3)
print(json.dumps(out))
