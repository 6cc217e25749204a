"""Only full training steps of the head (forward_train + backward + clip + AdamW), for rocprofv3 kernel statistics per step.

    python tools/train_prof_step.py [--problem cfg2_s] [--iters 20] [--warm 3]
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
ap.add_argument('--gt', type=int, default=40)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--warm', type=int, default=3)
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
gtc = synthetic.make_train_gt(a.gt, 3)
gt, labels = [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])]
feat = torch.from_numpy(prob['feat']).to(dev).requires_grad_(True)
props = [torch.from_numpy(p) for p in prob['proposals']]
metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
opt = torch.optim.AdamW([p for p in head.parameters() if p.requires_grad], lr=1e-6, fused=True)      # (torch's single-kernel AdamW)


def full_step():
    losses = head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=True)
    for p in head.parameters():
        p.grad = None
    feat.grad = None
    sum(losses.values()).backward()
    torch.nn.utils.clip_grad_norm_(head.parameters(), 35.0)
    opt.step()


for _ in range(a.warm):
    full_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    full_step()
torch.cuda.synchronize()
print(json.dumps(dict(problem=a.problem, steps=a.iters + a.warm, ms_per_step=round((time.perf_counter() - t0) / a.iters * 1e3, 3))))
