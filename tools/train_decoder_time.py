"""GPU wall time of the C-issued decoder + branches alone (forward, backward), with and without the side streams (MV2D_TD_SERIAL=1).

    python tools/train_decoder_time.py [--problem cfg2_s]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import configs, registry, synthetic, train  # noqa: E402
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
    cfg['use_denoise'] = False
head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
head = head.to(dev)
gtc = synthetic.make_train_gt(40, 3)
gt, labels = [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])]
feat = torch.from_numpy(prob['feat']).to(dev).requires_grad_(True)
props = [torch.from_numpy(p) for p in prob['proposals']]
metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
got, orig = {}, train.TrainDecoder.__call__


def rec(self, *args, **kw):
    got['a'], got['k'] = args, kw
    return orig(self, *args, **kw)


train.TrainDecoder.__call__ = rec
head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=True)
train.TrainDecoder.__call__ = orig
dec = head._train_decoder
args = [t.detach().clone().requires_grad_(True) for t in got['a'][:3]] + list(got['a'][3:])
g = None
tf, tb, hf, hb = [], [], [], []
for it in range(a.iters + 3):
    for p in head.parameters():
        p.grad = None
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    e[0].record()
    all_cls, all_reg = dec(*args, **got['k'])
    e[1].record()
    t1 = time.perf_counter()
    if g is None:
        g = (torch.randn_like(all_cls), torch.randn_like(all_reg))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    e[1].record()
    torch.autograd.backward([all_cls, all_reg], g)
    e[2].record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    if it >= 3:
        hf.append((t1 - t0) * 1e3); hb.append((t3 - t2) * 1e3)
        tb.append(e[1].elapsed_time(e[2]))
for it in range(a.iters):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    with torch.no_grad():
        pass
    all_cls, all_reg = dec(*args, **got['k'])
    e[1].record()
    torch.cuda.synchronize()
    tf.append(e[0].elapsed_time(e[1]))
med = lambda v: round(sorted(v)[len(v) // 2], 3)  # noqa: E731
print(json.dumps(dict(problem=a.problem, serial=os.environ.get('MV2D_TD_SERIAL', '0'), fused=os.environ.get('MV2D_TRAIN_FUSED', '1'),
                      forward_gpu_ms=med(tf), backward_gpu_ms=med(tb), forward_host_ms=med(hf), backward_host_ms=med(hb))))
