"""The decoder issued from C against the per-operator autograd graph on the SAME inputs and the same upstream gradient (no matching in between).

    python tools/train_fused_unit.py [--problem cfg2_s]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import configs, registry, synthetic, train  # noqa: E402
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
    cfg['use_denoise'] = False
head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
head = head.to(dev).eval()
gtc = synthetic.make_train_gt(40, 3)
gt, labels = [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])]
feat = torch.from_numpy(prob['feat']).to(dev).requires_grad_(True)
props = [torch.from_numpy(p) for p in prob['proposals']]
metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]

captured = {}
orig = train.TrainDecoder.__call__


def rec(self, *args, **kw):
    captured['args'], captured['kw'] = args, kw
    return orig(self, *args, **kw)


train.TrainDecoder.__call__ = rec
head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=True)
train.TrainDecoder.__call__ = orig
dec = head._train_decoder
ref, key_in, val_in, row_ptr, col = captured['args'][:5]
rest = captured['args'][5:]
torch.manual_seed(0)
res = {}
g_cls = g_reg = None
for tag, fused in (('a', False), ('b', False), ('c', True), ('d', True)):
    dec.fused = fused
    for p in head.parameters():
        p.grad = None
    r_, k_, v_ = ref.detach().clone().requires_grad_(True), key_in.detach().clone().requires_grad_(True), val_in.detach().clone().requires_grad_(True)
    all_cls, all_reg = dec(r_, k_, v_, row_ptr, col, *rest, **captured['kw'])
    if g_cls is None:
        g_cls, g_reg = torch.randn_like(all_cls), torch.randn_like(all_reg)
    ((all_cls * g_cls).sum() + (all_reg * g_reg).sum()).backward()
    torch.cuda.synchronize()
    res[tag] = dict(cls=all_cls.detach(), reg=all_reg.detach(), ref=r_.grad, key=k_.grad, val=v_.grad,
                      **{n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})
rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp(min=1e-20))  # noqa: E731
for x, y in (('b', 'a'), ('d', 'c'), ('c', 'a')):
    errs = {n: rel(res[x][n], res[y][n]) for n in res[y] if n in res[x]}
    top = sorted(errs.items(), key=lambda t: -t[1])[:40]
    print(json.dumps(dict(pair=x + y, problem=a.problem, T=int(ref.shape[0]), S=int(key_in.shape[0]), missing=[n for n in res[y] if n not in res[x]],
                          cls=errs['cls'], reg=errs['reg'], key=errs['key'], val=errs['val'], ref=errs['ref'], worst=top)))
