"""Training-step timing of the head: forward_train + backward on one synthetic sample, both routes.

    python tools/bench_train.py [--problem cfg2_s|cfg3_t] [--gt 40] [--iters 20]

Prints one JSON line: ms per forward (engine route, no gradient), ms per forward+backward (autograd route) and its split.  The timed
region includes the host side of the step (Hungarian assignment on the host, the syncs it needs) — that is what a training loop pays.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--problem', default='cfg2_s')
    ap.add_argument('--gt', type=int, default=40)
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
    gtc = synthetic.make_train_gt(a.gt, 3)
    gt, labels = [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])]
    feat = torch.from_numpy(prob['feat']).to(dev).requires_grad_(True)      # the backbone's output: the step includes its gradient
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]

    def step(autograd, backward):
        losses = head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=autograd)
        if backward:
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

    opt = torch.optim.AdamW([p for p in head.parameters() if p.requires_grad], lr=1e-6, fused=True)      # (torch's single-kernel AdamW)

    def full_step():
        losses = step(True, True)
        torch.nn.utils.clip_grad_norm_(head.parameters(), 35.0)
        opt.step()
        return losses

    fwd_engine = timed(lambda: step(False, False), a.iters)
    fwd_autograd = timed(lambda: step(True, False), a.iters)
    fwd_bwd = timed(lambda: step(True, True), a.iters)
    full = timed(full_step, a.iters)
    eng = head.engine(feat.device, metas)
    infer = timed(lambda: eng.results(eng.run(feat.detach(), props, metas)), a.iters)
    R = sum(len(p) for p in prob['proposals'])
    print(json.dumps(dict(metric='head training step', problem=a.problem, kind=kind, queries=R, gt_boxes=a.gt,
                          denoising_queries=10 * a.gt if getattr(head, 'use_denoise', False) else 0,
                          inference_ms=round(infer, 3), forward_engine_route_ms=round(fwd_engine, 3),
                          forward_autograd_route_ms=round(fwd_autograd, 3), forward_backward_ms=round(fwd_bwd, 3), step_with_clip_and_adamw_ms=round(full, 3),
                          note='one sample per step, eager launches, Hungarian assignment on the host inside the timed region; clip_grad_norm_ + torch AdamW(fused=True)')))


if __name__ == '__main__':
    main()
