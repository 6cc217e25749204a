"""Golden vectors of the training targets / losses from the UNMODIFIED reference (build container only).

    python -B -m oracle.gen_golden_train      # writes tests/golden/train_loss.npz

Runs the reference's own HungarianAssigner3D.assign, CrossAttentionBoxHead.loss (-> loss_single -> get_targets -> _get_target_single)
and dn_loss_single on seeded synthetic head outputs and ground truth (mv2d_amd/synthetic.make_train_case; inputs are regenerated from
the seed by the tests, only outputs are stored).  The mmdet classes those functions call are restated in oracle/_stubs_train.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from mv2d_amd import configs, synthetic  # noqa: E402
from oracle import _stubs_train  # noqa: E402
from oracle.gen_golden import build_reference_head  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'train_loss.npz')


def main():
    (S_cls, T_cls), Assigner = _stubs_train.install('/root/reference')
    head = build_reference_head('S', S_cls, T_cls, synthetic.make_head_state(seed=0), 2)
    cfg = configs.roi_head_cfg_s()['bbox_head']
    bh = _stubs_train.arm_bbox_head(head.bbox_head, Assigner, configs.TRAIN_CFG_RCNN, cfg['loss_cls'], cfg['loss_bbox'])
    assert [float(x) for x in bh.code_weights] == cfg['code_weights']
    rec = {}
    for name, (R, G, seed) in synthetic.TRAIN_CASES.items():
        c = synthetic.make_train_case(R, G, seed)
        cls, box = torch.from_numpy(c['cls']), torch.from_numpy(c['box'])
        gt = _stubs_train.GtBoxes(torch.from_numpy(c['gt_bottom']))
        labels = torch.from_numpy(c['gt_labels'])
        match, lc, lb = [], [], []
        for l in range(cls.shape[0]):
            gtc = torch.cat((gt.gravity_center, gt.tensor[:, 3:]), 1)
            res = bh.assigner.assign(box[l], cls[l], gtc, labels)
            match.append((res.gt_inds - 1).numpy())
            out = bh.loss([gt], [labels], {'cls_scores': [cls[l]], 'bbox_preds': [box[l]]})
            lc.append(float(out['loss_cls']))
            lb.append(float(out['loss_bbox']))
        rec[name + '.match'] = np.stack(match).astype(np.int32)
        rec[name + '.loss'] = np.stack([lc, lb], 1).astype(np.float32)
        # denoising loss: the first n rows as denoising queries with given targets (labels == 10 are negatives)
        n = min(R, c['known_labels'].shape[0])
        for neg in (False, True):
            dn = [bh.dn_loss_single(cls[l][:n].clone(), box[l][:n].clone(), torch.from_numpy(c['known_bboxs'][:n]).clone(),
                                    torch.from_numpy(c['known_labels'][:n]), c['dn_num_tgt'], configs.POINT_CLOUD_RANGE, 0.6, neg_bbox_loss=neg)
                  for l in range(cls.shape[0])]
            rec[name + ('.dn_neg' if neg else '.dn')] = np.array([[float(a), float(b)] for a, b in dn], np.float32)
        print(name, rec[name + '.loss'][-1], rec[name + '.dn'][-1], 'matched', int((rec[name + '.match'][-1] >= 0).sum()))
    # --- prepare_for_dn: the reference's own method on the S head (training branch).  `.cuda()` (no GPU here) and `torch.rand_like`
    # (the noise, regenerated from the seed by the tests) are patched for the call; everything else runs unmodified.
    head.train()
    head.use_denoise = True
    for name, (R, G, seed, scalar, nscale, split) in synthetic.DN_CASES.items():
        c = synthetic.make_train_case(R, G, seed)
        rnd = torch.from_numpy(synthetic.make_dn_noise(G * scalar, seed))
        head.denoise_scalar, head.denoise_noise_scale, head.denoise_split, head.denoise_noise_trans = scalar, nscale, split, 0.0
        meta = dict(gt_bboxes_3d=_stubs_train.GtBoxes(torch.from_numpy(c['gt_bottom'])), gt_labels_3d=torch.from_numpy(c['gt_labels']))
        ref = torch.from_numpy(synthetic.make_dn_noise(R, seed + 100))
        cuda, rand_like = torch.Tensor.cuda, torch.rand_like
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.rand_like = lambda t, *a, **k: rnd.to(t.dtype)
        try:
            padded, attn_mask, md = head.prepare_for_dn(1, ref, [meta], R)
        finally:
            torch.Tensor.cuda, torch.rand_like = cuda, rand_like
        rec[name + '.padded'] = padded.numpy()
        rec[name + '.attn_mask'] = np.packbits(attn_mask.numpy())
        rec[name + '.known_labels'] = md['known_lbs_bboxes'][0].numpy()
        rec[name + '.known_bboxs'] = md['known_lbs_bboxes'][1].numpy()
        rec[name + '.map_known_indice'] = md['map_known_indice'].numpy()
        rec[name + '.known_indice'] = md['known_indice'].numpy()
        rec[name + '.pad_size'] = np.int64(md['pad_size'])
        print(name, padded.shape, attn_mask.shape, 'negatives', int((md['known_lbs_bboxes'][0] == 10).sum()))
    # --- forward_train of the whole head (forward only): the reference's losses on seeded problems.  Dropout is switched off (p = 0: the
    # golden must not depend on torch's dropout stream); `.cuda()` / `torch.rand_like` patched as above.
    import torch.nn as nn
    for name, (prob_name, kind, G, seed) in synthetic.FWD_TRAIN_CASES.items():
        prob = synthetic.make_problem(prob_name, seed=0)
        with_dn = kind.endswith('+DN')
        kind = kind[0]
        h = build_reference_head(kind, S_cls, T_cls, synthetic.make_head_state(seed=0), prob['views_per_frame'], train_cfg=configs.TRAIN_CFG_RCNN)
        if with_dn:
            h.use_denoise = True
        cfgk = (configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t())['bbox_head']
        _stubs_train.arm_bbox_head(h.bbox_head, Assigner, configs.TRAIN_CFG_RCNN, cfgk['loss_cls'], cfgk['loss_bbox'])
        h.train()
        for m in h.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
            if isinstance(m, nn.MultiheadAttention):
                m.dropout = 0.0                      # the attention-probability dropout (attn_drop) is a float attribute, not a module
        gtc = synthetic.make_train_gt(G, seed)
        gt = _stubs_train.GtBoxes(torch.from_numpy(gtc['gt_bottom']))
        labels = torch.from_numpy(gtc['gt_labels'])
        rnd = torch.from_numpy(synthetic.make_dn_noise(G * 10, seed))
        metas = [dict(m, box_type_3d=(lambda b, d: b)) for m in prob['img_metas']]
        props = [torch.from_numpy(p) for p in prob['proposals']]
        captured = {}
        orig = h._bbox_forward_train

        def wrapped(*a, **k):
            r = orig(*a, **k)
            captured['res'] = r
            return r
        h._bbox_forward_train = wrapped
        cuda, rand_like = torch.Tensor.cuda, torch.rand_like
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.rand_like = lambda t, *a, **k: rnd.to(t.dtype)
        try:
            with torch.no_grad():
                losses = h.forward_train([torch.from_numpy(prob['feat'])], metas, props, None, None, None, None, [gt], [labels], None)
        finally:
            torch.Tensor.cuda, torch.rand_like = cuda, rand_like
        for k, v in losses.items():
            rec[f'{name}.loss.{k}'] = np.float32(float(v))
        # the same forward with autograd: gradients of sum(losses) w.r.t. the decoder / branches / query_embedding parameters, stored as
        # (L2 norm, projection on a seeded probe vector) per parameter
        for q in h.parameters():
            q.grad = None
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.rand_like = lambda t, *a, **k: rnd.to(t.dtype)
        try:
            xg = torch.from_numpy(prob['feat']).clone().requires_grad_(True)
            losses_g = h.forward_train([xg], metas, props, None, None, None, None, [gt], [labels], None)
        finally:
            torch.Tensor.cuda, torch.rand_like = cuda, rand_like
        sum(losses_g.values()).backward()
        names, norms, projs = [], [], []
        for pn, q in h.named_parameters():
            if q.grad is not None:
                names.append(pn)
                norms.append(float(q.grad.double().norm()))
                projs.append(float((q.grad.double().flatten() * torch.from_numpy(synthetic.grad_probe(pn, q.numel())).double()).sum()))
        rec[f'{name}.dfeat_norm'] = np.float64(float(xg.grad.double().norm()))
        rec[f'{name}.dfeat_proj'] = np.float64(float((xg.grad.double().flatten() * torch.from_numpy(synthetic.grad_probe('feat', xg.numel())).double()).sum()))
        rec[f'{name}.dfeat_view_norms'] = xg.grad.double().flatten(1).norm(dim=1).numpy()
        rec[f'{name}.grad_names'] = np.array(names)
        rec[f'{name}.grad_norm'] = np.array(norms)
        rec[f'{name}.grad_proj'] = np.array(projs)
        print('   grads:', len(names), 'params, max norm', max(norms))
        res = captured['res']
        rec[f'{name}.cls'] = torch.stack(res['pred']['cls_scores']).detach().numpy()
        rec[f'{name}.reg'] = torch.stack(res['pred']['bbox_preds']).detach().numpy()
        gtc9 = torch.cat((gt.gravity_center, gt.tensor[:, 3:]), 1)
        rec[f'{name}.match'] = np.stack([(h.bbox_head.assigner.assign(b.detach(), c.detach(), gtc9, labels).gt_inds - 1).numpy()
                                         for c, b in zip(res['pred']['cls_scores'], res['pred']['bbox_preds'])]).astype(np.int32)
        md = res.get('dn_mask_dict')
        if md:
            rec[f'{name}.dn_cls'] = md['output_known_lbs_bboxes'][0][:, 0].detach().numpy()
            rec[f'{name}.dn_reg'] = md['output_known_lbs_bboxes'][1][:, 0].detach().numpy()
        print(name, {k: round(float(v), 5) for k, v in list(losses.items())[-4:]}, 'rows', rec[f'{name}.cls'].shape)
    np.savez_compressed(OUT, **rec)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
