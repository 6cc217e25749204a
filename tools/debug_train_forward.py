"""Debug helper: training forward vs the goldens of the reference's forward_train, stage by stage (run on the GPU box)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from conftest import load_golden  # noqa: E402
from mv2d_amd import configs, registry, synthetic, train  # noqa: E402
import mv2d_amd.plugin  # noqa: F401,E402

DEV = 'cuda'
gold = load_golden('train_loss')
for name, (prob_name, kind, G, seed) in synthetic.FWD_TRAIN_CASES.items():
    prob = synthetic.make_problem(prob_name, seed=0)
    with_dn, kind = kind.endswith('+DN'), kind[0]
    cfg = configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t()
    if kind == 'T':
        cfg['num_views'] = prob['views_per_frame']
    if with_dn:
        cfg['use_denoise'] = True
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    head = head.to(DEV)
    gtc = synthetic.make_train_gt(G, seed)
    rnd = torch.from_numpy(synthetic.make_dn_noise(G * 10, seed)).to(DEV)
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    eng = head.engine(feat.device, metas)
    out = eng.run(feat, [p[:, :6] for p in props], metas)
    R = out['R']
    gt = torch.from_numpy(gtc['gt']).to(DEV)
    labels = torch.from_numpy(gtc['gt_labels']).to(DEV)
    if getattr(head, 'use_denoise', False):
        padded, _, md = train.prepare_for_dn(out['ws']['ref'][:R], gt, labels, head.denoise_scalar, head.denoise_noise_scale,
                                             head.denoise_noise_trans, head.denoise_split, 10, list(head.pc_range), rnd=rnd, dense_mask=False)
        pad = md['pad_size']
        cls, reg = eng.train_forward(out, padded[0, :pad], md['dn_single'])
    else:
        pad = 0
        cls, reg = eng.train_forward(out)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    gc, gr = torch.from_numpy(gold[name + '.cls']), torch.from_numpy(gold[name + '.reg'])
    print(name, 'R', R, 'pad', pad, 'matched rows: cls', rel(cls[:, pad:].cpu(), gc), 'reg', rel(reg[:, pad:].cpu(), gr),
          '| inference rows cls', rel(out['ws']['cls'][:, :R].cpu(), gc))
    if pad:
        dc, dr = torch.from_numpy(gold[name + '.dn_cls']), torch.from_numpy(gold[name + '.dn_reg'])
        print('   dn rows: cls', rel(cls[:, :pad].cpu(), dc), 'reg', rel(reg[:, :pad].cpu(), dr), 'per layer cls',
              [round(rel(cls[l, :pad].cpu(), dc[l]), 5) for l in range(cls.shape[0])])
        d = (cls[0, :pad].cpu() - dc[0]).abs().max(1).values
        print('   layer-0 dn row errors (first 12):', [round(float(x), 4) for x in d[:12]], 'worst row', int(d.argmax()))
    losses = head.forward_train([feat], metas, props, None, None, None, None, [gt], [labels], None, dn_noise=rnd)
    for k in sorted(losses):
        w = float(gold[f'{name}.loss.{k}'])
        print(f'   {k:18s} {float(losses[k]):.6f} ref {w:.6f} rel {abs(float(losses[k]) - w) / max(abs(w), 1e-9):.2e}')
