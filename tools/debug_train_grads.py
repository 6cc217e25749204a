"""Debug helper: autograd route of forward_train vs the reference goldens (rows, losses, per-parameter gradient errors)."""
import sys
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from conftest import load_golden  # noqa: E402
from mv2d_amd import configs, registry, synthetic  # noqa: E402
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
    feat = torch.from_numpy(prob['feat']).to(DEV).requires_grad_(True)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    cap = {}
    from mv2d_amd import train
    orig = train.TrainDecoder.__call__

    def wrapped(self, *a, **k):
        r = orig(self, *a, **k)
        cap['rows'] = r
        cap['pad'] = a[5] if len(a) > 5 else k.get('pad', 0)
        return r
    train.TrainDecoder.__call__ = wrapped
    losses = head.forward_train([feat], metas, props, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])],
                                None, dn_noise=rnd)
    train.TrainDecoder.__call__ = orig
    cls, reg = cap['rows']
    pad = cap['pad']
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    print(name, 'pad', pad, 'rows cls', rel(cls[:, pad:].detach().cpu(), torch.from_numpy(gold[name + '.cls'])), 'reg',
          rel(reg[:, pad:].detach().cpu(), torch.from_numpy(gold[name + '.reg'])))
    if pad:
        print('   dn rows cls', rel(cls[:, :pad].detach().cpu(), torch.from_numpy(gold[name + '.dn_cls'])), 'reg',
              rel(reg[:, :pad].detach().cpu(), torch.from_numpy(gold[name + '.dn_reg'])))
    bad = [(k, float(losses[k].detach()), float(gold[f'{name}.loss.{k}'])) for k in sorted(losses)]
    print('   worst loss rel', max(abs(a - b) / max(abs(b), 1e-2) for _, a, b in bad))
    sum(losses.values()).backward()
    params = dict(head.named_parameters())
    rows = []
    for n, norm, proj in zip(gold[name + '.grad_names'], gold[name + '.grad_norm'], gold[name + '.grad_proj']):
        g = params[str(n)].grad
        if g is None:
            rows.append((9.9, str(n), 'NO GRAD', norm, 0, 0))
            continue
        g = g.double().cpu()
        gp = float((g.flatten() * torch.from_numpy(synthetic.grad_probe(str(n), g.numel())).double()).sum())
        rows.append((abs(float(g.norm()) - norm) / max(norm, 1e-9), str(n), float(g.norm()), norm, gp, proj))
    gf = feat.grad.double().cpu()
    print('   dfeat norm', float(gf.norm()), 'ref', float(gold[name + '.dfeat_norm']), 'proj', float((gf.flatten() * torch.from_numpy(synthetic.grad_probe('feat', gf.numel())).double()).sum()), 'ref', float(gold[name + '.dfeat_proj']))
    print('   per-view', [round(float(x), 5) for x in gf.flatten(1).norm(dim=1)], 'ref', [round(float(x), 5) for x in gold[name + '.dfeat_view_norms']])
    rows.sort(reverse=True)
    for r in rows[:5]:
        print('   ', r)
    print('    median norm err', sorted(x[0] for x in rows)[len(rows) // 2])
