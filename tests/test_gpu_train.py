"""Training targets / losses of the box head on the GPU: HIP kernels through the C-ABI vs the oracle restatement and the
goldens recorded from the reference's own assigner / loss code (tests/golden/train_loss.npz, oracle/gen_golden_train.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from mv2d_amd import configs, synthetic

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _case(name):
    R, G, seed = synthetic.TRAIN_CASES[name]
    c = synthetic.make_train_case(R, G, seed)
    return c, {k: torch.from_numpy(v).to(DEV) for k, v in c.items() if isinstance(v, np.ndarray)}


def _dropout_off(head):
    """The training goldens were produced with dropout switched off on the reference (nn.Dropout.p = 0, nn.MultiheadAttention.dropout = 0,
    oracle/gen_golden_train.py); the head under test gets the same treatment."""
    for m in head.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    return head


def _head_loss():
    from mv2d_amd import train
    cfg = configs.roi_head_cfg_s()['bbox_head']
    return train.HeadLoss(num_classes=10, loss_cls=cfg['loss_cls'], loss_bbox=cfg['loss_bbox'], code_weights=cfg['code_weights'],
                          train_cfg=configs.TRAIN_CFG_RCNN, device=DEV)


@pytest.mark.parametrize('name', list(synthetic.TRAIN_CASES))
def test_match_cost_and_assignment(name):
    from oracle import mv2d_oracle as O
    c, d = _case(name)
    hl = _head_loss()
    if c['gt'].shape[0] and c['cls'].shape[1]:
        cost = hl.assigner.cost(d['box'], d['cls'], d['gt'], d['gt_labels']).cpu()
        for l in range(cost.shape[0]):
            want = O.match_cost(torch.from_numpy(c['box'][l]), torch.from_numpy(c['cls'][l]), torch.from_numpy(c['gt']),
                                torch.from_numpy(c['gt_labels']))
            assert torch.allclose(cost[l], want, rtol=2e-5, atol=2e-5), (l, float((cost[l] - want).abs().max()))
    match = hl.assigner.assign(d['box'], d['cls'], d['gt'], d['gt_labels'])
    assert match.dtype == torch.int32 and np.array_equal(match.cpu().numpy(), load_golden('train_loss')[name + '.match'])
    one = hl.assigner.assign(d['box'][2], d['cls'][2], d['gt'], d['gt_labels'])          # the reference's per-layer call shape
    assert torch.equal(one, match[2])


def test_match_cost_nan_to_num():
    """nan / +-inf costs become 100 / 100 / -100 (hungarian_assigner_3d.py:136)."""
    from mv2d_amd import ops
    cls = torch.zeros(1, 3, 10, device=DEV)
    box = torch.zeros(1, 3, 10, device=DEV)
    box[0, 0, 0] = float('nan')
    box[0, 1, 1] = float('inf')
    gt = torch.tensor([[0., 0, 0, 1, 1, 1, 0, 0, 0]], device=DEV)
    cost = ops.match_cost(cls, box, gt, torch.zeros(1, dtype=torch.int32, device=DEV)).cpu()
    assert cost[0, 0, 0] == 100.0 and cost[0, 1, 0] == 100.0 and torch.isfinite(cost[0, 2, 0])


@pytest.mark.parametrize('name', list(synthetic.TRAIN_CASES))
def test_head_loss_matches_reference_and_autograd(name):
    from oracle import mv2d_oracle as O
    c, d = _case(name)
    gold = load_golden('train_loss')
    hl = _head_loss()
    cls = d['cls'].clone().requires_grad_(True)
    box = d['box'].clone().requires_grad_(True)
    losses, total, match = hl.loss(cls, box, d['gt'], d['gt_labels'])
    L = cls.shape[0]
    got = torch.stack([torch.stack([losses[f'l{l}.loss_cls'], losses[f'l{l}.loss_bbox']]) for l in range(L)]).detach().cpu().numpy()
    np.testing.assert_allclose(got, gold[name + '.loss'] * 0.1, rtol=1e-5, atol=1e-7)      # stage_loss_weights = 0.1
    total.backward()
    # gradient: torch autograd (fp64) through the oracle restatement with the same assignment
    ocls = torch.from_numpy(c['cls']).double().requires_grad_(True)
    obox = torch.from_numpy(c['box']).double().requires_grad_(True)
    tot = 0
    for l in range(L):
        lc, lb, _ = O.loss_single(ocls[l], obox[l], torch.from_numpy(c['gt']).double(), torch.from_numpy(c['gt_labels']),
                                  match=match[l].cpu().long())
        tot = tot + 0.1 * (lc + lb)
    tot.backward()
    assert abs(float(total.detach()) - float(tot.detach())) <= 1e-5 * max(1.0, abs(float(tot.detach())))
    for g, w in ((cls.grad, ocls.grad), (box.grad, obox.grad)):
        w = w.float()
        assert float((g.cpu() - w).abs().max()) <= 1e-5 * max(float(w.abs().max()), 1e-6), name


@pytest.mark.parametrize('name', ['small', 'mid'])
def test_bbox_head_loss_interface_matches_reference(name):
    """CrossAttentionBoxHead.loss / dn_loss_single (the reference's per-layer interface, cross_attention_head.py:436-463, 476-538) against the
    reference's own values in the golden (layer by layer, unit stage weight) and with gradients flowing."""
    c, d = _case(name)
    gold = load_golden('train_loss')
    from mv2d_amd import registry
    head = registry.build_head(configs.roi_head_cfg_s(), train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN).to(DEV)
    bh = head.bbox_head
    L = d['cls'].shape[0]
    for l in range(L):
        cls = d['cls'][l:l + 1].clone().requires_grad_(True)
        box = d['box'][l:l + 1].clone().requires_grad_(True)
        out = bh.loss([d['gt']], [d['gt_labels']], dict(cls_scores=cls, bbox_preds=box))
        got = np.array([float(out['loss_cls']), float(out['loss_bbox'])])
        np.testing.assert_allclose(got, gold[name + '.loss'][l], rtol=1e-5, atol=1e-7)
        (out['loss_cls'] + out['loss_bbox']).backward()
        assert float(cls.grad.abs().sum()) > 0 and float(box.grad.abs().sum()) > 0
    n = c['known_labels'].shape[0]
    lc, lb = bh.dn_loss_single(d['cls'][0, :n].contiguous(), d['box'][0, :n].contiguous(), d['known_bboxs'], d['known_labels'], c['dn_num_tgt'],
                               split=0.6)
    np.testing.assert_allclose(np.array([float(lc), float(lb)]), gold[name + '.dn'][0], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('neg', [False, True])
@pytest.mark.parametrize('name', ['small', 'mid', 'few_queries'])
def test_dn_loss_matches_reference_and_autograd(name, neg):
    from oracle import mv2d_oracle as O
    c, d = _case(name)
    gold = load_golden('train_loss')[name + ('.dn_neg' if neg else '.dn')]
    hl = _head_loss()
    n = c['known_labels'].shape[0]
    cls = d['cls'][:, :n].contiguous().requires_grad_(True)
    box = d['box'][:, :n].contiguous().requires_grad_(True)
    losses, total = hl.dn_loss(cls, box, d['known_bboxs'], d['known_labels'], c['dn_num_tgt'], 0.6, neg_bbox_loss=neg)
    L = cls.shape[0]
    got = torch.stack([torch.stack([losses[f'l{l}.dn_loss_cls'], losses[f'l{l}.dn_loss_bbox']]) for l in range(L)]).detach().cpu().numpy()
    np.testing.assert_allclose(got, gold * 0.1, rtol=1e-5, atol=1e-7)
    total.backward()
    ocls = torch.from_numpy(c['cls'][:, :n]).double().requires_grad_(True)
    obox = torch.from_numpy(c['box'][:, :n]).double().requires_grad_(True)
    tot = 0
    for l in range(L):
        lc, lb = O.dn_loss_single(ocls[l], obox[l], torch.from_numpy(c['known_bboxs']).double(), torch.from_numpy(c['known_labels']),
                                  c['dn_num_tgt'], 0.6, neg_bbox_loss=neg)
        tot = tot + 0.1 * (lc + lb)
    tot.backward()
    for g, w in ((cls.grad, ocls.grad), (box.grad, obox.grad)):
        w = w.float()
        assert float((g.cpu() - w).abs().max()) <= 1e-5 * max(float(w.abs().max()), 1e-6)


def test_set_loss_is_deterministic_and_rejects_bad_factors():
    from mv2d_amd import _lib, ops
    c, d = _case('mid')
    hl = _head_loss()
    match = hl.assigner.assign(d['box'], d['cls'], d['gt'], d['gt_labels'])
    lab = d['gt_labels'].to(torch.int32)
    a = ops.set_loss(d['cls'], d['box'], match, d['gt'], lab, hl.code_weights, None, 45.0, 45.0)
    b = ops.set_loss(d['cls'], d['box'], match, d['gt'], lab, hl.code_weights, None, 45.0, 45.0)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    with pytest.raises(_lib.Mv2dHipError):
        ops.set_loss(d['cls'], d['box'], match, d['gt'], lab, hl.code_weights, None, 0.0, 45.0)


@pytest.mark.parametrize('name', list(synthetic.DN_CASES))
def test_prepare_for_dn_matches_reference(name):
    """Host prepare_for_dn (HIP kernel for the noised queries) vs the goldens of the reference's own method."""
    from mv2d_amd import train
    gold = load_golden('train_loss')
    R, G, seed, scalar, nscale, split = synthetic.DN_CASES[name]
    c = synthetic.make_train_case(R, G, seed)
    rnd = torch.from_numpy(synthetic.make_dn_noise(G * scalar, seed)).to(DEV)
    ref = torch.from_numpy(synthetic.make_dn_noise(R, seed + 100)).to(DEV)
    padded, attn_mask, md = train.prepare_for_dn(ref, torch.from_numpy(c['gt']), torch.from_numpy(c['gt_labels']), scalar, nscale, 0.0, split,
                                                 rnd=rnd)
    assert md['pad_size'] == int(gold[name + '.pad_size'])
    want = gold[name + '.padded']
    assert padded.shape == want.shape
    assert float((padded.cpu() - torch.from_numpy(want)).abs().max()) <= 2e-7 if want.size else True    # fp32 ulp (fused multiply-add)
    assert np.array_equal(np.packbits(attn_mask.cpu().numpy()), gold[name + '.attn_mask'])
    kl, kb = md['known_lbs_bboxes']
    assert np.array_equal(kl.cpu().numpy(), gold[name + '.known_labels']) and np.array_equal(kb.cpu().numpy(), gold[name + '.known_bboxs'], equal_nan=True)
    assert np.array_equal(md['map_known_indice'].cpu().numpy(), gold[name + '.map_known_indice'])
    assert np.array_equal(md['known_indice'].cpu().numpy(), gold[name + '.known_indice'])
    # drawn on the device when no noise is given: same structure, labels either the ground truth's or the negative class
    p2, _, md2 = train.prepare_for_dn(ref, torch.from_numpy(c['gt']), torch.from_numpy(c['gt_labels']), scalar, nscale, 0.0, split, dense_mask=False)
    assert p2.shape == padded.shape and torch.equal(p2[0, md['pad_size']:], ref)
    if G:
        l2 = md2['known_lbs_bboxes'][0].cpu()
        assert bool(((l2 == torch.from_numpy(c['gt_labels']).repeat(scalar)) | (l2 == 10)).all())


@pytest.mark.parametrize('R,pad,single', [(82, 70, 7), (150, 110, 11), (33, 0, 1), (64, 64, 16), (300, 290, 29), (17, 3, 3)])
def test_self_attn_with_denoising_mask(R, pad, single):
    """ops.self_attn_dn evaluates prepare_for_dn's attn_mask from (pad_size, single_pad): equal to dense masked attention in fp64."""
    import math
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    g = torch.Generator().manual_seed(R + pad)
    qkv = torch.randn(R, 768, generator=g).to(DEV)
    out = ops.self_attn_dn(qkv, pad, single)
    _, mask, _, _, _ = O.prepare_for_dn(torch.zeros(R - pad, 3), torch.ones(single if pad else 0, 9), torch.zeros(single if pad else 0).long(),
                                        torch.zeros(pad, 3), scalar=pad // single if pad else 0, noise_scale=0.0)
    q, k, v = [t.double().cpu().view(R, 8, 32).transpose(0, 1) for t in qkv.split(256, 1)]
    logits = (q @ k.transpose(1, 2) / math.sqrt(32)).masked_fill(mask[None], float('-inf'))
    want = (torch.softmax(logits, -1) @ v).transpose(0, 1).reshape(R, 256)
    assert float((out.cpu().double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    out3 = ops.self_attn_dn(qkv, pad, single, impl='x3')    # the split-precision kernel with the same mask (unit-variance q, k: logits ~ +-15)
    assert float((out3.cpu().double() - want).abs().max()) <= 6e-5 * float(want.abs().max())
    if pad == 0:
        assert torch.equal(out, ops.self_attn(qkv, impl='f32'))          # no denoising rows: the inference kernel's arithmetic
        assert torch.equal(out3, ops.self_attn(qkv, impl='x3'))


@pytest.mark.parametrize('name', list(synthetic.FWD_TRAIN_CASES))
def test_forward_train_losses_match_reference(name):
    """The whole training forward (engine + denoising rows + assignment + losses) vs the reference's forward_train (dropout off)."""
    from mv2d_amd import registry
    import mv2d_amd.plugin  # noqa: F401
    gold = load_golden('train_loss')
    prob_name, kind, G, seed = synthetic.FWD_TRAIN_CASES[name]
    prob = synthetic.make_problem(prob_name, seed=0)
    with_dn, kind = kind.endswith('+DN'), kind[0]
    cfg = configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t()
    if kind == 'T':
        cfg['num_views'] = prob['views_per_frame']
    if with_dn:
        cfg['use_denoise'] = True
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    head = _dropout_off(head.to(DEV))
    gtc = synthetic.make_train_gt(G, seed)
    rnd = torch.from_numpy(synthetic.make_dn_noise(G * 10, seed)).to(DEV)
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    losses = head.forward_train([feat], metas, props, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])],
                                None, dn_noise=rnd, autograd=False)
    want = {k[len(name) + 6:]: float(v) for k, v in gold.items() if k.startswith(name + '.loss.')}
    assert set(losses) == set(want), (sorted(losses), sorted(want))
    for k, v in want.items():
        assert abs(float(losses[k].detach()) - v) <= 2e-3 * max(abs(v), 1e-2), (k, float(losses[k].detach()), v)
    # the rows behind the losses: [denoising queries | the sample's queries] of every layer vs the reference's training forward
    from mv2d_amd import train
    eng = head.engine(feat.device, metas)
    out = eng.run(feat, [p[:, :6] for p in props], metas)
    R, pad = out['R'], 0
    if kind == 'T' or with_dn:
        gt, labels = torch.from_numpy(gtc['gt']).to(DEV), torch.from_numpy(gtc['gt_labels']).to(DEV)
        padded, _, md = train.prepare_for_dn(out['ws']['ref'][:R], gt, labels, head.denoise_scalar, head.denoise_noise_scale,
                                             head.denoise_noise_trans, head.denoise_split, 10, list(head.pc_range), rnd=rnd, dense_mask=False)
        pad = md['pad_size']
        cls, reg = eng.train_forward(out, padded[0, :pad], md['dn_single'])
        assert pad == 10 * G
        for got, key in ((cls[:, :pad], '.dn_cls'), (reg[:, :pad], '.dn_reg')):
            w = torch.from_numpy(gold[name + key])
            assert float((got.cpu() - w).abs().max()) <= 2e-3 * float(w.abs().max()), key
    else:
        cls, reg = eng.train_forward(out)
    for got, key in ((cls[:, pad:], '.cls'), (reg[:, pad:], '.reg')):
        w = torch.from_numpy(gold[name + key])
        assert float((got.cpu() - w).abs().max()) <= 2e-3 * float(w.abs().max()), key
    assert torch.equal(cls[:, pad:], out['ws']['cls'][:, :R]) if pad == 0 else True


@pytest.mark.parametrize('name', list(synthetic.FWD_TRAIN_CASES))
def test_forward_train_gradients_match_reference(name):
    """forward_train through the autograd route (torch linears around the HIP attention forward / backward and the HIP loss kernel):
    losses and the gradients of every parameter of the head (decoder, branches, query embedding, query generator, PE) vs the reference's autograd (goldens store the norm and a
    seeded projection of each gradient), and the gradient w.r.t. the input feature map."""
    from mv2d_amd import registry
    import mv2d_amd.plugin  # noqa: F401
    gold = load_golden('train_loss')
    prob_name, kind, G, seed = synthetic.FWD_TRAIN_CASES[name]
    prob = synthetic.make_problem(prob_name, seed=0)
    with_dn, kind = kind.endswith('+DN'), kind[0]
    cfg = configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t()
    if kind == 'T':
        cfg['num_views'] = prob['views_per_frame']
    if with_dn:
        cfg['use_denoise'] = True
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    head = _dropout_off(head.to(DEV))
    gtc = synthetic.make_train_gt(G, seed)
    rnd = torch.from_numpy(synthetic.make_dn_noise(G * 10, seed)).to(DEV)
    feat = torch.from_numpy(prob['feat']).to(DEV).requires_grad_(True)      # the backbone's output: it gets a gradient too
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    # the Hungarian assignment is discontinuous: a near-tie can flip under the 5e-4 difference of the rows.  The comparison of losses and
    # gradients therefore uses the reference's assignment; the head's own assignment is checked separately.
    hl = head._head_loss(torch.device(DEV, torch.cuda.current_device()))
    own, want_match = {}, torch.from_numpy(gold[name + '.match']).to(DEV)
    orig_assign = hl.assigner.assign

    def assign(*a, **k):
        own['match'] = orig_assign(*a, **k)
        own['cost'] = hl.assigner.cost(a[0].contiguous(), a[1].contiguous(), a[2].contiguous(), a[3])
        return want_match
    hl.assigner.assign = assign
    losses = head.forward_train([feat], metas, props, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])],
                                None, dn_noise=rnd)
    hl.assigner.assign = orig_assign
    # ... so the head's own assignment is judged by its cost: as good as the reference's on the head's own cost matrix (a near-tie reroutes
    # a whole chain of pairs at no cost)
    def total(m):
        c = torch.gather(own['cost'], 2, m.clamp(min=0).long()[..., None])[..., 0]
        return (c * (m >= 0)).sum(1)
    c_own, c_ref = total(own['match']), total(want_match)
    assert bool((c_own <= c_ref + 1e-4 * c_ref.abs()).all()) and bool(((c_ref - c_own).abs() <= 2e-3 * c_ref.abs()).all()), (c_own, c_ref)
    for k in losses:
        v = float(gold[f'{name}.loss.{k}'])
        assert abs(float(losses[k].detach()) - v) <= 2e-3 * max(abs(v), 1e-2), (k, float(losses[k].detach()), v)
    sum(losses.values()).backward()
    params = dict(head.named_parameters())
    names = [str(n) for n in gold[name + '.grad_names']]
    assert len(names) == 232                                # every parameter of the head: decoder, branches, query embedding, query generator, PE
    worst, errs, top = (0.0, None), [], float(gold[name + '.grad_norm'].max())
    for n, norm, proj in zip(names, gold[name + '.grad_norm'], gold[name + '.grad_proj']):
        g = params[n].grad
        assert g is not None, n
        if norm < 1e-5 * top:
            continue                                      # numerically zero in the reference (layer-0 self attention: every query row equal)
        g = g.double().cpu()
        got_norm = float(g.norm())
        got_proj = float((g.flatten() * torch.from_numpy(synthetic.grad_probe(n, g.numel())).double()).sum())
        # a projection on a unit-variance probe has standard deviation |g - g_ref|: both numbers bound the relative error of the gradient
        e = max(abs(got_norm - norm), abs(got_proj - proj) / 3.0) / norm
        errs.append(e)
        if e > worst[0]:
            worst = (e, n)
    # gradient w.r.t. the feature map (RoIAlign backward + the gathered key rows): norm, probe projection, per-view norms
    gf = feat.grad.double().cpu()
    fn = float(gold[name + '.dfeat_norm'])
    assert abs(float(gf.norm()) - fn) <= 2e-2 * fn
    assert abs(float((gf.flatten() * torch.from_numpy(synthetic.grad_probe('feat', gf.numel())).double()).sum()) - float(gold[name + '.dfeat_proj'])) <= 8e-2 * fn      # (fp32 atomics in the RoIAlign backward + sign flips of the L1 term: seen at 7e-2 once in ~10 runs)
    assert torch.allclose(gf.flatten(1).norm(dim=1), torch.from_numpy(gold[name + '.dfeat_view_norms']), rtol=3e-2, atol=1e-3 * fn)
    errs.sort()
    # bf16-rounded K / V in both attentions (as in the inference engine), everything else fp32.  The L1 term's gradient is sign(pred - target):
    # a coordinate within 1e-3 of its target can flip, which moves a handful of small-norm gradients by several per cent on the 12-row cases
    assert worst[0] <= 0.15 and errs[len(errs) // 2] <= 1e-2, (worst, errs[len(errs) // 2])


def test_roi_align_backward_matches_autograd_of_the_oracle():
    """ops.RoIAlignRows (mv2d_roi_align / mv2d_roi_align_bwd) vs torch autograd through the oracle's RoIAlign: full map and compacted map."""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    g = torch.Generator().manual_seed(5)
    V, H, W = 2, 9, 13
    feat = torch.randn(V, 256, H, W, generator=g)
    rois = torch.tensor([[0, 10., 12., 90., 70.], [1, -20., 30., 60., 150.], [1, 100., 5., 207., 143.], [0, 150., 100., 230., 160.],
                         [0, 33., 40., 36., 44.]])
    gout = torch.randn(rois.shape[0], 256, 7, 7, generator=g)
    f0 = feat.clone().requires_grad_(True)
    want = O.roi_align(f0, rois)
    want.backward(gout)
    rows = feat.permute(0, 2, 3, 1).reshape(V * H * W, 256).to(DEV).requires_grad_(True)
    out = ops.RoIAlignRows.apply(rows, None, rois.to(DEV), H, W)                      # [R,49,256]
    got = out.view(-1, 7, 7, 256).permute(0, 3, 1, 2)
    assert float((got.detach().cpu() - want.detach()).abs().max()) <= 1e-5 * float(want.detach().abs().max())
    out.backward(gout.permute(0, 2, 3, 1).reshape(-1, 49, 256).to(DEV).contiguous())
    wg = f0.grad.permute(0, 2, 3, 1).reshape(V * H * W, 256)
    assert float((rows.grad.cpu() - wg).abs().max()) <= 1e-5 * float(wg.abs().max())
    # compacted map: only the positions with a non-zero reference gradient are listed (others: index -1, never read with weight > 0)
    keep = wg.abs().sum(1) > 0
    index = torch.full((V * H * W,), -1, dtype=torch.int32)
    index[keep] = torch.arange(int(keep.sum()), dtype=torch.int32)
    small = feat.permute(0, 2, 3, 1).reshape(V * H * W, 256)[keep].to(DEV).requires_grad_(True)
    out2 = ops.RoIAlignRows.apply(small, index.to(DEV), rois.to(DEV), H, W, rows.detach())
    assert torch.allclose(out2, out.detach(), rtol=0, atol=1e-6)
    out2.backward(gout.permute(0, 2, 3, 1).reshape(-1, 49, 256).to(DEV).contiguous())
    assert float((small.grad.cpu() - wg[keep]).abs().max()) <= 1e-5 * float(wg.abs().max())


@pytest.mark.parametrize('kind,prob_name', [('T', 'micro_t'), ('S', 'micro_s')])
def test_a_few_optimizer_steps_reduce_the_loss(kind, prob_name):
    """The gradients are usable: AdamW on the head's parameters lowers the training loss of a fixed sample (fixed denoising noise)."""
    from mv2d_amd import registry
    import mv2d_amd.plugin  # noqa: F401
    prob = synthetic.make_problem(prob_name, seed=0)
    with_dn, kind = kind.endswith('+DN'), kind[0]
    cfg = configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t()
    if kind == 'T':
        cfg['num_views'] = prob['views_per_frame']
    if with_dn:
        cfg['use_denoise'] = True
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    head = _dropout_off(head.to(DEV))
    gtc = synthetic.make_train_gt(5, 31)
    gt, labels = [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])]
    rnd = torch.from_numpy(synthetic.make_dn_noise(50, 31)).to(DEV)
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    opt = torch.optim.AdamW([p for p in head.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.0)
    hist = []
    for _ in range(12):
        losses = head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, dn_noise=rnd)
        total = sum(losses.values())
        opt.zero_grad()
        total.backward()
        torch.nn.utils.clip_grad_norm_(head.parameters(), 35.0)               # the reference's optimizer_config (grad_clip max_norm = 35)
        opt.step()
        hist.append(float(total.detach()))
    assert all(h == h for h in hist) and min(hist[-3:]) < 0.8 * hist[0], hist


@pytest.mark.parametrize('kind,prob_name', [('T', 'micro_t'), ('S', 'micro_s')])
def test_forward_train_without_ground_truth(kind, prob_name):
    """A frame without objects: every query is background, no denoising rows, the box loss is zero, gradients still flow (both routes)."""
    from mv2d_amd import registry
    import mv2d_amd.plugin  # noqa: F401
    prob = synthetic.make_problem(prob_name, seed=0)
    with_dn, kind = kind.endswith('+DN'), kind[0]
    cfg = configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t()
    if kind == 'T':
        cfg['num_views'] = prob['views_per_frame']
    if with_dn:
        cfg['use_denoise'] = True
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    head = _dropout_off(head.to(DEV))
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    gt, labels = [torch.zeros(0, 9)], [torch.zeros(0, dtype=torch.long)]
    a = head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=False)
    b = head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=True)
    assert set(a) == set(b) == {f'l{i}.{k}' for i in range(6) for k in ('loss_cls', 'loss_bbox')}
    for k in a:
        assert float(a[k].detach()) == float(a[k].detach()) and abs(float(a[k].detach()) - float(b[k].detach())) <= 5e-3 * max(abs(float(a[k].detach())), 1e-3)
        if k.endswith('loss_bbox'):
            assert float(a[k].detach()) == 0.0 and float(b[k].detach()) == 0.0
        else:
            assert float(a[k].detach()) > 0.0
    sum(b.values()).backward()
    g = head.bbox_head.cls_branches[5][6].weight.grad
    assert g is not None and float(g.abs().sum()) > 0 and bool(torch.isfinite(g).all())


def _t_head(prob):
    from mv2d_amd import registry
    import mv2d_amd.plugin  # noqa: F401
    cfg = configs.roi_head_cfg_t()
    cfg['num_views'] = prob['views_per_frame']
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    return head.to(DEV)


def test_fallback_key_for_a_roi_without_visible_keys():
    """Training-time rule of the two-frame head (RH/mv2d_t_head.py:80-82): a RoI whose every key is masked gets the key at map position
    (view 0, 0, 0) instead of a NaN row.  The padded-strip problem of tests/golden/nanrow_t.npz has such a RoI: in eval the whole frame is
    NaN (reference behaviour); forward_train must give finite losses and gradients on both routes, the CSR helper is checked exactly."""
    from mv2d_amd import train
    rp = torch.tensor([0, 2, 2, 5, 5], dtype=torch.int32, device=DEV)
    col = torch.tensor([7, 9, 1, 2, 3], dtype=torch.int32, device=DEV)
    rp2, col2, empty = train.fallback_key_csr(rp, col, 42)
    assert rp2.tolist() == [0, 2, 3, 6, 7] and col2.tolist() == [7, 9, 42, 1, 2, 3, 42] and empty.tolist() == [False, True, False, True]
    g = load_golden('nanrow_t')
    prob = dict(kind='T', views_per_frame=2, img_metas=synthetic.make_img_metas(2, 128, 96, frames=1, pad_w=192, yaw_step_deg=40.0),
                proposals=[g['proposals_v0'], g['proposals_v1']], feat=synthetic.make_feat(2, 8, 12, seed=22))
    head = _dropout_off(_t_head(prob))
    gtc = synthetic.make_train_gt(5, 3)
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(np.asarray(p)) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    args = ([feat], metas, props, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])], None)
    rnd = torch.from_numpy(synthetic.make_dn_noise(5 * 10, 3)).to(DEV)            # the same denoising noise on both routes
    fwd = head.forward_train(*args, dn_noise=rnd, autograd=False)
    assert all(bool(torch.isfinite(v).all()) for v in fwd.values()), fwd
    losses = head.forward_train(*args, dn_noise=rnd, autograd=True)
    assert all(bool(torch.isfinite(v).all()) for v in losses.values())
    for k in fwd:                                                     # the two routes agree (bf16 K/V on the autograd route only)
        assert abs(float(fwd[k]) - float(losses[k])) <= 5e-3 * max(abs(float(fwd[k])), 1e-2), (k, float(fwd[k]), float(losses[k]))
    sum(losses.values()).backward()
    gr = [p.grad for p in head.bbox_head.parameters() if p.requires_grad and p.grad is not None]
    assert gr and all(bool(torch.isfinite(x).all()) for x in gr)


def test_fallback_key_row_comes_from_the_input_map():
    """The fallback key of a RoI without visible keys is map position (view 0, 0, 0) (RH/mv2d_t_head.py:80-82).  With the masked transposition that row of the
    engine's position-major copy is only written when some RoI rectangle covers it: here a first frame (other boxes, a map of 1000s) does cover it, the frame
    under test does not -- the row must still be taken from THIS frame's input map (round-6 fix of an advisor finding), i.e. the losses equal those of a fresh head."""
    g = load_golden('nanrow_t')
    prob = dict(kind='T', views_per_frame=2, img_metas=synthetic.make_img_metas(2, 128, 96, frames=1, pad_w=192, yaw_step_deg=40.0),
                proposals=[g['proposals_v0'], g['proposals_v1']], feat=synthetic.make_feat(2, 8, 12, seed=22))
    gtc = synthetic.make_train_gt(5, 3)
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(np.asarray(p)) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    rnd = torch.from_numpy(synthetic.make_dn_noise(5 * 10, 3)).to(DEV)
    mk = lambda f, pr: ([f], metas, pr, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])], None)      # noqa: E731
    fresh = _dropout_off(_t_head(prob)).forward_train(*mk(feat, props), dn_noise=rnd, autograd=False)
    head = _dropout_off(_t_head(prob))
    cover = [torch.tensor([[0., 0., 40., 40., 0.9, 1.], [20., 30., 70., 90., 0.8, 2.]]), torch.tensor([[10., 20., 60., 80., 0.7, 3.]])]
    eng = head.engine(feat.device, metas)
    eng.run(torch.full_like(feat, 1000.0), cover, prob['img_metas'])              # leaves 1000s in the workspace's position-major row 0
    torch.cuda.synchronize()
    again = head.forward_train(*mk(feat, props), dn_noise=rnd, autograd=False)
    assert all(bool(torch.isfinite(v).all()) for v in again.values())
    for k in fresh:
        assert float(fresh[k]) == float(again[k]), (k, float(fresh[k]), float(again[k]))


def test_training_route_applies_the_configured_dropout():
    """The shipped configs put dropout 0.1 on both attentions' output paths and in the FFN: in training mode the autograd route applies it
    (two draws differ, and differ from the dropout-free losses); in eval mode nothing is dropped."""
    import warnings
    prob = synthetic.make_problem('cfg1_t', seed=0)
    head = _t_head(prob)
    gtc = synthetic.make_train_gt(9, 5)
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    args = ([feat], metas, props, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])], None)

    rnd = torch.from_numpy(synthetic.make_dn_noise(9 * 10, 5)).to(DEV)            # fixed denoising noise: only the dropout draws differ

    def total(seed):
        torch.manual_seed(seed)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return float(sum(head.forward_train(*args, dn_noise=rnd, autograd=True).values()).detach())
    head.train()
    a, b = total(1), total(2)
    assert a != b
    head.eval()
    c, d = total(1), total(2)
    assert c == d and c != a



# ---- round 3: the dense operators of the training route on the HIP kernels (mv2d_amd/autograd_ops.py) ------------------------------------
@pytest.mark.parametrize('M,K,N,act,bias', [(300, 256, 256, 0, True), (37, 1040, 512, 1, True), (1000, 256, 2048, 1, True), (513, 2048, 256, 0, True),
                                            (300, 256, 10, 0, True), (84, 256, 3, 0, False), (0, 256, 256, 0, True), (5000, 192, 1024, 1, True)])
def test_hip_linear_forward_backward_vs_fp64_autograd(M, K, N, act, bias):
    """LinearFn: y = act(x W^T + b) and dx, dW, db against torch autograd in fp64 (every product runs on mv2d_gemm_f32x3: fp32 operands read in
    place in either orientation, split precision; ragged sizes exercise the edge tiles)."""
    from mv2d_amd.autograd_ops import linear
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(100 + M + K)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).to(dev).requires_grad_(True) if bias else None
    dy = torch.randn(M, N, generator=g).to(dev)
    y = linear(x, W, b, act)
    y.backward(dy)
    xd, Wd = x.detach().double().requires_grad_(True), W.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    yr = torch.nn.functional.linear(xd, Wd, bd)
    if act:
        # the ReLU mask of the HIP result: a pre-activation within rounding of zero may fall on either side; the gradients are compared
        # under the same mask (one flipped element moves a dW entry by |dy x|, far above the arithmetic error this test bounds)
        mask = (y.detach() > 0).double()
        assert float((torch.relu(yr.detach()) - yr.detach() * mask).abs().max()) < 1e-4
        yr = yr * mask
    yr.backward(dy.double())
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-30)) if r.numel() else 0.0   # noqa: E731
    errs = dict(y=rel(y, yr), dx=rel(x.grad, xd.grad), dW=rel(W.grad, Wd.grad))
    if bias:
        errs['db'] = rel(b.grad, bd.grad)
    print(M, K, N, act, {k: f'{v:.1e}' for k, v in errs.items()})
    assert all(v < 5e-5 for v in errs.values()), errs


def test_hip_linear_relu_backward_with_misaligned_gradient():
    """The incoming gradient of a ReLU linear may be a contiguous slice with a storage offset (rows of 10 floats behind 3 pad rows: 120 bytes,
    not 16-byte aligned): the ReLU-mask kernel of mv2d_linear_bwd_x3 must not use 16-byte accesses on it."""
    from mv2d_amd.autograd_ops import linear
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    M, K, N = 77, 256, 10
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).to(dev).requires_grad_(True)
    dy_all = torch.randn(M + 3, N, generator=g).to(dev)
    dy = dy_all[3:]
    assert dy.is_contiguous() and dy.data_ptr() % 16 != 0
    y = linear(x, W, b, 1)
    y.backward(dy)
    xd, Wd, bd = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    yr = torch.nn.functional.linear(xd, Wd, bd) * (y.detach() > 0).double()
    yr.backward(dy.double())
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())   # noqa: E731
    errs = dict(dx=rel(x.grad, xd.grad), dW=rel(W.grad, Wd.grad), db=rel(b.grad, bd.grad))
    assert all(v < 5e-5 for v in errs.values()), errs


@pytest.mark.parametrize('M', [1, 64, 300, 2401])
def test_hip_layer_norm_forward_backward_vs_fp64_autograd(M):
    from mv2d_amd.autograd_ops import layer_norm
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7 + M)
    x = (torch.randn(M, 256, generator=g) * 3 + 0.5).to(dev).requires_grad_(True)
    w = (torch.rand(256, generator=g) + 0.5).to(dev).requires_grad_(True)
    b = torch.randn(256, generator=g).to(dev).requires_grad_(True)
    dy = torch.randn(M, 256, generator=g).to(dev)
    y = layer_norm(x, w, b)
    y.backward(dy)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.layer_norm(xd, (256,), wd, bd)
    yr.backward(dy.double())
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())   # noqa: E731
    errs = dict(y=rel(y, yr), dx=rel(x.grad, xd.grad), dw=rel(w.grad, wd.grad), db=rel(b.grad, bd.grad))
    print(M, {k: f'{v:.1e}' for k, v in errs.items()})
    assert all(v < 2e-5 for v in errs.values()), errs


def test_hip_matmul_nt_both_operands_differentiable():
    from mv2d_amd.autograd_ops import matmul_nt_ad
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    A = torch.randn(130, 32, generator=g).to(dev).requires_grad_(True)
    B = torch.randn(1777, 32, generator=g).to(dev).requires_grad_(True)
    dC = torch.randn(130, 1777, generator=g).to(dev)
    C = matmul_nt_ad(A, B)
    C.backward(dC)
    Ad, Bd = A.detach().double().requires_grad_(True), B.detach().double().requires_grad_(True)
    Cr = Ad @ Bd.t()
    Cr.backward(dC.double())
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())   # noqa: E731
    assert rel(C, Cr) < 5e-5 and rel(A.grad, Ad.grad) < 5e-5 and rel(B.grad, Bd.grad) < 5e-5


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K,pad', [(37, 10, 70, 0), (300, 256, 256, 0), (129, 65, 1040, 3), (256, 256, 14700, 0), (5, 3, 7, 1)])
def test_gemm_f32x3_all_orientations(ta, tb, M, N, K, pad):
    """mv2d_gemm_f32x3 directly: C = act(op(A) op(B)^T + bias) for every (trans_a, trans_b), ragged sizes, row strides that are not multiples
    of 4 floats (`pad`: the scalar load path), and a contraction long enough for the split-K route (slabs + fixed-order sum), vs fp64."""
    from mv2d_amd.autograd_ops import matmul_nt
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + 2 * ta + tb)
    def operand(rows, cols, trans):
        shape = (cols, rows) if trans else (rows, cols)
        full = torch.randn(shape[0], shape[1] + pad, generator=g).to(dev)
        return full[:, :shape[1]]                                              # a view with row stride shape[1] + pad
    A, B = operand(M, K, ta), operand(N, K, tb) * 0.1
    bias = torch.randn(N, generator=g).to(dev)
    for act in (0, 1):
        C = matmul_nt(A, B, bias, act, trans_a=ta, trans_b=tb)
        ref = (A.double().t() if ta else A.double()) @ (B.double() if tb else B.double().t()) + bias.double()
        if act:
            mask = (C > 0).double()
            assert float((torch.relu(ref) - ref * mask).abs().max()) < 1e-4 * float(ref.abs().max())
            ref = ref * mask
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        assert C.shape == (M, N) and err < 5e-5, (ta, tb, M, N, K, act, err)


def test_training_step_launches_no_blas_kernel():
    """The autograd route of forward_train + backward: the only matrix products are the HIP GEMM's (kernel names of a profiled step
    contain no rocBLAS / hipBLASLt 'Cijk_' / 'gemm' symbol from outside libmv2d_hip)."""
    from torch.profiler import ProfilerActivity, profile
    from mv2d_amd import configs, registry
    import mv2d_amd.plugin  # noqa: F401
    dev = 'cuda'
    prob = synthetic.make_problem('cfg1_t', seed=0)
    cfg = configs.roi_head_cfg_t()
    cfg['num_views'] = prob['views_per_frame']
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    head = head.to(dev)
    gtc = synthetic.make_train_gt(9, 3)
    feat = torch.from_numpy(prob['feat']).to(dev).requires_grad_(True)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]

    def step():
        losses = head.forward_train([feat], metas, props, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])],
                                    None, autograd=True)
        sum(losses.values()).backward()
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step()
        torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages() if getattr(e, 'device_time_total', 0) > 0 or getattr(e, 'cuda_time_total', 0) > 0}
    blas = sorted(n for n in names if 'Cijk' in n or 'rocblas' in n.lower() or 'hipblas' in n.lower() or 'miopen' in n.lower())
    assert not blas, blas
    assert any('gemm_f32x3_kernel' in n for n in names), sorted(names)[:40]


@pytest.mark.parametrize('p_drop', [0.1, 0.5])
def test_sparse_attention_probability_dropout_forward_backward(p_drop):
    """Attention-probability dropout inside the sparse attention kernels (nn.MultiheadAttention(dropout=p) in training,
    MU/petr_transformer.py:404-418): with the mask the kernels derive from (seed, p) restated in numpy (ops.attn_drop_mask), the output and
    dq / dK / dV equal fp64 autograd of softmax -> mask / (1 - p) -> value sum; the keep rate is 1 - p; p = 0 is the plain kernel."""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    dev = torch.device('cuda:0')
    g = np.random.Generator(np.random.PCG64(77))
    R, S, seed = 61, 700, 123456789
    allowed = torch.from_numpy(g.random((R, S)) < 0.08)
    allowed[5] = False
    q = torch.from_numpy(g.standard_normal((R, 256)).astype(np.float32) * 0.3).to(dev).requires_grad_(True)
    K = torch.from_numpy(g.standard_normal((S, 256)).astype(np.float32)).to(dev).to(torch.bfloat16).float().requires_grad_(True)
    V = torch.from_numpy(g.standard_normal((S, 256)).astype(np.float32)).to(dev).to(torch.bfloat16).float().requires_grad_(True)
    row_ptr, col = O.csr_from_allowed(allowed)
    row_ptr, col = row_ptr.to(dev), col.to(dev)
    nnz = int(col.numel())
    dout = torch.from_numpy(g.standard_normal((R, 256)).astype(np.float32)).to(dev)
    out = ops.SparseCrossAttention.apply(q, K, V, row_ptr, col, False, None, p_drop, seed)
    out.backward(dout)
    # reference: dense fp64 with the same mask
    m = ops.attn_drop_mask(nnz, seed, p_drop)                                    # [nnz, 8] in CSR order
    keep = float((m > 0).mean())
    assert abs(keep - (1 - p_drop)) < 0.02, keep
    rows = torch.repeat_interleave(torch.arange(R), allowed.sum(1))
    M = torch.zeros(8, R, S, dtype=torch.float64)
    M[:, rows, col.cpu().long()] = torch.from_numpy(m.T.astype(np.float64))
    qd, Kd, Vd = (t.detach().double().cpu().requires_grad_(True) for t in (q, K, V))
    lg = torch.einsum('rhd,shd->hrs', qd.view(R, 8, 32), Kd.view(S, 8, 32)).masked_fill(~allowed[None], float('-inf'))
    P = torch.softmax(lg, -1)
    P = torch.where(allowed.any(1)[None, :, None], P, torch.zeros_like(P)) * M
    ref = torch.einsum('hrs,shd->rhd', P, Vd.view(S, 8, 32)).reshape(R, 256)
    ref.backward(dout.double().cpu())
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())   # noqa: E731
    errs = dict(out=rel(out, ref), dq=rel(q.grad, qd.grad), dK=rel(K.grad, Kd.grad), dV=rel(V.grad, Vd.grad))
    print(p_drop, 'keep rate', round(keep, 4), {k: f'{v:.1e}' for k, v in errs.items()})
    assert all(v < 5e-5 for v in errs.values()), errs
    out0 = ops.SparseCrossAttention.apply(q.detach(), K.detach(), V.detach(), row_ptr, col, False, None, 0.0, seed)
    assert torch.equal(out0, ops.sparse_xattn(q.detach(), K.detach().to(torch.bfloat16), V.detach().to(torch.bfloat16), row_ptr, col, empty_nan=False))


# ---- round 5: decoder + branches issued from C (csrc/train_decoder.hip: mv2d_train_decoder_* / mv2d_train_heads_*) -------------------------
def _captured_decoder_call(prob_name, kind, train_mode=False, denoise=False):
    """A head on a synthetic problem and the arguments its ``TrainDecoder`` received in one forward_train (no denoising queries)."""
    from mv2d_amd import registry, train
    import mv2d_amd.plugin  # noqa: F401
    prob = synthetic.make_problem(prob_name, seed=0)
    cfg = configs.roi_head_cfg_s() if kind == 'S' else configs.roi_head_cfg_t()
    if kind == 'T':
        cfg['num_views'] = prob['views_per_frame']
        cfg['use_denoise'] = denoise
    head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
    head = head.to(DEV)
    head.train(train_mode)
    gtc = synthetic.make_train_gt(9, 5)
    feat = torch.from_numpy(prob['feat']).to(DEV)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
    got, orig = {}, train.TrainDecoder.__call__

    def rec(self, *a, **k):
        got['a'], got['k'] = a, k
        return orig(self, *a, **k)
    train.TrainDecoder.__call__ = rec
    try:
        head.forward_train([feat], metas, props, None, None, None, None, [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])], None,
                           autograd=True)
    finally:
        train.TrainDecoder.__call__ = orig
    return head, got['a'], got['k']


def _decoder_outputs_and_grads(head, a, k, fused, g=None, batched_dense=None):
    dec = head._train_decoder
    dec.fused = fused
    dec.batched_dense = fused if batched_dense is None else batched_dense
    for p in head.parameters():
        p.grad = None
    ref, key_in, val_in = (t.detach().clone().requires_grad_(True) for t in a[:3])
    all_cls, all_reg = dec(ref, key_in, val_in, *a[3:], **k)
    if g is None:
        gen = torch.Generator(device='cpu').manual_seed(3)
        g = (torch.randn(all_cls.shape, generator=gen).to(DEV), torch.randn(all_reg.shape, generator=gen).to(DEV))
    f = (all_cls * g[0]).sum() + (all_reg * g[1]).sum()
    f.backward()
    grads = {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}
    grads.update(ref=ref.grad, key_in=key_in.grad, val_in=val_in.grad)
    return all_cls.detach(), all_reg.detach(), grads, g, float(f.detach())


@pytest.mark.parametrize('prob_name,kind', [('cfg1_s', 'S'), ('cfg1_t', 'T'), ('nc6_s', 'S')])
def test_decoder_issued_from_c_equals_the_operator_graph(prob_name, kind):
    """mv2d_train_decoder_fwd/_bwd + mv2d_train_heads_fwd/_bwd (one autograd node each, launch sequences issued from C, weight-gradient
    products and the key side on side streams) against the per-operator autograd graph of rounds 3-4 on the same inputs and the same
    upstream gradient, dropout off: the same outputs (bitwise) and the same gradient for every parameter, the key / value rows and the
    reference points."""
    head, a, k = _captured_decoder_call(prob_name, kind)
    cls0, reg0, g0, g, _ = _decoder_outputs_and_grads(head, a, k, False)
    cls1, reg1, g1, _, _ = _decoder_outputs_and_grads(head, a, k, True, g)
    cls2, reg2, g2, _, _ = _decoder_outputs_and_grads(head, a, k, True, g)
    assert torch.equal(cls1, cls2) and torch.equal(reg1, reg2) and all(torch.equal(g1[n], g2[n]) for n in g1), 'not deterministic'
    # the same kernels on the same values in the same arithmetic order: the forward is bit-identical (so no ReLU unit switches between the
    # routes), the gradients differ by summation order only
    assert torch.equal(cls1, cls0) and torch.equal(reg1, reg0)
    assert set(g0) == set(g1)
    top = max(float(v.abs().max()) for n, v in g0.items() if n not in ('key_in', 'val_in', 'ref'))
    worst = (0.0, '')
    for n, v in g0.items():
        m = float(v.abs().max())
        if m < 1e-5 * top:
            assert float(g1[n].abs().max()) < 1e-4 * top, n        # numerically zero in both (layer-0 self attention: every value row equal)
            continue
        worst = max(worst, (float((g1[n] - v).abs().max()) / m, n))
    assert worst[0] <= 1e-4, worst


@pytest.mark.parametrize('route', ['c_entry', 'batched_dense'])
def test_denoising_rows_equal_the_per_head_loop(route):
    """With denoising queries (the two-frame head's training recipe) the first `pad` rows of the cross attention are a dense block over the keys
    any RoI sees.  'c_entry': the whole decoder incl. that block from mv2d_train_decoder_* (batched products, its own softmax + dropout
    kernel); 'batched_dense': the per-operator graph with the block batched over the heads -- both against the per-head loop of rounds 3-4:
    the same outputs (softmax summation order: a few ulp) and gradients (by norm: a ReLU unit within rounding of zero may switch)."""
    head, a, k = _captured_decoder_call('cfg1_t', 'T', denoise=True)
    assert k.get('dn_keys') is not None
    cls0, reg0, g0, g, _ = _decoder_outputs_and_grads(head, a, k, False)
    cls1, reg1, g1, _, _ = _decoder_outputs_and_grads(head, a, k, route == 'c_entry', g, batched_dense=True)
    assert float((cls1 - cls0).abs().max()) <= 2e-5 * float(cls0.abs().max()) and float((reg1 - reg0).abs().max()) <= 2e-5 * float(reg0.abs().max())
    top = max(float(v.norm()) for n, v in g0.items() if n not in ('key_in', 'val_in', 'ref'))
    errs = {n: float((g1[n] - v).norm()) / float(v.norm()) for n, v in g0.items() if float(v.norm()) >= 1e-5 * top}
    worst = max(errs.items(), key=lambda t: t[1])
    assert worst[1] <= 5e-2 and sorted(errs.values())[len(errs) // 2] <= 2e-3, (worst, sorted(errs.values())[len(errs) // 2])


@pytest.mark.parametrize('prob_name,kind,denoise', [('cfg1_s', 'S', False), ('cfg1_t', 'T', True)])
def test_decoder_issued_from_c_dropout_masks_of_forward_and_backward_agree(prob_name, kind, denoise):
    """Training mode (dropout 0.1 on the attention probabilities, both attentions' output paths and twice in the FFN -- configs/mv2d/exp/*:67-79):
    the backward regenerates the masks of the forward from (seed, layer, site, element).  With the mask counter pinned, f is a fixed
    piecewise-smooth function of the parameters, so a central difference along the gradient direction must reproduce |grad|; two different
    draws give different outputs; eval mode drops nothing."""
    head, a, k = _captured_decoder_call(prob_name, kind, train_mode=True, denoise=denoise)
    dec = head._train_decoder
    names = [n for n, _ in head.named_parameters() if 'transformer.decoder' in n]
    P = dict(head.named_parameters())

    def run(counter):
        dec._drop_calls = counter
        torch.manual_seed(11)
        return _decoder_outputs_and_grads(head, a, k, True, run.g)
    run.g = None
    cls_a, _, grads, run.g, f0 = run(0)
    cls_b, _, _, _, _ = run(0)
    cls_c, _, _, _, _ = run(5)
    assert torch.equal(cls_a, cls_b) and not torch.equal(cls_a, cls_c)
    gn = float(torch.sqrt(sum((grads[n].double() ** 2).sum() for n in names)))
    eps = 2e-3
    fs = []
    for sgn in (1.0, -1.0):
        with torch.no_grad():
            for n in names:
                P[n].add_(grads[n], alpha=sgn * eps / gn)
        fs.append(run(0)[4])
        with torch.no_grad():
            for n in names:
                P[n].add_(grads[n], alpha=-sgn * eps / gn)
    fd = (fs[0] - fs[1]) / (2 * eps)
    assert abs(fd - gn) <= 5e-2 * gn, (fd, gn, f0, fs)
    head.eval()
    cls_e, _, _, _, _ = run(0)
    cls_f, _, _, _, _ = run(5)
    assert torch.equal(cls_e, cls_f) and not torch.equal(cls_e, cls_a)


@pytest.mark.parametrize('n,nk,p', [(37, 203, 0.0), (400, 1501, 0.0), (64, 128, 0.25)])
def test_dense_block_of_the_denoising_rows_batched_over_heads(n, nk, p):
    """DenseHeadsAttnFn (one batched launch per product, mv2d_gemm_f32x3_batched) against fp64 torch attention per head: output and the
    gradients of q, k, v; with dropout the mask is read back from the saved probabilities (kept = non-zero), so the backward sees the
    forward's mask."""
    from mv2d_amd.autograd_ops import DenseHeadsAttnFn
    g = torch.Generator(device='cpu').manual_seed(n + nk)
    q, k, v, go = (torch.randn(r, 256, generator=g).to(DEV) for r in (n, nk, nk, n))
    q = q * 0.3
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    torch.manual_seed(5)
    out = DenseHeadsAttnFn.apply(qa, ka, va, 8, p)
    out.backward(go)
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    S = torch.einsum('nhd,mhd->hnm', qd.view(n, 8, 32), kd.view(nk, 8, 32))
    P = torch.softmax(S, -1)
    if p > 0:
        # the mask of the run above: regenerate it with the same seed on the same shape (padded to a multiple of 4 columns)
        torch.manual_seed(5)
        nkp = (nk + 3) & ~3
        keep = (torch.nn.functional.dropout(torch.ones(8, n, nkp, device=DEV), p, True) != 0)[..., :nk].double()
        P = P * keep / (1 - p)
    ref = torch.einsum('hnm,mhd->nhd', P, vd.view(nk, 8, 32)).reshape(n, 256)
    ref.backward(go.double())
    for got, want, name in ((out, ref, 'out'), (qa.grad, qd.grad, 'dq'), (ka.grad, kd.grad, 'dk'), (va.grad, vd.grad, 'dv')):
        err = float((got.double() - want).abs().max() / want.abs().max())
        assert err < 1e-4, (name, err)


# ---- round 5: the extended product entries and the small fused nodes, one by one ---------------------------------------------------------
def _ptr(t):
    return 0 if t is None else t.data_ptr()


@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (300, 256, 2048), (37, 10, 256), (5000, 256, 256)])
def test_gemm_ex_alpha_accumulate_bf16(M, N, K):
    """mv2d_gemm_f32x3_ex: C = (A B^T + bias) * alpha, C += ..., bf16 output -- against fp64 (split precision: ~1e-5 of the output maximum)."""
    from mv2d_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    A, B, bias, C0 = (torch.randn(s, generator=g).to(DEV) for s in ((M, K), (N, K), (N,), (M, N)))
    ws = torch.empty(max(int(lib.mv2d_gemm_f32x3_ws_bytes(M, N, K)), 1) + 512, device=DEV, dtype=torch.uint8)
    want = (A.double() @ B.double().t() + bias.double()) * 0.25
    scale = float(want.abs().max())

    def run(acc, bf16, out):
        _lib.check(lib.mv2d_gemm_f32x3_ex(A.data_ptr(), K, 0, B.data_ptr(), K, 0, bias.data_ptr(), 0, 0.25, acc, bf16, out.data_ptr(), N, M, N, K,
                                          ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), 'gemm_ex')
        return out
    out = run(0, 0, torch.empty(M, N, device=DEV))
    assert float((out.double() - want).abs().max()) <= 3e-5 * scale
    acc = run(1, 0, C0.clone())
    assert float((acc.double() - (want + C0.double())).abs().max()) <= 3e-5 * scale
    b16 = run(0, 1, torch.empty(M, N, device=DEV, dtype=torch.bfloat16))
    # the fp32 value rounded once (a long contraction runs in one pass here, split-K above: the two fp32 values differ in their last bits)
    assert bool(((b16.float() - out).abs() <= 2.0 ** -8 * out.abs() + 1e-6 * scale).all())


@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (300, 2048, 256), (300, 10, 256), (14700, 256, 256), (700, 256, 2048)])
def test_wgrad_with_bias_gradient_and_masked_dgrad(M, N, K):
    """mv2d_wgrad_f32x3 (dW = g^T x, db = column sums of g inside the product's kernel when it runs in one pass, a separate sum after split-K)
    and mv2d_dgrad_relu_f32x3 (dx = (g W) alpha where y > 0) against fp64."""
    from mv2d_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device='cpu').manual_seed(M * 3 + N)
    g, x, W, y = (torch.randn(s, generator=gen).to(DEV) for s in ((M, N), (M, K), (N, K), (M, K)))
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(int(lib.mv2d_gemm_f32x3_ws_bytes(N, K, M)), 1) + 512, device=DEV, dtype=torch.uint8)
    cs = torch.empty((max(int(lib.mv2d_colsum_scratch_rows(M)), 1), N), device=DEV)
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    _lib.check(lib.mv2d_wgrad_f32x3(g.data_ptr(), x.data_ptr(), dW.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), ws.numel(), cs.data_ptr(), st), 'wgrad')
    wW, wb = g.double().t() @ x.double(), g.double().sum(0)
    assert float((dW.double() - wW).abs().max()) <= 3e-5 * float(wW.abs().max())
    assert float((db.double() - wb).abs().max()) <= 1e-5 * float(wb.abs().max()) + 1e-4
    dx = torch.empty(M, K, device=DEV)
    _lib.check(lib.mv2d_dgrad_relu_f32x3(g.data_ptr(), W.data_ptr(), y.data_ptr(), 1.25, dx.data_ptr(), M, N, K, st), 'dgrad_relu')
    want = torch.where(y > 0, (g.double() @ W.double()) * 1.25, torch.zeros((), device=DEV, dtype=torch.float64))
    assert float((dx.double() - want).abs().max()) <= 3e-5 * float(want.abs().max())
    assert bool((dx[y <= 0] == 0).all())


def test_box_code_node_equals_the_torch_expression():
    """BoxCodeFn (mv2d_box_code_fwd / _bwd) against the reference's expression (cross_attention_head.py:216-238; velocities of the rows >= pad
    divided by dt, RH/mv2d_t_head.py:136-140) under torch autograd: boxes, d t, d reference points -- incl. reference points outside (0, 1)."""
    from mv2d_amd.autograd_ops import BoxCodeFn
    L, T, pad, dt = 6, 77, 13, 0.5
    rng = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    gen = torch.Generator(device='cpu').manual_seed(2)
    t = torch.randn(L, T, 10, generator=gen).to(DEV)
    ref = (torch.rand(T, 3, generator=gen) * 1.2 - 0.1).to(DEV)
    ref[0, 0], ref[1, 1] = 0.0, 1.0
    go = torch.randn(L, T, 10, generator=gen).to(DEV)
    t1, r1 = t.clone().requires_grad_(True), ref.clone().requires_grad_(True)
    out = BoxCodeFn.apply(t1, r1, rng, pad, dt)
    out.backward(go)
    t2, r2 = t.double().clone().requires_grad_(True), ref.double().clone().requires_grad_(True)
    r = r2.clamp(0, 1)
    inv = torch.log(r.clamp(min=1e-5) / (1 - r).clamp(min=1e-5))
    lo, hi = torch.tensor(rng[:3], device=DEV, dtype=torch.float64), torch.tensor(rng[3:], device=DEV, dtype=torch.float64)
    cxyz = (torch.cat([t2[..., 0:2], t2[..., 4:5]], -1) + inv).sigmoid() * (hi - lo) + lo
    vel = torch.cat([t2[:, :pad, 8:10], t2[:, pad:, 8:10] / dt], 1)
    want = torch.cat([cxyz[..., 0:2], t2[..., 2:4], cxyz[..., 2:3], t2[..., 5:8], vel], -1)
    want.backward(go.double())
    assert float((out.double() - want).abs().max()) <= 2e-5
    assert float((t1.grad.double() - t2.grad).abs().max()) <= 2e-5 * float(t2.grad.abs().max())
    assert float((r1.grad.double() - r2.grad).abs().max()) <= 1e-4 * float(r2.grad.abs().max())


def test_position_embedding_node_gradient():
    """PosEmbFn: forward = mv2d_posemb3d (the inference kernel), backward from the saved embedding -- against torch autograd of pos2posemb3d
    (MU/pe.py:20-33)."""
    import math
    from mv2d_amd.autograd_ops import PosEmbFn
    gen = torch.Generator(device='cpu').manual_seed(4)
    ref = torch.rand(123, 3, generator=gen).to(DEV)
    go = torch.randn(123, 384, generator=gen).to(DEV)
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = (10000 ** (2 * (dim_t // 2) / 128)).to(DEV)
    r1 = ref.clone().requires_grad_(True)
    PosEmbFn.apply(r1, dim_t).backward(go)
    r2 = ref.double().clone().requires_grad_(True)

    def emb(p):
        p = (p * (2 * math.pi))[..., None] / dim_t.double()
        return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)
    want = torch.cat((emb(r2[..., 1]), emb(r2[..., 0]), emb(r2[..., 2])), dim=-1)
    want.backward(go.double())
    assert float((PosEmbFn.apply(ref, dim_t).double() - want).abs().max()) <= 2e-4        # (fp32 argument of sin / cos at up to 2 pi)
    assert float((r1.grad.double() - r2.grad).abs().max()) <= 2e-4 * float(r2.grad.abs().max())


def test_center2lidar_node_and_unfolding_node():
    """Center2LidarFn against the torch expression of the reference (query_generator.py:333-341, mv2d_s_head.py:146-152) under autograd;
    Im2Col3x3Fn against pad + nine shifted views (what F.unfold does), forward and gradient."""
    import torch.nn.functional as F
    from mv2d_amd.autograd_ops import Center2LidarFn, Im2Col3x3Fn
    gen = torch.Generator(device='cpu').manual_seed(8)
    R, rng = 57, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    c = (torch.randn(R, 3, generator=gen) * torch.tensor([300.0, 200.0, 10.0]) + torch.tensor([400.0, 200.0, 30.0])).to(DEV)
    minv = torch.randn(R, 16, generator=gen).to(DEV)
    go = torch.randn(R, 3, generator=gen).to(DEV)
    c1 = c.clone().requires_grad_(True)
    ref = Center2LidarFn.apply(c1, minv, rng)
    ref.backward(go)
    c2 = c.double().clone().requires_grad_(True)
    hom = torch.cat([c2[:, :2] * c2[:, 2:3], c2[:, 2:3], torch.ones_like(c2[:, :1])], 1)
    xyz = (minv.double().view(R, 4, 4) * hom[:, None, :]).sum(-1)[:, :3]
    lo, hi = torch.tensor(rng[:3], device=DEV, dtype=torch.float64), torch.tensor(rng[3:], device=DEV, dtype=torch.float64)
    want = (xyz - lo) / (hi - lo)
    want.backward(go.double())
    assert float((ref.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    assert float((c1.grad.double() - c2.grad).abs().max()) <= 2e-6 * float(c2.grad.abs().max())
    x = torch.randn(9, 49, 256, generator=gen).to(DEV)
    gc = torch.randn(9 * 49, 2304, generator=gen).to(DEV)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    cols = Im2Col3x3Fn.apply(x1)
    cols.backward(gc)
    xp = F.pad(x2.view(9, 7, 7, 256), (0, 0, 1, 1, 1, 1))
    want = torch.cat([xp[:, ky:ky + 7, kx:kx + 7] for ky in range(3) for kx in range(3)], -1).reshape(9 * 49, 2304)
    want.backward(gc)
    assert torch.equal(cols, want)
    assert float((x1.grad - x2.grad).abs().max()) <= 1e-5 * float(x2.grad.abs().max())


def _dense_drop_mask(n, nk, p, seed):
    """keep / scale factors [8, n, nk] of mv2d_dense_attn_* (csrc/dense_attn.hip: murmur3 finaliser of the counter (head n + query) nkp + key)"""
    nkp = (nk + 31) & ~31
    pf = np.float32(p)
    t = float(pf) * 4294967296.0
    thr = (0xffffffff if t >= 4294967295.0 else int(t)) or 1
    idx = ((np.arange(8)[:, None, None] * n + np.arange(n)[None, :, None]) * nkp + np.arange(nk)[None, None, :]).astype(np.uint64)
    with np.errstate(over='ignore'):
        u = ((idx * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff)).astype(np.uint32) ^ np.uint32(seed & 0xffffffff)
        u ^= u >> np.uint32(16); u = ((u.astype(np.uint64) * np.uint64(0x85EBCA6B)) & np.uint64(0xffffffff)).astype(np.uint32)
        u ^= u >> np.uint32(13); u = ((u.astype(np.uint64) * np.uint64(0xC2B2AE35)) & np.uint64(0xffffffff)).astype(np.uint32)
        u ^= u >> np.uint32(16)
    return np.where(u >= np.uint32(thr), np.float32(1.0) / (np.float32(1.0) - pf), np.float32(0.0)).astype(np.float64)


@pytest.mark.parametrize('n,nk,p', [(37, 203, 0.0), (400, 1501, 0.0), (64, 128, 0.25), (100, 4097, 0.1), (16, 31, 0.0)])
def test_dense_block_without_materialised_logits(n, nk, p):
    """mv2d_dense_attn_fwd / _bwd (the logits of a tile live in MFMA accumulators only) against fp64 attention per head with the kernel's own
    dropout mask (numpy restatement of the counter hash): output, log-sum-exp, dq, dk, dv."""
    from mv2d_amd.autograd_ops import FlashDenseAttnFn
    g = torch.Generator(device='cpu').manual_seed(n * 7 + nk)
    q, k, v, go = (torch.randn(r, 256, generator=g).to(DEV) for r in (n, nk, nk, n))
    q = q * 0.3
    qa, ka, va = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = FlashDenseAttnFn.apply(qa, ka, va, p, 1234)
    out.backward(go)
    assert torch.equal(out, FlashDenseAttnFn.apply(q, k, v, p, 1234))
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    P = torch.softmax(torch.einsum('nhd,mhd->hnm', qd.view(n, 8, 32), kd.view(nk, 8, 32)), -1)
    if p > 0:
        m = torch.from_numpy(_dense_drop_mask(n, nk, p, 1234)).to(DEV)
        assert abs(float((m > 0).double().mean()) - (1 - p)) < 0.02
        P = P * m
    ref = torch.einsum('hnm,mhd->nhd', P, vd.view(nk, 8, 32)).reshape(n, 256)
    ref.backward(go.double())
    for got, want, name in ((out, ref, 'out'), (qa.grad, qd.grad, 'dq'), (ka.grad, kd.grad, 'dk'), (va.grad, vd.grad, 'dv')):
        err = float((got.double() - want).abs().max() / want.abs().max())
        assert err < 1e-4, (name, err)
