"""The oracle (oracle/mv2d_oracle.py) against golden vectors produced by the UNMODIFIED reference
(oracle/gen_golden.py, run in the build container).  CPU only.

Integer / boolean stage outputs must be bit-exact; float stages within 2e-5 relative-to-max (the
restatement re-associates nothing but does use explicit matmul+softmax instead of
F.multi_head_attention_forward, so last-ulp differences exist)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, unpack_bits
from mv2d_amd import synthetic
from oracle import mv2d_oracle as O


def close(a, b, tol=2e-5):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-6)
    err = np.abs(a - b).max() / scale
    assert err <= tol, err


def run(name, prob=None):
    prob = prob or synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=0)
    st = {}
    feat = torch.from_numpy(prob['feat'])
    props = [torch.from_numpy(p) for p in prob['proposals']]
    if prob['kind'] == 'T':
        O.forward_t(sd, feat, props, prob['img_metas'], num_views=prob['views_per_frame'], stages=st)
    else:
        O.forward_s(sd, feat, props, prob['img_metas'], stages=st)
    return st


def check_common(st, g):
    np.testing.assert_array_equal(st['K_roi'].numpy(), g['K_roi'])
    np.testing.assert_array_equal(st['E'].numpy(), g['E'])
    close(st['intr'], g['intr'], 1e-7)
    close(st['center_pred'], g['center_pred'])
    close(st['xyz'], g['xyz'])
    close(st['ref'].numpy().reshape(-1, 3), g['ref'].reshape(-1, 3))
    close(st['cls'].numpy().reshape(g['cls'].shape) if st['cls'].numel() == g['cls'].size else st['cls'], g['cls'], 1e-4)


@pytest.mark.parametrize('name,seed', [('micro_t', 0), ('cfg1_t', 0), ('cfg3_t', 0), ('cfg5_t', 0), ('cfg3_t', 2), ('cfg5_t_dup', 0)])      # round 6: a third seed; near-duplicate boxes
def test_t_path_matches_reference(name, seed):
    g = load_golden(name if seed == 0 else f'{name}_seed{seed}')
    st = run(name, synthetic.make_problem(name, seed=seed))
    check_common(st, g)
    ffr = unpack_bits(g['feat_for_rois'], g['feat_for_rois_shape'])
    np.testing.assert_array_equal(st['feat_for_rois'].numpy(), ffr)                 # bit-exact bool
    blocked = unpack_bits(g['blocked_attn'], g['blocked_shape'])
    np.testing.assert_array_equal((~st['feat_for_rois'])[:, st['roi_mask']].numpy(), blocked)
    np.testing.assert_array_equal(st['key_padding'].numpy(), g['key_padding'])
    reg = st['reg'].numpy().reshape(g['reg'].shape)
    if name in ('cfg3_t', 'cfg5_t', 'cfg5_t_dup'):
        # two frames: the golden 'reg' is CrossAttentionBoxHead.forward's output, BEFORE RH/mv2d_t_head.py:136-140 divides the velocities
        # by dt = 0.5 s (the golden 'boxes' are after it)
        reg = np.concatenate([reg[..., :8], reg[..., 8:] * 0.5], -1)
    close(reg, g['reg'], 1e-4)
    # integer decode outputs: bit-exact
    idx = (st['bbox_index'] * 10 + st['labels']).numpy()
    # the reference applies the centre-range filter after top-k; compare on the kept set
    np.testing.assert_array_equal(st['labels'].numpy(), g['labels'])
    close(st['scores'], g['scores'], 1e-4)
    close(st['boxes'], g['boxes'], 1e-4)
    assert set(idx.tolist()) <= set(g['topk_index'].tolist())
    if name == 'micro_t':
        close(st['pe'], g['pe'], 1e-5)
        close(st['roi_feats'], g['roi_align'][:, :256], 1e-6)
        close(st['qpos'], g['qpos'][0], 1e-5)
        close(st['outs_dec'], g['outs_dec'][:, 0], 1e-4)
        cap = st['capture']
        allowed = ~st['blocked'].numpy()
        lg = cap[0]['logits'].numpy()
        assert np.abs(lg - g['logits_l0'])[:, allowed].max() <= 1e-4 * np.abs(g['logits_l0']).max()
        for l in range(6):
            close(cap[l]['attn_mean'], g['attn_mean'][l], 1e-4)


@pytest.mark.parametrize('name,seed', [('micro_s', 0), ('cfg1_s', 0), ('cfg2_s', 0), ('nc6_s', 0), ('cfg2_s_nc6', 0),      # nc6_s / cfg2_s_nc6: up to 6 correlated RoIs per query
                                       ('cfg2_s_nc6', 2), ('cfg2_s_r450', 0), ('cfg2_s_dup', 0)])                       # round 6: third seed, R = 450, near-duplicate boxes
def test_s_path_matches_reference(name, seed):
    g = load_golden(name if seed == 0 else f'{name}_seed{seed}')
    st = run(name, synthetic.make_problem(name, seed=seed))
    check_common(st, g)
    if name.endswith('_dup'):
        # a fifth of the boxes are copies: an IoU tie between IDENTICAL boxes is broken by the reference's unstable argsort (RH/utils/box_correlation.py:370;
        # observed: towards the higher index) and by the oracle towards the lower one -- the matched RoI may differ, its box (and so its features) may not
        rois = O.bbox2roi([torch.from_numpy(p)[:, :4] for p in synthetic.make_problem(name, seed=seed)['proposals']]).numpy()
        assert np.array_equal(rois[st['corr'].numpy()], rois[g['corr']]) and (st['corr'].numpy() != g['corr']).any()
    else:
        np.testing.assert_array_equal(st['corr'].numpy(), g['corr'])                # bit-exact int64
    np.testing.assert_array_equal(st['corr_mask'].numpy(), g['corr_mask'])          # bit-exact bool
    close(st['reg'].numpy().reshape(g['reg'].shape), g['reg'], 1e-4)
    np.testing.assert_array_equal(st['labels'].numpy(), g['labels'])
    close(st['scores'], g['scores'], 1e-4)
    close(st['boxes'], g['boxes'], 1e-4)
    if name == 'micro_s':
        close(st['pe'], g['pe'], 1e-5)
        close(torch.cat([st['roi_feats'], st['roi_pe']], 1), g['roi_align'], 1e-5)
        close(st['outs_dec'], g['outs_dec'][:, :, 0], 1e-4)


def test_two_frame_velocity_dt():
    g = load_golden('twoframe_t')
    prob = dict(kind='T', views_per_frame=2,
                img_metas=synthetic.make_img_metas(2, 128, 192, frames=2, yaw_step_deg=40.0),
                proposals=synthetic.make_proposals(4, 4, 128, 192, seed=11),
                feat=synthetic.make_feat(4, 8, 12, seed=12))
    st = run(None, prob)
    np.testing.assert_array_equal(st['feat_for_rois'].numpy(), unpack_bits(g['feat_for_rois'], g['feat_for_rois_shape']))
    # the golden 'reg' is CrossAttentionBoxHead.forward's output, i.e. BEFORE RH/mv2d_t_head.py:136-140 divides
    # (vx, vy) by dt = 0.5 s; the golden 'boxes' are after it.
    reg = st['reg'].numpy().reshape(g['reg'].shape)
    close(reg[..., :8], g['reg'][..., :8], 1e-4)
    close(reg[..., 8:] * 0.5, g['reg'][..., 8:], 1e-4)
    np.testing.assert_array_equal(st['labels'].numpy(), g['labels'])
    close(st['boxes'], g['boxes'], 1e-4)


@pytest.mark.parametrize('kind', ['t', 's'])
def test_empty_detections_dummy_proposal(kind):
    g = load_golden('empty_' + kind)
    prob = synthetic.make_problem('micro_' + kind, seed=0)
    prob['proposals'] = [np.zeros((0, 6), np.float32) for _ in prob['proposals']]
    st = run(None, prob)
    assert st['rois'].shape == (1, 5)
    close(st['cls'].numpy().reshape(g['cls'].shape), g['cls'], 1e-4)
    close(st['boxes'], g['boxes'], 1e-4)
    np.testing.assert_array_equal(st['labels'].numpy(), g['labels'])


def test_fully_masked_row_is_nan_in_reference():
    """A query whose every key is padding-masked: the reference (nn.MultiheadAttention) yields NaN and the
    NaN poisons every query through the next self-attention.  The oracle reproduces that; the
    HIP path deliberately emits a zero attention output for such a row instead."""
    g = load_golden('nanrow_t')
    prob = dict(kind='T', views_per_frame=2,
                img_metas=synthetic.make_img_metas(2, 128, 96, frames=1, pad_w=192, yaw_step_deg=40.0),
                proposals=[g['proposals_v0'], g['proposals_v1']],
                feat=synthetic.make_feat(2, 8, 12, seed=22))
    st = run(None, prob)
    assert st['blocked'].all(1).any()
    assert np.isnan(g['cls']).sum() == g['cls'].size - 0 or np.isnan(g['cls']).any()
    assert torch.isnan(st['cls'][-1]).all()
    assert st['boxes'].shape[0] == 0 and g['boxes'].shape[0] == 0


# ---- training targets / losses: oracle restatement vs the reference's own assigner / loss code ------------------------
TRAIN = load_golden('train_loss')


@pytest.mark.parametrize('name', list(synthetic.TRAIN_CASES))
def test_training_targets_and_losses_match_reference(name):
    R, G, seed = synthetic.TRAIN_CASES[name]
    c = synthetic.make_train_case(R, G, seed)
    cls, box = torch.from_numpy(c['cls']), torch.from_numpy(c['box'])
    gt, labels = torch.from_numpy(c['gt']), torch.from_numpy(c['gt_labels'])
    n = c['known_labels'].shape[0]
    for l in range(cls.shape[0]):
        lc, lb, match = O.loss_single(cls[l], box[l], gt, labels)
        assert np.array_equal(match.numpy().astype(np.int32), TRAIN[name + '.match'][l])
        np.testing.assert_allclose([float(lc), float(lb)], TRAIN[name + '.loss'][l], rtol=2e-6, atol=1e-7)
        for neg in (False, True):
            dc, db = O.dn_loss_single(cls[l][:n], box[l][:n], torch.from_numpy(c['known_bboxs']), torch.from_numpy(c['known_labels']),
                                      c['dn_num_tgt'], 0.6, neg_bbox_loss=neg)
            np.testing.assert_allclose([float(dc), float(db)], TRAIN[name + ('.dn_neg' if neg else '.dn')][l], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('name', list(synthetic.DN_CASES))
def test_prepare_for_dn_matches_reference(name):
    R, G, seed, scalar, nscale, split = synthetic.DN_CASES[name]
    c = synthetic.make_train_case(R, G, seed)
    rnd = torch.from_numpy(synthetic.make_dn_noise(G * scalar, seed))
    ref = torch.from_numpy(synthetic.make_dn_noise(R, seed + 100))
    padded, attn_mask, kl, kb, pad = O.prepare_for_dn(ref, torch.from_numpy(c['gt']), torch.from_numpy(c['gt_labels']), rnd, scalar, nscale,
                                                      0.0, split)
    assert pad == int(TRAIN[name + '.pad_size']) == G * scalar
    assert np.array_equal(padded.numpy(), TRAIN[name + '.padded'])
    assert np.array_equal(np.packbits(attn_mask.numpy()), TRAIN[name + '.attn_mask'])
    assert np.array_equal(kl.numpy(), TRAIN[name + '.known_labels']) and np.array_equal(kb.numpy(), TRAIN[name + '.known_bboxs'], equal_nan=True)
    assert np.array_equal(TRAIN[name + '.map_known_indice'], np.arange(pad))


@pytest.mark.parametrize('name', ['cfg1_t'])
def test_oracle_logits_on_allowed_pairs(name):
    """The oracle's layer-0 cross-attention logits / head-averaged weights against tests/golden/attn_pairs.npz (reference outputs on the
    allowed pairs, oracle/gen_golden_logits.py)."""
    ap = load_golden('attn_pairs')
    st = run(name)
    pairs = ap[f'{name}/pairs']
    rr, kk = pairs[:, 0], pairs[:, 1]
    allowed = ~st['blocked'].numpy()
    np.testing.assert_array_equal(np.stack(np.nonzero(allowed), 1), pairs)
    cap = st['capture']
    close(cap[0]['logits'].numpy()[:, rr, kk], ap[f'{name}/logits'], 1e-4)
    for l in range(6):
        close(cap[l]['attn_mean'].numpy()[rr, kk], ap[f'{name}/attn'][l], 1e-4)


@pytest.mark.parametrize('name,src', [('cfg1_s_allm', 'cfg1_s'), ('nc6_s_allm', 'nc6_s')])
def test_all_matched_correlation_mode_on_the_s_head_matches_reference(name, src):
    """correlation_mode='all_matched' through the S head (round 6; RH/utils/box_correlation.py:165-193 with the branch :305-338): the reference's own
    compacted id lists [R, n_c] (up to 33 RoIs per query on the overlapping rig) and its class logits against the oracle's restatement (topk=None)."""
    g = load_golden(name)
    prob = synthetic.make_problem(src, seed=0)
    sd = synthetic.make_head_state(seed=0)
    st = {}
    O.forward_s(sd, torch.from_numpy(prob['feat']), [torch.from_numpy(p) for p in prob['proposals']], prob['img_metas'], topk=None, stages=st)
    np.testing.assert_array_equal(st['corr_mask'].numpy(), g['corr_mask'])
    np.testing.assert_array_equal(st['corr'].numpy() * g['corr_mask'], g['corr'] * g['corr_mask'])
    close(st['cls'].numpy().reshape(g['cls'].shape), g['cls'], 1e-4)
    if src == 'nc6_s':
        assert g['corr_mask'].sum(1).max() > 7                      # more RoIs per query than topk_matched:1 can list: the mode is not vacuous here


@pytest.mark.parametrize('name,src', [('cfg1_t_allm', 'cfg1_t'), ('nc6_t_allm', 'nc6_s')])
def test_all_matched_correlation_mode_matches_reference(name, src):
    """correlation_mode='all_matched' (RH/utils/box_correlation.py:305-338; no shipped config) through the T head: the goldens come from the
    reference's own all_matched branch; the oracle's restatement (topk=None) reproduces its boolean cell masks bit for bit and its class
    logits; topk_matched with k = 1 gives OTHER masks on the overlapping rig (the mode is not vacuous there)."""
    g = load_golden(name)
    prob = synthetic.make_problem(src, seed=0)
    sd = synthetic.make_head_state(seed=0)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    st = {}
    O.forward_t(sd, torch.from_numpy(prob['feat']), props, prob['img_metas'], num_views=prob['views_per_frame'], topk=None, stages=st)
    ffr = unpack_bits(g['feat_for_rois'], g['feat_for_rois_shape'])
    np.testing.assert_array_equal(st['feat_for_rois'].numpy(), ffr)
    close(st['cls'].numpy().reshape(g['cls'].shape), g['cls'], 1e-4)
    if src == 'nc6_s':
        st1 = {}
        O.forward_t(sd, torch.from_numpy(prob['feat']), props, prob['img_metas'], num_views=prob['views_per_frame'], topk=1, stages=st1)
        assert int((st1['feat_for_rois'].numpy() != ffr).sum()) > 0
