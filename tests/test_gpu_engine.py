"""End-to-end parity of the fused HIP engine (mv2d_amd.engine.HeadEngine) against the oracle on the same seeded
inputs (-m gpu).  Integer / boolean stages (correlated-RoI lists, key list, CSR) are bit-exact; float stages are
bounded by the bf16 rounding of the key side + bf16 MFMA of the PE / conv GEMMs (tolerances written below,
relative to the stage's max magnitude); the final top-k is compared gap-aware: every index whose oracle score is
separated from the K-th score by more than the score tolerance must be selected."""
import numpy as np
import pytest
import torch

from mv2d_amd import synthetic

pytestmark = pytest.mark.gpu

# engine vs oracle on the same inputs; bounds = ~2 x the largest measured value (round 4, fp16 key side: see the dicts this file prints under -s;
# round 3 with bf16 keys: center 3.3e-4, ref 1.2e-4, qpos 2.2e-4, pe 4.3e-3, roi_feat 2.7e-3, outs 5.2e-4, cls 2.3e-4, reg 4.3e-4)
TOL = dict(center=1.3e-4, ref=4e-5, qpos=4e-5, pe=1e-3, outs=2e-4, cls=5e-5, reg=1.5e-4, roi_feat=7e-4, score=5e-4)


def relmax(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def run_both(name, prob=None, state_seed=0):
    from mv2d_amd.engine import HeadEngine
    from oracle import mv2d_oracle as O
    prob = prob or synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=state_seed)
    dev = torch.device('cuda:0')
    eng = HeadEngine(sd, prob['kind'], dev, num_views=prob['views_per_frame'])
    feat = torch.from_numpy(prob['feat'])
    props = [torch.from_numpy(p) for p in prob['proposals']]
    out = eng.run(feat.to(dev), props, prob['img_metas'], keep_stages=True)
    torch.cuda.synchronize()
    st = {}
    if prob['kind'] == 'T':
        O.forward_t(sd, feat, props, prob['img_metas'], num_views=prob['views_per_frame'], stages=st)
    else:
        O.forward_s(sd, feat, props, prob['img_metas'], stages=st)
    return eng, out, st


def check_topk(out, st, eng):
    n = int(out['count'].item())
    got = set((out['bbox_index'][:n] * 10 + out['labels'][:n]).cpu().tolist())
    cls = st['cls'][-1].reshape(-1, 10)
    scores = cls.sigmoid().view(-1)
    k = min(300, scores.numel())
    top, idx = scores.topk(k)
    kth = float(top[-1])
    # candidates kept by the centre-range filter in the oracle
    exp = set((st['bbox_index'] * 10 + st['labels']).tolist())
    if k < scores.numel():
        margin = TOL['score'] * float(top[0])
        must = {int(i) for i, s in zip(idx.tolist(), top.tolist()) if s - kth > margin and int(i) in exp}
    else:
        must = exp
    missing = must - got
    assert not missing, f'{len(missing)} confidently-top-k indices missing'
    # scores of common entries agree
    gs = dict(zip((out['bbox_index'][:n] * 10 + out['labels'][:n]).cpu().tolist(), out['scores'][:n].cpu().tolist()))
    common = [i for i in exp if i in gs]
    err = max(abs(gs[i] - float(scores[i])) for i in common) / float(top[0])
    assert err < TOL['score'], err
    return len(exp & got) / max(len(exp), 1)


@pytest.mark.parametrize('name', ['micro_t', 'cfg1_t', 'cfg3_t'])
def test_engine_t_path(name):
    check_t_path(name)


def check_t_path(name, prob=None, state_seed=0):
    from oracle import mv2d_oracle as O
    eng, out, st = run_both(name, prob, state_seed)
    metas = (prob or synthetic.make_problem(name, seed=0))['img_metas']
    s = out['stages']
    R = out['R']
    # --- integer / boolean stages: bit-exact
    ffr = st['feat_for_rois']
    pad = st['key_padding']
    V, h, w = ffr.shape[1:]
    assert torch.equal(s['roi_mask'].cpu().bool().view(V, h, w), st['roi_mask'])
    keep = (st['roi_mask'] & ~O.padding_mask(metas, h, w)).view(-1)
    S = int(s['S_dev'])
    assert S == int(keep.sum())
    assert torch.equal(s['s2pos'][:S].cpu().long(), keep.nonzero()[:, 0])
    allowed = (ffr & ~O.padding_mask(metas, h, w)[None]).view(R, -1)[:, keep]
    rp, col = O.csr_from_allowed(allowed)
    assert torch.equal(s['row_ptr'].cpu(), rp)
    assert torch.equal(s['col_idx'][:int(rp[-1])].cpu(), col)
    # --- float stages
    errs = dict(center=relmax(s['center'], st['center_pred']), ref=relmax(s['ref'], st['ref']),
                qpos=relmax(s['qpos'], st['qpos']))
    pe_ref = st['pe'].permute(0, 2, 3, 1).reshape(-1, 256)[keep]
    errs['pe'] = relmax(s['pe'][:S], pe_ref)
    errs['outs'] = relmax(s['outs'], st['outs_dec'])
    errs['cls'] = relmax(s['cls'], st['cls'])
    errs['reg'] = relmax(s['reg'], st['reg'])
    print(name, 'R', R, 'S', S, 'nnz', int(rp[-1]), {k: f'{v:.2e}' for k, v in errs.items()})
    for k, v in errs.items():
        assert v < TOL[k], (k, v)
    frac = check_topk(out, st, eng)
    print(name, 'top-k overlap', frac)
    assert frac > 0.9


def eng_metas(name):
    return synthetic.make_problem(name, seed=0)['img_metas']


@pytest.mark.parametrize('name', ['micro_s', 'cfg1_s', 'cfg2_s', 'nc6_s'])
def test_engine_s_path(name):
    check_s_path(name)


def check_s_path(name, prob=None, state_seed=0):
    eng, out, st = run_both(name, prob, state_seed)
    s = out['stages']
    R = out['R']
    corr, cmask = st['corr'], st['corr_mask']
    exp_cols = []
    for r in range(R):
        for j in range(corr.shape[1]):
            if cmask[r, j]:
                exp_cols += [int(corr[r, j]) * 49 + c for c in range(49)]
    nnz = int(s['nnz'][0])
    assert nnz == len(exp_cols)
    assert s['col_idx'][:nnz].cpu().tolist() == exp_cols
    errs = dict(center=relmax(s['center'], st['center_pred']), ref=relmax(s['ref'], st['ref']), qpos=relmax(s['qpos'], st['qpos']),
                outs=relmax(s['outs'], st['outs_dec']), cls=relmax(s['cls'], st['cls']), reg=relmax(s['reg'], st['reg']))
    rf = st['roi_feats'].flatten(2).transpose(1, 2)
    errs['roi_feat'] = relmax(s['roi_feat'].float(), rf)
    print(name, 'R', R, 'nnz', nnz, {k: f'{v:.2e}' for k, v in errs.items()})
    for k, v in errs.items():
        assert v < TOL.get(k, 1e-2), (k, v)
    frac = check_topk(out, st, eng)
    print(name, 'top-k overlap', frac)
    assert frac > 0.9


@pytest.mark.parametrize('seed', [1, 2, 3, 4, 5])
@pytest.mark.parametrize('name', ['cfg1_t', 'cfg1_s', 'nc6_s'])
def test_engine_other_seeds(name, seed):
    """The stage-by-stage comparison with the oracle on OTHER random draws than the goldens' seed 0: different proposals / features
    (problem seed) and different head weights (state seed); integer stages bit-exact, float stages within the same bounds."""
    prob = synthetic.make_problem(name, seed=seed)
    (check_t_path if prob['kind'] == 'T' else check_s_path)(name, prob, state_seed=seed)


@pytest.mark.parametrize('name,seed', [('cfg1_t', 1), ('cfg1_t', 2), ('cfg1_t', 3), ('cfg1_s', 1), ('cfg1_s', 2), ('cfg1_s', 3), ('nc6_s', 1), ('nc6_s', 2),
                                       ('nc6_s', 3), ('cfg2_s', 2), ('cfg2_s', 3), ('cfg3_t', 2)])      # the last three: FULL size, other weights + inputs
def test_index_exact_route_other_seeds(name, seed):
    """HeadEngine(exact=True) on other random draws: the ranked (query, class) list of the decode equals the oracle's (fp32 restatement of the
    reference) up to fp32-rounding ties, scores at fp32 rounding level."""
    from mv2d_amd.engine import HeadEngine
    from oracle import mv2d_oracle as O
    prob = synthetic.make_problem(name, seed=seed)
    sd = synthetic.make_head_state(seed=seed)
    dev = torch.device('cuda:0')
    eng = HeadEngine(sd, prob['kind'], dev, num_views=prob['views_per_frame'], exact=True)
    feat = torch.from_numpy(prob['feat'])
    props = [torch.from_numpy(p) for p in prob['proposals']]
    out = eng.run(feat.to(dev), props, prob['img_metas'])
    torch.cuda.synchronize()
    st = {}
    if prob['kind'] == 'T':
        O.forward_t(sd, feat, props, prob['img_metas'], num_views=prob['views_per_frame'], stages=st)
    else:
        O.forward_s(sd, feat, props, prob['img_metas'], stages=st)
    n = int(out['count'].item())
    got = (out['bbox_index'][:n] * 10 + out['labels'][:n]).cpu().numpy()
    ref = (st['bbox_index'] * 10 + st['labels']).numpy()
    assert n == len(ref)
    moved = int((got != ref).sum())
    err = float((out['scores'][:n].cpu() - st['scores']).abs().max())
    print(f'[index parity, exact route vs oracle] {name} seed {seed}: {moved}/{n} ranked (query, class) indices differ, max score err {err:.1e}')
    assert err < 3e-6                       # round 5 (fp16 pairs on the query side): measured <= 4e-7
    # the oracle is an fp32 pipeline of its own: like the reference against itself (tests/golden/refnoise.npz: 0-4 ranks across gaps <= 4.1e-8) it may
    # order exact score ties differently; every moved entry must sit on such a tie
    assert moved <= 4
    sc = st['scores'].numpy()
    for i in np.nonzero(got != ref)[0]:
        j = int(np.nonzero(ref == got[i])[0][0]) if (ref == got[i]).any() else None
        assert j is not None and abs(float(sc[i]) - float(sc[j])) < 2e-7


def test_engine_two_frame_velocity_and_empty():
    from mv2d_amd.engine import HeadEngine
    from oracle import mv2d_oracle as O
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    prob = dict(kind='T', views_per_frame=2, img_metas=synthetic.make_img_metas(2, 128, 192, frames=2, yaw_step_deg=40.0),
                proposals=synthetic.make_proposals(4, 4, 128, 192, seed=11), feat=synthetic.make_feat(4, 8, 12, seed=12))
    eng, out, st = run_both(None, prob)
    assert relmax(out['stages']['reg'], st['reg']) < TOL['reg']                      # includes (vx, vy) / dt
    # empty detections -> one dummy proposal in view 0
    prob = synthetic.make_problem('micro_t', seed=0)
    prob['proposals'] = [np.zeros((0, 6), np.float32) for _ in prob['proposals']]
    eng, out, st = run_both(None, prob)
    assert out['R'] == 1
    assert relmax(out['stages']['cls'], st['cls']) < TOL['cls']
    b, s_, l = eng.results(out)
    assert b.shape[0] == st['boxes'].shape[0]
    assert torch.equal(l.cpu(), st['labels'])


def test_engine_is_deterministic_and_reusable():
    """Same frame twice through the same engine/workspace -> bitwise identical outputs (no atomics in the float path)."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem('cfg1_t', seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    eng = HeadEngine(sd, 'T', dev, num_views=2)
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    o1 = eng.run(feat, props, prob['img_metas'])
    c1, r1 = o1['cls'].clone(), o1['reg'].clone()
    b1 = [t.clone() for t in eng.results(o1)]
    o2 = eng.run(feat, props, prob['img_metas'])
    b2 = eng.results(o2)
    assert torch.equal(c1, o2['cls']) and torch.equal(r1, o2['reg'])
    for a, b in zip(b1, b2):
        assert torch.equal(a, b)


@pytest.mark.parametrize('kind,name', [('T', 'cfg1_t'), ('S', 'cfg1_s')])
def test_engine_varying_roi_count_shares_one_storage(kind, name):
    """Real frames have a different number of RoIs every time: the engine must serve them from ONE storage bucket (views per R)
    and every frame must equal what a fresh engine computes for it, also when a smaller frame follows a larger one (graphs too)."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    vpf = prob['views_per_frame']
    feat = torch.from_numpy(prob['feat']).to(dev)
    full = [torch.from_numpy(p) for p in prob['proposals']]
    variants = [full, [p[:max(1, p.shape[0] - 2 - i)] for i, p in enumerate(full)], [p[:max(1, p.shape[0] // 2)] for p in full], full]
    eng = HeadEngine(sd, kind, dev, num_views=vpf)
    for use_graph in (False, True):
        for props in variants:
            out = eng.run(feat, props, prob['img_metas'], use_graph=use_graph)
            got = [out['cls'].clone(), out['reg'].clone()] + [t.clone() for t in eng.results(out)]
            fresh = HeadEngine(sd, kind, dev, num_views=vpf)
            ref = fresh.run(feat, props, prob['img_metas'])
            want = [ref['cls'], ref['reg']] + list(fresh.results(ref))
            for a, b in zip(got, want):
                assert torch.equal(a, b), (use_graph, out['R'])
    # all four RoI counts fall into one bucket: one workspace, and (with use_graph) ONE captured graph replayed for every count
    assert len(eng._ws_base) == 1 and len(eng._ws) == 1
    assert len(next(iter(eng._ws.values()))['graphs']) == 1


def _varied_samples(name, n):
    """n different samples of one workload: own features, own proposals (different counts), own camera rig / timestamps"""
    out = []
    for i in range(n):
        prob = synthetic.make_problem(name, seed=10 * i)
        props = [torch.from_numpy(p[:max(1, p.shape[0] - (2 * i + j) % 5)]) for j, p in enumerate(prob['proposals'])]
        metas = [dict(m) for m in prob['img_metas']]
        if i:
            import numpy as np
            rot = np.eye(4); a = 0.03 * i
            rot[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]; rot[0, 3] = 0.2 * i
            for m in metas:
                e2 = np.asarray(m['extrinsics'], dtype=np.float64).T @ rot              # lidar -> camera of a slightly moved rig
                m['extrinsics'] = np.ascontiguousarray(e2.T)
                m['lidar2img'] = np.asarray(m['intrinsics'], dtype=np.float64) @ e2
                if 'timestamp' in m:
                    m['timestamp'] = float(m['timestamp']) * (1.0 + 0.1 * i)
        out.append((torch.from_numpy(prob['feat']), props, metas, prob['views_per_frame']))
    return out


@pytest.mark.parametrize('kind,name,n', [('S', 'cfg1_s', 3), ('T', 'cfg1_t', 3), ('S', 'cfg2_s', 2), ('T', 'cfg3_t', 2), ('S', 'cfg1_s', 16), ('T', 'cfg1_t', 16)])      # 16: the samples per launch of the bench
def test_engine_batch_of_samples_equals_single_runs(kind, name, n):
    """run_batch puts several samples through ONE sequence of launches; nothing may leak between samples: every sample's
    outputs are bitwise what a single-sample run gives (box correlation, self attention, top-k, dt stay inside a sample)."""
    from mv2d_amd.engine import HeadEngine
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    samples = _varied_samples(name, n)
    vpf = samples[0][3]
    eng = HeadEngine(sd, kind, dev, num_views=vpf)
    feats = [s[0].to(dev) for s in samples]
    for use_graph in (False, True):
        out = eng.run_batch(feats, [s[1] for s in samples], [s[2] for s in samples], use_graph=use_graph)
        res = [[t.clone() for t in r] for r in eng.results_batch(out)]
        cls, reg, grp = out['cls'].clone(), out['reg'].clone(), out['grp_start'].tolist()
        assert len(res) == n and grp[-1] == out['R']
        single = HeadEngine(sd, kind, dev, num_views=vpf)
        for b, (f, props, metas, _) in enumerate(samples):
            ref = single.run(feats[b], props, metas)
            assert ref['R'] == grp[b + 1] - grp[b]
            assert torch.equal(cls[:, grp[b]:grp[b + 1]], ref['cls']) and torch.equal(reg[:, grp[b]:grp[b + 1]], ref['reg']), (use_graph, b)
            for a, w in zip(res[b], single.results(ref)):
                assert torch.equal(a, w), (use_graph, b)
    # a stacked [B*V,256,h,w] map is accepted as well
    out = eng.run_batch(torch.cat(feats, 0), [s[1] for s in samples], [s[2] for s in samples])
    assert torch.equal(out['cls'], cls)


@pytest.mark.parametrize('kind', ['S', 'T'])
def test_engine_batch_edge_cases(kind):
    """A batch mixing the edge cases: a sample without any detection (dummy proposal in ITS first view), a sample with an empty view and a
    single box, a full sample; 'nan' semantics stay inside the sample that triggers them.  Each sample == its single-sample run."""
    from mv2d_amd.engine import HeadEngine
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    base = synthetic.make_problem('cfg1_' + kind.lower(), seed=0)
    full = [torch.from_numpy(p) for p in base['proposals']]
    none = [p[:0] for p in full]
    ragged = [full[0][:0], full[1][:1]] + [p[:3] for p in full[2:]]
    plist = [none, full, ragged, none]
    metas = [base['img_metas']] * len(plist)
    feats = [torch.from_numpy(synthetic.make_problem('cfg1_' + kind.lower(), seed=s)['feat']).to(dev) for s in (0, 3, 4, 5)]
    eng = HeadEngine(sd, kind, dev, num_views=base['views_per_frame'])
    out = eng.run_batch(feats, plist, metas)
    grp = out['grp_start'].tolist()
    assert [grp[i + 1] - grp[i] for i in range(4)] == [1, sum(len(p) for p in full), sum(len(p) for p in ragged), 1]
    res = eng.results_batch(out)
    single = HeadEngine(sd, kind, dev, num_views=base['views_per_frame'])
    for b in range(4):
        ref = single.run(feats[b], plist[b], metas[b])
        a, w = out['cls'][:, grp[b]:grp[b + 1]], ref['cls']
        assert torch.equal(torch.isnan(a), torch.isnan(w)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(w)), b
        for x, y in zip(res[b], single.results(ref)):
            assert torch.equal(x, y), b


@pytest.mark.parametrize('kind,name', [('S', 'cfg1_s'), ('T', 'cfg1_t')])
def test_engine_map_kernels_fused_or_separate_bitwise_equal(kind, name):
    """The per-head query / context maps of the tile cross attention run inside the neighbouring row kernels (chosen for launches of <= 512
    rows, i.e. here) or as separate kernels (batches): bitwise the same results, which is why the choice may depend on the launch size; the
    option is part of the hipGraph key.  The T path's block order (queries by smallest key) is a speed-only option: same results without it."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    eng = HeadEngine(sd, kind, dev, num_views=prob['views_per_frame'])
    grouped = None
    if kind == 'T':
        # round 6: the opt-in shared-tile cross attention (csrc/xattn_group.hip) visits the keys of a softmax row in the order of its group's union:
        # equal to the per-query kernels to fp32 rounding, not bitwise; eager and graph replay agree bitwise
        eng.group_xattn = True
        grouped = {k: v.clone() for k, v in eng.run(feat, props, prob['img_metas']).items() if k in ('cls', 'reg')}
        g2 = eng.run(feat, props, prob['img_metas'], use_graph=True)
        assert torch.equal(g2['cls'], grouped['cls']) and torch.equal(g2['reg'], grouped['reg'])
        eng.group_xattn = None
    out = eng.run(feat, props, prob['img_metas'])
    o2 = eng.run(feat, props, prob['img_metas'], use_graph=True)
    assert torch.equal(o2['cls'], out['cls'])
    assert eng.fuse_maps is None and out['R'] <= 512
    if grouped is not None:
        for k in ('cls', 'reg'):
            err = float((grouped[k] - out[k]).abs().max()) / float(out[k].abs().max())
            assert err < 2e-5, (k, err)
    ref = {k: out[k].clone() for k in ('cls', 'reg')}
    for forced in (False, True):
        eng.fuse_maps = forced
        sep = eng.run(feat, props, prob['img_metas'], use_graph=True)                # (a new graph: the option is in the key)
        assert torch.equal(sep['cls'], ref['cls']) and torch.equal(sep['reg'], ref['reg'])
    eng.fuse_maps = None
    if kind == 'T':
        assert eng.q_order
        eng.q_order = False
        nat = eng.run(feat, props, prob['img_metas'])
        assert torch.equal(nat['cls'], ref['cls']) and torch.equal(nat['reg'], ref['reg'])


@pytest.mark.parametrize('kind,name', [('S', 'nc6_s'), ('T', 'cfg1_t')])
def test_engine_route_options(kind, name):
    """The remaining attributes of HeadEngine that select a route (DESIGN.md "Switches"), each against the default setting on the same
    inputs: xattn_waves (waves per query of the tile kernel: the tiles of a row are dealt to the waves and their partial softmax sums merged at
    the end, so the summation ORDER depends on it -- fp32 rounding differences only), keep_sine_rows (the training route's extra output of pe_inputs: inference results untouched), exact_skip on the
    index-exact route (stages left at the default route's single rounding: class logits stay within the default route's bound of the
    full index-exact result), force_nc (bench only, S path: every query lists exactly n_c RoIs = 49 n_c keys)."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    mk = lambda **kw: HeadEngine(sd, kind, dev, num_views=prob['views_per_frame'], **kw)
    eng = mk()
    ref = eng.run(feat, props, prob['img_metas'])
    ref = {k: ref[k].clone() for k in ('cls', 'reg')}
    for nw in (1, 4):
        eng.xattn_waves = nw
        got = eng.run(feat, props, prob['img_metas'], use_graph=True)
        e = max(relmax(got['cls'], ref['cls']), relmax(got['reg'], ref['reg']))
        print(f'[xattn_waves {name}] {nw} vs 2 waves: {e:.2e}')
        assert e < 2e-5, (nw, e)                                                       # measured 6.6e-6 (nc6_s, 1 vs 2 waves)
    eng.xattn_waves = 2
    eng.keep_sine_rows = True
    got = eng.run(feat, props, prob['img_metas'], use_graph=True)
    assert torch.equal(got['cls'], ref['cls']) and torch.equal(got['reg'], ref['reg'])
    ws = eng._ws[next(iter(eng._ws))]
    a2 = ws['A2'][:int(ws['S_dev'].item())].float()                                  # sine / cosine rows of the listed positions
    assert a2.shape[0] > 0 and bool(torch.isfinite(a2).all()) and 0.5 < float(a2.abs().max()) <= 1.0
    # index-exact route: full split precision vs. the shipped setting (conv at single precision) vs. everything skipped (= the default route's arithmetic)
    ex = mk(exact=True)
    assert ex.exact_skip == frozenset({'conv'})
    shipped = ex.run(feat, props, prob['img_metas'])['cls'].clone()
    ex.exact_skip = frozenset()
    full = ex.run(feat, props, prob['img_metas'])['cls'].clone()
    ex.exact_skip = frozenset({'conv', 'pe', 'attn'})
    none = ex.run(feat, props, prob['img_metas'])['cls'].clone()
    e_ship, e_none = relmax(shipped, full), relmax(none, full)
    print(f'[exact_skip {name}] cls vs full split precision: conv skipped {e_ship:.2e}, all skipped {e_none:.2e}, default route {relmax(ref["cls"], full):.2e}')
    assert e_ship < 1e-5 and e_none < TOL['cls']
    if kind == 'S':
        f = mk()
        f.force_nc = 3
        out = f.run(feat, props, prob['img_metas'])
        rp = f._ws[next(iter(f._ws))]['row_ptr'].cpu().numpy()
        R = out['R']
        assert (np.diff(rp[:R + 1]) == 3 * 49).all()


@pytest.mark.parametrize('kind,name', [('S', 'nc6_s'), ('S', 'cfg2_s'), ('T', 'cfg1_t')])
def test_round6_storage_options(kind, name):
    """The two storage choices of round 6 against their predecessors on the same inputs.  pe_at_positions (S path: the PE rows written at their map
    positions, RoIAlign reads them without the position -> row table): BITWISE the results of the compacted rows, also when the map still holds the rows
    of another frame (stale rows only ever meet weight 0).  lo8_rows (the lo halves of the key / value rows as e4m3 bytes instead of fp16): class logits
    within 1e-6 of their scale, the same ranked (query, class) indices."""
    from mv2d_amd.engine import HeadEngine
    dev = torch.device('cuda:0')
    sd = synthetic.make_head_state(seed=0)
    probs = [synthetic.make_problem(name, seed=s_) for s_ in (0, 3)]
    mk = lambda **kw: HeadEngine(sd, kind, dev, num_views=probs[0]['views_per_frame'], **kw)
    keys = ('cls', 'reg', 'boxes', 'scores', 'labels', 'bbox_index', 'count')

    def frames(eng, order):
        outs = []
        for i in order:
            p_ = probs[i]
            o = eng.run(torch.from_numpy(p_['feat']).to(dev), [torch.from_numpy(x) for x in p_['proposals']], p_['img_metas'])
            torch.cuda.synchronize()
            outs.append({k: o[k].clone() for k in keys})
        return outs
    a = mk()
    assert a.exact and a.lo8_rows and a.pe_at_positions
    ra = frames(a, (0, 1, 0))
    ws = a._ws[next(iter(a._ws))]
    assert ws['xk_lo'].dtype == torch.uint8 and (ws['pe_pos'] is not None) == (kind == 'S')
    for k in keys:
        assert torch.equal(ra[0][k], ra[2][k]), k                                       # the first frame again, after another one went through the same buffers
    b = mk()
    b.pe_at_positions = False
    rb = frames(b, (0, 1))
    for i in (0, 1):
        for k in keys:
            assert torch.equal(ra[i][k], rb[i][k]), (i, k)
    c = mk()
    c.lo8_rows = False
    rc = frames(c, (0, 1))
    assert c._ws[next(iter(c._ws))]['xk_lo'].dtype == ops_key16()
    for i in (0, 1):
        e = relmax(ra[i]['cls'], rc[i]['cls'])
        n = int(rc[i]['count'].item())
        print(f'[lo8 rows vs fp16 lo rows] {name} frame {i}: cls {e:.2e}, ranked indices equal: {bool(torch.equal(ra[i]["bbox_index"][:n], rc[i]["bbox_index"][:n]))}')
        assert e < 1e-6, e
        assert int(ra[i]['count'].item()) == n and torch.equal(ra[i]['bbox_index'][:n], rc[i]['bbox_index'][:n]) and torch.equal(ra[i]['labels'][:n], rc[i]['labels'][:n])


@pytest.mark.parametrize('kind,name', [('S', 'cfg1_s'), ('T', 'cfg1_t')])
def test_lo8_saturation_is_reported(kind, name):
    """Feature values beyond +-224 do not fit the fixed scale of the e4m3 lo rows (csrc/common.h "lo8"): those elements keep their hi halves only.  The row
    producers raise a device flag, the synchronising accessor turns it into a RuntimeWarning; frames inside the range raise none."""
    import warnings
    from mv2d_amd.engine import HeadEngine
    dev = torch.device('cuda:0')
    prob = synthetic.make_problem(name, seed=0)
    eng = HeadEngine(synthetic.make_head_state(seed=0), kind, dev, num_views=prob['views_per_frame'])
    props = [torch.from_numpy(p) for p in prob['proposals']]
    feat = torch.from_numpy(prob['feat']).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        eng.results(eng.run(feat, props, prob['img_metas']))
    with pytest.warns(RuntimeWarning, match='lo halves saturated'):
        eng.results(eng.run(feat * 400.0, props, prob['img_metas']))
    with warnings.catch_warnings():                                                     # the flag is per frame
        warnings.simplefilter('error')
        eng.results(eng.run(feat, props, prob['img_metas']))


def ops_key16():
    from mv2d_amd import ops
    return ops.key16_dtype()


@pytest.mark.parametrize('kind,name,n', [('S', 'cfg1_s', 1), ('S', 'nc6_s', 3), ('T', 'cfg1_t', 2)])
def test_decode_writes_the_all_gather_payload(kind, name, n):
    """run(..., payload=buf): the decode kernel writes the wire rows of the per-step all-gather itself (one launch less per frame); they equal
    what mv2d_pack_detections makes of the decoded boxes, also under hipGraph replay (the buffer's address is part of the graph key)."""
    from mv2d_amd import ops
    from mv2d_amd.engine import HeadEngine
    dev = torch.device('cuda:0')
    sd = synthetic.make_head_state(seed=0)
    probs = [synthetic.make_problem(name, seed=s) for s in range(n)]
    eng = HeadEngine(sd, kind, dev, num_views=probs[0]['views_per_frame'])
    feats = [torch.from_numpy(p['feat']).to(dev) for p in probs]
    props = [[torch.from_numpy(x) for x in p['proposals']] for p in probs]
    metas = [p['img_metas'] for p in probs]
    for use_graph in (False, True):
        pay = torch.full((n, 300 * 11 + 1), -7.0, device=dev)
        out = eng.run_batch(feats, props, metas, use_graph=use_graph, payload=pay) if n > 1 else eng.run(feats[0], props[0], metas[0], use_graph=use_graph, payload=pay)
        ref = torch.empty_like(pay)
        boxes, scores, labels = (out[k] if n > 1 else out[k][None] for k in ('boxes', 'scores', 'labels'))
        ops.pack_detections(boxes.contiguous(), scores.contiguous(), labels.contiguous(), out['count'], ref)
        torch.cuda.synchronize()
        assert torch.equal(pay, ref) and int(out['count'].sum()) > 0
    plain = eng.run_batch(feats, props, metas) if n > 1 else eng.run(feats[0], props[0], metas[0])       # without a payload: same decoded boxes
    assert torch.equal(plain['boxes'], out['boxes'])


def test_engine_stays_finite_when_features_exceed_the_fp16_range():
    """Range guard of the fp16 key side (csrc/common.h): feature values beyond +-65504 SATURATE in the key / value rows and RoI cells instead of
    becoming inf -- every output of the frame stays finite (an inf in a key row would turn its softmax rows into NaN and, through the next self
    attention, the whole frame).  The index-exact route carries the same guard in its hi + lo pairs."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem('micro_t', seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    feat = torch.from_numpy(prob['feat']).to(dev) * 4.0e4                             # |values| up to ~1.6e5
    assert float(feat.abs().max()) > 65504.0
    props = [torch.from_numpy(p) for p in prob['proposals']]
    for exact in (False, True):
        eng = HeadEngine(sd, 'T', dev, num_views=prob['views_per_frame'], exact=exact)
        out = eng.run(feat, props, prob['img_metas'], keep_stages=True)
        torch.cuda.synchronize()
        R = out['R']
        assert bool(torch.isfinite(out['cls'][:, :R]).all()) and bool(torch.isfinite(out['reg'][:, :R]).all()), exact
        S = int(out['stages']['S_dev'])
        assert bool(torch.isfinite(out['stages']['Xk'][:S].float()).all()) and float(out['stages']['Xk'][:S].float().abs().max()) == 65504.0


@pytest.mark.parametrize('name', ['cfg1_t', 'cfg1_s'])
def test_layer0_self_attention_fold(name):
    """HeadEngine.fold_sa0 (round 5): the decoder starts from target = 0, so the value rows of layer 0's self attention all equal the value
    bias and its context is that bias for every query -- the engine feeds it to the out-projection directly.  Against the launched in-projection +
    attention core (whose softmax rows sum to 1 +- 1e-7): class logits within 1e-6 of their scale, identical decoded indices; also as a batch
    and under hipGraph replay."""
    from mv2d_amd.engine import HeadEngine
    dev = torch.device('cuda:0')
    prob = synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=0)
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    outs = {}
    for fold in (True, False):
        eng = HeadEngine(sd, prob['kind'], dev, num_views=prob['views_per_frame'])
        eng.fold_sa0 = fold
        o = eng.run(feat, props, prob['img_metas'])
        torch.cuda.synchronize()
        outs[fold] = {k: o[k].clone() for k in ('cls', 'reg', 'bbox_index', 'labels', 'count')}
        if fold:
            g = eng.run(feat, props, prob['img_metas'], use_graph=True)
            torch.cuda.synchronize()
            assert torch.equal(g['cls'], outs[True]['cls']) and torch.equal(g['bbox_index'], outs[True]['bbox_index'])
    a, b = outs[True], outs[False]
    err = float((a['cls'] - b['cls']).abs().max() / b['cls'].abs().max())
    print(f'[fold_sa0] {name}: cls deviation {err:.1e}')
    assert err < 1e-6
    n = int(b['count'].item())
    assert int(a['count'].item()) == n and torch.equal(a['bbox_index'][:n], b['bbox_index'][:n]) and torch.equal(a['labels'][:n], b['labels'][:n])


@pytest.mark.parametrize('name', ['micro_t', 'micro_s'])
def test_poisoned_feature_cell_stays_visible(name):
    """A NaN in the feature map (bad input frame) must reach the outputs like it does in the reference (torch propagates NaN through conv / linear /
    attention): the key / value rows, RoI cells and the PE gate read that cell, so the queries that attend to it -- and, through the next self
    attention, the whole frame -- become NaN.  Both modes: the index-exact route (NaN-preserving ReLU + saturating conversions that keep NaN in
    every split) and the opt-in key16 mode (its PE hidden layer turns a NaN into 0 -- common.h pack_k16x2_relu -- but the FEATURE ROW itself carries
    the NaN into the key / value rows)."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    props = [torch.from_numpy(p) for p in prob['proposals']]
    b = prob['proposals'][0][0]
    cx, cy = int((b[0] + b[2]) / 2 / 16), int((b[1] + b[3]) / 2 / 16)              # a cell inside the first RoI of view 0
    feat = torch.from_numpy(prob['feat']).to(dev).clone()
    feat[0, :, cy, cx] = float('nan')
    for exact in (True, False):
        eng = HeadEngine(sd, prob['kind'], dev, num_views=prob['views_per_frame'], exact=exact)
        out = eng.run(feat, props, prob['img_metas'])
        torch.cuda.synchronize()
        R = out['R']
        assert bool(torch.isnan(out['cls'][-1, :R]).any()), (name, exact, 'the poisoned cell did not reach the last layer')
        assert int(out['count'].item()) < 10 * R


def test_engine_full_size_properties_cfg5():
    """BASELINE.json's largest configuration (R101 1600x640, 12 views, 900 queries) through size-independent properties:
    CSR well-formedness, idempotence (same frame twice -> bitwise identical), fork/no-fork equality, and the decode kernel
    against a host re-computation of top-k on the engine's own logits (bit-exact integers)."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem('cfg5_t', seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    eng = HeadEngine(sd, 'T', dev, num_views=6)
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    out = eng.run(feat, props, prob['img_metas'], keep_stages=True)
    st = out['stages']
    R = out['R']
    assert R == 900
    S, nnz, ovf = int(st['S_dev']), int(st['nnz'][0]), int(st['nnz'][1])
    assert ovf == 0 and 0 < S <= 12 * 40 * 100 and nnz > R
    rp = st['row_ptr'].cpu().long()
    col = st['col_idx'][:nnz].cpu().long()
    assert rp[0] == 0 and rp[-1] == nnz and bool((rp[1:] >= rp[:-1]).all())
    assert int(col.min()) >= 0 and int(col.max()) < S
    seg = torch.repeat_interleave(torch.arange(R), rp[1:] - rp[:-1])
    same_row = seg[1:] == seg[:-1]
    assert bool((col[1:][same_row] > col[:-1][same_row]).all())                    # strictly ascending keys inside every row
    s2 = st['s2pos'][:S].cpu().long()
    assert bool((s2[1:] > s2[:-1]).all())                                          # key list in row-major (v,y,x) order
    assert torch.equal(st['pos2s'].cpu().long()[s2], torch.arange(S))
    assert bool(torch.isfinite(out['cls']).all()) and bool(torch.isfinite(out['reg']).all())
    c1, r1 = out['cls'].clone(), out['reg'].clone()
    b1 = [t.clone() for t in eng.results(out)]
    # decode == host top-k on the same logits (integers bit-exact)
    cl = c1[-1].cpu()
    idx = torch.argsort(cl.view(-1), descending=True, stable=True)[:300]          # ties (9000 fp32 logits: ~1 expected): lower flat index first,
    scores = cl.view(-1)[idx]                                                      # the order the decode kernel defines
    n = int(out['count'].item())
    bp = r1[-1].cpu()[idx // 10]
    keep = ((bp[:, 0].abs() <= 61.2) & (bp[:, 1].abs() <= 61.2) & (bp[:, 4].abs() <= 10.0))
    assert n == int(keep.sum())
    assert torch.equal(out['labels'][:n].cpu(), (idx % 10)[keep])
    assert torch.equal(out['bbox_index'][:n].cpu(), (idx // 10)[keep])
    # idempotence + the two-stream fork is only a schedule: results identical
    eng.fork_qg = False
    out2 = eng.run(feat, props, prob['img_metas'])
    torch.cuda.synchronize()
    assert torch.equal(c1, out2['cls']) and torch.equal(r1, out2['reg'])
    for a, b in zip(b1, eng.results(out2)):
        assert torch.equal(a, b)
    # hipGraph replay == eager
    eng.fork_qg = True
    out3 = eng.run(feat, props, prob['img_metas'], use_graph=True)
    out3 = eng.run(feat, props, prob['img_metas'], use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(c1, out3['cls']) and torch.equal(r1, out3['reg'])


def test_stream_planner_returns_concurrent_streams():
    """mv2d_amd.streams: the chosen streams overlap pairwise (spin chains finish in ~the time of one chain)."""
    import time
    from mv2d_amd import _lib, streams
    lib = _lib.load()
    pool = streams.concurrent_streams(4, 'cuda:0')
    assert 1 <= len(pool) <= 4 and len({s.cuda_stream for s in pool}) == len(pool)

    def timed(group):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in group:
            for _ in range(10):
                lib.mv2d_spin(40, s.cuda_stream)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    single = min(timed(pool[:1]) for _ in range(3))
    assert single > 10 * 40e-6 * 0.9                                   # the spin kernel really spins
    if len(pool) >= 2:
        assert min(timed(pool[:2]) for _ in range(3)) < 1.5 * single


@pytest.mark.parametrize('kind', ['t', 's'])
def test_engine_ragged_views(kind):
    """Ragged input: a view without any 2-D detection beside views with detections, a single box in another, and a box that leaves
    the image — against the oracle (the reference's bbox2roi skips empty views; RoI indices shift accordingly)."""
    prob = synthetic.make_problem('cfg1_' + kind, seed=0)
    props = [np.asarray(p).copy() for p in prob['proposals']]
    props[0] = props[0][:0]                                   # no detection in view 0
    props[1] = props[1][:7]
    props[1][3, :4] = [350.0, 180.0, 430.0, 240.0]            # sticks out of the 400x224 image into the padding
    if len(props) > 2:
        props[2] = props[2][:1]
    prob = dict(prob, proposals=props)
    eng, out, st = run_both(None, prob)
    s, R = out['stages'], out['R']
    assert R == sum(len(p) for p in props) == st['rois'].shape[0]
    assert torch.equal(s['rois'][:R].cpu(), st['rois'])
    errs = dict(center=relmax(s['center'][:R], st['center_pred']), ref=relmax(s['ref'][:R], st['ref']),
                cls=relmax(out['cls'][:, :R], st['cls']), reg=relmax(out['reg'][:, :R], st['reg']))
    for k, v in errs.items():
        assert v < TOL[k], (k, v)
    assert check_topk(out, st, eng) > 0.9


@pytest.mark.parametrize('kind', ['S', 'T'])
def test_sine_table_with_padded_views_and_mixed_geometry_batches(kind):
    """adapt_pos3d(sine) is read from a per-(weights, geometry) table (LOG.md section 8).  With a padded view the PE rows still equal the
    oracle's (which evaluates the branch per frame like the reference), and a batch whose samples differ in padding geometry (one table row
    per position of the whole batch then) == the two single runs, bitwise."""
    from mv2d_amd import engine as E
    from oracle import mv2d_oracle as O
    sd_np = synthetic.make_head_state(seed=0)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    name = 'cfg1_s' if kind == 'S' else 'cfg1_t'
    prob = synthetic.make_problem(name, seed=0)
    nv = prob['views_per_frame']
    b = E.HeadEngine(sd, kind, 'cuda', num_views=nv)
    feat_h = torch.from_numpy(prob['feat'])
    feat = feat_h.cuda()
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = prob['img_metas']
    narrow = [dict(m, img_shape=(m['img_shape'][0], m['img_shape'][1] - 48, 3)) for m in metas]      # a padded strip on the right
    for ms in (metas, narrow):
        ob = b.run(feat, props, ms, keep_stages=True)
        st = {}
        (O.forward_t if kind == 'T' else O.forward_s)(sd_np, feat_h, props, ms, stages=st, **({'num_views': nv} if kind == 'T' else {}))
        S = int(ob['stages']['S_dev'])
        pos = ob['stages']['s2pos'][:S].cpu().long()
        pe_ref = st['pe'].permute(0, 2, 3, 1).reshape(-1, 256)[pos]
        assert relmax(ob['stages']['pe'][:S], pe_ref) < TOL['pe']
        assert relmax(ob['cls'], st['cls']) < TOL['cls']
    # a batch of two samples with different padding geometry == the two single runs (bitwise, as for the default path)
    single = [b.run(feat, props, ms)['cls'].clone() for ms in (metas, narrow)]
    R = single[0].shape[1]
    ob = b.run_batch([feat, feat], [props, props], [metas, narrow])
    assert torch.equal(ob['cls'][:, :R], single[0][:, :R]) and torch.equal(ob['cls'][:, R:2 * R], single[1][:, :R])


@pytest.mark.parametrize('kind', ['S', 'T'])
def test_sine_table_follows_the_weights(kind):
    """The folded sine branch is keyed on the weights as well as on the geometry: load_state() with other adapt_pos3d weights must give
    exactly what a fresh engine with those weights gives (hipGraph replay included), and never the stale table."""
    from mv2d_amd import engine as E
    sd = {k: torch.from_numpy(v).clone() for k, v in synthetic.make_head_state(seed=0).items()}
    name = 'cfg1_s' if kind == 'S' else 'cfg1_t'
    prob = synthetic.make_problem(name, seed=0)
    nv = prob['views_per_frame']
    feat = torch.from_numpy(prob['feat']).cuda()
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = prob['img_metas']
    eng = E.HeadEngine(sd, kind, 'cuda', num_views=nv)
    first = eng.run(feat, props, metas, use_graph=True)['cls'].clone()
    sd2 = dict(sd)
    for k in ('position_encoding.adapt_pos3d.0.weight', 'position_encoding.adapt_pos3d.2.weight', 'position_encoding.adapt_pos3d.2.bias'):
        sd2[k] = sd[k] * 1.5 + 0.01
    eng.load_state(sd2)
    again = eng.run(feat, props, metas, use_graph=True)['cls'].clone()
    fresh = E.HeadEngine(sd2, kind, 'cuda', num_views=nv).run(feat, props, metas)['cls']
    assert torch.equal(again, fresh)
    assert not torch.equal(again, first)



@pytest.mark.parametrize('name,kind', [('cfg1_s', 'S'), ('cfg1_t', 'T')])
def test_last_stage_heads_option(name, kind):
    """HeadEngine.last_stage_heads (opt-in): only the branches of the last decoder layer are evaluated; the decoded result and the last
    layer's logits / box codes are bitwise those of the default (all six layers, like the reference's forward)."""
    from mv2d_amd.engine import HeadEngine
    prob = synthetic.make_problem(name, seed=0)
    sd = synthetic.make_head_state(seed=0)
    dev = torch.device('cuda:0')
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    a = HeadEngine(sd, kind, dev, num_views=prob['views_per_frame'])
    b = HeadEngine(sd, kind, dev, num_views=prob['views_per_frame'])
    b.last_stage_heads = True
    for use_graph in (False, True):
        oa = a.run(feat, props, prob['img_metas'], use_graph=use_graph)
        ob = b.run(feat, props, prob['img_metas'], use_graph=use_graph)
        for k in ('boxes', 'scores', 'labels', 'bbox_index', 'count'):
            assert torch.equal(oa[k], ob[k]), k
        assert torch.equal(oa['cls'][-1], ob['cls'][-1]) and torch.equal(oa['reg'][-1], ob['reg'][-1])
    ok = b.run(feat, props, prob['img_metas'], keep_stages=True)                  # a keep_stages run evaluates all layers again
    assert torch.equal(ok['cls'], oa['cls'])


@pytest.mark.parametrize('name,n', [('micro_t', 1), ('cfg1_t', 3), ('cfg3_t', 2), ('nc6_s', 3), ('cfg1_s', 2)])
def test_query_order_of_the_attention_blocks(name, n):
    """The launch order of the per-query attention blocks: T path -- mv2d_xattn_query_order ranks the queries of every sample by their smallest
    key (csrc/xattn_order.hip); S path -- by the smallest RoI they list (own or matched; from the correlation lists, inside the CSR launch), with the
    rows that list more RoIs first when matched RoIs are rare in the sample.  The tile kernel launched in that order gives bitwise the rows of the natural order (the order only decides which blocks share an L2)."""
    from mv2d_amd import ops
    from mv2d_amd.engine import HeadEngine
    dev = torch.device('cuda:0')
    sd = synthetic.make_head_state(seed=0)
    probs = [synthetic.make_problem(name, seed=s) for s in range(n)]
    kind = probs[0]['kind']
    eng = HeadEngine(sd, kind, dev, num_views=probs[0]['views_per_frame'])
    feats = [torch.from_numpy(p['feat']).to(dev) for p in probs]
    props = [[torch.from_numpy(x) for x in p['proposals']] for p in probs]
    metas = [p['img_metas'] for p in probs]
    out = eng.run_batch(feats, props, metas) if n > 1 else eng.run(feats[0], props[0], metas[0])
    torch.cuda.synchronize()
    ws = out['ws']
    Rl = ws['x'].shape[0]                                        # rows of the launch (RoI-count bucket)
    rp, ci = ws['row_ptr'][:Rl + 1].cpu().numpy(), ws['col_idx'].cpu().numpy()
    perm = ws['q_order'].cpu().numpy()
    grp = list(ws['grp_start_h'].numpy()) + [Rl]
    assert sorted(perm[:Rl].tolist()) == list(range(Rl)) and int(ws['qt_ctl'][1].item()) == 0
    for a, b in zip(grp[:-2], grp[1:-1]):                        # the order stays inside a sample and is ascending in the smallest key / RoI
        assert sorted(perm[a:b].tolist()) == list(range(a, b))
        if kind == 'T':
            firsts = [ci[rp[r]] if rp[r + 1] > rp[r] else 2 ** 31 - 1 for r in perm[a:b]]
        else:
            # S path (round 5): a sample with fewer than one matched RoI per two queries is ordered (more RoIs listed first, then the smallest
            # listed RoI), otherwise by the smallest listed RoI alone
            ncs = [(rp[r + 1] - rp[r]) // 49 for r in range(a, b)]
            length_major = 2 * (sum(ncs) - (b - a)) < (b - a)
            firsts = [((-((rp[r + 1] - rp[r]) // 49)) if length_major else 0, int(ci[rp[r]:rp[r + 1]:49].min()) // 49) for r in perm[a:b]]
        assert firsts == sorted(firsts)
    assert perm[grp[-2]:Rl].tolist() == list(range(grp[-2], Rl))        # bucket-padding rows keep their places
    # (the S path runs the one-launch cross attention, which keeps Qt on chip: map the last layer's query rows here for the three-kernel form)
    ops.xattn_qmap(ws['q'], eng.w[f'ca_mapA{eng.L - 1}'], ws['Qt'], R=Rl)
    z_t = ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], R=Rl, empty_nan=False, waves=2)
    z_o = ops.xattn_tile(ws['Qt'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], R=Rl, empty_nan=False, waves=2, order=ws['q_order'])
    assert torch.equal(z_o, z_t)
    if kind == 'S':
        c_t = ops.xattn_fused(ws['q'], eng.w['ca_mapA0'], eng.w['ca_mapB0'], eng.w['ca_v_b0'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], R=Rl,
                              Xk_lo=ws['xk_lo'], Xv_lo=ws['xv_lo'])
        c_o = ops.xattn_fused(ws['q'], eng.w['ca_mapA0'], eng.w['ca_mapB0'], eng.w['ca_v_b0'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], R=Rl,
                              Xk_lo=ws['xk_lo'], Xv_lo=ws['xv_lo'], order=ws['q_order'])
        assert torch.equal(c_o.view(torch.int32), c_t.view(torch.int32))
    eng2 = HeadEngine(sd, kind, dev, num_views=probs[0]['views_per_frame'])
    eng2.q_order = False
    out2 = eng2.run_batch(feats, props, metas) if n > 1 else eng2.run(feats[0], props[0], metas[0])
    assert torch.equal(out2['cls'], out['cls']) and torch.equal(out2['boxes'], out['boxes'])


def test_t_head_with_expand_stride_0_reads_transposed_rows():
    """BoxCorrelation's default expand_stride = 0 (the shipped T configs use 2): a bilinear RoIAlign tap may then lie one cell OUTSIDE its RoI's rectangle, i.e.
    outside roi_mask, so the masked transposition of the feature map (round 5) must stay off on the T path (round-6 fix of an advisor finding: it read rows that
    this frame never wrote).  The engine first runs another frame on the same workspace (stale rows everywhere), then the frame under test; its integer outputs
    and class logits must be those of the oracle with the same expand_stride."""
    from mv2d_amd.engine import HeadEngine
    from oracle import mv2d_oracle as O
    dev = torch.device('cuda:0')
    prob = synthetic.make_problem('cfg1_t', seed=0)
    other = synthetic.make_problem('cfg1_t', seed=7)
    sd = synthetic.make_head_state(seed=0)
    eng = HeadEngine(sd, 'T', dev, num_views=prob['views_per_frame'], expand_stride=0)
    tp = lambda pr: [torch.from_numpy(np.asarray(p)) for p in pr['proposals']]      # noqa: E731
    eng.run(torch.from_numpy(other['feat'] * 50.0).to(dev), tp(other), other['img_metas'])
    out = eng.run(torch.from_numpy(prob['feat']).to(dev), tp(prob), prob['img_metas'])
    torch.cuda.synchronize()
    st = {}
    O.forward_t(sd, torch.from_numpy(prob['feat']), tp(prob), prob['img_metas'], num_views=prob['views_per_frame'], expand_stride=0, stages=st)
    R = out['R']
    ref_cls = st['cls'].reshape(out['cls'][:, :R].shape)
    err = float((out['cls'][:, :R].cpu() - ref_cls).abs().max() / ref_cls.abs().max())
    assert err < 1e-5, err
    n = int(out['count'].item())
    assert n == st['labels'].numel()
    assert torch.equal(out['labels'][:n].cpu(), st['labels']) and torch.equal(out['bbox_index'][:n].cpu(), st['bbox_index'])
