"""Seeded synthetic inputs and weights for the MV2D hot path.

Everything here is platform-stable (numpy PCG64 by seed) so that fixtures, tests, the oracle and
bench.py regenerate identical inputs/weights on any box instead of storing them.

* camera rig: ring of yaw-spaced pinhole cameras, f = 0.8*W, principal point at the image centre,
  ``extrinsics`` stored as the TRANSPOSED lidar->cam matrix and ``lidar2img = intrinsics @ extrinsics.T``
  (the img_metas contract of mmdet3d_plugin/datasets/custom_nuscenes_dataset.py:141-150 and
  mmdet3d_plugin/datasets/pipelines/transform_3d.py:587-591); a previous frame is the same rig
  translated by 0.4 m with timestamp 0.5 s.
* proposals: per view ``n`` boxes (x1,y1,x2,y2,score,label) like mmdet3d_plugin/models/detectors/mv2d.py:60-86.
* weights: state-dict keyed exactly like the reference ``roi_head`` sub-module.
"""
import math
from collections import OrderedDict

import numpy as np

EMBED = 256
NUM_LAYERS = 6
FFN_DIM = 2048
NUM_CLASSES = 10
CODE_SIZE = 10
DEPTH_NUM = 64


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def make_img_metas(views_per_frame, img_h, img_w, frames=1, pad_w=None, pad_h=None, yaw_step_deg=None, ego=0.0, baseline=0.0):
    """Return a list of per-view img_meta dicts (len = views_per_frame*frames).  ego: extra ego motion between the frames (metres along
    the driving direction) and time-stamp jitter of the previous frame -- every real two-frame sample has its own."""
    pad_w = img_w if pad_w is None else pad_w
    pad_h = img_h if pad_h is None else pad_h
    metas = []
    nv = views_per_frame * frames
    for f in range(frames):
        for v in range(views_per_frame):
            # full ring by default; rigs with < 6 cameras use a 40 degree step so that neighbouring views
            # overlap (horizontal fov at f = 0.8 W is 64 degrees) and the epipolar correlation is exercised
            step = (2.0 * math.pi / views_per_frame) if yaw_step_deg is None else math.radians(yaw_step_deg)
            yaw = v * step
            K = np.eye(4, dtype=np.float64)
            K[0, 0] = K[1, 1] = 0.8 * img_w
            K[0, 2] = img_w / 2.0
            K[1, 2] = img_h / 2.0
            c, s = math.cos(yaw), math.sin(yaw)
            # lidar (x fwd, y left, z up) -> camera (x right, y down, z fwd), then yaw about lidar z
            R0 = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], dtype=np.float64)
            Rz = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float64)
            T = np.eye(4, dtype=np.float64)
            T[:3, :3] = R0 @ Rz
            T[:3, 3] = [baseline * v, 1.5, -0.5 - (0.4 + ego) * f]
            metas.append(dict(
                intrinsics=K,
                extrinsics=T.T.copy(),
                lidar2img=K @ T,
                pad_shape=(pad_h, pad_w, 3),
                img_shape=(img_h, img_w, 3),
                num_views=nv,
                timestamp=(0.5 + 0.02 * ego) * f,
            ))
    return metas


def make_proposals(num_views, n_per_view, img_h, img_w, seed, wh_lo=(16.0, 16.0), wh_hi=(116.0, 96.0)):
    """List of ``num_views`` float32 arrays [n,6] = (x1,y1,x2,y2,score,label)."""
    g = _rng(seed)
    if isinstance(n_per_view, int):
        n_per_view = [n_per_view] * num_views
    out = []
    for v in range(num_views):
        n = n_per_view[v]
        xy = g.random((n, 2)) * np.array([max(img_w - 120.0, 1.0), max(img_h - 100.0, 1.0)])
        wh = g.random((n, 2)) * (np.array(wh_hi) - np.array(wh_lo)) + np.array(wh_lo)
        score = g.random((n, 1))
        label = g.integers(0, NUM_CLASSES, (n, 1)).astype(np.float64)
        out.append(np.concatenate([xy, xy + wh, score, label], 1).astype(np.float32))
    return out


def make_feat(num_views, h, w, seed, channels=EMBED):
    """Stride-16 FPN map [V, 256, h, w] float32 ~ N(0,1)."""
    g = _rng(seed)
    return g.standard_normal((num_views, channels, h, w), dtype=np.float32)


def _xavier(g, shape, gain=1.0):
    fan_out = shape[0]
    fan_in = int(np.prod(shape[1:]))
    rf = 1
    if len(shape) > 2:
        rf = int(np.prod(shape[2:]))
        fan_in = shape[1] * rf
        fan_out = shape[0] * rf
    a = gain * math.sqrt(6.0 / (fan_in + fan_out))
    return ((g.random(shape, dtype=np.float32) * 2.0 - 1.0) * a).astype(np.float32)


def _bias(g, n, a=0.05):
    return ((g.random(n, dtype=np.float32) * 2.0 - 1.0) * a).astype(np.float32)


def _ln(g, n):
    w = (0.8 + 0.4 * g.random(n, dtype=np.float32)).astype(np.float32)
    b = _bias(g, n, 0.1)
    return w, b


def make_head_state(seed=0, num_layers=NUM_LAYERS):
    """OrderedDict[str, np.ndarray] with the reference ``roi_head.*`` state-dict key layout.

    Xavier-uniform matrices like PETRTransformer.init_weights
    (mmdet3d_plugin/models/utils/petr_transformer.py:65-71); biases / LayerNorm affine are small
    non-trivial values so that every bias path is exercised by the parity tests.
    """
    g = _rng(seed)
    C, F = EMBED, FFN_DIM
    sd = OrderedDict()
    dec = 'bbox_head.transformer.decoder.'
    for i in range(num_layers):
        p = f'{dec}layers.{i}.'
        for a in (0, 1):
            sd[p + f'attentions.{a}.attn.in_proj_weight'] = _xavier(g, (3 * C, C))
            sd[p + f'attentions.{a}.attn.in_proj_bias'] = _bias(g, 3 * C)
            sd[p + f'attentions.{a}.attn.out_proj.weight'] = _xavier(g, (C, C))
            sd[p + f'attentions.{a}.attn.out_proj.bias'] = _bias(g, C)
        sd[p + 'ffns.0.layers.0.0.weight'] = _xavier(g, (F, C))
        sd[p + 'ffns.0.layers.0.0.bias'] = _bias(g, F)
        sd[p + 'ffns.0.layers.1.weight'] = _xavier(g, (C, F))
        sd[p + 'ffns.0.layers.1.bias'] = _bias(g, C)
        for n in range(3):
            w, b = _ln(g, C)
            sd[p + f'norms.{n}.weight'] = w
            sd[p + f'norms.{n}.bias'] = b
    w, b = _ln(g, C)
    sd[dec + 'post_norm.weight'] = w
    sd[dec + 'post_norm.bias'] = b
    sd['bbox_head.query_embedding.0.weight'] = _xavier(g, (C, C * 3 // 2))
    sd['bbox_head.query_embedding.0.bias'] = _bias(g, C)
    sd['bbox_head.query_embedding.2.weight'] = _xavier(g, (C, C))
    sd['bbox_head.query_embedding.2.bias'] = _bias(g, C)
    for l in range(num_layers):
        p = f'bbox_head.cls_branches.{l}.'
        sd[p + '0.weight'] = _xavier(g, (C, C)); sd[p + '0.bias'] = _bias(g, C)
        sd[p + '1.weight'], sd[p + '1.bias'] = _ln(g, C)
        sd[p + '3.weight'] = _xavier(g, (C, C)); sd[p + '3.bias'] = _bias(g, C)
        sd[p + '4.weight'], sd[p + '4.bias'] = _ln(g, C)
        sd[p + '6.weight'] = _xavier(g, (NUM_CLASSES, C))
        # bias_init_with_prob(0.01) (cross_attention_head.py:193-197) plus a per-class spread
        sd[p + '6.bias'] = (np.full(NUM_CLASSES, -math.log((1 - 0.01) / 0.01), np.float32) + _bias(g, NUM_CLASSES, 0.5))
    for l in range(num_layers):
        p = f'bbox_head.reg_branches.{l}.'
        sd[p + '0.weight'] = _xavier(g, (C, C)); sd[p + '0.bias'] = _bias(g, C)
        sd[p + '2.weight'] = _xavier(g, (C, C)); sd[p + '2.bias'] = _bias(g, C)
        sd[p + '4.weight'] = _xavier(g, (CODE_SIZE, C)); sd[p + '4.bias'] = _bias(g, CODE_SIZE, 0.3)
    sd['bbox_head.code_weights'] = np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.5, 1.5, 2.0, 2.0], np.float32)
    # query generator (mmdet3d_plugin/models/roi_heads/utils/query_generator.py)
    q = 'query_generator.'
    sd[q + 'shared_convs.0.conv.weight'] = _xavier(g, (C, C, 3, 3))
    sd[q + 'shared_convs.0.conv.bias'] = _bias(g, C)
    sd[q + 'shared_fcs.0.weight'] = _xavier(g, (1024, C))
    sd[q + 'shared_fcs.0.bias'] = _bias(g, 1024)
    sd[q + 'extra_enc.0.weight'] = _xavier(g, (512, 1024 + 16))
    sd[q + 'extra_enc.0.bias'] = _bias(g, 512)
    sd[q + 'extra_enc.2.weight'] = _xavier(g, (C, 512))
    sd[q + 'extra_enc.2.bias'] = _bias(g, C)
    # (u, v, depth) in the 7x7 RoI frame: centre the prediction inside the RoI at ~25 m
    wc = _xavier(g, (3, C))
    wc[:2] *= 4.0
    wc[2] *= 24.0
    sd[q + 'fc_center.weight'] = wc
    sd[q + 'fc_center.bias'] = np.array([3.5, 3.5, 25.0], np.float32)
    # 3D position-aware key embedding (mmdet3d_plugin/models/utils/pe.py:50-82)
    pe = 'position_encoding.'
    sd[pe + 'position_encoder.0.weight'] = _xavier(g, (4 * C, 3 * DEPTH_NUM, 1, 1))
    sd[pe + 'position_encoder.0.bias'] = _bias(g, 4 * C)
    sd[pe + 'position_encoder.2.weight'] = _xavier(g, (C, 4 * C, 1, 1))
    sd[pe + 'position_encoder.2.bias'] = _bias(g, C)
    sd[pe + 'adapt_pos3d.0.weight'] = _xavier(g, (4 * C, C * 3 // 2, 1, 1))
    sd[pe + 'adapt_pos3d.0.bias'] = _bias(g, 4 * C)
    sd[pe + 'adapt_pos3d.2.weight'] = _xavier(g, (C, 4 * C, 1, 1))
    sd[pe + 'adapt_pos3d.2.bias'] = _bias(g, C)
    sd[pe + 'fpe.conv_reduce.weight'] = _xavier(g, (C, C, 1, 1))
    sd[pe + 'fpe.conv_reduce.bias'] = _bias(g, C)
    sd[pe + 'fpe.conv_expand.weight'] = _xavier(g, (C, C, 1, 1))
    sd[pe + 'fpe.conv_expand.bias'] = _bias(g, C)
    return sd


# ---------------------------------------------------------------------------------------------
# The five BASELINE.json configs as concrete synthetic problems.
# ---------------------------------------------------------------------------------------------
WORKLOADS = {
    # name: (head kind, views/frame, frames, img_h, img_w, pad_w, boxes/view)
    'micro_t': ('T', 2, 1, 128, 192, None, 6),
    'micro_s': ('S', 2, 1, 128, 192, None, 6),
    'cfg1_s': ('S', 2, 1, 224, 400, 416, 25),
    'cfg1_t': ('T', 2, 1, 224, 400, 416, 25),
    'cfg2_s': ('S', 6, 1, 512, 1408, None, 50),
    'cfg3_t': ('T', 6, 2, 512, 1408, None, 25),
    'cfg5_t': ('T', 6, 2, 640, 1600, None, 75),
    # S path with MANY correlated RoIs per query (the n_c sweep): six cameras 8 degrees apart with a lateral baseline, so that
    # every view overlaps every other one and the reference's epipolar correlation (top-1 per other view) yields up to 1 + 5 RoIs per query
    'nc6_s': ('S', 6, 1, 224, 400, 416, 14),
    # the same overlapping rig at the HEADLINE size (BASELINE.json configs[1]: 6 cams 1408x512, 300 RoIs): the non-trivial S workload of the bench
    # line (round 4) -- every query reads its own RoI plus up to five matched ones instead of the 1.01 RoIs of the ring rig
    'cfg2_s_nc6': ('S', 6, 1, 512, 1408, None, 50),
    # round 6 -- the S path at the cap of SURVEY 8.0 (75 boxes per view, R = 450), and near-duplicate 2-D boxes (a fifth of every view's boxes
    # are copies of others: half of them exact, half a quarter pixel larger) at the headline S size and at the cfg-5 T size: ties and near-ties
    # in the IoU ranking of the box correlation (RH/utils/box_correlation.py:370-374)
    'cfg2_s_r450': ('S', 6, 1, 512, 1408, None, 75),
    'cfg2_s_dup': ('S', 6, 1, 512, 1408, None, 50),
    'cfg5_t_dup': ('T', 6, 2, 640, 1600, None, 75),
}
# rigs that are not a full ring: yaw step in degrees and lateral camera spacing in metres
RIG = {'nc6_s': (8.0, 0.35), 'cfg2_s_nc6': (8.0, 0.35), 'cfg2_s_dup': (8.0, 0.35)}       # (the duplicates of cfg2_s_dup on the overlapping rig: matched RoIs exist)


def make_problem(name, seed=0, with_feat=True, ego=0.0):
    """Return dict(kind, feat [V,256,h,w] f32, proposals list[V] of [n,6] f32, img_metas list[V]).  with_feat=False: no feature map
    (bench.py draws the maps of its rotating frame sets on the device); ego: see make_img_metas."""
    kind, vpf, frames, H, W, pad_w, n = WORKLOADS[name]
    pw = W if pad_w is None else pad_w
    V = vpf * frames
    yaw, base = RIG.get(name, ((40.0 if vpf < 6 else None), 0.0))
    metas = make_img_metas(vpf, H, W, frames, pad_w=pw, yaw_step_deg=yaw, ego=ego, baseline=base)
    props = make_proposals(V, n, H, W, seed + 1)
    if name.endswith('_dup'):
        # the last fifth of every view's boxes become copies of its first ones: even copies exact, odd ones a quarter pixel LARGER (x2, y2 + 0.25: a near
        # tie in the IoU ranking; a shifted copy of equal size would tie EXACTLY with a different box whenever the epipolar rectangle contains both, and
        # the reference's unstable argsort, RH/utils/box_correlation.py:370, would then decide which features a query reads)
        for v_ in range(V):
            k = n // 5
            props[v_][n - k:, :4] = props[v_][:k, :4]
            props[v_][n - k + 1::2, 2:4] += np.float32(0.25)
    feat = make_feat(V, H // 16, pw // 16, seed + 2) if with_feat else None
    return dict(kind=kind, feat=feat, proposals=props, img_metas=metas, name=name,
                views_per_frame=vpf, frames=frames)


# ---- training targets / losses: head outputs + ground truth -----------------------------------------------------------
TRAIN_CASES = {'small': (40, 7, 3), 'mid': (300, 45, 5), 'few_queries': (5, 12, 7), 'no_gt': (30, 0, 9), 'nan_velocity': (60, 11, 13)}


def make_train_case(R, G, seed, num_layers=NUM_LAYERS, num_classes=10):
    """Seeded head outputs of `num_layers` decoder layers (cls [L,R,C] logits, box [L,R,10] codes) and ground truth: `gt_bottom` [G,9]
    bottom-centre boxes (x, y, z, w, l, h, yaw, vx, vy) as the dataset stores them, `gt` [G,9] with the gravity centre (what the loss
    sees), labels, and denoising targets (`known_bboxs` [n,9] gravity-centre boxes, `known_labels` [n], label == num_classes = negative).
    The first min(R, G) queries sit near a ground-truth box so that the assignment is not arbitrary."""
    g = _rng(seed)
    gt = np.zeros((G, 9), np.float32)
    gt[:, 0:2] = g.uniform(-50, 50, (G, 2))
    gt[:, 2] = g.uniform(-3, 0, G)
    gt[:, 3:6] = g.uniform(0.5, 5.0, (G, 3))
    gt[:, 6] = g.uniform(-np.pi, np.pi, G)
    gt[:, 7:9] = g.uniform(-3, 3, (G, 2))
    if seed == 13 and G > 2:
        gt[2, 7:9] = np.nan                                  # nuScenes boxes without a velocity estimate
    labels = g.integers(0, num_classes, G).astype(np.int64)
    grav = gt.copy()
    grav[:, 2] += gt[:, 5] * 0.5
    code = np.stack([grav[:, 0], grav[:, 1], np.log(grav[:, 3]), np.log(grav[:, 4]), grav[:, 2], np.log(grav[:, 5]),
                     np.sin(grav[:, 6]), np.cos(grav[:, 6]), np.nan_to_num(grav[:, 7]), np.nan_to_num(grav[:, 8])], 1).astype(np.float32)
    box = np.zeros((num_layers, R, 10), np.float32)
    box[..., 0:2] = g.uniform(-50, 50, (num_layers, R, 2))
    box[..., 2:4] = g.uniform(-0.5, 1.5, (num_layers, R, 2))
    box[..., 4] = g.uniform(-3, 2, (num_layers, R))
    box[..., 5] = g.uniform(-0.5, 1.5, (num_layers, R))
    box[..., 6:8] = g.uniform(-1, 1, (num_layers, R, 2))
    box[..., 8:10] = g.uniform(-3, 3, (num_layers, R, 2))
    cls = g.normal(-2.0, 1.5, (num_layers, R, num_classes)).astype(np.float32)
    n = min(R, G)
    if n:
        pick = g.permutation(G)[:n]
        for l in range(num_layers):
            box[l, :n] = code[pick] + g.normal(0, 0.3 / (l + 1), (n, 10)).astype(np.float32)
            cls[l, np.arange(n), labels[pick]] += 2.0
    nk = min(R, 24)
    known = np.zeros((nk, 9), np.float32)
    known[:, 0:2] = g.uniform(-50, 50, (nk, 2))
    known[:, 2] = g.uniform(-2, 2, nk)
    known[:, 3:6] = g.uniform(0.5, 5.0, (nk, 3))
    known[:, 6] = g.uniform(-np.pi, np.pi, nk)
    known[:, 7:9] = g.uniform(-3, 3, (nk, 2))
    known_labels = g.integers(0, num_classes + 1, nk).astype(np.int64)
    if nk:
        known_labels[0] = num_classes
    return dict(cls=cls, box=box.astype(np.float32), gt_bottom=gt, gt=grav.astype(np.float32), gt_labels=labels, known_bboxs=known,
                known_labels=known_labels, dn_num_tgt=max(nk * 2 // 3, 1))


# prepare_for_dn cases: name -> (R, G, seed, denoise_scalar, denoise_noise_scale, denoise_split)
DN_CASES = {'dn_default': (12, 7, 3, 10, 1.0, 0.75), 'dn_two_frames': (40, 11, 13, 10, 1.25, 0.6), 'dn_no_noise': (9, 4, 5, 3, 0.0, 0.75),
            'dn_no_gt': (6, 0, 9, 10, 1.0, 0.75)}


def make_dn_noise(n, seed):
    """[n,3] uniform in [0,1): stands for torch.rand_like in prepare_for_dn (and, with another seed, for reference points)."""
    return _rng(seed + 7919).random((n, 3)).astype(np.float32)


# ---- nuScenes I/O contract: a synthetic info record of the dataset pkl ----------------------------------------------------
NUSC_SENSORS = ['CAM_FRONT', 'CAM_FRONT_RIGHT', 'CAM_FRONT_LEFT', 'CAM_BACK', 'CAM_BACK_LEFT', 'CAM_BACK_RIGHT']
NUSC_AUG_CONF = dict(resize_lim=(0.8, 1.0), final_dim=(512, 1408), bot_pct_lim=(0.0, 0.0), rot_lim=(0.0, 0.0), H=900, W=1600,
                     rand_flip=True)                                   # configs/mv2d/data/two_frames.py:23-31
NUSC_AUG_CONF_SMALL = dict(resize_lim=(0.8, 1.0), final_dim=(48, 128), bot_pct_lim=(0.0, 0.2), rot_lim=(-5.4, 5.4), H=90, W=160,
                           rand_flip=True)                             # same structure at 1/10 size, with rotation, for the image tests


def _rand_rotation(g):
    q, r = np.linalg.qr(g.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def make_nusc_info(seed, n_sweeps=6, incomplete_sweep=None):
    """One key-frame record shaped like the info pkl the reference's dataset reads (custom_nuscenes_dataset.py:117-150) plus the camera
    sweeps written by tools/generate_sweep_pkl.py (per sensor: data_path, timestamp in us, float32 lidar2img / intrinsics / extrinsics)."""
    g = _rng(seed)
    t0 = 1533151603547590 + int(g.integers(0, 10 ** 6))
    cams = {}
    for i, s in enumerate(NUSC_SENSORS):
        f = float(g.uniform(1100, 1300))
        cams[s] = dict(data_path=f'samples/{s}/frame_{seed}_{i}.jpg', timestamp=t0 - int(g.integers(0, 50000)),
                       sensor2lidar_rotation=_rand_rotation(g), sensor2lidar_translation=g.uniform(-2, 2, 3),
                       cam_intrinsic=np.array([[f, 0, 800 + g.uniform(-30, 30)], [0, f, 450 + g.uniform(-30, 30)], [0, 0, 1]]))
    sweeps = []
    for k in range(n_sweeps):
        sw = {}
        for i, s in enumerate(NUSC_SENSORS):
            if incomplete_sweep == k and i >= 4:
                continue
            sw[s] = dict(data_path=f'sweeps/{s}/sweep_{seed}_{k}_{i}.jpg', timestamp=t0 - (k + 1) * 83000 - int(g.integers(0, 20000)),
                         lidar2img=g.normal(size=(4, 4)).astype(np.float32), intrinsics=g.normal(size=(4, 4)).astype(np.float32),
                         extrinsics=g.normal(size=(4, 4)).astype(np.float32))
        sweeps.append(sw)
    return dict(token=f'token{seed}', lidar_path=f'samples/LIDAR_TOP/{seed}.bin', sweeps=sweeps, timestamp=t0, cams=cams)


def make_view_images(n, h, w, seed):
    g = _rng(seed + 31)
    return [g.integers(0, 256, (h, w, 3)).astype(np.float32) for _ in range(n)]


def fake_image(path, h=90, w=160):
    """A seeded uint8 image per file name (stands for reading the file in the I/O tests)."""
    import zlib
    g = _rng(zlib.crc32(path.encode()) % (2 ** 31))
    return g.integers(0, 256, (h, w, 3)).astype(np.uint8)


NUSC_CASES = {
    'test_two_frames': dict(seed=1, sweeps=dict(sweeps_num=1, to_float32=True, pad_empty_sweeps=True, sweep_range=[3, 27]),
                            conf=NUSC_AUG_CONF_SMALL, training=False, n_sweeps=20),
    'test_few_sweeps': dict(seed=2, sweeps=dict(sweeps_num=1, to_float32=True, pad_empty_sweeps=True, sweep_range=[3, 27]),
                            conf=NUSC_AUG_CONF_SMALL, training=False, n_sweeps=1),
    'test_no_sweeps_padded': dict(seed=3, sweeps=dict(sweeps_num=1, to_float32=True, pad_empty_sweeps=True, sweep_range=[3, 27]),
                                  conf=NUSC_AUG_CONF_SMALL, training=False, n_sweeps=0),
    'train_random_sweep': dict(seed=4, sweeps=dict(sweeps_num=1, to_float32=True, pad_empty_sweeps=True, test_mode=False, sweep_range=[3, 27]),
                               conf=NUSC_AUG_CONF_SMALL, training=True, n_sweeps=30),
    'train_incomplete_sweep': dict(seed=5, sweeps=dict(sweeps_num=2, to_float32=False, pad_empty_sweeps=True, test_mode=False,
                                                       sweep_range=[3, 8]),
                                   conf=NUSC_AUG_CONF_SMALL, training=True, n_sweeps=9, incomplete_sweep=5),
}


# forward_train cases: name -> (problem of make_problem, head kind, number of ground-truth boxes, seed)
FWD_TRAIN_CASES = {'train_micro_t': ('micro_t', 'T', 5, 21), 'train_cfg1_t': ('cfg1_t', 'T', 9, 22), 'train_micro_s': ('micro_s', 'S', 4, 23),
                   'train_cfg1_s': ('cfg1_s', 'S', 30, 24),
                   # the S head with denoising queries (use_denoise=True: not a shipped config, but what MV2DSHead's training branch does)
                   'train_micro_s_dn': ('micro_s', 'S+DN', 6, 25)}


def make_train_gt(G, seed):
    """Ground truth of a training sample: `gt_bottom` [G,9] bottom-centre boxes inside the point-cloud range, `gt` with the gravity
    centre, labels."""
    g = _rng(seed + 5000)
    gt = np.zeros((G, 9), np.float32)
    gt[:, 0:2] = g.uniform(-45, 45, (G, 2))
    gt[:, 2] = g.uniform(-3, 0, G)
    gt[:, 3:6] = g.uniform(0.5, 5.0, (G, 3))
    gt[:, 6] = g.uniform(-np.pi, np.pi, G)
    gt[:, 7:9] = g.uniform(-3, 3, (G, 2))
    grav = gt.copy()
    grav[:, 2] += gt[:, 5] * 0.5
    return dict(gt_bottom=gt, gt=grav, gt_labels=g.integers(0, 10, G).astype(np.int64))


def grad_probe(name, n):
    """A seeded probe vector per parameter name: gradient goldens store (norm, grad . probe) instead of the full gradient."""
    import zlib
    return _rng(zlib.crc32(name.encode()) % (2 ** 31)).normal(size=n).astype(np.float32)


class RecordingBoxes:
    """Stands for mmdet3d's LiDARInstance3DBoxes in the I/O tests: records the rotate / scale calls it receives."""

    def __init__(self):
        self.calls = []

    def rotate(self, angle):
        self.calls.append(('rotate', float(angle)))

    def scale(self, ratio):
        self.calls.append(('scale', float(ratio)))


def make_boxes_2d(num_views, seed, w=160, h=90):
    """Per-view 2-D ground truth at the raw image size: boxes [n,4] (x1, y1, x2, y2) fp64 like the COCO-style annotation arrays, labels,
    the index of the matching 3-D box, ignore boxes."""
    g = _rng(seed + 900)
    out = dict(gt_bboxes_2d=[], gt_labels_2d=[], gt_bboxes_2d_to_3d=[], gt_bboxes_ignore=[])
    for _ in range(num_views):
        n, m = int(g.integers(0, 7)), int(g.integers(0, 3))
        for key, cnt in (('gt_bboxes_2d', n), ('gt_bboxes_ignore', m)):
            xy = np.stack([g.uniform(-5, w - 10, cnt), g.uniform(-5, h - 8, cnt)], 1)
            wh = np.stack([g.uniform(2, 70, cnt), g.uniform(2, 50, cnt)], 1)
            out[key].append(np.concatenate([xy, xy + wh], 1))
        out['gt_labels_2d'].append(g.integers(0, 10, n))
        out['gt_bboxes_2d_to_3d'].append(g.integers(-1, 12, n))
    return out


def make_ann_2d_case(info, seed, n_boxes=7):
    """COCO-style 2-D annotations for the six images of an info record, consistent with a set of 3-D boxes: every annotation's
    ``bbox_cam3d`` starts with the box centre in that camera's frame.  Includes an ignored, a crowd, an out-of-image, a degenerate and an
    unknown-category annotation."""
    g = _rng(seed + 1700)
    centers = np.concatenate([g.uniform(-30, 30, (n_boxes, 2)), g.uniform(-2, 1, (n_boxes, 1))], 1)
    labels3d = g.integers(0, 10, n_boxes)
    cat_ids = [11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
    cat2label = {c: i for i, c in enumerate(cat_ids)}
    images = {}
    for _cam, c in info['cams'].items():
        r = np.linalg.inv(c['sensor2lidar_rotation'])
        t = c['sensor2lidar_translation'] @ r.T
        cam = centers @ r.T - t                                   # lidar -> camera (row vectors), as get_data_info builds lidar2cam
        anns = []
        for j in g.permutation(n_boxes)[:int(g.integers(2, n_boxes + 1))]:
            x, y = float(g.uniform(0, 1400)), float(g.uniform(0, 800))
            w, h = float(g.uniform(5, 300)), float(g.uniform(5, 200))
            anns.append(dict(bbox=[x, y, w, h], area=w * h, category_id=cat_ids[int(labels3d[j])],
                             bbox_cam3d=[*cam[j].tolist(), 1.0, 2.0, 3.0], iscrowd=0))
        anns.append(dict(bbox=[10., 10., 50., 50.], area=2500., category_id=11, bbox_cam3d=[0, 0, 0, 1, 1, 1], ignore=True))
        anns.append(dict(bbox=[100., 100., 80., 60.], area=4800., category_id=12, bbox_cam3d=[0, 0, 0, 1, 1, 1], iscrowd=1))
        anns.append(dict(bbox=[1700., 100., 80., 60.], area=4800., category_id=12, bbox_cam3d=[0, 0, 0, 1, 1, 1]))
        anns.append(dict(bbox=[200., 100., 0.5, 60.], area=30., category_id=12, bbox_cam3d=[0, 0, 0, 1, 1, 1]))
        anns.append(dict(bbox=[300., 100., 40., 60.], area=2400., category_id=99, bbox_cam3d=[0, 0, 0, 1, 1, 1]))
        order = g.permutation(len(anns))
        images[c['data_path']] = (dict(width=1600, height=900, file_name=c['data_path']), [anns[i] for i in order])
    return dict(images=images, centers_lidar=centers, gt_labels_3d=labels3d, cat_ids=cat_ids, cat2label=cat2label)
