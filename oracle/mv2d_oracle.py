"""ORACLE — CPU restatement of the MV2D sparse cross-attention decoder hot path.

THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product path (``mv2d_amd``) never does and
fails loudly when the HIP extension is missing.

Every function restates, in plain torch-CPU fp32/fp64 arithmetic with the reference's own dtypes and
operation order, one function of the reference (tusen-ai/MV2D, paths relative to the reference root):

    RH = mmdet3d_plugin/models/roi_heads      MU = mmdet3d_plugin/models/utils
    CB = mmdet3d_plugin/core/bbox

Pinning status: the reference ships no tests / golden vectors.  The oracle is pinned
against OUTPUTS OF THE REFERENCE ITSELF: ``oracle/gen_golden.py`` imports the unmodified reference
modules (under a stub layer for the absent OpenMMLab packages) in the build container and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against those vectors.
Arithmetic that lives in un-vendored third-party code is restated here from the pinned versions and is
"parity unpinned" by any reference-side test:  mmcv==1.6.1 ``RoIAlign`` (roi_align), mmcv==1.6.1
``BaseTransformerLayer``/``FFN`` (decoder_layer), mmdet==2.25.1 ``bbox2roi`` / ``inverse_sigmoid``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
POST_RANGE = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]
NUM_HEADS = 8


# --------------------------------------------------------------------------------------------
# small third-party helpers (mmdet==2.25.1)
# --------------------------------------------------------------------------------------------
def bbox2roi(proposals):
    """mmdet.core.bbox2roi (call site RH/mv2d_head.py:110): [R,5] = (view, x1, y1, x2, y2)."""
    out = []
    for i, b in enumerate(proposals):
        if b.size(0) > 0:
            out.append(torch.cat([b.new_full((b.size(0), 1), i), b[:, :4]], -1))
        else:
            out.append(b.new_zeros((0, 5)))
    return torch.cat(out, 0)


def inverse_sigmoid(x, eps=1e-5):
    """mmdet.models.utils.transformer.inverse_sigmoid (call sites cross_attention_head.py:219, MU/pe.py:130)."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def with_dummy_proposal(proposals):
    """RH/mv2d_head.py:105-108: one dummy box in view 0 when there is no 2-D detection at all."""
    if sum(len(p) for p in proposals) == 0:
        dummy = torch.tensor([[0, 50, 50, 100, 100, 0]], dtype=proposals[0].dtype)
        proposals = [dummy] + list(proposals[1:])
    return proposals


# --------------------------------------------------------------------------------------------
# a3: per-RoI camera  (RH/mv2d_head.py:51-72)   a5: intrinsics feature (RH/mv2d_head.py:95-101)
# --------------------------------------------------------------------------------------------
def get_box_params(proposals, intrinsics, extrinsics, roi_size=(7, 7)):
    Ks, Es = [], []
    for bbox, K, E in zip(proposals, intrinsics, extrinsics):
        K = torch.from_numpy(np.asarray(K)).double().repeat(bbox.shape[0], 1, 1)
        E = torch.from_numpy(np.asarray(E)).double().repeat(bbox.shape[0], 1, 1)
        wh_bbox = bbox[:, 2:4] - bbox[:, :2]
        wh_roi = wh_bbox.new_tensor(roi_size)
        scale = wh_roi[None] / wh_bbox
        K[:, :2, 2] = K[:, :2, 2] - bbox[:, :2] - 0.5 / scale
        K[:, :2] = K[:, :2] * scale[..., None]
        Ks.append(K)
        Es.append(E)
    return torch.cat(Ks, 0), torch.cat(Es, 0)


def process_intrins_feat(rois, intrinsics, scale=0.1, min_size=4):
    f = intrinsics.view(intrinsics.shape[0], 16).clone().float() * scale
    wh = rois[:, 3:5] - rois[:, 1:3]
    f[(wh < min_size).any(1)] = 0
    return f


# --------------------------------------------------------------------------------------------
# a4: RoIAlign  (mmcv==1.6.1 ops/csrc/common/cuda/roi_align_cuda_kernel.cuh semantics:
#     aligned=True, pool_mode='avg', sampling_ratio<=0 -> adaptive grid) — THIRD PARTY, parity unpinned.
#     call site RH/mv2d_head.py:114-115, config CFG-T:49-53
# --------------------------------------------------------------------------------------------
def roi_align(feat, rois, out_size=7, spatial_scale=1.0 / 16, sampling_ratio=-1):
    V, C, H, W = feat.shape
    R = rois.size(0)
    P = out_size
    out = feat.new_zeros((R, C, P, P))
    f32 = torch.float32
    for r in range(R):
        b = int(rois[r, 0])
        # all arithmetic in fp32 like the kernel (T = float)
        x1 = rois[r, 1].to(f32) * spatial_scale - 0.5
        y1 = rois[r, 2].to(f32) * spatial_scale - 0.5
        x2 = rois[r, 3].to(f32) * spatial_scale - 0.5
        y2 = rois[r, 4].to(f32) * spatial_scale - 0.5
        rw, rh = x2 - x1, y2 - y1
        bw, bh = rw / P, rh / P
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rh) / P))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rw) / P))
        count = max(gh * gw, 1)
        if gh <= 0 or gw <= 0:
            continue
        ph = torch.arange(P, dtype=f32)[:, None]
        iy = torch.arange(gh, dtype=f32)[None]
        ix = torch.arange(gw, dtype=f32)[None]
        ys = (y1 + ph * bh + (iy + 0.5) * bh / gh).reshape(-1)
        xs = (x1 + ph * bw + (ix + 0.5) * bw / gw).reshape(-1)
        vy = (ys >= -1.0) & (ys <= H)
        vx = (xs >= -1.0) & (xs <= W)
        yc = ys.clamp(min=0)
        xc = xs.clamp(min=0)
        y0 = yc.to(torch.int64)
        x0 = xc.to(torch.int64)
        ytop = y0 >= H - 1
        xtop = x0 >= W - 1
        y0 = torch.where(ytop, torch.full_like(y0, H - 1), y0)
        x0 = torch.where(xtop, torch.full_like(x0, W - 1), x0)
        y1i = torch.where(ytop, y0, y0 + 1).clamp(max=H - 1)
        x1i = torch.where(xtop, x0, x0 + 1).clamp(max=W - 1)
        yc = torch.where(ytop, y0.to(f32), yc)
        xc = torch.where(xtop, x0.to(f32), xc)
        ly, lx = yc - y0, xc - x0
        hy, hx = 1.0 - ly, 1.0 - lx
        fm = feat[b]
        a = fm[:, y0][:, :, x0]
        bq = fm[:, y0][:, :, x1i]
        c = fm[:, y1i][:, :, x0]
        d = fm[:, y1i][:, :, x1i]
        w1 = hy[:, None] * hx[None]
        w2 = hy[:, None] * lx[None]
        w3 = ly[:, None] * hx[None]
        w4 = ly[:, None] * lx[None]
        val = w1 * a + w2 * bq + w3 * c + w4 * d
        val = val * (vy[:, None] & vx[None])
        out[r] = val.view(C, P, gh, P, gw).sum((2, 4)) / count
    return out


# --------------------------------------------------------------------------------------------
# a6/a7/a8: QueryGenerator  (RH/utils/query_generator.py:343-405, 333-341), ref-point normalisation
#           (RH/mv2d_t_head.py:51-57; the .clamp at :57 is NOT in-place -> no clamp)
# --------------------------------------------------------------------------------------------
def query_generator(sd, bbox_feats, K_roi, E, intr_feat, prefix='query_generator.'):
    w = lambda n: sd[prefix + n]
    x = F.relu(F.conv2d(bbox_feats, w('shared_convs.0.conv.weight'), w('shared_convs.0.conv.bias'), padding=1))
    x = F.avg_pool2d(x, 7).flatten(1)
    x = F.relu(F.linear(x, w('shared_fcs.0.weight'), w('shared_fcs.0.bias')))
    x = torch.cat([x, intr_feat], 1).clamp(min=-5e3, max=5e3)
    x = F.relu(F.linear(x, w('extra_enc.0.weight'), w('extra_enc.0.bias')))
    x = F.relu(F.linear(x, w('extra_enc.2.weight'), w('extra_enc.2.bias')))
    center_pred = F.linear(x, w('fc_center.weight'), w('fc_center.bias'))          # (u, v, depth) in RoI frame
    return center_pred, center2lidar(center_pred, K_roi, E)


def center2lidar(center_pred, K_roi, E):
    c_img = torch.cat([center_pred[:, :2] * center_pred[:, 2:3], center_pred[:, 2:3]], 1)
    c_hom = torch.cat([c_img, c_img.new_ones((c_img.shape[0], 1))], 1)
    lidar2img = torch.bmm(K_roi, E.transpose(1, 2))
    img2lidar = torch.inverse(lidar2img).float()
    return torch.bmm(img2lidar, c_hom[..., None])[:, :3, 0]


def normalize_ref(xyz, pc_range=PC_RANGE):
    ref = xyz.clone()
    ref[..., 0:1] = (ref[..., 0:1] - pc_range[0]) / (pc_range[3] - pc_range[0])
    ref[..., 1:2] = (ref[..., 1:2] - pc_range[1]) / (pc_range[4] - pc_range[1])
    ref[..., 2:3] = (ref[..., 2:3] - pc_range[2]) / (pc_range[5] - pc_range[2])
    return ref


# --------------------------------------------------------------------------------------------
# a9-a11: BoxCorrelation  (RH/utils/box_correlation.py)
# --------------------------------------------------------------------------------------------
def view_transforms(img_metas):
    """trans[a, b] = lidar2img[b] @ inv(lidar2img[a])  (box_correlation.py:117-122), fp64 [V,V,4,4]."""
    l2i = torch.stack([torch.from_numpy(np.asarray(m['lidar2img'])) for m in img_metas], 0).double()
    i2l = torch.inverse(l2i)
    return torch.matmul(l2i[None], i2l[:, None])


def sample_points_in_rois(rois, sample_size=4):
    """box_correlation.py:196-209 -> [R, 16, 3] (view, x, y), grid includes the box edges."""
    xs = ys = torch.linspace(0, 1, sample_size)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    coords_roi = torch.stack([gx, gy], -1)
    wh = rois[:, 3:5] - rois[:, 1:3]
    coords_img = rois[:, None, None, 1:3] + wh[:, None, None] * coords_roi[None]
    pts = coords_img.reshape(rois.size(0), sample_size * sample_size, 2)
    return torch.cat([rois[:, None, 0:1].expand_as(pts[..., 0:1]), pts], -1)


def lid_depths(num_depth=8, depth_start=0.5, depth_end=70):
    """box_correlation.py:221-225 (fp32 tensor arithmetic, python-float bin size)."""
    index = torch.arange(0, num_depth, 1).float()
    bin_size = (depth_end - depth_start) / (num_depth * (1 + num_depth))
    return depth_start + bin_size * index * (index + 1)


def epipolar_in_each_view(points, image_shape, trans, num_depth=8, depth_start=0.5, depth_end=70):
    """box_correlation.py:212-257 -> projected uv fp32 [N,V,D,2], valid [N,V,D]."""
    N = points.size(0)
    d = lid_depths(num_depth, depth_start, depth_end)[None].expand(N, num_depth)
    p2d = torch.cat([points[:, None, 1:3].expand(N, num_depth, 2), d[..., None]], -1).to(trans.dtype)
    hom = torch.cat([p2d[..., :2] * p2d[..., 2:3], p2d[..., 2:3], p2d.new_ones((N, num_depth, 1))], -1)
    view_ids = points[:, 0].long()
    tm = trans[view_ids]                                                           # [N, V, 4, 4]
    cam = torch.matmul(tm[:, :, None], hom[:, None, ..., None])[..., :3, 0]       # [N, V, D, 3]
    uv = cam[..., :2] / cam[..., 2:3].clamp_min(1e-2)
    valid = torch.ones_like(uv[..., 0], dtype=torch.bool)
    valid[cam[..., 2] < depth_start] = 0
    in_x = (0 <= uv[..., 0]) & (uv[..., 0] <= image_shape[1] - 1)
    in_y = (0 <= uv[..., 1]) & (uv[..., 1] <= image_shape[0] - 1)
    valid = valid & in_x & in_y
    valid[torch.arange(N), view_ids] = 0
    return uv.float(), valid


def box_iou(a, b, eps=1e-4):
    """box_correlation.py:385-398; a [n,4], b [m,4] -> [n,m] fp32."""
    a = a[:, None, :]
    b = b[None, :, :]
    xy_start = torch.maximum(a[..., 0:2], b[..., 0:2])
    xy_end = torch.minimum(a[..., 2:4], b[..., 2:4])
    wh = torch.maximum(xy_end - xy_start, a.new_tensor(0))
    inter = wh.prod(-1)
    area_a = (a[..., 2:4] - a[..., 0:2]).prod(-1)
    area_b = (b[..., 2:4] - b[..., 0:2]).prod(-1)
    union = area_a + area_b - inter
    return inter / (union + eps)


def epipolar_in_box(rois, num_per_view, image_shape, trans, topk, iou_thr=0.0, ratio=0.0,
                    sample_size=4, num_depth=8, depth_start=0.5, depth_end=70):
    """box_correlation.py:260-382, 'topk_matched:k:thr:ratio' mode; topk=None: 'all_matched' (:305-338: every RoI of a view the epipolar
    points reach whose IoU with their bounding box is > 0, in the view's RoI order).

    Returns ragged python lists: for each RoI r a list of (roi_id, keep) in (view-major, IoU-rank) order —
    exactly the valid prefix the reference builds by pad_sequence/flatten (:376-380).  Ties in the IoU sort
    are broken towards the lower index (stable); the reference's argsort is unstable there.
    """
    R = rois.size(0)
    V = trans.size(0)
    if R == 0:
        return []
    pts = sample_points_in_rois(rois, sample_size)
    n_pts = pts.size(1)
    uv, valid = epipolar_in_each_view(pts.reshape(R * n_pts, 3), image_shape, trans, num_depth, depth_start, depth_end)
    uv = uv.view(R, n_pts, V, num_depth, 2).permute(0, 2, 1, 3, 4).reshape(R, V, n_pts * num_depth, 2)
    valid = valid.view(R, n_pts, V, num_depth).permute(0, 2, 1, 3).reshape(R, V, n_pts * num_depth)
    starts = np.concatenate([[0], np.cumsum(num_per_view)]).astype(int)
    out = [[] for _ in range(R)]
    for r in range(R):
        for v in range(V):
            n_v = starts[v + 1] - starts[v]
            if n_v == 0 or not bool(valid[r, v].any()):
                continue
            rv = rois[starts[v]:starts[v + 1]]                                    # [n_v, 5]
            u, w_ = uv[r, v, :, 0], uv[r, v, :, 1]
            hit = ((rv[:, None, 1] <= u[None]) & (u[None] <= rv[:, None, 3]) &
                   (rv[:, None, 2] <= w_[None]) & (w_[None] <= rv[:, None, 4]) & valid[r, v][None])
            if not bool(hit.any()):
                continue
            m = valid[r, v]
            p = uv[r, v]
            pmax = torch.where(m[:, None], p, torch.full_like(p, -1e4)).max(0)[0]
            pmin = torch.where(m[:, None], p, torch.full_like(p, 1e4)).min(0)[0]
            t_roi = torch.cat([pmin, pmax])[None]                                 # [1,4]
            iou = box_iou(t_roi, rv[:, 1:])[0]                                    # [n_v]
            if topk is None:                                                      # all_matched (:330-331): all_mask = iou > 0, RoI order
                order = torch.arange(n_v)
                top_iou = iou
                keep = iou > 0
            else:
                order = torch.argsort(iou, descending=True, stable=True)[:topk]
                top_iou = iou[order]
                keep = ((top_iou > ratio * top_iou.max()) | (top_iou > iou_thr)) & (top_iou > 0)
            for j in range(order.numel()):
                out[r].append((int(order[j]) + int(starts[v]), bool(keep[j])))
    return out


def gen_box_roi_correlation(rois, num_per_view, img_metas, topk=1):
    """S-path: box_correlation.py:165-193 -> corr [R,n_c] int64 (pad id 0), mask [R,n_c] bool."""
    R = rois.size(0)
    if rois.numel() == 0:
        return rois.new_zeros((0, 0), dtype=torch.int64), rois.new_zeros((0, 0), dtype=torch.bool)
    trans = view_transforms(img_metas)
    ep = epipolar_in_box(rois, num_per_view, img_metas[0]['pad_shape'], trans, topk)
    rows = [[r] + [i for (i, k) in ep[r] if k] for r in range(R)]
    n_c = max(len(x) for x in rows)
    corr = torch.zeros((R, n_c), dtype=torch.int64)
    mask = torch.zeros((R, n_c), dtype=torch.bool)
    for r, row in enumerate(rows):
        corr[r, :len(row)] = torch.tensor(row, dtype=torch.int64)
        mask[r, :len(row)] = True
    return corr, mask


def feat_in_rois_mask(rois, V, h, w, stride, expand_stride):
    """box_correlation.py:101-115 own-view in-RoI cell mask [R,V,h,w] bool (fp32 compares)."""
    ys = (torch.arange(h, dtype=torch.float32) + 0.5) * stride - 0.5
    xs = (torch.arange(w, dtype=torch.float32) + 0.5) * stride - 0.5
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    coords = torch.stack([gx, gy], -1)[None]                                      # [1,h,w,2] (x,y)
    box = rois[:, None, None]
    inb = ((coords[..., 0:2] + 0.5 * stride + expand_stride * stride >= box[..., 1:3]) &
           (coords[..., 0:2] - 0.5 * stride - expand_stride * stride <= box[..., 3:5])).all(-1)
    out = torch.zeros((rois.size(0), V, h, w), dtype=torch.bool)
    out[torch.arange(rois.size(0)), rois[:, 0].long()] = inb
    return out


def gen_box_correlation(rois, num_per_view, img_metas, h, w, stride=16, expand_stride=2, topk=20):
    """T-path: box_correlation.py:95-162 -> feat_in_corr_rois [R,V,h,w] bool."""
    V = len(img_metas)
    fin = feat_in_rois_mask(rois, V, h, w, stride, expand_stride)
    trans = view_transforms(img_metas)
    ep = epipolar_in_box(rois, num_per_view, img_metas[0]['pad_shape'], trans, topk)
    out = torch.zeros_like(fin)
    for r in range(rois.size(0)):
        ids = [r] + [i for (i, k) in ep[r] if k]
        out[r] = fin[ids].any(0)
    return out


# --------------------------------------------------------------------------------------------
# a2: PE  (MU/pe.py:84-169) + SinePositionalEncoding3D (MU/positional_encoding.py:58-96)
# --------------------------------------------------------------------------------------------
def padding_mask(img_metas, h, w):
    """RH/mv2d_t_head.py:69-76 / MU/pe.py:146-155: nearest-interpolated outside-image mask [V,h,w] bool."""
    V = len(img_metas)
    ph, pw, _ = img_metas[0]['pad_shape']
    m = torch.ones((1, V, ph, pw), dtype=torch.float32)
    for i in range(V):
        ih, iw, _ = img_metas[i]['img_shape']
        m[0, i, :ih, :iw] = 0
    return F.interpolate(m, size=(h, w)).to(torch.bool)[0]


def sine_pe3d(mask, num_feats=128, temperature=10000, scale=2 * math.pi, eps=1e-6, stride=16):
    """MU/positional_encoding.py:58-96 with normalize=True, offset=0; mask [B,V,h,w] bool -> [B,V,384,h,w]."""
    not_mask = 1 - mask.to(torch.int)
    n_e = not_mask.cumsum(1, dtype=torch.float32)
    y_e = not_mask.cumsum(2, dtype=torch.float32)
    x_e = not_mask.cumsum(3, dtype=torch.float32)
    if stride > 0:
        y_e = (y_e - 0.5) * stride
        x_e = (x_e - 0.5) * stride
    n_e = n_e / (n_e[:, -1:, :, :] + eps) * scale
    y_e = y_e / (y_e[:, :, -1:, :] + eps) * scale
    x_e = x_e / (x_e[:, :, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    B, N, H, W = mask.shape

    def emb(e):
        p = e[..., None] / dim_t
        return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=4).view(B, N, H, W, -1)
    return torch.cat((emb(n_e), emb(y_e), emb(x_e)), dim=4).permute(0, 1, 4, 2, 3)


def pe_frustum_input(img_metas, H, W, depth_num=64, depth_start=1, position_range=POST_RANGE):
    """MU/pe.py:84-130: inverse-sigmoid'ed normalised frustum coords, fp32 [V, 3*D, H, W] (channel = k*3+axis)."""
    eps = 1e-3
    pad_h, pad_w, _ = img_metas[0]['pad_shape']
    V = len(img_metas)
    coords_h = (torch.arange(H).double() + 0.5) * pad_h / H - 0.5
    coords_w = (torch.arange(W).double() + 0.5) * pad_w / W - 0.5
    index = torch.arange(0, depth_num, 1).double()
    bin_size = (position_range[3] - depth_start) / (depth_num * (1 + depth_num))
    coords_d = depth_start + bin_size * index * (index + 1)
    D = coords_d.shape[0]
    coords = torch.stack(torch.meshgrid([coords_w, coords_h, coords_d], indexing='ij')).permute(1, 2, 3, 0)
    coords = torch.cat((coords, torch.ones_like(coords[..., :1])), -1)
    coords[..., :2] = coords[..., :2] * torch.maximum(coords[..., 2:3], torch.ones_like(coords[..., 2:3]) * eps)
    img2lidars = np.asarray([np.linalg.inv(m['lidar2img']) for m in img_metas])
    img2lidars = coords.new_tensor(img2lidars)
    coords = coords.view(1, 1, W, H, D, 4, 1).repeat(1, V, 1, 1, 1, 1, 1)
    img2lidars = img2lidars.view(1, V, 1, 1, 1, 4, 4).repeat(1, 1, W, H, D, 1, 1)
    c3 = torch.matmul(img2lidars, coords).squeeze(-1)[..., :3]
    c3[..., 0:1] = (c3[..., 0:1] - position_range[0]) / (position_range[3] - position_range[0])
    c3[..., 1:2] = (c3[..., 1:2] - position_range[1]) / (position_range[4] - position_range[1])
    c3[..., 2:3] = (c3[..., 2:3] - position_range[2]) / (position_range[5] - position_range[2])
    c3 = c3.permute(0, 1, 4, 5, 3, 2).contiguous().view(V, -1, H, W)
    return inverse_sigmoid(c3).float()


def pe_map(sd, feat, img_metas, prefix='position_encoding.', stride=16, return_parts=False):
    """PE.forward (MU/pe.py:137-169): [V,256,h,w] key position embedding for the WHOLE map."""
    w = lambda n: sd[prefix + n]
    V, C, H, W = feat.shape
    masks = padding_mask(img_metas, H, W)[None]
    x3 = pe_frustum_input(img_metas, H, W)
    p = F.conv2d(F.relu(F.conv2d(x3, w('position_encoder.0.weight'), w('position_encoder.0.bias'))),
                 w('position_encoder.2.weight'), w('position_encoder.2.bias'))
    gate = torch.sigmoid(F.conv2d(F.relu(F.conv2d(feat, w('fpe.conv_reduce.weight'), w('fpe.conv_reduce.bias'))),
                                  w('fpe.conv_expand.weight'), w('fpe.conv_expand.bias')))
    p = p * gate
    sin = sine_pe3d(masks, stride=stride).flatten(0, 1)
    s = F.conv2d(F.relu(F.conv2d(sin, w('adapt_pos3d.0.weight'), w('adapt_pos3d.0.bias'))),
                 w('adapt_pos3d.2.weight'), w('adapt_pos3d.2.bias'))
    if return_parts:
        return p + s, x3, sin
    return p + s


# --------------------------------------------------------------------------------------------
# a13: pos2posemb3d (MU/pe.py:21-33) + query embedding MLP (cross_attention_head.py:118-122,199-200)
# --------------------------------------------------------------------------------------------
def pos2posemb3d(pos, num_pos_feats=128, temperature=10000):
    pos = pos * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)

    def emb(p):
        p = p[..., None] / dim_t
        return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)
    return torch.cat((emb(pos[..., 1]), emb(pos[..., 0]), emb(pos[..., 2])), dim=-1)


def query_embedding(sd, ref, prefix='bbox_head.'):
    e = pos2posemb3d(ref)
    e = F.relu(F.linear(e, sd[prefix + 'query_embedding.0.weight'], sd[prefix + 'query_embedding.0.bias']))
    return F.linear(e, sd[prefix + 'query_embedding.2.weight'], sd[prefix + 'query_embedding.2.bias'])


# --------------------------------------------------------------------------------------------
# a16-a19: decoder  (MU/petr_transformer.py:317-370, 426-513, 569-593; mmcv BaseTransformerLayer/FFN)
# --------------------------------------------------------------------------------------------
def _mha(q_in, k_in, v_in, w_in, b_in, w_out, b_out, blocked=None, num_heads=NUM_HEADS, want=None):
    """torch.nn.MultiheadAttention math on [Lq,C] x [Lk,C]; ``blocked`` [Lq,Lk] bool True = excluded.

    A fully blocked row yields NaN exactly like torch."""
    C = q_in.shape[-1]
    d = C // num_heads
    q = F.linear(q_in, w_in[:C], b_in[:C])
    k = F.linear(k_in, w_in[C:2 * C], b_in[C:2 * C])
    v = F.linear(v_in, w_in[2 * C:], b_in[2 * C:])
    Lq, Lk = q.shape[0], k.shape[0]
    qh = q.view(Lq, num_heads, d).transpose(0, 1) * math.sqrt(1.0 / d)
    kh = k.view(Lk, num_heads, d).transpose(0, 1)
    vh = v.view(Lk, num_heads, d).transpose(0, 1)
    logits = torch.bmm(qh, kh.transpose(1, 2))                                    # [H, Lq, Lk]
    if blocked is not None:
        logits = logits.masked_fill(blocked[None], float('-inf'))
    attn = torch.softmax(logits, -1)
    ctx = torch.bmm(attn, vh).transpose(0, 1).reshape(Lq, C)
    if want is not None:
        want['logits'] = logits
        want['attn_mean'] = attn.mean(0)
        want['q'] = q
        want['k'] = k
        want['v'] = v
    return F.linear(ctx, w_out, b_out)


def decoder(sd, query_pos, key_in, val_in, blocked, num_layers=6, prefix='bbox_head.transformer.decoder.',
            self_blocked=None, capture=None):
    """6 post-norm layers (self_attn, norm, cross_attn, norm, ffn, norm) + shared post_norm on every
    intermediate.  query_pos [Q,C]; key_in = memory + key_pos [S,C]; val_in = memory [S,C];
    blocked [Q,S] bool (attn_mask OR key_padding_mask).  Returns [L,Q,C]."""
    x = torch.zeros_like(query_pos)                                               # target = zeros (cross_attention_head.py:32)
    outs = []
    for i in range(num_layers):
        p = f'{prefix}layers.{i}.'
        g = lambda n: sd[p + n]
        # FlattenMHSelfAttention: q = k = x + qpos, v = x, residual
        qk = x + query_pos
        x = x + _mha(qk, qk, x, g('attentions.0.attn.in_proj_weight'), g('attentions.0.attn.in_proj_bias'),
                     g('attentions.0.attn.out_proj.weight'), g('attentions.0.attn.out_proj.bias'), self_blocked)
        x = F.layer_norm(x, (x.shape[-1],), g('norms.0.weight'), g('norms.0.bias'))
        want = {} if (capture is not None) else None
        x = x + _mha(x + query_pos, key_in, val_in, g('attentions.1.attn.in_proj_weight'),
                     g('attentions.1.attn.in_proj_bias'), g('attentions.1.attn.out_proj.weight'),
                     g('attentions.1.attn.out_proj.bias'), blocked, want=want)
        if capture is not None:
            capture.append(want)
        x = F.layer_norm(x, (x.shape[-1],), g('norms.1.weight'), g('norms.1.bias'))
        hdn = F.relu(F.linear(x, g('ffns.0.layers.0.0.weight'), g('ffns.0.layers.0.0.bias')))
        x = x + F.linear(hdn, g('ffns.0.layers.1.weight'), g('ffns.0.layers.1.bias'))
        x = F.layer_norm(x, (x.shape[-1],), g('norms.2.weight'), g('norms.2.bias'))
        outs.append(F.layer_norm(x, (x.shape[-1],), sd[prefix + 'post_norm.weight'], sd[prefix + 'post_norm.bias']))
    return torch.stack(outs)


def decoder_s(sd, query_pos, roi_k_in, roi_v_in, corr, cmask, num_layers=6,
              prefix='bbox_head.transformer.decoder.'):
    """S-path decoder (RH/mv2d_s_head.py:181-192 + cross_attention_head.py:24-49): batch = R queries, each
    with its own key set = 49 cells of each correlated RoI.  roi_k_in/roi_v_in [R,49,C] (cell-major),
    corr [R,n_c] int64, cmask [R,n_c] bool.  Self-attention couples all R queries (flatten)."""
    R, n_c = corr.shape
    C = query_pos.shape[-1]
    x = torch.zeros_like(query_pos)
    outs = []
    # memory layout of the reference: [n_c*49, R, C] with key index = j*49 + cell
    kin = roi_k_in[corr].reshape(R, n_c * 49, C)
    vin = roi_v_in[corr].reshape(R, n_c * 49, C)
    blocked = (~cmask)[:, :, None].expand(R, n_c, 49).reshape(R, n_c * 49)
    d = C // NUM_HEADS
    for i in range(num_layers):
        p = f'{prefix}layers.{i}.'
        g = lambda n: sd[p + n]
        qk = x + query_pos
        x = x + _mha(qk, qk, x, g('attentions.0.attn.in_proj_weight'), g('attentions.0.attn.in_proj_bias'),
                     g('attentions.0.attn.out_proj.weight'), g('attentions.0.attn.out_proj.bias'))
        x = F.layer_norm(x, (C,), g('norms.0.weight'), g('norms.0.bias'))
        w_in, b_in = g('attentions.1.attn.in_proj_weight'), g('attentions.1.attn.in_proj_bias')
        q = F.linear(x + query_pos, w_in[:C], b_in[:C]).view(R, NUM_HEADS, 1, d) * math.sqrt(1.0 / d)
        k = F.linear(kin, w_in[C:2 * C], b_in[C:2 * C]).view(R, -1, NUM_HEADS, d).transpose(1, 2)
        v = F.linear(vin, w_in[2 * C:], b_in[2 * C:]).view(R, -1, NUM_HEADS, d).transpose(1, 2)
        logits = torch.matmul(q, k.transpose(2, 3)).masked_fill(blocked[:, None, None, :], float('-inf'))
        ctx = torch.matmul(torch.softmax(logits, -1), v).reshape(R, C)
        x = x + F.linear(ctx, g('attentions.1.attn.out_proj.weight'), g('attentions.1.attn.out_proj.bias'))
        x = F.layer_norm(x, (C,), g('norms.1.weight'), g('norms.1.bias'))
        hdn = F.relu(F.linear(x, g('ffns.0.layers.0.0.weight'), g('ffns.0.layers.0.0.bias')))
        x = x + F.linear(hdn, g('ffns.0.layers.1.weight'), g('ffns.0.layers.1.bias'))
        x = F.layer_norm(x, (C,), g('norms.2.weight'), g('norms.2.bias'))
        outs.append(F.layer_norm(x, (C,), sd[prefix + 'post_norm.weight'], sd[prefix + 'post_norm.bias']))
    return torch.stack(outs)


# --------------------------------------------------------------------------------------------
# a14: per-layer heads (cross_attention_head.py:216-238)   a20: velocity / dt (RH/mv2d_t_head.py:130-142)
# --------------------------------------------------------------------------------------------
def pred_heads(sd, outs_dec, ref, pc_range=PC_RANGE, prefix='bbox_head.'):
    L = outs_dec.shape[0]
    cls_all, reg_all = [], []
    reference = inverse_sigmoid(ref.clone())
    for l in range(L):
        c = lambda n: sd[f'{prefix}cls_branches.{l}.{n}']
        r = lambda n: sd[f'{prefix}reg_branches.{l}.{n}']
        h = outs_dec[l]
        y = F.relu(F.layer_norm(F.linear(h, c('0.weight'), c('0.bias')), (256,), c('1.weight'), c('1.bias')))
        y = F.relu(F.layer_norm(F.linear(y, c('3.weight'), c('3.bias')), (256,), c('4.weight'), c('4.bias')))
        cls = F.linear(y, c('6.weight'), c('6.bias'))
        t = F.relu(F.linear(h, r('0.weight'), r('0.bias')))
        t = F.relu(F.linear(t, r('2.weight'), r('2.bias')))
        t = F.linear(t, r('4.weight'), r('4.bias'))
        t[..., 0:2] = (t[..., 0:2] + reference[..., 0:2]).sigmoid()
        t[..., 4:5] = (t[..., 4:5] + reference[..., 2:3]).sigmoid()
        cls_all.append(cls)
        reg_all.append(t)
    cls_all = torch.stack(cls_all)
    reg_all = torch.stack(reg_all)
    reg_all[..., 0:1] = reg_all[..., 0:1] * (pc_range[3] - pc_range[0]) + pc_range[0]
    reg_all[..., 1:2] = reg_all[..., 1:2] * (pc_range[4] - pc_range[1]) + pc_range[1]
    reg_all[..., 4:5] = reg_all[..., 4:5] * (pc_range[5] - pc_range[2]) + pc_range[2]
    return cls_all, reg_all


def mean_time_delta(img_metas, num_views):
    ts = np.array([m['timestamp'] for m in img_metas])
    return ts[num_views:].mean() - ts[:num_views].mean()


# --------------------------------------------------------------------------------------------
# a21: NMS-free decode (CB/coders/nms_free_coder.py:49-102, CB/util.py:60-87, cross_attention_head.py:357-377)
# --------------------------------------------------------------------------------------------
def decode(cls_scores, bbox_preds, max_num=300, num_classes=10, post_center_range=POST_RANGE):
    max_num = min(max_num, cls_scores.numel())
    scores, idx = cls_scores.sigmoid().view(-1).topk(max_num)
    labels = idx % num_classes
    bbox_index = idx // num_classes
    bp = bbox_preds[bbox_index]
    rot = torch.atan2(bp[..., 6:7], bp[..., 7:8])
    boxes = torch.cat([bp[..., 0:1], bp[..., 1:2], bp[..., 4:5], bp[..., 2:3].exp(), bp[..., 3:4].exp(),
                       bp[..., 5:6].exp(), rot, bp[:, 8:9], bp[:, 9:10]], -1)
    rng = torch.tensor(post_center_range)
    mask = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
    boxes, scores, labels, bbox_index = boxes[mask], scores[mask], labels[mask], bbox_index[mask]
    boxes = boxes.clone()
    boxes[:, 2] = boxes[:, 2] - boxes[:, 5] * 0.5
    return boxes, scores, labels, bbox_index


# --------------------------------------------------------------------------------------------
# f1 ("next"): post-decoder 3-D NMS + packing (DET/mv2d.py:265-287 -> mmdet3d==1.0.0 box3d_multiclass_nms, THIRD PARTY,
#     parity unpinned).  nms_thr = 1.0: nms_bev keeps everything in descending score order.
# --------------------------------------------------------------------------------------------
def post_nms_pack(boxes, scores, labels, num_classes=10, score_thr=0.0, max_num=300):
    sc = scores.new_zeros((len(scores), num_classes + 1)).scatter_(1, labels[:, None], scores[:, None])
    ob, os_, ol = [], [], []
    for i in range(num_classes):
        inds = sc[:, i] > score_thr
        if not inds.any():
            continue
        s_i = sc[inds, i]
        order = torch.argsort(s_i, descending=True, stable=True)
        ob.append(boxes[inds][order]); os_.append(s_i[order]); ol.append(torch.full((int(inds.sum()),), i, dtype=torch.long))
    if not ob:
        return boxes.new_zeros((0, boxes.shape[1])), scores.new_zeros((0,)), labels.new_zeros((0,))
    ob, os_, ol = torch.cat(ob), torch.cat(os_), torch.cat(ol)
    if ob.shape[0] > max_num:
        inds = torch.argsort(os_, descending=True, stable=True)[:max_num]
        ob, os_, ol = ob[inds], os_[inds], ol[inds]
    return ob, os_, ol


def fpn_neck(x_lvl, w_lat, b_lat, w_fpn, b_fpn):
    """f2: mmdet==2.25.1 FPN with start_level == end_level, num_outs == 1 (THIRD PARTY, parity unpinned; configs/mv2d/exp/*:32-39,
    DET/mv2d.py:122-127): one lateral 1x1 conv + one 3x3 fpn conv, ConvModule(norm_cfg=None, act_cfg=None) = plain Conv2d with bias."""
    return F.conv2d(F.conv2d(x_lvl, w_lat, b_lat), w_fpn, b_fpn, padding=1)


def process_2d_detections(results, min_bbox_size=0):
    """DET/mv2d.py:60-86."""
    dets = [torch.cat([torch.cat([torch.tensor(b), torch.full((len(b), 1), label_id, dtype=torch.float)], dim=1)
                       for label_id, b in enumerate(res)], dim=0) for res in results]
    if min_bbox_size > 0:
        dets = [d[((d[:, 2:4] - d[:, 0:2]) >= min_bbox_size).all(dim=1)] for d in dets]
    return dets


# --------------------------------------------------------------------------------------------
# a1: orchestration — MV2DTHead / MV2DSHead.simple_test  (RH/mv2d_head.py:249-267)
# --------------------------------------------------------------------------------------------
def _to_t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v for k, v in sd.items()}


def masked_cross_attention(q, K, V, allowed):
    """Attention core of PETRMultiheadAttention (MU/petr_transformer.py:487-513 -> nn.MultiheadAttention with a boolean attn_mask):
    q [Q,256] already scaled by 1/sqrt(32), K / V [S,256], allowed [Q,S] bool; 8 heads x 32; rows without an allowed key give 0
    here (the module itself gives NaN).  Differentiable: torch autograd of this function is the oracle of mv2d_sparse_xattn_bwd."""
    Q, S = q.shape[0], K.shape[0]
    qh = q.view(Q, 8, 32).transpose(0, 1)
    kh = K.view(S, 8, 32).transpose(0, 1)
    vh = V.view(S, 8, 32).transpose(0, 1)
    logits = (qh @ kh.transpose(1, 2)).masked_fill(~allowed[None], float('-inf'))
    att = torch.softmax(logits, -1)
    att = torch.where(allowed.any(1)[None, :, None], att, torch.zeros_like(att))
    return (att @ vh).transpose(0, 1).reshape(Q, 256)


def csr_from_allowed(allowed):
    """allowed [Q,S] bool -> (row_ptr int32 [Q+1], col_idx int32 [nnz]) row-major."""
    counts = allowed.sum(1)
    row_ptr = torch.zeros(allowed.shape[0] + 1, dtype=torch.int32)
    row_ptr[1:] = counts.cumsum(0)
    col = allowed.nonzero()[:, 1].to(torch.int32)
    return row_ptr, col


def forward_t(sd, feat, proposals, img_metas, num_views=6, expand_stride=2, topk=20, stages=None):
    """MV2DTHead path (RH/mv2d_t_head.py:26-142).  feat [V,256,h,w] f32 torch; proposals list of [n,6]."""
    sd = _to_t(sd)
    st = {} if stages is None else stages
    V, C, h, w = feat.shape
    proposals = with_dummy_proposal([p for p in proposals])
    rois = bbox2roi(proposals)
    K_roi, E = get_box_params(proposals, [m['intrinsics'] for m in img_metas], [m['extrinsics'] for m in img_metas])
    pe = pe_map(sd, feat, img_metas)
    roi_feats = roi_align(feat, rois)                                             # only the feature half is used
    intr = process_intrins_feat(rois, K_roi)
    center_pred, xyz = query_generator(sd, roi_feats, K_roi, E, intr)
    ref = normalize_ref(xyz)
    num_per_view = [len(p) for p in proposals]
    ffr = gen_box_correlation(rois, num_per_view, img_metas, h, w, 16, expand_stride, topk)
    pad = padding_mask(img_metas, h, w)
    roi_mask = ffr.any(0)                                                         # [V,h,w]
    mem = feat.permute(0, 2, 3, 1)[roi_mask]                                      # [S,C] row-major (v,y,x)
    mpe = pe.permute(0, 2, 3, 1)[roi_mask]
    kpm = pad[roi_mask]
    blocked = (~ffr)[:, roi_mask] | kpm[None]
    qpos = query_embedding(sd, ref)
    cap = [] if stages is not None else None
    outs = decoder(sd, qpos, mem + mpe, mem, blocked, capture=cap)
    cls_all, reg_all = pred_heads(sd, outs, ref)
    if len(img_metas) > num_views:
        dt = mean_time_delta(img_metas, num_views)
        reg_all = torch.cat([reg_all[..., :8], reg_all[..., 8:] / dt], -1)
    boxes, scores, labels, bidx = decode(cls_all[-1], reg_all[-1])
    st.update(rois=rois, K_roi=K_roi, E=E, pe=pe, roi_feats=roi_feats, intr=intr, center_pred=center_pred, xyz=xyz,
              ref=ref, feat_for_rois=ffr, roi_mask=roi_mask, key_padding=kpm, blocked=blocked, qpos=qpos,
              outs_dec=outs, cls=cls_all, reg=reg_all, boxes=boxes, scores=scores, labels=labels, bbox_index=bidx,
              capture=cap)
    return boxes, scores, labels


def forward_s(sd, feat, proposals, img_metas, topk=1, stages=None):
    """MV2DSHead eval path (RH/mv2d_s_head.py:122-211, branch :181-192)."""
    sd = _to_t(sd)
    st = {} if stages is None else stages
    V, C, h, w = feat.shape
    proposals = with_dummy_proposal([p for p in proposals])
    rois = bbox2roi(proposals)
    K_roi, E = get_box_params(proposals, [m['intrinsics'] for m in img_metas], [m['extrinsics'] for m in img_metas])
    pe = pe_map(sd, feat, img_metas)
    roi_all = roi_align(torch.cat([feat, pe], 1), rois)                           # [R,512,7,7]
    roi_feats, roi_pe = roi_all.split([C, C], 1)
    intr = process_intrins_feat(rois, K_roi)
    center_pred, xyz = query_generator(sd, roi_feats, K_roi, E, intr)
    ref = normalize_ref(xyz)
    corr, cmask = gen_box_roi_correlation(rois, [len(p) for p in proposals], img_metas, topk)
    qpos = query_embedding(sd, ref)
    rf = roi_feats.flatten(2).transpose(1, 2)                                     # [R,49,C] cell-major
    rp = roi_pe.flatten(2).transpose(1, 2)
    outs = decoder_s(sd, qpos, rf + rp, rf, corr, cmask)
    cls_all, reg_all = pred_heads(sd, outs, ref)
    boxes, scores, labels, bidx = decode(cls_all[-1], reg_all[-1])
    st.update(rois=rois, K_roi=K_roi, E=E, pe=pe, roi_feats=roi_feats, roi_pe=roi_pe, intr=intr,
              center_pred=center_pred, xyz=xyz, ref=ref, corr=corr, corr_mask=cmask, qpos=qpos, outs_dec=outs,
              cls=cls_all, reg=reg_all, boxes=boxes, scores=scores, labels=labels, bbox_index=bidx)
    return boxes, scores, labels


# --------------------------------------------------------------------------------------------
# training targets and losses of the box head
# --------------------------------------------------------------------------------------------
CODE_WEIGHTS = [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.5, 1.5, 2.0, 2.0]      # configs/mv2d/exp/*:90


def normalize_bbox(b):
    """CB/util.py:37-58: (cx, cy, cz, w, l, h, yaw, vx, vy) -> (cx, cy, log w, log l, cz, log h, sin, cos, vx, vy)."""
    return torch.cat([b[..., 0:1], b[..., 1:2], b[..., 3:4].log(), b[..., 4:5].log(), b[..., 2:3], b[..., 5:6].log(),
                      b[..., 6:7].sin(), b[..., 6:7].cos(), b[..., 7:9]], -1)


def focal_loss_cost(cls_pred, gt_labels, weight=2.0, alpha=0.25, gamma=2.0, eps=1e-12):
    """mmdet==2.25.1 FocalLossCost._focal_loss_cost (third party, restated; cfg `cls_cost`, configs/mv2d/exp/*:138)."""
    p = cls_pred.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    return (pos[:, gt_labels] - neg[:, gt_labels]) * weight


def match_cost(bbox_pred, cls_pred, gt_bboxes, gt_labels, reg_weight=0.25):
    """CB/assigners/hungarian_assigner_3d.py:120-134: focal cost + L1 cdist on the first 8 codes (CB/match_costs/match_cost.py:24), nan_to_num."""
    cost = focal_loss_cost(cls_pred, gt_labels) + torch.cdist(bbox_pred[:, :8], normalize_bbox(gt_bboxes)[:, :8], p=1) * reg_weight
    return torch.nan_to_num(cost, nan=100.0, posinf=100.0, neginf=-100.0)


def hungarian_assign(bbox_pred, cls_pred, gt_bboxes, gt_labels):
    """HungarianAssigner3D.assign (:65-150) + PseudoSampler: match [R] = index of the assigned gt or -1 (background)."""
    from scipy.optimize import linear_sum_assignment
    R, G = bbox_pred.size(0), gt_bboxes.size(0)
    match = torch.full((R,), -1, dtype=torch.long)
    if R == 0 or G == 0:
        return match
    rows, cols = linear_sum_assignment(match_cost(bbox_pred, cls_pred, gt_bboxes, gt_labels).detach().cpu())
    match[torch.from_numpy(rows)] = torch.from_numpy(cols)
    return match


def sigmoid_focal_loss_sum(pred, labels, num_classes, alpha=0.25, gamma=2.0):
    """mmdet==2.25.1 py_sigmoid_focal_loss summed over all elements (third party, restated); label == num_classes is background."""
    t = F.one_hot(labels, num_classes + 1)[:, :num_classes].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    return (F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw).sum()


def loss_single(cls_scores, bbox_preds, gt_bboxes, gt_labels, match=None, num_classes=10, code_weights=CODE_WEIGHTS,
                loss_cls_weight=2.0, loss_bbox_weight=0.25):
    """CrossAttentionBoxHead.loss_single for one sample (RH/bbox_heads/cross_attention_head.py:380-434 with _get_target_single
    :234-302): cls_scores [R,C], bbox_preds [R,10], gt_bboxes [G,9], gt_labels [G] -> (loss_cls, loss_bbox, match)."""
    R = cls_scores.size(0)
    if match is None:
        match = hungarian_assign(bbox_preds, cls_scores, gt_bboxes, gt_labels)
    pos = match >= 0
    labels = torch.full((R,), num_classes, dtype=torch.long)
    labels[pos] = gt_labels[match[pos]]
    targets = torch.zeros(R, 9, dtype=bbox_preds.dtype)
    if gt_bboxes.size(0):
        targets[pos] = gt_bboxes[match[pos]].to(bbox_preds.dtype)
    weights = torch.zeros_like(bbox_preds)
    weights[pos] = 1.0
    num_pos = int(pos.sum())
    cls_avg = max(num_pos * 1.0, 1)                          # bg_cls_weight = 0 (:153), sync_cls_avg_factor False
    loss_cls = sigmoid_focal_loss_sum(cls_scores, labels, num_classes) / cls_avg * loss_cls_weight
    nt = normalize_bbox(targets)
    ok = torch.isfinite(nt).all(-1)
    weights = weights * torch.tensor(code_weights, dtype=bbox_preds.dtype)
    loss_bbox = ((bbox_preds[ok, :10] - nt[ok, :10]).abs() * weights[ok, :10]).sum() / max(num_pos, 1) * loss_bbox_weight
    return torch.nan_to_num(loss_cls), torch.nan_to_num(loss_bbox), match


def dn_loss_single(cls_scores, bbox_preds, known_bboxs, known_labels, num_total_pos, split, num_classes=10,
                   code_weights=CODE_WEIGHTS, loss_cls_weight=2.0, loss_bbox_weight=0.25, neg_bbox_loss=False):
    """CrossAttentionBoxHead.dn_loss_single (:477-538): rows are the denoising queries, targets given (no matching)."""
    cls_avg = max(num_total_pos * 3.14159 / 6 * split * split * split, 1)
    loss_cls = sigmoid_focal_loss_sum(cls_scores, known_labels.long(), num_classes) / cls_avg * loss_cls_weight
    known_bboxs = known_bboxs.clone()
    if not neg_bbox_loss:
        known_bboxs[known_labels == num_classes] = 0
    nt = normalize_bbox(known_bboxs)
    ok = torch.isfinite(nt).all(-1)
    weights = torch.ones_like(bbox_preds) * torch.tensor(code_weights, dtype=bbox_preds.dtype)
    weights[:, 6:8] = 0
    loss_bbox = ((bbox_preds[ok, :10] - nt[ok, :10]).abs() * weights[ok, :10]).sum() / max(num_total_pos, 1) * loss_bbox_weight
    return torch.nan_to_num(loss_cls), torch.nan_to_num(loss_bbox)


def prepare_for_dn(reference_points, gt_bboxes, gt_labels, rnd, scalar=10, noise_scale=1.0, noise_trans=0.0, split=0.75,
                   num_classes=10, pc_range=PC_RANGE, eps=1e-4):
    """MV2DSHead.prepare_for_dn, training branch, one sample (RH/mv2d_s_head.py:39-121).  `rnd` [G*scalar,3] in [0,1) replaces
    torch.rand_like.  Returns (padded reference points [1,pad+R,3], attn_mask bool [T,T], known_labels, known_bboxs, pad_size)."""
    G, R = gt_bboxes.size(0), reference_points.size(0)
    known_labels = gt_labels.repeat(scalar, 1).view(-1).long()
    known_bboxs = gt_bboxes.repeat(scalar, 1)
    center = known_bboxs[:, :3].clone()
    scale = known_bboxs[:, 3:6].clone()
    if noise_scale > 0:
        diff = scale / 2 + noise_trans
        rand_prob = rnd * 2 - 1.0
        center += torch.mul(rand_prob, diff) * noise_scale
        for k in range(3):
            center[..., k:k + 1] = (center[..., k:k + 1] - pc_range[k]) / (pc_range[k + 3] - pc_range[k])
        center = center.clamp(min=0.0 + eps, max=1.0 - eps)
        known_labels[torch.norm(rand_prob, 2, 1) > split] = num_classes
    single_pad = G
    pad = single_pad * scalar
    padded = torch.cat([torch.zeros(pad, 3), reference_points], 0).unsqueeze(0)
    if pad:
        padded[0, :pad] = center                       # map_known_indice = arange(G) + single_pad * i: the identity for one sample
    T = pad + R
    attn_mask = torch.zeros(T, T, dtype=torch.bool)
    attn_mask[pad:, :pad] = True
    for i in range(scalar):
        attn_mask[single_pad * i:single_pad * (i + 1), single_pad * (i + 1):pad] = True
        attn_mask[single_pad * i:single_pad * (i + 1), :single_pad * i] = True
    return padded, attn_mask, known_labels, known_bboxs, pad


# ---- rotated BEV NMS for nms_thr < 1 (mmdet3d box3d_multiclass_nms -> nms_bev -> mmcv nms_rotated; third party, parity unpinned) -------
def _bev_corners(b):
    import numpy as np
    cx, cy, dx, dy, yaw = float(b[0]), float(b[1]), float(b[3]), float(b[4]), float(b[6])
    c, s = np.cos(yaw), np.sin(yaw)
    loc = np.array([[dx / 2, dy / 2], [-dx / 2, dy / 2], [-dx / 2, -dy / 2], [dx / 2, -dy / 2]])
    return loc @ np.array([[c, s], [-s, c]]) + np.array([cx, cy])


def rotated_iou_bev(a, b):
    """IoU of the rotated rectangles (x, y, dx, dy, yaw) of two 9-code boxes (fp64; convex clipping, one half plane at a time)."""
    import numpy as np
    P, Q = _bev_corners(a), _bev_corners(b)
    poly = [tuple(p) for p in P]
    for e in range(4):
        q0, q1 = Q[e], Q[(e + 1) % 4]
        ex, ey = q1 - q0
        side = lambda p: ex * (p[1] - q0[1]) - ey * (p[0] - q0[0])  # noqa: E731
        out = []
        for i in range(len(poly)):
            s_, t_ = poly[i], poly[(i + 1) % len(poly)]
            ds, dt = side(s_), side(t_)
            if ds >= 0:
                out.append(s_)
            if (ds >= 0) != (dt >= 0):
                u = ds / (ds - dt)
                out.append((s_[0] + u * (t_[0] - s_[0]), s_[1] + u * (t_[1] - s_[1])))
        poly = out
        if not poly:
            break
    inter = 0.5 * abs(sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1] for i in range(len(poly)))) if poly else 0.0
    aa, ab = abs(float(a[3]) * float(a[4])), abs(float(b[3]) * float(b[4]))
    return inter / max(aa + ab - inter, 1e-8)


def nms_bev(boxes, scores, labels, thr):
    """Per class, greedy in score order (ties: lower index first): keep mask [n] bool."""
    import numpy as np
    boxes, scores, labels = np.asarray(boxes, np.float64), np.asarray(scores, np.float64), np.asarray(labels)
    keep = np.ones(len(scores), bool)
    for c in np.unique(labels):
        idx = [i for i in np.argsort(-scores, kind='stable') if labels[i] == c]
        for a_, i in enumerate(idx):
            if not keep[i]:
                continue
            for j in idx[a_ + 1:]:
                if keep[j] and rotated_iou_bev(boxes[j], boxes[i]) > thr:
                    keep[j] = False
    return keep

