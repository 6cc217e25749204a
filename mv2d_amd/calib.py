"""Per-frame calibration tables derived from ``img_metas`` on the host (tiny, O(V + V*h*w) numbers).

Everything the reference derives from img_metas with host numpy / tiny tensor ops is computed here once per
frame with the SAME operations and dtypes, then shipped to the GPU in one pinned copy:

* per-view intrinsics / extrinsics fp64 (RH/mv2d_head.py:58-59),
* img2lidar = np.linalg.inv(lidar2img) fp64 (MU/pe.py:111-114),
* trans[a,b] = lidar2img[b] @ inverse(lidar2img[a]) fp64 (RH/utils/box_correlation.py:117-122),
* frustum pixel-centre / LID depth tables fp64 (MU/pe.py:93-104),
* padding mask (nearest interpolation, RH/mv2d_t_head.py:69-76) and the normalised (view, y, x) cumsum embeds of
  SinePositionalEncoding3D (MU/positional_encoding.py:62-77),
* constant tables: linspace(0,1,4), LID depths of BoxCorrelation (RH/utils/box_correlation.py:198,221-225),
  dim_t = 10000 ** (2*(i//2)/128) (MU/pe.py:24-25).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def constant_tables(sample_size=4, num_depth=8, depth_start=0.5, depth_end=70.0, num_pos_feats=128, temperature=10000):
    lin = torch.linspace(0, 1, sample_size)
    index = torch.arange(0, num_depth, 1).float()
    bin_size = (depth_end - depth_start) / (num_depth * (1 + num_depth))
    depths = depth_start + bin_size * index * (index + 1)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    return dict(lin=lin.contiguous(), depths=depths.contiguous(), dim_t=dim_t.contiguous())


def geometry_tables(img_metas):
    """The part of the calibration that changes with every frame (camera matrices; on the two-frame path the previous frame's extrinsics
    carry the ego motion): per-view K / E, img2lidar, the view-to-view transforms, time stamps.  A few 4x4 inverses: tens of microseconds."""
    V = len(img_metas)
    viewK = torch.from_numpy(np.stack([np.asarray(m['intrinsics'], dtype=np.float64) for m in img_metas])).reshape(V, 16)
    viewE = torch.from_numpy(np.stack([np.asarray(m['extrinsics'], dtype=np.float64) for m in img_metas])).reshape(V, 16)
    l2i_np = np.stack([np.asarray(m['lidar2img'], dtype=np.float64) for m in img_metas])
    img2lidar = torch.from_numpy(np.asarray([np.linalg.inv(x) for x in l2i_np])).reshape(V, 16)
    l2i = torch.from_numpy(l2i_np)
    # torch.inverse on a handful of 4x4 matrices: with the default intra-op thread pool the call costs 30-70 ms of thread wake-ups
    # (measured; 0.03 ms on one thread, bitwise the same result) -- a per-frame host cost on the two-frame path
    nt = torch.get_num_threads()
    if nt > 1:
        torch.set_num_threads(1)
    try:
        trans = torch.matmul(l2i[None], torch.inverse(l2i)[:, None]).reshape(V, V, 16)
    finally:
        if nt > 1:
            torch.set_num_threads(nt)
    ts = np.array([mm.get('timestamp', 0.0) for mm in img_metas])
    return dict(viewK=viewK.contiguous(), viewE=viewE.contiguous(), img2lidar=img2lidar.contiguous(), trans=trans.contiguous(), timestamps=ts)


def geometry_tables_batch(metas_list):
    """geometry_tables of B samples (V views each) with ONE batched call per operation — bitwise the same matrices as B per-sample
    calls (checked in tests/test_calib_cpu.py), a fraction of the Python / dispatch time: this runs for every frame.
    Returns (mats [3, B*V, 4, 4] fp64 = intrinsics | extrinsics | lidar2img as given, img2lidar [B*V,16], trans [B, V, V, 16],
    timestamps [B, V])."""
    B, V = len(metas_list), len(metas_list[0])
    flat = [m for metas in metas_list for m in metas]
    mats = np.stack([np.stack([np.asarray(m[k]) for m in flat]) for k in ('intrinsics', 'extrinsics', 'lidar2img')]).astype(np.float64, copy=False)
    img2lidar = torch.from_numpy(np.linalg.inv(mats[2])).reshape(B * V, 16)
    l2i = torch.from_numpy(mats[2]).view(B, V, 4, 4)
    nt = torch.get_num_threads()
    if nt > 1:
        torch.set_num_threads(1)                       # see geometry_tables
    try:
        inv = torch.inverse(l2i.view(-1, 4, 4)).view(B, V, 4, 4)
        trans = torch.matmul(l2i[:, None], inv[:, :, None]).reshape(B, V, V, 16)
    finally:
        if nt > 1:
            torch.set_num_threads(nt)
    ts = np.array([m.get('timestamp', 0.0) for m in flat], dtype=np.float64).reshape(B, V)
    return mats, img2lidar, trans, ts


def shape_tables(shapes, h, w, stride=16, depth_num=64, depth_start=1, position_range=(-61.2, -61.2, -10.0, 61.2, 61.2, 10.0),
                 eps=1e-6, scale=2 * math.pi):
    """The part that depends only on the padding geometry of the rig — shapes = ((pad_h, pad_w), ((img_h, img_w) per view)) — and on
    the map size: frustum tables, padding mask, the cumsum embeds of SinePositionalEncoding3D.  Constant for a deployment; callers
    cache it by its arguments (the ones-image it interpolates is V x pad_h x pad_w floats)."""
    (pad_h, pad_w), img_shapes = shapes
    V = len(img_shapes)
    coords_h = (torch.arange(h).double() + 0.5) * pad_h / h - 0.5
    coords_w = (torch.arange(w).double() + 0.5) * pad_w / w - 0.5
    index = torch.arange(0, depth_num, 1).double()
    bin_size = (position_range[3] - depth_start) / (depth_num * (1 + depth_num))
    coords_d = depth_start + bin_size * index * (index + 1)
    m = torch.ones((1, V, pad_h, pad_w), dtype=torch.float32)
    for i in range(V):
        ih, iw = img_shapes[i]
        m[0, i, :ih, :iw] = 0
    pad = F.interpolate(m, size=(h, w)).to(torch.bool)                                  # [1,V,h,w]
    not_mask = 1 - pad.to(torch.int)
    n_e = not_mask.cumsum(1, dtype=torch.float32)
    y_e = not_mask.cumsum(2, dtype=torch.float32)
    x_e = not_mask.cumsum(3, dtype=torch.float32)
    y_e = (y_e - 0.5) * stride
    x_e = (x_e - 0.5) * stride
    n_e = n_e / (n_e[:, -1:, :, :] + eps) * scale
    y_e = y_e / (y_e[:, :, -1:, :] + eps) * scale
    x_e = x_e / (x_e[:, :, :, -1:] + eps) * scale
    embeds = torch.stack([n_e[0].reshape(-1), y_e[0].reshape(-1), x_e[0].reshape(-1)]).contiguous()   # [3,P]
    return dict(coords_w=coords_w.contiguous(), coords_h=coords_h.contiguous(), coords_d=coords_d.contiguous(),
                pad_mask=pad[0].reshape(-1).to(torch.uint8).contiguous(), embeds=embeds, pad_h=int(pad_h), pad_w=int(pad_w))


def meta_shapes(img_metas):
    """Hashable padding geometry of a sample's views: ((pad_h, pad_w), ((img_h, img_w), ...))."""
    pad_h, pad_w = img_metas[0]['pad_shape'][:2]
    return (int(pad_h), int(pad_w)), tuple((int(m['img_shape'][0]), int(m['img_shape'][1])) for m in img_metas)


def frame_tables(img_metas, h, w, stride=16, depth_num=64, depth_start=1, position_range=(-61.2, -61.2, -10.0, 61.2, 61.2, 10.0),
                 eps=1e-6, scale=2 * math.pi):
    """Host tensors for one frame (dict of contiguous CPU tensors): geometry_tables + shape_tables."""
    out = geometry_tables(img_metas)
    out.update(shape_tables(meta_shapes(img_metas), h, w, stride, depth_num, depth_start, position_range, eps, scale))
    return out
