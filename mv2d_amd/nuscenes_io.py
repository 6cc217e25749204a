"""nuScenes I/O contract on the input side of the hot path — host code, numpy only.

What the head consumes per view is ``img_metas[v]`` = {intrinsics 4x4, extrinsics 4x4 (lidar->camera, stored transposed), lidar2img 4x4,
timestamp, img_shape, pad_shape}; this module builds those from the on-disk records exactly as the reference's dataset / pipeline do:

* ``camera_geometry``        — ``CustomNuScenesDataset.get_data_info`` (test mode), mmdet3d_plugin/datasets/custom_nuscenes_dataset.py:100-163
* ``sweep_camera_record``    — ``add_frame`` of tools/generate_sweep_pkl.py:32-82 (the sweep cameras in the key frame's lidar frame)
* ``append_sweeps``          — ``LoadMultiViewImageFromMultiSweepsFiles.__call__``, mmdet3d_plugin/datasets/pipelines/loading.py:53-163
* ``sample_augmentation`` / ``image_aug_matrix`` / ``resize_crop_flip`` — ``ResizeCropFlipImageMono`` (incl. the 2-D box branch),
  mmdet3d_plugin/datasets/pipelines/transform_3d.py:456-591
* ``split_view_metas``       — the per-view split of ``MV2D.simple_test``, mmdet3d_plugin/models/detectors/mv2d.py:232-246

* ``global_rot_scale_trans`` — ``GlobalRotScaleTransImage`` (transform_3d.py:822-904); ``center_match`` — custom_nuscenes_dataset.py:199-208;
  ``pad_multi_view`` / ``normalize_multiview`` — ``PadMultiViewImage`` / ``NormalizeMultiviewImage`` (transform_3d.py:121-203)

* ``parse_ann_2d`` / ``attach_2d_annotations`` — ``get_ann_info_2d`` and the training branch of ``get_data_info``
  (custom_nuscenes_dataset.py:165-196,262-322)

Out of this slice: image decoding itself (an ``imread`` callable is injected), reading the COCO-style 2-D annotation files of the training branch, and the
result JSON (``_format_bbox`` / nuScenes eval live in mmdet3d and the nuscenes devkit, not in the reference tree).
"""
import numpy as np

SENSORS = ['CAM_FRONT', 'CAM_FRONT_RIGHT', 'CAM_FRONT_LEFT', 'CAM_BACK', 'CAM_BACK_LEFT', 'CAM_BACK_RIGHT']


def _lidar2cam(sensor2lidar_rotation, sensor2lidar_translation):
    """The 4x4 the reference calls ``lidar2cam_rt`` (row-vector convention: points_h @ lidar2cam_rt = points in the camera frame)."""
    r = np.linalg.inv(sensor2lidar_rotation)
    t = sensor2lidar_translation @ r.T
    rt = np.eye(4)
    rt[:3, :3] = r.T
    rt[3, :3] = -t
    return rt


def _viewpad(intrinsic):
    intrinsic = np.asarray(intrinsic)
    pad = np.eye(4)
    pad[:intrinsic.shape[0], :intrinsic.shape[1]] = intrinsic
    return pad


def camera_geometry(info):
    """One key-frame record of the info pkl -> the geometry part of ``get_data_info``'s input_dict (fp64 matrices, in camera order)."""
    out = dict(sample_idx=info['token'], pts_filename=info['lidar_path'], sweeps=info['sweeps'], timestamp=info['timestamp'] / 1e6,
               img_timestamp=[], img_filename=[], lidar2img=[], intrinsics=[], extrinsics=[])
    for _cam, c in info['cams'].items():
        out['img_timestamp'].append(c['timestamp'] / 1e6)
        out['img_filename'].append(c['data_path'])
        rt = _lidar2cam(c['sensor2lidar_rotation'], c['sensor2lidar_translation'])
        pad = _viewpad(c['cam_intrinsic'])
        out['lidar2img'].append(pad @ rt.T)
        out['intrinsics'].append(pad)
        out['extrinsics'].append(rt)        # lidar -> camera, transposed (the reference's comment at :150)
    out['img_info'] = info
    return out


def quaternion_rotation_matrix(q):
    """(w, x, y, z) -> 3x3 rotation (what pyquaternion's ``Quaternion(q).rotation_matrix`` returns; the quaternion is normalised first)."""
    w, x, y, z = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def sweep_camera_record(sensor2ego_rotation, sensor2ego_translation, ego2global_rotation, ego2global_translation, cam_intrinsic,
                        key_l2e_r_mat, key_l2e_t, key_e2g_r_mat, key_e2g_t):
    """A sweep camera in the key frame's lidar frame (``add_frame``): rotations of the sweep as quaternions (w, x, y, z), the key frame's
    lidar->ego and ego->global as 3x3 matrices + translations.  Returns sensor2lidar_rotation / _translation (fp64) and the float32
    intrinsics / extrinsics / lidar2img the loader appends."""
    s2e = quaternion_rotation_matrix(sensor2ego_rotation)
    e2g = quaternion_rotation_matrix(ego2global_rotation)
    back = np.linalg.inv(key_e2g_r_mat).T @ np.linalg.inv(key_l2e_r_mat).T
    R = (s2e.T @ e2g.T) @ back
    T = (np.asarray(sensor2ego_translation) @ e2g.T + np.asarray(ego2global_translation)) @ back
    T -= np.asarray(key_e2g_t) @ back + np.asarray(key_l2e_t) @ np.linalg.inv(key_l2e_r_mat).T
    rec = dict(sensor2lidar_rotation=R.T, sensor2lidar_translation=T)
    rt = _lidar2cam(rec['sensor2lidar_rotation'], rec['sensor2lidar_translation'])
    pad = _viewpad(np.array(cam_intrinsic))
    rec['intrinsics'] = pad.astype(np.float32)
    rec['extrinsics'] = rt.astype(np.float32)
    rec['lidar2img'] = (pad @ rt.T).astype(np.float32)
    return rec


def append_sweeps(results, sweeps_num=5, sweep_range=(3, 27), sweeps_id=None, sensors=SENSORS, test_mode=True, pad_empty_sweeps=False,
                  prob=1.0, to_float32=False, imread=None, rng=np.random):
    """``LoadMultiViewImageFromMultiSweepsFiles``: appends the chosen previous frame(s) — images (through ``imread(path)``; None keeps the
    paths only), file names, geometry — and turns ``results['timestamp']`` into the per-image time offsets to the lidar timestamp.
    ``rng`` supplies ``random()`` / ``choice()`` for the training-time choice (the reference uses the global numpy state)."""
    imgs = list(results.get('img', []))
    lidar_ts = results['timestamp']
    ts = [lidar_ts - t for t in results['img_timestamp']]
    out_imgs, out_ts = list(imgs), list(ts)
    nums = len(results['img_timestamp'])
    if pad_empty_sweeps and len(results['sweeps']) == 0:
        for _ in range(sweeps_num):
            out_imgs.extend(imgs)
            mean_time = (sweep_range[0] + sweep_range[1]) / 2.0 * 0.083
            out_ts.extend([t + mean_time for t in ts])
            for j in range(nums):
                results['filename'].append(results['filename'][j])
                for k in ('lidar2img', 'intrinsics', 'extrinsics'):
                    results[k].append(np.copy(results[k][j]))
    else:
        n_sw = len(results['sweeps'])
        mid = [int((sweep_range[0] + sweep_range[1]) / 2) - 1]
        if sweeps_id:
            choices = sweeps_id
        elif n_sw <= sweeps_num:
            choices = np.arange(n_sw)
        elif test_mode:
            choices = mid
        elif rng.random() < prob:
            cand = list(range(sweep_range[0], min(sweep_range[1], n_sw))) if sweep_range[0] < n_sw else list(range(*sweep_range))
            choices = rng.choice(cand, sweeps_num, replace=False)
        else:
            choices = mid
        for idx in choices:
            si = min(idx, n_sw - 1)
            sweep = results['sweeps'][si]
            if len(sweep.keys()) < len(sensors):
                sweep = results['sweeps'][si - 1]
            results['filename'].extend([sweep[s]['data_path'] for s in sensors])
            if imread is not None:
                for s in sensors:
                    im = imread(sweep[s]['data_path'])
                    out_imgs.append(im.astype(np.float32) if to_float32 else im)
            out_ts.extend([lidar_ts - sweep[s]['timestamp'] / 1e6 for s in sensors])
            for s in sensors:
                results['lidar2img'].append(sweep[s]['lidar2img'])
                results['intrinsics'].append(sweep[s]['intrinsics'])
                results['extrinsics'].append(sweep[s]['extrinsics'])
    results['img'] = out_imgs
    results['timestamp'] = out_ts
    return results


def sample_augmentation(conf, training, rng=np.random):
    """``ResizeCropFlipImage._sample_augmentation``: (resize, resize_dims, crop, flip, rotate); draws in the reference's order."""
    H, W = conf['H'], conf['W']
    fH, fW = conf['final_dim']
    if training:
        resize = rng.uniform(*conf['resize_lim'])
        resize_dims = (int(W * resize), int(H * resize))
        newW, newH = resize_dims
        crop_h = int((1 - rng.uniform(*conf['bot_pct_lim'])) * newH) - fH
        crop_w = int(rng.uniform(0, max(0, newW - fW)))
        crop = (crop_w, crop_h, crop_w + fW, crop_h + fH)
        flip = bool(conf['rand_flip'] and rng.choice([0, 1]))
        rotate = rng.uniform(*conf['rot_lim'])
    else:
        resize = max(fH / H, fW / W)
        resize_dims = (int(W * resize), int(H * resize))
        newW, newH = resize_dims
        crop_h = int((1 - np.mean(conf['bot_pct_lim'])) * newH) - fH
        crop_w = int(max(0, newW - fW) / 2)
        crop = (crop_w, crop_h, crop_w + fW, crop_h + fH)
        flip, rotate = False, 0
    return resize, resize_dims, crop, flip, rotate


def image_aug_matrix(resize, crop, flip, rotate):
    """The 3x3 ``ida_mat`` of ``_img_transform`` (fp32 like the reference's torch.Tensor arithmetic): pixel (u, v, 1) of the source image
    -> pixel of the resized / cropped / flipped / rotated image."""
    f32 = np.float32
    rot = np.eye(2, dtype=f32) * f32(resize)
    tran = np.zeros(2, f32) - np.array(crop[:2], f32)
    if flip:
        A = np.array([[-1, 0], [0, 1]], f32)
        b = np.array([crop[2] - crop[0], 0], f32)
        rot = A @ rot
        tran = A @ tran + b
    h = rotate / 180 * np.pi
    A = np.array([[np.cos(h), np.sin(h)], [-np.sin(h), np.cos(h)]], f32)
    b = np.array([crop[2] - crop[0], crop[3] - crop[1]], f32) / f32(2)
    b = A @ (-b) + b
    rot = A @ rot
    tran = A @ tran + b
    m = np.eye(3, dtype=f32)
    m[:2, :2] = rot
    m[:2, 2] = tran
    return m


def _aug_boxes_2d(boxes, resize, crop, flip, rotate, filter_small):
    """The 2-D box side of ``ResizeCropFlipImageMono`` (transform_3d.py:607-664) for one view: resize, crop + clip, optional flip, rotate
    (bounding box of the rotated corners, clipped).  Returns (boxes, keep) with ``keep`` the index list of the surviving input rows
    (area > 64 after the crop and again after the rotation) when ``filter_small`` — the ignore boxes are only filtered after the crop and
    neither clipped nor filtered after the rotation, as in the reference."""
    b = boxes * resize
    b[:, 0::2] = np.clip(b[:, 0::2], crop[0], crop[2]) - crop[0]
    b[:, 1::2] = np.clip(b[:, 1::2], crop[1], crop[3]) - crop[1]
    keep = np.nonzero((b[:, 2:] - b[:, :2]).prod(1) > 64)[0]
    b = b[keep]
    if flip:
        f = b.copy()
        wd = crop[2] - crop[0]
        f[..., 0::4] = wd - b[..., 2::4]
        f[..., 2::4] = wd - b[..., 0::4]
        b = f
    h = rotate / 180 * np.pi
    A = np.array([[np.cos(h), np.sin(h)], [-np.sin(h), np.cos(h)]], np.float32)
    t = np.array([crop[2] - crop[0], crop[3] - crop[1]], np.float32) / np.float32(2)
    t = A @ (-t) + t
    corners = np.stack([b[:, 0], b[:, 1], b[:, 0], b[:, 3], b[:, 2], b[:, 3], b[:, 2], b[:, 1]], axis=1).reshape(-1, 4, 2)
    corners = corners @ A.T + t[None, None]
    b = np.concatenate([corners.min(1), corners.max(1)], axis=1)
    if filter_small:
        b[:, 0::2] = np.clip(b[:, 0::2], 0, crop[2] - crop[0])
        b[:, 1::2] = np.clip(b[:, 1::2], 0, crop[3] - crop[1])
        k2 = np.nonzero((b[:, 2:] - b[:, :2]).prod(1) > 64)[0]
        b, keep = b[k2], keep[k2]
    return b, keep


def resize_crop_flip(results, conf, training=False, rng=np.random, transform_images=True, with_bbox_2d=False, num_views=6):
    """``ResizeCropFlipImageMono.__call__`` (``with_bbox_2d=False``): one augmentation for all views; intrinsics[:3,:3] <- ida @ intrinsics
    [:3,:3] in place, lidar2img rebuilt as intrinsics @ extrinsics.T; the images go through PIL as in the reference when
    ``transform_images``."""
    resize, resize_dims, crop, flip, rotate = sample_augmentation(conf, training, rng)
    ida = image_aug_matrix(resize, crop, flip, rotate)
    if transform_images and results.get('img'):
        from PIL import Image
        new = []
        for im in results['img']:
            p = Image.fromarray(np.uint8(im)).resize(resize_dims).crop(crop)
            if flip:
                p = p.transpose(method=Image.FLIP_LEFT_RIGHT)
            new.append(np.array(p.rotate(rotate)).astype(np.float32))
        results['img'] = new
    for i in range(len(results['intrinsics'])):
        results['intrinsics'][i][:3, :3] = ida @ results['intrinsics'][i][:3, :3]
    results['lidar2img'] = [results['intrinsics'][i] @ results['extrinsics'][i].T for i in range(len(results['extrinsics']))]
    if with_bbox_2d:
        # the per-view 2-D ground truth of the key frame's views (transform_3d.py:593-674)
        out = dict(gt_bboxes_2d=[], gt_labels_2d=[], gt_bboxes_2d_to_3d=[], gt_bboxes_ignore=[])
        for i in range(min(len(results['intrinsics']), num_views)):
            b, keep = _aug_boxes_2d(results['gt_bboxes_2d'][i], resize, crop, flip, rotate, True)
            ign, _ = _aug_boxes_2d(results['gt_bboxes_ignore'][i], resize, crop, flip, rotate, False)
            out['gt_bboxes_2d'].append(b)
            out['gt_labels_2d'].append(results['gt_labels_2d'][i][keep])
            out['gt_bboxes_2d_to_3d'].append(results['gt_bboxes_2d_to_3d'][i][keep])
            out['gt_bboxes_ignore'].append(ign)
        results.update(out)
    return results


def split_view_metas(img_metas_views, num_views):
    """One sample's collated img_metas (lists over views) -> the per-view dicts the head takes (``MV2D.simple_test``)."""
    out = []
    for j in range(num_views):
        m = dict(num_views=num_views)
        for k, v in img_metas_views.items():
            if isinstance(v, list):
                m[k] = v[j]
            elif k == 'ori_shape':
                m[k] = v[:3]
            else:
                m[k] = v
        out.append(m)
    return out


def global_rot_scale_trans(results, rot_range=(-0.3925, 0.3925), scale_ratio_range=(0.95, 1.05), reverse_angle=False, rng=np.random):
    """``GlobalRotScaleTransImage.__call__`` (mmdet3d_plugin/datasets/pipelines/transform_3d.py:822-904): a random rotation about z and a
    random isotropic scale of the lidar frame, folded into every view's ``lidar2img`` / ``extrinsics`` (fp32, like the reference's torch
    arithmetic) and applied to ``results['gt_bboxes_3d']`` through its own ``rotate`` / ``scale`` methods (mmdet3d's box class, third
    party).  Returns (rot_angle passed to the boxes, scale_ratio)."""
    f32 = np.float32
    angle = rng.uniform(*rot_range)
    c, s = f32(np.cos(f32(angle))), f32(np.sin(f32(angle)))
    rot = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], f32)
    rot_inv = rot if reverse_angle else np.linalg.inv(rot).astype(f32)

    def fold(m_inv):
        for v in range(len(results['lidar2img'])):
            results['lidar2img'][v] = np.asarray(results['lidar2img'][v], f32) @ m_inv
            results['extrinsics'][v] = m_inv.T @ np.asarray(results['extrinsics'][v], f32)
    fold(rot_inv)
    box_angle = -angle if reverse_angle else angle
    results['gt_bboxes_3d'].rotate(np.array(box_angle))
    ratio = rng.uniform(*scale_ratio_range)
    fold(np.linalg.inv(np.diag(np.array([ratio, ratio, ratio, 1], f32))).astype(f32))
    results['gt_bboxes_3d'].scale(ratio)
    return box_angle, ratio


def center_match(bboxes_a, bboxes_b):
    """``CustomNuScenesDataset.center_match`` (custom_nuscenes_dataset.py:199-208): for every camera-frame 2-D annotation the index of the
    3-D box whose centre coincides (L1 distance <= 1e-3), else -1."""
    cts_a, cts_b = bboxes_a[:, :3], bboxes_b[:, :3]
    if len(cts_a) == 0 or len(cts_b) == 0:
        return np.zeros(len(cts_a), dtype=np.int32) - 1
    dist = np.abs(cts_a[:, None] - cts_b[None]).sum(-1)
    match = dist.argmin(1)
    match[dist.min(1) > 1e-3] = -1
    return match


def pad_multi_view(results, size=None, size_divisor=None, pad_val=0):
    """``PadMultiViewImage`` (transform_3d.py:121-170; mmcv's ``impad`` / ``impad_to_multiple`` restated: constant padding at the bottom /
    right): records ``img_shape`` (before) and ``pad_shape`` (after) — the two shapes the head's padding mask is built from."""
    assert (size is None) != (size_divisor is None)
    out = []
    for img in results['img']:
        h, w = img.shape[:2]
        H, W = size if size is not None else (int(np.ceil(h / size_divisor)) * size_divisor, int(np.ceil(w / size_divisor)) * size_divisor)
        p = np.full((H, W) + img.shape[2:], pad_val, dtype=img.dtype)
        p[:h, :w] = img
        out.append(p)
    results['img_shape'] = [img.shape for img in results['img']]
    results['img'] = out
    results['pad_shape'] = [img.shape for img in out]
    results['pad_fixed_size'], results['pad_size_divisor'] = size, size_divisor
    return results


def normalize_multiview(results, mean, std, to_rgb=True):
    """``NormalizeMultiviewImage`` (transform_3d.py:173-203; mmcv's ``imnormalize`` restated: optional BGR->RGB, (img - mean) * (1 / std) in
    fp32)."""
    mean, std = np.array(mean, np.float32), np.array(std, np.float32)
    stdinv = (1 / np.float64(std.reshape(1, -1))).astype(np.float32)
    out = []
    for img in results['img']:
        x = np.asarray(img, np.float32)
        if to_rgb:
            x = x[..., ::-1]
        out.append((x - mean.reshape(1, -1)) * stdinv)
    results['img'] = out
    results['img_norm_cfg'] = dict(mean=mean, std=std, to_rgb=to_rgb)
    return results


def parse_ann_2d(img_info_2d, ann_info_2d, cat_ids, cat2label):
    """``CustomNuScenesDataset.get_ann_info_2d`` (custom_nuscenes_dataset.py:262-322): the COCO-style per-image annotation list (dicts with
    ``bbox`` = x, y, w, h, ``area``, ``category_id``, ``bbox_cam3d``, optional ``ignore`` / ``iscrowd``) -> boxes (x1, y1, x2, y2), labels,
    camera-frame 3-D boxes and ignore boxes.  Boxes outside the image, degenerate boxes and unknown categories are dropped."""
    boxes, labels, ignore, cam3d = [], [], [], []
    for ann in ann_info_2d:
        if ann.get('ignore', False):
            continue
        x1, y1, w, h = ann['bbox']
        inter_w = max(0, min(x1 + w, img_info_2d['width']) - max(x1, 0))
        inter_h = max(0, min(y1 + h, img_info_2d['height']) - max(y1, 0))
        if inter_w * inter_h == 0 or ann['area'] <= 0 or w < 1 or h < 1 or ann['category_id'] not in cat_ids:
            continue
        box = [x1, y1, x1 + w, y1 + h]
        if ann.get('iscrowd', False):
            ignore.append(box)
        else:
            boxes.append(box)
            labels.append(cat2label[ann['category_id']])
            cam3d.append(np.array(ann['bbox_cam3d']).reshape(1, -1).squeeze())
    return dict(bboxes_cam=np.array(cam3d, dtype=np.float32) if cam3d else np.zeros((0, 6), np.float32),
                bboxes_2d=np.array(boxes, dtype=np.float32) if boxes else np.zeros((0, 4), np.float32),
                gt_bboxes_ignore=np.array(ignore, dtype=np.float32) if ignore else np.zeros((0, 4), np.float32),
                labels=np.array(labels, dtype=np.int64))


def attach_2d_annotations(input_dict, anns_2d, centers_lidar, gt_labels_3d):
    """Training branch of ``get_data_info`` (custom_nuscenes_dataset.py:165-196): per view, the 2-D annotations of that image
    (``parse_ann_2d`` results, in ``img_filename`` order) and the index of the 3-D box each belongs to — the camera-frame centre of the
    annotation against the lidar-frame gravity centres moved into that camera (``center_match``).  Returns the four lists the reference
    puts into ``ann_info``; raises if a matched pair disagrees on the label (the reference asserts)."""
    out = dict(gt_bboxes_2d=[], gt_labels_2d=[], gt_bboxes_2d_to_3d=[], gt_bboxes_ignore=[])
    hom = np.concatenate([centers_lidar, np.ones((len(centers_lidar), 1))], axis=1)
    for cam_i, ann in enumerate(anns_2d):
        lidar2cam = input_dict['extrinsics'][cam_i].T
        centers_cam = (hom @ lidar2cam.T)[:, :3]
        match = center_match(ann['bboxes_cam'], centers_cam)
        if not (ann['labels'][match > -1] == np.asarray(gt_labels_3d)[match[match > -1]]).all():
            raise AssertionError('a 2-D annotation and the 3-D box at the same centre disagree on the class')
        out['gt_bboxes_2d'].append(ann['bboxes_2d'])
        out['gt_bboxes_2d_to_3d'].append(match)
        out['gt_labels_2d'].append(ann['labels'])
        out['gt_bboxes_ignore'].append(ann['gt_bboxes_ignore'])
    return out
