"""Output side of the nuScenes I/O contract: what the reference's dataset class does with the detector's results,
and the remaining input-side readers.  Host-side data plumbing (numpy / json), no GPU involved.

* ``format_results`` / ``format_bbox`` — ``CustomNuScenesDataset.format_results`` (mmdet3d_plugin/datasets/custom_nuscenes_dataset.py:324-370):
  the result list (bare dicts or ``{'pts_bbox': ...}`` wrappers, detectors/mv2d.py:283-292) -> the nuScenes detection submission
  ``results_nusc.json``.  The reference inherits ``_format_bbox`` from mmdet3d 1.0.0's ``NuScenesDataset`` (``output_to_nusc_box``,
  ``lidar_nusc_box_to_global``: LiDAR -> ego -> global with pyquaternion / the devkit's ``Box``, class-range filter, attribute rule);
  mmdet3d, nuscenes-devkit and pyquaternion are third party and absent from the reference tree, so that part is restated here with
  plain numpy quaternions and is PARITY UNPINNED.
* ``metrics_detail`` — the bookkeeping of ``_evaluate_single`` (:403-426) on the devkit's ``metrics_summary.json``; ``evaluate`` runs the
  devkit's ``NuScenesEval`` when it is installed and says so when it is not.
* ``Coco2D`` / ``load_annotations_2d`` / ``impath_to_ann2d`` — :73-98, the COCO-json reader of the 2-D annotations (pycocotools' index
  restated: images, annotations per image, categories by name); feeds ``nuscenes_io.parse_ann_2d``.
* ``COLLECT_MONO3D_META_KEYS`` / ``collect_mono3d`` — ``CollectMono3D`` (pipelines/formatting.py:27-45): which keys travel as img_metas.
"""
import json
import os
import tempfile

import numpy as np

# nuscenes.eval.detection configs/detection_cvpr_2019.json: class_range; mmdet3d NuScenesDataset.DefaultAttribute / ErrNameMapping
CLASS_RANGE = {'car': 50, 'truck': 50, 'bus': 50, 'trailer': 50, 'construction_vehicle': 50, 'pedestrian': 40, 'motorcycle': 40,
               'bicycle': 40, 'traffic_cone': 30, 'barrier': 30}
DEFAULT_ATTRIBUTE = {'car': 'vehicle.parked', 'pedestrian': 'pedestrian.moving', 'trailer': 'vehicle.parked', 'truck': 'vehicle.parked',
                     'bus': 'vehicle.moving', 'motorcycle': 'cycle.without_rider', 'construction_vehicle': 'vehicle.parked',
                     'bicycle': 'cycle.without_rider', 'barrier': '', 'traffic_cone': ''}
ERR_NAME_MAPPING = {'trans_err': 'mATE', 'scale_err': 'mASE', 'orient_err': 'mAOE', 'vel_err': 'mAVE', 'attr_err': 'mAAE'}
CLASSES = ('car', 'truck', 'trailer', 'bus', 'construction_vehicle', 'bicycle', 'motorcycle', 'pedestrian', 'traffic_cone', 'barrier')
MODALITY = dict(use_camera=True, use_lidar=False, use_radar=False, use_map=False, use_external=True)

COLLECT_MONO3D_META_KEYS = ('filename', 'ori_shape', 'img_shape', 'lidar2img', 'depth2img', 'cam2img', 'pad_shape', 'scale_factor', 'flip',
                            'pcd_horizontal_flip', 'pcd_vertical_flip', 'box_mode_3d', 'box_type_3d', 'img_norm_cfg', 'pcd_trans',
                            'sample_idx', 'pcd_scale_factor', 'pcd_rotation', 'pcd_rotation_angle', 'pts_filename',
                            'transformation_3d_flow', 'trans_mat', 'affine_aug', 'intrinsics', 'extrinsics', 'timestamp')


# ---- quaternions (w, x, y, z), Hamilton convention like pyquaternion ------------------------------------------------------------------
def quat_axis_z(angle):
    return np.array([np.cos(angle / 2.0), 0.0, 0.0, np.sin(angle / 2.0)])


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def quat_rotation_matrix(q):
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)


def _box_arrays(det):
    """boxes_3d: a LiDARInstance3DBoxes-like object (gravity_center, dims, yaw, tensor) or a [n,9] array of the head's own boxes
    (x, y, z_bottom, dx, dy, dz, yaw, vx, vy — cross_attention_head.py:372-373 moves z to the bottom face)."""
    b = det['boxes_3d']
    if hasattr(b, 'gravity_center'):
        return _np(b.gravity_center), _np(b.dims), _np(b.yaw), _np(b.tensor)[:, 7:9]
    t = _np(b).reshape(-1, 9).astype(np.float64)
    centre = t[:, :3].copy()
    centre[:, 2] += 0.5 * t[:, 5]
    return centre, t[:, 3:6], t[:, 6], t[:, 7:9]


def output_to_nusc_boxes(det):
    """mmdet3d 1.0.0 ``output_to_nusc_box``: list of dict(center, wlh, orientation (quaternion), velocity (3), label, score) in the LiDAR
    frame; dims (dx, dy, dz) -> nuScenes (w, l, h) = (dy, dx, dz), orientation = rotation about +z by yaw."""
    centre, dims, yaw, vel = _box_arrays(det)
    scores, labels = _np(det['scores_3d']), _np(det['labels_3d'])
    out = []
    for i in range(len(scores)):
        out.append(dict(center=np.asarray(centre[i], np.float64), wlh=np.asarray(dims[i], np.float64)[[1, 0, 2]], orientation=quat_axis_z(float(yaw[i])),
                        velocity=np.array([float(vel[i][0]), float(vel[i][1]), 0.0]), label=int(labels[i]), score=float(scores[i])))
    return out


def _rotate(box, q):
    R = quat_rotation_matrix(q)
    box['center'] = R @ box['center']
    box['orientation'] = quat_mul(np.asarray(q, np.float64), box['orientation'])
    box['velocity'] = R @ box['velocity']


def lidar_boxes_to_global(info, boxes, classes=CLASSES, class_range=CLASS_RANGE):
    """mmdet3d ``lidar_nusc_box_to_global``: LiDAR -> ego (rotate, translate), drop boxes beyond their class's evaluation range in the ego
    frame, ego -> global."""
    out = []
    for box in boxes:
        _rotate(box, info['lidar2ego_rotation'])
        box['center'] = box['center'] + np.asarray(info['lidar2ego_translation'], np.float64)
        if np.linalg.norm(box['center'][:2], 2) > class_range[classes[box['label']]]:
            continue
        _rotate(box, info['ego2global_rotation'])
        box['center'] = box['center'] + np.asarray(info['ego2global_translation'], np.float64)
        out.append(box)
    return out


def attribute_of(name, velocity):
    """mmdet3d NuScenesDataset._format_bbox's attribute rule."""
    if np.sqrt(velocity[0] ** 2 + velocity[1] ** 2) > 0.2:
        if name in ('car', 'construction_vehicle', 'bus', 'truck', 'trailer'):
            return 'vehicle.moving'
        if name in ('bicycle', 'motorcycle'):
            return 'cycle.with_rider'
        return DEFAULT_ATTRIBUTE[name]
    if name == 'pedestrian':
        return 'pedestrian.standing'
    if name == 'bus':
        return 'vehicle.stopped'
    return DEFAULT_ATTRIBUTE[name]


def format_bbox(results, data_infos, jsonfile_prefix, classes=CLASSES, modality=MODALITY, class_range=CLASS_RANGE):
    """list of dict(boxes_3d, scores_3d, labels_3d) (one per sample, in data_infos order) -> ``<prefix>/results_nusc.json``; returns its path."""
    annos = {}
    for sample_id, det in enumerate(results):
        info = data_infos[sample_id]
        boxes = lidar_boxes_to_global(info, output_to_nusc_boxes(det), classes, class_range)
        token = info['token']
        annos[token] = [dict(sample_token=token, translation=b['center'].tolist(), size=b['wlh'].tolist(), rotation=b['orientation'].tolist(),
                             velocity=b['velocity'][:2].tolist(), detection_name=classes[b['label']], detection_score=b['score'],
                             attribute_name=attribute_of(classes[b['label']], b['velocity'])) for b in boxes]
    os.makedirs(jsonfile_prefix, exist_ok=True)
    path = os.path.join(jsonfile_prefix, 'results_nusc.json')
    with open(path, 'w') as fh:
        json.dump({'meta': modality, 'results': annos}, fh)
    return path


def format_results(results, data_infos, jsonfile_prefix=None, **kw):
    """CustomNuScenesDataset.format_results (:324-370): accepts both result formats; returns (result_files, tmp_dir)."""
    assert isinstance(results, list), 'results must be a list'
    assert len(results) == len(data_infos), 'The length of results is not equal to the dataset len: {} != {}'.format(len(results), len(data_infos))
    tmp_dir = None
    if jsonfile_prefix is None:
        tmp_dir = tempfile.TemporaryDirectory()
        jsonfile_prefix = os.path.join(tmp_dir.name, 'results')
    if not ('pts_bbox' in results[0] or 'img_bbox' in results[0]):
        return format_bbox(results, data_infos, jsonfile_prefix, **kw), tmp_dir
    files = {}
    for name in results[0]:
        if name in ('pts_bbox', 'img_bbox'):
            files[name] = format_bbox([out[name] for out in results], data_infos, os.path.join(jsonfile_prefix, name), **kw)
    return files, tmp_dir


def metrics_detail(metrics, classes=CLASSES, result_name='pts_bbox'):
    """The dict ``_evaluate_single`` builds from the devkit's metrics_summary.json (:403-426)."""
    detail, prefix = {}, f'{result_name}_NuScenes'
    for name in classes:
        for k, v in metrics['label_aps'][name].items():
            detail['{}/{}_AP_dist_{}'.format(prefix, name, k)] = float('{:.4f}'.format(v))
        for k, v in metrics['label_tp_errors'][name].items():
            detail['{}/{}_{}'.format(prefix, name, k)] = float('{:.4f}'.format(v))
        for k, v in metrics['tp_errors'].items():
            detail['{}/{}'.format(prefix, ERR_NAME_MAPPING[k])] = float('{:.4f}'.format(v))
    detail['{}/NDS'.format(prefix)] = metrics['nd_score']
    detail['{}/mAP'.format(prefix)] = metrics['mean_ap']
    return detail


def evaluate(results, data_infos, version, data_root, jsonfile_prefix=None, result_names=('pts_bbox',), eval_version='detection_cvpr_2019'):
    """CustomNuScenesDataset.evaluate (:428-456): submission json + the devkit's NuScenesEval.  The devkit is third party: without it the
    json files are still written and an ImportError names what is missing."""
    files, tmp_dir = format_results(results, data_infos, jsonfile_prefix)
    try:
        from nuscenes import NuScenes
        from nuscenes.eval.detection.config import config_factory
        from nuscenes.eval.detection.evaluate import NuScenesEval
    except ImportError as e:
        raise ImportError(f'nuscenes-devkit is not installed: the submission files are at {files}; run NuScenesEval on them') from e
    out = {}
    for name, path in (files.items() if isinstance(files, dict) else [('pts_bbox', files)]):
        if isinstance(files, dict) and name not in result_names:
            continue
        output_dir = os.path.dirname(path)
        nusc = NuScenes(version=version, dataroot=data_root, verbose=False)
        NuScenesEval(nusc, config=config_factory(eval_version), result_path=path, eval_set={'v1.0-mini': 'mini_val', 'v1.0-trainval': 'val'}[version],
                     output_dir=output_dir, verbose=False).main(render_curves=False)
        with open(os.path.join(output_dir, 'metrics_summary.json')) as fh:
            out.update(metrics_detail(json.load(fh), result_name=name))
    if tmp_dir is not None:
        tmp_dir.cleanup()
    return out


# ---- 2-D annotations: the COCO json (pycocotools' index, restated) ----------------------------------------------------------------------
class Coco2D:
    """What ``load_annotations_2d`` / ``impath_to_ann2d`` use of pycocotools.COCO (:73-98): image list, annotations per image, category ids
    by class name; ``cat2label`` in the order of ``classes``."""

    def __init__(self, ann_file, classes=CLASSES, data_prefix='./data/nuscenes/'):
        with open(ann_file) as fh:
            d = json.load(fh)
        self.imgs = {im['id']: im for im in d.get('images', [])}
        self.anns_of = {}
        for a in d.get('annotations', []):
            self.anns_of.setdefault(a['image_id'], []).append(a)
        by_name = {c['name']: c['id'] for c in d.get('categories', [])}
        self.cat_ids = [by_name[n] for n in classes if n in by_name]
        self.cat2label = {cid: i for i, cid in enumerate(self.cat_ids)}
        self.impath_to_imgid, self.imgid_to_dataid, self.data_infos_2d = {}, {}, []
        all_ids = []
        for i in self.imgs:                                              # get_img_ids(): dataset order
            info = dict(self.imgs[i])
            info['filename'] = info['file_name']
            self.impath_to_imgid[data_prefix + info['file_name']] = i
            self.imgid_to_dataid[i] = len(self.data_infos_2d)
            self.data_infos_2d.append(info)
            all_ids += [a['id'] for a in self.anns_of.get(i, [])]
        assert len(set(all_ids)) == len(all_ids), f"Annotation ids in '{ann_file}' are not unique!"

    def impath_to_ann2d(self, impath):
        """(img_info, ann_info) of one image path, as ``impath_to_ann2d`` hands them to ``get_ann_info_2d`` (= nuscenes_io.parse_ann_2d)."""
        img_id = self.impath_to_imgid[impath]
        return self.data_infos_2d[self.imgid_to_dataid[img_id]], list(self.anns_of.get(img_id, []))


def collect_mono3d(results, keys, meta_keys=COLLECT_MONO3D_META_KEYS):
    """CollectMono3D / Collect3D: the listed meta keys that are present travel as ``img_metas``, ``keys`` are passed through."""
    out = {'img_metas': {k: results[k] for k in meta_keys if k in results}}
    for k in keys:
        out[k] = results[k]
    return out
