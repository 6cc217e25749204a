"""Output side of the nuScenes I/O contract (mv2d_amd/nuscenes_out.py): submission json, the pts_bbox wrapper, the
metrics bookkeeping, the COCO-json reader of the 2-D annotations, the CollectMono3D key bundle.  The box conversion restates third-party
code (mmdet3d 1.0.0 / nuscenes-devkit / pyquaternion: parity unpinned) and is checked against direct rotation-matrix composition."""
import json
import math
import os

import numpy as np
import torch

from mv2d_amd import nuscenes_out as N


def _rz(a):
    return np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.0]])


def _q_from_R(R):
    w = math.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


def test_quaternions_are_rotations():
    g = np.random.Generator(np.random.PCG64(1))
    for _ in range(20):
        q = g.standard_normal(4); q /= np.linalg.norm(q)
        p = g.standard_normal(4); p /= np.linalg.norm(p)
        assert np.allclose(N.quat_rotation_matrix(N.quat_mul(q, p)), N.quat_rotation_matrix(q) @ N.quat_rotation_matrix(p), atol=1e-12)
    assert np.allclose(N.quat_rotation_matrix(N.quat_axis_z(0.7)), _rz(0.7), atol=1e-12)


def _info(g, token):
    a, b = g.random() * 0.2 - 0.1, g.random() * 6.28
    return dict(token=token, lidar2ego_rotation=_q_from_R(_rz(a)).tolist(), lidar2ego_translation=[0.9, 0.0, 1.8],
                ego2global_rotation=_q_from_R(_rz(b)).tolist(), ego2global_translation=(g.random(3) * 100).tolist())


def test_submission_json_and_wrapper(tmp_path):
    g = np.random.Generator(np.random.PCG64(2))
    infos = [_info(g, f'tok{i}') for i in range(3)]
    results = []
    for i in range(3):
        n = 6
        b = np.zeros((n, 9), np.float32)
        b[:, :2] = g.random((n, 2)) * 30 - 15
        b[0, :2] = [70.0, 0.0]                                      # beyond every class range -> dropped
        b[:, 2] = -1.0
        b[:, 3:6] = g.random((n, 3)) * 3 + 0.5
        b[:, 6] = g.random(n) * 6 - 3
        b[:, 7:9] = g.random((n, 2)) * 2 - 1
        b[1, 7:9] = 0.0                                             # standing
        results.append(dict(boxes_3d=torch.from_numpy(b), scores_3d=torch.from_numpy(g.random(n).astype(np.float32)),
                            labels_3d=torch.tensor([0, 7, 3, 5, 8, 2])))
    path, tmp = N.format_results(results, infos, str(tmp_path / 'plain'))
    assert tmp is None and path.endswith('results_nusc.json')
    sub = json.load(open(path))
    assert sub['meta'] == N.MODALITY and set(sub['results']) == {'tok0', 'tok1', 'tok2'}
    for i, info in enumerate(infos):
        annos = sub['results'][info['token']]
        assert len(annos) == 5                                      # the far box is gone
        b = results[i]['boxes_3d'].numpy().astype(np.float64)
        R1, R2 = N.quat_rotation_matrix(info['lidar2ego_rotation']), N.quat_rotation_matrix(info['ego2global_rotation'])
        for a, k in zip(annos, range(1, 6)):
            centre = b[k, :3] + [0, 0, 0.5 * b[k, 5]]
            want = R2 @ (R1 @ centre + info['lidar2ego_translation']) + info['ego2global_translation']
            assert np.allclose(a['translation'], want, atol=1e-9)
            assert np.allclose(a['size'], b[k, [4, 3, 5]])            # (w, l, h) = (dy, dx, dz)
            assert np.allclose(N.quat_rotation_matrix(a['rotation']), R2 @ R1 @ _rz(b[k, 6]), atol=1e-9)
            assert np.allclose(a['velocity'], (R2 @ R1 @ [b[k, 7], b[k, 8], 0.0])[:2], atol=1e-9)
            assert a['detection_name'] == N.CLASSES[int(results[i]['labels_3d'][k])] and a['sample_token'] == info['token']
        assert annos[0]['attribute_name'] == 'pedestrian.standing'  # label 7, zero velocity
    # the reference's wrapper format (detectors/mv2d.py:283-292)
    files, _ = N.format_results([dict(pts_bbox=r) for r in results], infos, str(tmp_path / 'wrapped'))
    assert set(files) == {'pts_bbox'} and json.load(open(files['pts_bbox'])) == sub
    assert os.path.basename(os.path.dirname(files['pts_bbox'])) == 'pts_bbox'
    # attribute rule
    assert N.attribute_of('car', [1, 0, 0]) == 'vehicle.moving' and N.attribute_of('car', [0, 0, 0]) == 'vehicle.parked'
    assert N.attribute_of('bicycle', [0.3, 0, 0]) == 'cycle.with_rider' and N.attribute_of('bus', [0, 0, 0]) == 'vehicle.stopped'
    assert N.attribute_of('barrier', [5, 0, 0]) == ''


def test_metrics_detail():
    m = dict(label_aps={c: {'0.5': 0.123456, '1.0': 0.5} for c in N.CLASSES}, label_tp_errors={c: {'trans_err': 0.33333} for c in N.CLASSES},
             tp_errors={'trans_err': 0.61, 'scale_err': 0.27, 'orient_err': 0.5, 'vel_err': 0.9, 'attr_err': 0.2}, nd_score=0.45, mean_ap=0.4)
    d = N.metrics_detail(m)
    assert d['pts_bbox_NuScenes/car_AP_dist_0.5'] == 0.1235 and d['pts_bbox_NuScenes/mATE'] == 0.61 and d['pts_bbox_NuScenes/NDS'] == 0.45
    assert d['pts_bbox_NuScenes/barrier_trans_err'] == 0.3333 and d['pts_bbox_NuScenes/mAP'] == 0.4


def test_coco_reader_feeds_the_2d_annotation_parser(tmp_path):
    from mv2d_amd import nuscenes_io
    coco = dict(images=[dict(id=7, file_name='samples/CAM_FRONT/a.jpg', width=1600, height=900), dict(id=3, file_name='samples/CAM_BACK/b.jpg', width=1600, height=900)],
                annotations=[dict(id=1, image_id=7, category_id=11, bbox=[100, 200, 50, 80], area=4000, iscrowd=0, bbox_cam3d=[1.0, 2.0, 14.0, 4.0, 1.8, 1.6, 0.3], center2d=[120.0, 230.0, 14.0]),
                             dict(id=2, image_id=7, category_id=12, bbox=[10, 20, 30, 40], area=1200, iscrowd=0, bbox_cam3d=[-3.0, 1.0, 9.0, 0.6, 0.6, 1.7, -1.0], center2d=[20.0, 30.0, 9.0]),
                             dict(id=5, image_id=3, category_id=11, bbox=[5, 5, 10, 10], area=100, iscrowd=1, bbox_cam3d=[0.0, 0.0, 30.0, 4.0, 1.8, 1.6, 0.0], center2d=[8.0, 8.0, 30.0])],
                categories=[dict(id=12, name='pedestrian'), dict(id=11, name='car'), dict(id=99, name='animal')])
    f = tmp_path / 'ann.json'
    f.write_text(json.dumps(coco))
    c = N.Coco2D(str(f))
    assert c.cat_ids == [11, 12] and c.cat2label == {11: 0, 12: 1}
    info, anns = c.impath_to_ann2d('./data/nuscenes/samples/CAM_FRONT/a.jpg')
    assert info['filename'] == 'samples/CAM_FRONT/a.jpg' and [a['id'] for a in anns] == [1, 2]
    assert c.impath_to_ann2d('./data/nuscenes/samples/CAM_BACK/b.jpg')[1][0]['iscrowd'] == 1
    ann = nuscenes_io.parse_ann_2d(info, anns, c.cat_ids, c.cat2label)
    assert ann['bboxes_2d'].shape == (2, 4) and ann['labels'].tolist() == [0, 1] and ann['bboxes_cam'].shape == (2, 7)


def test_collect_mono3d_bundle():
    res = dict(img=1, gt=2, intrinsics=np.eye(4), extrinsics=np.eye(4), timestamp=0.5, lidar2img=np.eye(4), pad_shape=(1, 2, 3), junk=5)
    out = N.collect_mono3d(res, ['img'])
    assert set(out) == {'img_metas', 'img'} and set(out['img_metas']) == {'intrinsics', 'extrinsics', 'timestamp', 'lidar2img', 'pad_shape'}
    assert len(N.COLLECT_MONO3D_META_KEYS) == 26
