"""nuScenes I/O contract: mv2d_amd.nuscenes_io vs goldens recorded from the reference's own dataset / pipeline classes
(oracle/gen_golden_io.py -> tests/golden/nusc_io.npz).  Host code only, runs on CPU."""
import copy

import numpy as np
import pytest

from conftest import load_golden
from mv2d_amd import nuscenes_io as nio
from mv2d_amd import synthetic

GOLD = load_golden('nusc_io')


def _eq(a, b, name):
    a = np.stack([np.asarray(x, np.float64) for x in a]) if len(a) else np.zeros((0,))
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))


@pytest.mark.parametrize('name', list(synthetic.NUSC_CASES))
def test_pipeline_geometry_matches_reference(name):
    kw = synthetic.NUSC_CASES[name]
    info = synthetic.make_nusc_info(kw['seed'], n_sweeps=kw.get('n_sweeps', 6), incomplete_sweep=kw.get('incomplete_sweep'))
    d = nio.camera_geometry(copy.deepcopy(info))
    for k in ('lidar2img', 'intrinsics', 'extrinsics'):
        _eq(d[k], GOLD[f'{name}.info.{k}'], k)
    assert np.array_equal(np.array(d['img_timestamp']), GOLD[f'{name}.info.img_timestamp'])
    assert d['timestamp'] == float(GOLD[f'{name}.info.timestamp'])
    d['img'] = [synthetic.fake_image(p).astype(np.float32) for p in d['img_filename']]
    d['filename'] = list(d['img_filename'])
    sw = dict(kw['sweeps'])
    np.random.seed(kw['seed'])
    d = nio.append_sweeps(d, imread=synthetic.fake_image, **sw)
    assert np.array_equal(np.array(d['timestamp']), GOLD[f'{name}.sweeps.timestamp'])
    assert list(d['filename']) == list(GOLD[f'{name}.sweeps.filename'])
    for k in ('lidar2img', 'intrinsics', 'extrinsics'):
        _eq(d[k], GOLD[f'{name}.sweeps.{k}'], 'sweeps.' + k)
    assert np.array_equal(np.array([float(np.asarray(im, np.float64).sum()) for im in d['img']]), GOLD[f'{name}.sweeps.img_sum'])
    np.random.seed(kw['seed'] + 1)
    d = nio.resize_crop_flip(d, kw['conf'], training=kw['training'])
    _eq(d['intrinsics'], GOLD[f'{name}.aug.intrinsics'], 'aug.intrinsics')
    _eq(d['lidar2img'], GOLD[f'{name}.aug.lidar2img'], 'aug.lidar2img')
    assert tuple(d['img'][0].shape) == tuple(GOLD[f'{name}.aug.img_shape'])
    assert np.array_equal(np.array([float(im.astype(np.float64).sum()) for im in d['img']]), GOLD[f'{name}.aug.img_sum'])


@pytest.mark.parametrize('training', [False, True])
def test_full_size_augmentation_matrix(training):
    np.random.seed(77)
    args = nio.sample_augmentation(synthetic.NUSC_AUG_CONF, training)
    want = GOLD[f'fullsize.{int(training)}.args']
    assert np.array_equal(np.array([args[0], *args[1], *args[2], float(args[3]), args[4]], np.float64), want)
    assert np.array_equal(nio.image_aug_matrix(args[0], args[2], args[3], args[4]), GOLD[f'fullsize.{int(training)}.ida'])
    if not training:                      # the shipped test pipeline: 1600x900 -> resize 0.88 -> crop rows 280.. -> 1408x512
        assert args[1] == (1408, 792) and args[2] == (0, 280, 1408, 792)


def test_split_view_metas_and_geometry_only_path():
    info = synthetic.make_nusc_info(8, n_sweeps=20)
    d = nio.camera_geometry(info)
    d['filename'] = list(d['img_filename'])
    d = nio.append_sweeps(d, sweeps_num=1, pad_empty_sweeps=True)                  # no images: paths and geometry only
    assert d['img'] == [] and len(d['lidar2img']) == 12 and len(d['timestamp']) == 12
    K0 = [k.copy() for k in d['intrinsics']]
    d = nio.resize_crop_flip(d, synthetic.NUSC_AUG_CONF, training=False, transform_images=False)
    ida = nio.image_aug_matrix(0.88, (0, 280, 1408, 792), False, 0)
    for k0, k1, e, l in zip(K0, d['intrinsics'], d['extrinsics'], d['lidar2img']):
        assert np.array_equal(k1[:3, :3], ida @ k0[:3, :3]) and np.array_equal(l, k1 @ e.T)
    metas = nio.split_view_metas(dict(lidar2img=d['lidar2img'], intrinsics=d['intrinsics'], extrinsics=d['extrinsics'],
                                      timestamp=d['timestamp'], ori_shape=(900, 1600, 3, 12), pad_shape=(512, 1408, 3), box_type_3d='x'), 12)
    assert len(metas) == 12 and metas[3]['num_views'] == 12 and metas[3]['ori_shape'] == (900, 1600, 3)
    assert metas[7]['timestamp'] == d['timestamp'][7] and metas[7]['pad_shape'] == (512, 1408, 3) and metas[0]['box_type_3d'] == 'x'
    assert np.array_equal(metas[11]['lidar2img'], d['lidar2img'][11])


def test_sweep_camera_record_against_homogeneous_composition():
    """add_frame (tools/generate_sweep_pkl.py:32-82) is a script (not importable): checked against composing 4x4 transforms directly,
    sensor -> ego(sweep) -> global -> ego(key) -> lidar(key), and against the identity case."""
    g = np.random.default_rng(5)

    def quat():
        q = g.normal(size=4)
        return q / np.linalg.norm(q)

    def hom(R, t):
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = R, t
        return m
    qs, qe, qkl, qke = quat(), quat(), quat(), quat()
    ts, te, tkl, tke = g.normal(size=3), g.normal(size=3) * 10, g.normal(size=3), g.normal(size=3) * 10
    K = np.array([[1250., 0, 800], [0, 1250., 450], [0, 0, 1]])
    Rkl, Rke = nio.quaternion_rotation_matrix(qkl), nio.quaternion_rotation_matrix(qke)
    rec = nio.sweep_camera_record(qs, ts, qe, te, K, Rkl, tkl, Rke, tke)
    s2l = np.linalg.inv(hom(Rkl, tkl)) @ np.linalg.inv(hom(Rke, tke)) @ hom(nio.quaternion_rotation_matrix(qe), te) @ \
        hom(nio.quaternion_rotation_matrix(qs), ts)
    assert np.allclose(rec['sensor2lidar_rotation'], s2l[:3, :3], atol=1e-12) and np.allclose(rec['sensor2lidar_translation'], s2l[:3, 3], atol=1e-9)
    l2i = nio._viewpad(K) @ np.linalg.inv(s2l)
    assert np.allclose(rec['lidar2img'], l2i.astype(np.float32), rtol=1e-5, atol=1e-3) and rec['lidar2img'].dtype == np.float32
    assert np.allclose(rec['extrinsics'].T, np.linalg.inv(s2l), atol=1e-5)
    R = nio.quaternion_rotation_matrix(qs)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.allclose(nio.quaternion_rotation_matrix([1, 0, 0, 0]), np.eye(3))
    assert np.allclose(nio.quaternion_rotation_matrix([np.cos(0.3), 0, 0, np.sin(0.3)])[:2, :2],
                       [[np.cos(0.6), -np.sin(0.6)], [np.sin(0.6), np.cos(0.6)]])


@pytest.mark.parametrize('reverse', [False, True])
def test_global_rot_scale_trans_matches_reference(reverse):
    info = synthetic.make_nusc_info(11, n_sweeps=0)
    d = nio.camera_geometry(info)
    box = synthetic.RecordingBoxes()
    d['gt_bboxes_3d'] = box
    np.random.seed(5)
    angle, ratio = nio.global_rot_scale_trans(d, reverse_angle=reverse)
    if reverse:          # no matrix inverse involved: bit-exact
        _eq(d['lidar2img'], GOLD['grst.1.lidar2img'], 'lidar2img')
        _eq(d['extrinsics'], GOLD['grst.1.extrinsics'], 'extrinsics')
    else:                # torch.inverse (LU, fp32) vs numpy's: one fp32 ulp
        assert np.allclose(np.stack(d['lidar2img']), GOLD['grst.0.lidar2img'], rtol=2e-6, atol=1e-6)
        assert np.allclose(np.stack(d['extrinsics']), GOLD['grst.0.extrinsics'], rtol=2e-6, atol=1e-6)
    assert [c[0] for c in box.calls] == ['rotate', 'scale']
    assert np.array_equal(np.array([v for _, v in box.calls]), GOLD[f'grst.{int(reverse)}.calls']) and box.calls[1][1] == ratio


def test_center_match_matches_reference():
    g = np.random.default_rng(3)
    b = g.normal(size=(9, 7))
    a = np.concatenate([b[[4, 1, 7]] + 1e-5, g.normal(size=(2, 7)), b[[2]] + 2e-3])
    m = nio.center_match(a, b)
    assert np.array_equal(m, GOLD['center_match.match']) and list(m[:3]) == [4, 1, 7] and m[-1] == -1
    assert np.array_equal(nio.center_match(a[:0], b), GOLD['center_match.empty_a'])
    assert np.array_equal(nio.center_match(a, b[:0]), GOLD['center_match.empty_b'])


def test_pad_and_normalize():
    imgs = synthetic.make_view_images(2, 50, 70, 1)
    r = nio.pad_multi_view(dict(img=[i.copy() for i in imgs]), size_divisor=32)
    assert r['img_shape'] == [(50, 70, 3)] * 2 and r['pad_shape'] == [(64, 96, 3)] * 2
    assert np.array_equal(r['img'][1][:50, :70], imgs[1]) and not r['img'][1][50:].any() and not r['img'][1][:, 70:].any()
    r = nio.pad_multi_view(dict(img=[imgs[0]]), size=(64, 80))
    assert r['pad_shape'] == [(64, 80, 3)] and r['pad_fixed_size'] == (64, 80)
    mean, std = [103.530, 116.280, 123.675], [57.375, 57.120, 58.395]
    n = nio.normalize_multiview(dict(img=[imgs[0]]), mean, std, to_rgb=False)
    want = (imgs[0].astype(np.float64) - np.array(mean)) / np.array(std)
    assert n['img'][0].dtype == np.float32 and np.allclose(n['img'][0], want, atol=1e-5)
    n2 = nio.normalize_multiview(dict(img=[imgs[0]]), mean, std, to_rgb=True)
    assert np.allclose(n2['img'][0], (imgs[0][..., ::-1].astype(np.float64) - np.array(mean)) / np.array(std), atol=1e-5)


@pytest.mark.parametrize('seed', [21, 22, 23])
def test_resize_crop_flip_with_2d_boxes_matches_reference(seed):
    info = synthetic.make_nusc_info(seed, n_sweeps=0)
    d = nio.camera_geometry(info)
    d['img'] = [synthetic.fake_image(p).astype(np.float32) for p in d['img_filename']]
    d.update(synthetic.make_boxes_2d(6, seed))
    np.random.seed(seed)
    d = nio.resize_crop_flip(d, synthetic.NUSC_AUG_CONF_SMALL, training=True, with_bbox_2d=True, num_views=6)
    for k in ('gt_bboxes_2d', 'gt_labels_2d', 'gt_bboxes_2d_to_3d', 'gt_bboxes_ignore'):
        for v in range(6):
            want = GOLD[f'box2d.{seed}.{k}.{v}']
            got = np.asarray(d[k][v])
            assert got.shape == want.shape, (k, v, got.shape, want.shape)
            assert np.allclose(got, want, rtol=1e-6, atol=1e-4), (k, v)


@pytest.mark.parametrize('seed', [31, 32])
def test_2d_annotations_parse_and_match_like_the_reference(seed):
    info = synthetic.make_nusc_info(seed, n_sweeps=0)
    case = synthetic.make_ann_2d_case(info, seed)
    d = nio.camera_geometry(copy.deepcopy(info))
    parsed = [nio.parse_ann_2d(*case['images'][p], case['cat_ids'], case['cat2label']) for p in d['img_filename']]
    for v, a in enumerate(parsed):
        for k in ('bboxes_cam', 'bboxes_2d', 'gt_bboxes_ignore', 'labels'):
            want = GOLD[f'ann2d.{seed}.parse.{v}.{k}']
            assert a[k].dtype == want.dtype and np.array_equal(a[k], want), (v, k)
    out = nio.attach_2d_annotations(d, parsed, case['centers_lidar'], case['gt_labels_3d'])
    n_matched = 0
    for v in range(6):
        for k in ('gt_bboxes_2d', 'gt_labels_2d', 'gt_bboxes_2d_to_3d', 'gt_bboxes_ignore'):
            assert np.array_equal(np.asarray(out[k][v]), GOLD[f'ann2d.{seed}.info.{v}.{k}']), (v, k)
        n_matched += int((out['gt_bboxes_2d_to_3d'][v] > -1).sum())
    assert n_matched > 0
