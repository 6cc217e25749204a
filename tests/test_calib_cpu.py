"""Host-side calibration tables (mv2d_amd/calib.py): the batched per-frame path of the engine gives bitwise the matrices of the per-sample
statement (which follows the reference's own numpy / torch calls: MU/pe.py:111-114, RH/utils/box_correlation.py:117-122), and
frame_tables is the composition of its geometry and shape parts."""
import numpy as np
import torch

from mv2d_amd import calib, synthetic


def test_batched_geometry_equals_per_sample():
    ms = [synthetic.make_problem('cfg3_t', seed=0, with_feat=False, ego=0.013 * (j + 1))['img_metas'] for j in range(5)]
    V = len(ms[0])
    mats, img2lidar, trans, ts = calib.geometry_tables_batch(ms)
    assert mats.shape == (3, 5 * V, 4, 4) and trans.shape == (5, V, V, 16)
    for b, m in enumerate(ms):
        g = calib.geometry_tables(m)
        assert torch.equal(trans[b], g['trans'])
        assert torch.equal(img2lidar[b * V:(b + 1) * V], g['img2lidar'])
        assert np.array_equal(mats[0, b * V:(b + 1) * V].reshape(V, 16), g['viewK'].numpy())
        assert np.array_equal(mats[1, b * V:(b + 1) * V].reshape(V, 16), g['viewE'].numpy())
        assert np.array_equal(ts[b], g['timestamps'])
    # the reference's own calls, one view / one sample at a time
    l2i = torch.from_numpy(np.stack([np.asarray(x['lidar2img'], dtype=np.float64) for x in ms[2]]))
    assert torch.equal(trans[2].view(V, V, 4, 4), torch.matmul(l2i[None], torch.inverse(l2i)[:, None]))
    assert np.array_equal(img2lidar[2 * V].view(4, 4).numpy(), np.linalg.inv(np.asarray(ms[2][0]['lidar2img'], dtype=np.float64)))


def test_frame_tables_is_geometry_plus_shape_tables():
    prob = synthetic.make_problem('cfg1_s', seed=0, with_feat=False)
    m = prob['img_metas']
    ft = calib.frame_tables(m, 14, 26)
    g = calib.geometry_tables(m)
    s = calib.shape_tables(calib.meta_shapes(m), 14, 26)
    for k in g:
        assert (torch.equal(ft[k], g[k]) if torch.is_tensor(g[k]) else np.array_equal(ft[k], g[k])), k
    for k in s:
        assert (torch.equal(ft[k], s[k]) if torch.is_tensor(s[k]) else ft[k] == s[k]), k
    assert torch.get_num_threads() >= 1


def test_ego_motion_changes_only_the_previous_frame():
    a = synthetic.make_problem('cfg3_t', seed=0, with_feat=False)['img_metas']
    b = synthetic.make_problem('cfg3_t', seed=0, with_feat=False, ego=0.05)['img_metas']
    nv = len(a) // 2
    for i in range(nv):
        assert np.array_equal(a[i]['lidar2img'], b[i]['lidar2img'])
    assert not np.array_equal(a[nv]['lidar2img'], b[nv]['lidar2img'])
