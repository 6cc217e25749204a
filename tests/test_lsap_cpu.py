"""mv2d_lsap_layers (host C++, csrc/lsap.hip) against scipy.optimize.linear_sum_assignment, which HungarianAssigner3D of the reference calls
(mmdet3d_plugin/core/bbox/assigners/hungarian_assigner_3d.py:137): the same assignment entry by entry -- also where several assignments
have the same cost (integer costs, constant matrices), since the scan order and the tie rule are the published ones.  No GPU needed."""
import ctypes

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from mv2d_amd import _lib


def lsap(cost, threads=0):
    cost = np.ascontiguousarray(cost, np.float32)
    L, R, G = cost.shape
    match = np.full((L, R), -7, np.int32)
    rc = _lib.load().mv2d_lsap_layers(cost.ctypes.data_as(ctypes.c_void_p), L, R, G, match.ctypes.data_as(ctypes.c_void_p), threads)
    return rc, match


def scipy_match(cost):
    L, R, G = cost.shape
    m = np.full((L, R), -1, np.int32)
    for l in range(L):
        r, c = linear_sum_assignment(cost[l].astype(np.float64))
        m[l, r] = c
    return m


@pytest.mark.parametrize('R,G', [(300, 40), (40, 300), (64, 64), (1, 5), (5, 1), (12, 9), (900, 120)])
@pytest.mark.parametrize('kind', ['float', 'int', 'const'])
def test_same_assignment_as_scipy(R, G, kind):
    rng = np.random.default_rng(R * 1000 + G)
    L = 6
    if kind == 'float':
        cost = rng.normal(size=(L, R, G)).astype(np.float32) * 3
    elif kind == 'int':
        cost = rng.integers(0, 4, size=(L, R, G)).astype(np.float32)          # many equal-cost assignments
    else:
        cost = np.full((L, R, G), 2.5, np.float32)
    for threads in (0, 1, 3):
        rc, m = lsap(cost, threads)
        assert rc == 0
        assert np.array_equal(m, scipy_match(cost)), (kind, R, G, threads)


def test_infinite_entries_and_errors():
    rng = np.random.default_rng(5)
    cost = rng.random((2, 30, 8)).astype(np.float32)
    cost[0, :10, 3] = np.inf                                                   # feasible: other rows can take column 3
    rc, m = lsap(cost)
    assert rc == 0 and np.array_equal(m, scipy_match(cost))
    bad = cost.copy()
    bad[1, 4, 2] = np.nan
    assert lsap(bad)[0] != 0                                                   # (scipy: ValueError "matrix contains invalid numeric entries")
    inf_col = cost.copy()
    inf_col[0, :, 5] = np.inf                                                  # no row can take column 5 of the transposed problem: infeasible
    with pytest.raises(ValueError):
        linear_sum_assignment(inf_col[0].astype(np.float64))
    assert lsap(inf_col)[0] != 0


def test_empty_shapes():
    rc, m = lsap(np.zeros((3, 0, 4), np.float32))
    assert rc == 0 and m.shape == (3, 0)
    rc, m = lsap(np.zeros((3, 5, 0), np.float32))
    assert rc == 0 and (m == -1).all()
