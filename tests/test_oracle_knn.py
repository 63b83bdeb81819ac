import numpy as np
from scipy.spatial import cKDTree

from util import oracle


def test_knn_oracle_matches_kdtree():
    rng = np.random.default_rng(0)
    for pts in (rng.uniform(-1, 1, (700, 3)), rng.normal(0, 1, (500, 3)) * [10, 1, 0.01]):
        pts = pts.astype(np.float32)
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
        want = (d[:, 1:] ** 2).mean(1)
        got = oracle.knn_dist2(pts)
        np.testing.assert_allclose(got, want, rtol=2e-5)


def test_knn_oracle_edge_cases():
    assert oracle.knn_dist2(np.zeros((0, 3), np.float32)).shape == (0,)
    # < 4 points: the empty neighbour slots keep FLT_MAX (simple_knn.cu:153): 3 points -> ~FLT_MAX/3, 2 points -> +inf
    assert (oracle.knn_dist2(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)) > 1e37).all()
    assert np.isinf(oracle.knn_dist2(np.array([[0, 0, 0], [1, 0, 0]], np.float32))).all()
    d = oracle.knn_dist2(np.array([[0, 0, 0]] * 4 + [[1, 0, 0]], np.float32))
    assert (d[:4] == 0).all() and abs(d[4] - 1.0) < 1e-6                                               # duplicates count, self excluded by index
