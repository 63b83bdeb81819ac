import numpy as np
import pytest
import torch

from util import oracle

pytestmark = pytest.mark.gpu


def _run(pts):
    from simple_knn._C import distCUDA2
    return distCUDA2(torch.tensor(np.ascontiguousarray(pts, np.float32), device="cuda")).cpu().numpy()


@pytest.mark.parametrize("name", ["uniform", "flat", "clustered", "dupes", "line", "keyframe"])
def test_knn_matches_oracle(name):
    rng = np.random.default_rng(0)
    pts = {
        "uniform": rng.uniform(-1, 1, (2400, 3)),
        "flat": rng.uniform(-1, 1, (9600, 3)) * [5, 1, 0.01],
        "clustered": np.concatenate([rng.normal(0, 0.01, (3000, 3)), rng.normal(5, 2, (3000, 3))]),
        "dupes": np.repeat(rng.uniform(-1, 1, (100, 3)), 5, 0),
        "line": np.stack([np.linspace(0, 1, 1000), np.zeros(1000), np.zeros(1000)], 1),
        # what the caller feeds it: a back-projected, randomly down-sampled RGB-D frame (gaussian_model.py:185-241)
        "keyframe": (lambda z: np.stack([(rng.uniform(0, 640, 9600) - 320) / 535 * z, (rng.uniform(0, 480, 9600) - 240) / 539 * z, z], 1))(rng.uniform(0.5, 4, 9600)),
    }[name].astype(np.float32)
    got, want = _run(pts), oracle.knn_dist2(pts)
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=0)   # exact 3-NN; only dx*dx+dy*dy+dz*dz rounding (fma) differs


def test_knn_edge_cases_and_large():
    assert _run(np.zeros((0, 3))).shape == (0,)
    for n in (1, 2, 3):
        pts = np.random.default_rng(n).uniform(-1, 1, (n, 3)).astype(np.float32)
        got, want = _run(pts), oracle.knn_dist2(pts)
        assert (np.isinf(got) == np.isinf(want)).all() and np.allclose(got[np.isfinite(want)], want[np.isfinite(want)], rtol=1e-6)
    from scipy.spatial import cKDTree
    pts = np.random.default_rng(1).uniform(-3, 3, (300000, 3)).astype(np.float32)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    np.testing.assert_allclose(_run(pts), (d[:, 1:] ** 2).mean(1), rtol=2e-5)
