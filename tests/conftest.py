import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    # the C restatement of the oracle is test infrastructure: build it if it is not there yet
    import subprocess
    if not os.path.exists(os.path.join(REPO, "oracle", "liblattice_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=False)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


# tolerances of the parity contract (BASELINE.json north_star):
#   population: exact.  eigen-features: |a-b| <= 1e-5*|b| + 1e-9.
#   centroid distance: 1e-9 relative plus the fp64 representation error of the coordinates
#   themselves: the reference forms every voxel centre in world coordinates (up to 2 ulp of the
#   coordinate each: cell*e, + min_corner, + e/2), sums k of them (numpy.mean) and subtracts; at
#   UTM-scale offsets its own value therefore carries tens of ulp(|coordinate|) of noise, while the GPU
#   path works in exact integer offsets from the home voxel.  a randomised sweep (tests/fuzz_parity.py)
#   saw differences up to 20 ulp; 64 ulp are allowed.
def assert_features_close(got, want, points, eig_rtol=1e-5, eig_atol=1e-9):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    coord_ulp = np.spacing(np.abs(points[:, :3]).max())
    for s in range(want.shape[1] // 4):
        g, w = got[:, 4 * s:4 * s + 4], want[:, 4 * s:4 * s + 4]
        assert np.array_equal(g[:, 0], w[:, 0]), "population differs at scale %d" % s
        tol = 1e-9 * np.abs(w[:, 1]) + 64 * coord_ulp + 1e-12
        bad = np.abs(g[:, 1] - w[:, 1]) > tol
        assert not bad.any(), "centroid differs at scale %d: max err %g" % (
            s, np.abs(g[:, 1] - w[:, 1]).max())
        for c in (2, 3):
            tol = eig_rtol * np.abs(w[:, c]) + eig_atol
            err = np.abs(g[:, c] - w[:, c])
            assert not (err > tol).any(), "eigen-feature %d differs at scale %d: max err %g" % (
                c, s, err.max())
