"""
CPU tests: the oracle (oracle/nimrud_oracle.py) against (a) the reference's own VoxelFilter
known-answer tests, restated from nimrud/utils/tests/geometry_tests.py:17-279, and (b) the golden
vectors captured from the imported reference (tests/golden/make_golden.py).
"""

import numpy as np
import pytest

from oracle import nimrud_oracle as oracle
from conftest import assert_features_close

PIPELINE_FIXTURES = ["g1_uniform.npz", "g2_scene.npz", "g3_offset.npz", "g4_lattice.npz"]


# ---- reference known-answer tests for the lattice (geometry_tests.py) ------------------------------

def test_lattice_init_validation():
    # geometry_tests.py:17-80
    rs = np.random.RandomState(10)
    for dim in (2, 3):
        with pytest.raises(ValueError):
            oracle.Lattice(rs.rand(1, dim) * 100, 0.5)
        pts = rs.rand(1000, dim) * 100
        lat = oracle.Lattice(pts, 0.5)
        assert np.array_equal(lat.minimum_corner, pts.min(0) - 0.25)
        assert np.array_equal(lat.maximum_corner, pts.max(0) + 0.25)
    for dim in (1, 4):
        with pytest.raises(ValueError):
            oracle.Lattice(rs.rand(1000, dim), 0.5)
    with pytest.raises(ValueError):
        oracle.Lattice(rs.rand(10), 0.5)
    with pytest.raises(ValueError):
        oracle.Lattice(rs.rand(10, 10, 10), 0.5)


def test_lattice_shift_and_overflow():
    # geometry_tests.py:84-138: 17 bits per axis at e = 0.001 over [0,100]; overflow -> ValueError
    for dim in (2, 3):
        pts = np.asarray([[0, 0, 0], [100, 100, 100]])[:, :dim]
        lat = oracle.Lattice(pts, 0.001)
        assert np.array_equal(lat.shifts, [17, 34][:dim - 1])
        assert np.array_equal(lat.widths, [17, 17, 17][:dim])
        with pytest.raises(ValueError):
            oracle.Lattice(pts, 0.00001 if dim == 3 else 0.00000001)


def test_lattice_masks():
    # geometry_tests.py:142-158
    for dim in (2, 3):
        pts = np.asarray([[0, 0, 0], [100, 100, 100]])[:, :dim]
        lat = oracle.Lattice(pts, 1)
        assert np.array_equal(lat.masks, [0b1111111, 0b11111110000000, 0b111111100000000000000][:dim])


def test_lattice_bounds():
    # geometry_tests.py:162-192
    for dim in (2, 3):
        lat = oracle.Lattice(np.asarray([[0, 0, 0], [100, 100, 100]])[:, :dim], 1)

        def ok(p):
            try:
                lat.check_in_bounds(p)
            except ValueError:
                return False
            return True
        assert ok(np.zeros((1, dim)) - 0.5)
        assert not ok(np.zeros((1, dim)) - 1.5)
        assert ok(np.zeros((1, dim)) + 100.5)
        assert not ok(np.zeros((1, dim)) + 101.5)
        assert not ok(np.zeros((1, dim + 1)))
        assert ok(np.zeros(dim))
        assert not ok(np.zeros(dim + 1))


def test_lattice_known_address():
    # geometry_tests.py:196-256: point (10,11,12) at e=1 -> address 198026 and back
    lat = oracle.Lattice(np.asarray([[0, 0, 0], [100, 100, 100]]), 1)
    assert lat.coordinate_to_address(np.arange(3) + 10)[0] == 198026
    assert np.allclose(lat.address_to_coordinate(198026).ravel(), np.arange(3) + 10)
    lat2 = oracle.Lattice(np.asarray([[0, 0], [100, 100]]), 1)
    assert np.allclose(lat2.address_to_coordinate(lat2.coordinate_to_address([10, 11])).ravel(),
                       [10, 11])


def test_lattice_unique():
    # geometry_tests.py:261-279
    for dim in (2, 3):
        lat = oracle.Lattice(np.asarray([[0, 0, 0], [100, 100, 100]])[:, :dim], 1)
        pts = np.concatenate([np.zeros((1, dim)) + off for off in np.arange(0, 20, 2)])
        assert np.array_equal(lat.unique_voxels(np.vstack((pts, pts))), pts)


# ---- golden vectors from the imported reference ---------------------------------------------------

@pytest.mark.parametrize("name", PIPELINE_FIXTURES)
def test_oracle_addresses_match_reference(golden, name):
    g = golden(name)
    for s, e in enumerate(g["edges"]):
        lat = oracle.Lattice(g["points"], e)
        assert np.array_equal(lat.unique_addresses(g["points"]), g["s%d_addresses" % s])
        assert np.array_equal(lat.minimum_corner, g["s%d_min_corner" % s])
        assert np.array_equal(lat.shifts, g["s%d_shifts" % s])
        assert np.array_equal(lat.widths, g["s%d_widths" % s])


@pytest.mark.parametrize("name", PIPELINE_FIXTURES)
def test_oracle_neighbors_match_reference(golden, name):
    g = golden(name)
    pts = g["points"]
    for s, (e, r) in enumerate(zip(g["edges"], g["radii"])):
        lat = oracle.Lattice(pts, e)
        voxels = lat.unique_voxels(pts)
        off, idx = g["s%d_nbr_offsets" % s], g["s%d_nbr_index" % s]
        n = len(off) - 1
        for lists in (oracle.ball_neighbors_kdtree(pts[:n], voxels, r),
                      oracle.ball_neighbors_bruteforce(pts[:n], voxels, r)):
            o2, i2 = oracle.neighbors_to_csr(lists)
            assert np.array_equal(o2, off)
            assert np.array_equal(i2, idx)
        # the C restatement's enumeration (what the full-size index test on the GPU box checks against): the
        # same lists, for the queries the reference's capture holds; and for ALL queries against the kd-tree
        o3, i3 = oracle.neighbor_lists_c(pts[:n], pts, e, r)
        assert np.array_equal(o3, off) and np.array_equal(i3, idx)
        o4, i4 = oracle.neighbor_lists_c(pts, pts, e, r)
        o5, i5 = oracle.neighbors_to_csr(oracle.ball_neighbors_kdtree(pts, voxels, r))
        assert np.array_equal(o4, o5) and np.array_equal(i4, i5)


@pytest.mark.parametrize("name", PIPELINE_FIXTURES)
def test_oracle_features_match_reference(golden, name):
    g = golden(name)
    pts = g["points"]
    got = oracle.process(pts, pts, g["edges"], g["radii"])
    want = g["features"]
    assert np.array_equal(got[:, ::4], want[:, ::4])
    assert np.abs(got - want).max() <= 1e-12
    fast = oracle.process_fast(pts, pts, g["edges"], g["radii"])
    assert_features_close(fast, want, pts)
    # the vectorised variant agrees with the faithful one far inside the contract
    assert np.abs(fast - want)[:, [1, 5 % want.shape[1]]].max() <= 1e-8


def test_oracle_operators_match_reference(golden):
    g = golden("g4_operators.npz")
    q = g["query"]
    for key in ("two", "three_collinear", "three", "four_coplanar", "plane_lattice", "line_lattice",
                "blob"):
        nb = g[key + "_points"]
        assert oracle.population(nb) == g[key + "_population"]
        assert abs(oracle.centroid(q, nb) - g[key + "_centroid"]) <= 1e-15
        assert np.abs(oracle.pca(nb) - g[key + "_pca"]).max() <= 1e-15
    # k = 0, 1: the reference raises FloatingPointError out of np.cov; the documented value is zeros
    for key, nb in (("empty", np.zeros((0, 3))), ("one", np.array([[1.0, 2.0, 3.0]]))):
        assert str(g[key + "_pca_raises"]) == "FloatingPointError"
        assert oracle.population(nb) == g[key + "_population"]
        assert abs(oracle.centroid(q, nb) - g[key + "_centroid"]) <= 1e-15
        assert np.array_equal(oracle.pca(nb), np.zeros(2))
        with pytest.raises(FloatingPointError):
            oracle.pca(nb, strict=True)


def test_oracle_forest_config5_matches_sklearn(golden):
    # the config 5 classifier (32 trees, depth <= 12, 5 classes; tests/golden/make_forest_c5.py)
    g = golden("g6_forest_c5.npz")
    assert len(g["roots"]) == 32 and g["value"].shape[1] == 5 and g["x"].shape == (4096, 20)
    model = {k: g[k] for k in ("left", "right", "feature", "threshold", "value", "roots", "classes")}
    proba = oracle.forest_predict_proba(model, g["x"])
    assert np.abs(proba - g["proba"]).max() <= 1e-15
    assert np.array_equal(oracle.forest_predict(model, g["x"]), g["label"])
    # depth of the flattened trees
    left, right = model["left"], model["right"]
    depth = np.zeros(len(left), dtype=np.int64)
    for node in range(len(left)):               # children come after their parent in sklearn's arrays
        for child in (left[node], right[node]):
            if child >= 0:
                depth[child] = depth[node] + 1
    assert depth.max() <= 12


def test_oracle_forest_matches_reference(golden):
    g = golden("g5_forest.npz")
    model = {k: g[k] for k in ("left", "right", "feature", "threshold", "value", "roots", "classes")}
    proba = oracle.forest_predict_proba(model, g["x"])
    assert np.abs(proba - g["proba"]).max() <= 1e-15
    assert np.array_equal(oracle.forest_predict(model, g["x"]), g["label"])


def test_oracle_knn_fallback_definition():
    # build-defined extension (config 4): rows with population >= k are untouched; the others use the
    # k nearest voxel centres within the radius factor (checked against a brute-force selection)
    from nimrud_amd import synth
    rs = np.random.RandomState(3)
    pts = np.concatenate((synth.uniform_cloud(1500, extent=1.5, seed=4), rs.rand(200, 3) * 6.0 - 2.0))
    e, r, k, factor = 0.1, 0.3, 6, 4.0
    plain = oracle.one_scale_fast(pts, pts, e, r)
    knn = oracle.one_scale_knn(pts, pts, e, r, k, radius_factor=factor)
    dense = plain[:, 0] >= k
    assert dense.sum() > 100 and (~dense).sum() > 50
    assert np.array_equal(knn[dense], plain[dense])
    assert np.array_equal(knn[:, 0], plain[:, 0])
    voxels = oracle.Lattice(pts, e).unique_voxels(pts)
    for row in np.nonzero(~dense)[0][:40]:
        d = np.linalg.norm(voxels - pts[row], axis=1)
        pick = np.argsort(d, kind="stable")[:k]
        nb = voxels[pick[d[pick] <= r * factor]]
        assert abs(knn[row, 1] - oracle.centroid(pts[row], nb)) < 1e-12
        assert np.abs(knn[row, 2:] - oracle.pca(nb)).max() < 1e-12


# ---- the plain-C restatement (oracle/lattice_oracle.c) ----------------------------------------------

@pytest.mark.parametrize("name", PIPELINE_FIXTURES)
def test_c_oracle_matches_reference(golden, name):
    g = golden(name)
    pts = g["points"]
    got = oracle.process_c(pts, pts, list(g["edges"]), list(g["radii"]))
    want = g["features"]
    assert np.array_equal(got[:, ::4], want[:, ::4])
    assert np.abs(got - want).max() <= 1e-12
    for s, e in enumerate(g["edges"]):
        _, m = oracle.one_scale_c(pts[:10], pts, e, g["radii"][s], return_voxel_count=True)
        assert m == len(g["s%d_addresses" % s])


def test_c_oracle_against_the_numpy_oracle_on_edge_cases():
    from nimrud_amd import synth
    search = synth.uniform_cloud(5000, extent=2.0, seed=201)
    rs = np.random.RandomState(202)
    query = rs.rand(1500, 3) * 4.0 - 1.0                      # partly outside the lattice
    query[:3] = [[-100.0, 0, 0], [0, 1e12, 0], [1.0, 1.0, -50.0]]
    for e, r in ((0.2, 0.6), (0.25, 0.3), (0.1, 0.52)):
        a = oracle.one_scale_c(query, search, e, r, threads=2)
        b = oracle.one_scale_fast(query, search, e, r)
        assert np.array_equal(a[:, 0], b[:, 0])
        assert np.abs(a - b).max() <= 1e-10
    wide = np.concatenate((search, rs.rand(5000, 2)), axis=1)  # strided rows
    assert np.array_equal(oracle.one_scale_c(wide, wide, 0.2, 0.6), oracle.one_scale_c(search, search, 0.2, 0.6))
