"""
GPU parity tests (run with `-m gpu` on an MI355X).  everything goes through the C ABI
(libnimrud_hip.so via nimrud_amd); the oracle and the golden vectors are only the checker.

bar: population and neighbor indices bit-exact; eigen-features |a-b| <= 1e-5*|b| + 1e-9; centroid
distance 1e-9 relative plus the fp64 representation error of the coordinates (conftest.py).
"""

import os

import numpy as np
import pytest
import torch

from conftest import assert_features_close
from nimrud_amd import synth
from nimrud_amd.minimal import classification, features, fields, multiscale
from nimrud_amd.utils import geometry
from oracle import nimrud_oracle as oracle

pytestmark = pytest.mark.gpu


def _device_runtime():
    from nimrud_amd import device
    return device.get_runtime()

PIPELINE_FIXTURES = ["g1_uniform.npz", "g2_scene.npz", "g3_offset.npz", "g4_lattice.npz"]


# ---- the reference's own VoxelFilter tests, against the GPU VoxelFilter ---------------------------

def test_voxel_init():
    # geometry_tests.py:17-80
    rs = np.random.RandomState(10)
    for dim in (2, 3):
        with pytest.raises(ValueError):
            geometry.VoxelFilter(rs.rand(1, dim) * 100, 0.5)
        pts = rs.rand(1000, dim) * 100
        vf = geometry.VoxelFilter(pts, 0.5)
        assert np.array_equal(vf.minimum_corner, pts.min(0) - 0.25)
        assert np.array_equal(vf.maximum_corner, pts.max(0) + 0.25)
        assert vf.edge_length == 0.5
    for dim in (1, 4):
        with pytest.raises(ValueError):
            geometry.VoxelFilter(rs.rand(1000, dim) * 100, 0.5)
    with pytest.raises(ValueError):
        geometry.VoxelFilter(rs.rand(10), 0.5)
    with pytest.raises(ValueError):
        geometry.VoxelFilter(rs.rand(10, 10, 10), 0.5)


def test_voxel_shift_masks_bounds():
    # geometry_tests.py:84-192
    for dim in (2, 3):
        pts = np.asarray([[0, 0, 0], [100, 100, 100]])[:, :dim]
        vf = geometry.VoxelFilter(pts, 0.001)
        assert np.array_equal(vf.shifts, [17, 34][:dim - 1])
        assert np.array_equal(vf.widths, [17, 17, 17][:dim])
        with pytest.raises(ValueError):
            geometry.VoxelFilter(pts, 0.00001 if dim == 3 else 0.00000001)
        vf = geometry.VoxelFilter(pts, 1)
        assert np.array_equal(vf.masks, [0b1111111, 0b11111110000000, 0b111111100000000000000][:dim])

        def ok(p):
            try:
                vf._check_in_bounds(p)
            except ValueError:
                return False
            return True
        assert ok(np.zeros((1, dim)) - 0.5)
        assert not ok(np.zeros((1, dim)) - 1.5)
        assert ok(np.zeros((1, dim)) + 0.5)
        assert ok(np.zeros((1, dim)) + 100.5)
        assert not ok(np.zeros((1, dim)) + 101.5)
        assert not ok(np.zeros((1, dim + 1)))
        assert ok(np.zeros(dim))
        assert not ok(np.zeros(dim + 1))


def test_voxel_address_transform_unique():
    # geometry_tests.py:196-279
    vf = geometry.VoxelFilter(np.asarray([[0, 0, 0], [100, 100, 100]]), 1)
    assert vf.coordinate_to_address(np.arange(3) + 10)[0] == 198026
    assert np.allclose(vf.address_to_coordinate(198026).flatten(), np.arange(3) + 10)
    vf2 = geometry.VoxelFilter(np.asarray([[0, 0], [100, 100]]), 1)
    assert np.allclose(vf2.address_to_coordinate(vf2.coordinate_to_address([10, 11]).flatten()),
                       [[10, 11]])
    for dim in (2, 3):
        vf = geometry.VoxelFilter(np.asarray([[0, 0, 0], [100, 100, 100]])[:, :dim], 1)
        pts = np.concatenate([np.zeros((1, dim)) + off for off in np.arange(0, 20, 2)])
        assert np.array_equal(vf.unique_voxels(np.vstack((pts, pts))), pts)


# ---- golden vectors captured from the reference ----------------------------------------------------

@pytest.mark.parametrize("name", PIPELINE_FIXTURES)
def test_golden_addresses_and_centres(golden, name):
    g = golden(name)
    pts = g["points"]
    for s, e in enumerate(g["edges"]):
        vf = geometry.VoxelFilter(pts, e)
        assert np.array_equal(vf.minimum_corner, g["s%d_min_corner" % s])
        assert np.array_equal(vf.unique_addresses(pts), g["s%d_addresses" % s])
        lat = oracle.Lattice(pts, e)
        assert np.array_equal(vf.coordinate_to_address(pts), lat.coordinate_to_address(pts))
        # centres must be bit-identical to the reference's (they feed the inclusion test)
        assert np.array_equal(vf.unique_voxels(pts), lat.unique_voxels(pts))


@pytest.mark.parametrize("name", PIPELINE_FIXTURES)
def test_golden_features(golden, name):
    g = golden(name)
    pts = g["points"]
    got = multiscale.process_single_core(pts, pts, list(g["edges"]), list(g["radii"]))
    assert_features_close(got, g["features"], pts)


@pytest.mark.parametrize("name", PIPELINE_FIXTURES)
def test_golden_neighbor_indices_bit_exact(golden, name):
    g = golden(name)
    pts = g["points"]
    for s, (e, r) in enumerate(zip(g["edges"], g["radii"])):
        want_off, want_idx = g["s%d_nbr_offsets" % s], g["s%d_nbr_index" % s]
        n = len(want_off) - 1
        off, idx = multiscale.neighbor_lists(pts[:n], pts, e, r)
        assert np.array_equal(off, want_off)
        assert np.array_equal(idx, want_idx)


def test_golden_operators(golden):
    g = golden("g4_operators.npz")
    q = g["query"]
    for key in ("two", "three_collinear", "three", "four_coplanar", "plane_lattice", "line_lattice",
                "blob"):
        nb = g[key + "_points"]
        assert features.population(nb) == g[key + "_population"]
        assert abs(features.centroid(q, nb) - g[key + "_centroid"]) <= 1e-12
        want = g[key + "_pca"]
        assert np.all(np.abs(features.pca(nb) - want) <= 1e-5 * np.abs(want) + 1e-9)
    for key, nb in (("empty", np.zeros((0, 3))), ("one", np.array([[1.0, 2.0, 3.0]]))):
        assert features.population(nb) == g[key + "_population"]
        assert abs(features.centroid(q, nb) - g[key + "_centroid"]) <= 1e-12
        assert np.array_equal(features.pca(nb), np.zeros(2))
        with pytest.raises(FloatingPointError):
            features.pca(nb, strict=True)
    idx = np.array([3, 1, 2])
    assert np.array_equal(features.take(idx, g["blob_points"]), g["blob_points"][idx])


def test_golden_forest(golden):
    g = golden("g5_forest.npz")
    model = classification.ForestModel.from_arrays(g)
    proba = model.predict_proba(g["x"])
    assert np.abs(proba - g["proba"]).max() <= 1e-15
    assert np.array_equal(model.predict(g["x"]), g["label"])


def test_forest_leaf_rows_packed_and_padded(golden):
    # nm_forest::leaf_stride: ForestModel hands over rows of 8 doubles, 64-byte aligned (one cache line per leaf);
    # a caller's packed (n_leaves, n_classes) table - leaf_stride 0 - must give the same numbers on every evaluator
    # (the stand-alone one streams trees through LDS above 4096 rows and walks from memory below), and a stride
    # below n_classes or unaligned 8-double rows are refused
    import ctypes, copy
    from nimrud_amd import _ffi, device as nm_device
    g = golden("g5_forest.npz")
    model = classification.ForestModel.from_arrays(g)
    rt = model.rt
    nc = model._c.n_classes
    assert model.leaf_stride == 8 and model._c.leaf_stride == 8
    packed = model.leaf_value[:, :nc].contiguous()
    rs = np.random.RandomState(5)
    for n in (len(g["x"]), 20000):
        x = g["x"] if n == len(g["x"]) else g["x"][rs.randint(0, len(g["x"]), n)]
        dx = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        want_p, want_l, _ = model._eval(dx, True, True)
        alt = _ffi.NmForest()
        ctypes.memmove(ctypes.byref(alt), ctypes.byref(model._c), ctypes.sizeof(_ffi.NmForest))
        alt.d_leaf_value = packed.data_ptr()
        alt.leaf_stride = 0
        proba = torch.empty_like(want_p)
        label = torch.empty_like(want_l)
        rt.check(rt.lib.nm_forest_eval(rt.ctx, ctypes.byref(alt), nm_device.ptr(dx), n, dx.shape[1],
                                       nm_device.ptr(proba), nm_device.ptr(label), rt.stream()))
        assert torch.equal(proba, want_p) and torch.equal(label, want_l)
    alt.leaf_stride = nc - 1
    with pytest.raises(ValueError):
        rt.check(rt.lib.nm_forest_eval(rt.ctx, ctypes.byref(alt), nm_device.ptr(dx), n, dx.shape[1],
                                       nm_device.ptr(proba), nm_device.ptr(label), rt.stream()))
    alt.leaf_stride = 8
    alt.d_leaf_value = model.leaf_value.data_ptr() + 8          # rows of 8 doubles, off their 64-byte alignment
    with pytest.raises(ValueError):
        rt.check(rt.lib.nm_forest_eval(rt.ctx, ctypes.byref(alt), nm_device.ptr(dx), n, dx.shape[1],
                                       nm_device.ptr(proba), nm_device.ptr(label), rt.stream()))


# ---- seeded clouds against the oracle ---------------------------------------------------------------

@pytest.mark.parametrize("ratio", [0.9, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 5.2])
def test_oracle_radius_ratios(ratio):
    # exercises every kernel variant: W = 3, 5, 7, 9 (LUT kernels) and W = 11 (generic kernel)
    pts = synth.uniform_cloud(6000, extent=4.0, seed=11)
    e = 0.2
    got = multiscale.process_single_core(pts, pts, [e], [ratio * e])
    want = oracle.process_fast(pts, pts, [e], [ratio * e])
    assert_features_close(got, want, pts)


def test_oracle_scene_three_scales():
    pts, _ = synth.scene_cloud(60000, extent=15.0, n_poles=12, n_spheres=3, seed=21)
    edges, radii = [0.10, 0.20, 0.40], [0.30, 0.60, 1.20]
    got, info = multiscale.process_gpu(torch.from_numpy(pts).cuda(), torch.from_numpy(pts).cuda(),
                                       edges, radii, return_info=True)
    want = oracle.process_fast(pts, pts, edges, radii)
    assert_features_close(got.cpu().numpy(), want, pts)
    for s, e in enumerate(edges):
        assert info[s].voxels == len(oracle.Lattice(pts, e).unique_addresses(pts))
        assert info[s].degenerate == int((want[:, 4 * s] < 2).sum())


def test_oracle_separate_query_cloud_partly_outside():
    search = synth.uniform_cloud(8000, extent=3.0, seed=31)
    rs = np.random.RandomState(32)
    query = rs.rand(3000, 3) * 5.0 - 1.0          # a third of them far outside the lattice
    query[:5] = [[-100.0, 0, 0], [0, 1e6, 0], [1.5, 1.5, -50.0], [1e9, 1e9, 1e9], [-3.0, -3.0, -3.0]]
    e, r = 0.25, 0.75
    got = multiscale.process_single_core(query, search, [e], [r])
    want = oracle.process_fast(query, search, [e], [r])
    assert (want[:, 0] == 0).sum() > 100 and (want[:, 0] == 1).sum() > 0
    assert_features_close(got, want, np.concatenate((query[5:], search)))


def test_oracle_sparse_cloud_many_passes():
    # isolated points far apart: every wave needs many passes of the search kernel
    rs = np.random.RandomState(41)
    pts = rs.rand(2000, 3) * 400.0
    pts = np.concatenate((pts, pts[:500] + 0.05), axis=0)
    e, r = 0.1, 0.3
    got, info = multiscale.process_gpu(torch.from_numpy(pts).cuda(), torch.from_numpy(pts).cuda(),
                                       [e], [r], return_info=True)
    want = oracle.process_fast(pts, pts, [e], [r])
    assert_features_close(got.cpu().numpy(), want, pts)
    assert info[0].extra_passes > 0


def test_index_builder_extremes():
    """the fused index builder at its corners, through the ladder call, against the C oracle:
    (a) every point in a superblock of its own, in an order that is not spatial for the coarser scales
        (each block creates as many leaves as it has points; other blocks wait for their publication);
    (b) a wide, nearly empty lattice (20+ address bits per axis: 64-bit superblock keys, long probe
        sequences) holding two dense clusters far apart;
    (c) one hundred thousand copies of a handful of cells (one leaf, maximal run lengths)."""
    rs = np.random.RandomState(733)
    # (a) 60 k isolated points on a jittered coarse grid, 40 cells apart at the finest scale
    g = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(40), indexing="ij"), -1).reshape(-1, 3)
    sparse = (g[rs.permutation(len(g))[:60000]] * 4.0 + rs.rand(60000, 3) * 0.5)
    # (b) two clusters 3 km apart at e = 0.05: widths of 17 bits and more
    a = synth.uniform_cloud(30000, extent=2.0, seed=734)
    b = synth.uniform_cloud(30000, extent=2.0, seed=735) + np.array([3000.0, 2500.0, 40.0])
    wide = np.concatenate((a, b))
    # (c) duplicates
    few = rs.rand(7, 3) * 0.3
    dup = few[rs.randint(0, 7, size=100000)] + rs.rand(100000, 3) * 1e-9
    for pts, edges, radii in ((sparse, [0.1, 0.2, 0.4], [0.3, 0.6, 1.2]),
                              (wide, [0.05, 0.1], [0.15, 0.3]),
                              (dup, [0.05, 0.1], [0.15, 0.3])):
        dev = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
        got, info = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
        want = oracle.process_c(pts, pts, edges, radii)
        assert_features_close(got.cpu().numpy(), want, pts)
        per_scale, info2 = multiscale.process_gpu(dev, dev, edges, radii, return_info=True, per_scale=True)
        assert torch.equal(got, per_scale)
        for x, y in zip(info, info2):
            assert x.voxels == y.voxels and x.leaves == y.leaves and x.voxels > 0


def test_strided_cloud_with_feature_columns():
    # (N, 3+F) clouds: geometry in the first three columns (minimal/README.md:38-40)
    pts = synth.uniform_cloud(5000, extent=3.0, seed=51)
    wide = np.concatenate((pts, np.random.RandomState(52).rand(5000, 4)), axis=1)
    a = multiscale.process_single_core(wide, wide, [0.2], [0.6])
    b = multiscale.process_single_core(pts, pts, [0.2], [0.6])
    assert np.array_equal(a, b)


def test_strict_raises_like_the_reference():
    pts = np.concatenate((synth.uniform_cloud(500, extent=1.0, seed=61), [[50.0, 50.0, 50.0]]))
    out = multiscale.process_single_core(pts, pts, [0.1], [0.3])
    assert out[-1, 0] == 1 and out[-1, 2] == 0 and out[-1, 3] == 0
    with pytest.raises(FloatingPointError):
        multiscale.process_single_core(pts, pts, [0.1], [0.3], strict=True)


def test_neighbor_lists_against_oracle():
    pts, _ = synth.scene_cloud(20000, extent=8.0, n_poles=5, n_spheres=2, seed=71)
    e, r = 0.1, 0.3
    off, idx = multiscale.neighbor_lists(pts[:3000], pts, e, r)
    voxels = oracle.Lattice(pts, e).unique_voxels(pts)
    want_off, want_idx = oracle.neighbors_to_csr(oracle.ball_neighbors_kdtree(pts[:3000], voxels, r))
    assert np.array_equal(off, want_off)
    assert np.array_equal(idx, want_idx)
    # and the fused path's population is the length of those lists
    feats = multiscale.process_single_core(pts[:3000], pts, [e], [r])
    assert np.array_equal(feats[:, 0], np.diff(off))


def test_config2_neighbor_indices_bit_exact_for_every_query():
    """north_star: "neighbor indices bit-exact".  config 2 at its stated size, finest scale (e = 0.10, r = 0.30):
    the index lists of ALL 10^6 queries - multiscale.neighbor_lists, i.e. nm_voxelize + nm_scale_neighbors
    through the C ABI - against the C oracle's enumeration ranked in the sorted unique addresses (np.unique,
    geometry.py:150; the oracle is pinned on the reference's captured lists in tests/test_oracle.py):
    offsets and indices equal, element for element."""
    pts, _, edges, radii = synth.make_config("c2_scene_1m")
    dev = torch.from_numpy(pts).cuda()
    off, idx = multiscale.neighbor_lists(dev, dev, edges[0], radii[0])
    want_off, want_idx = oracle.neighbor_lists_c(pts, pts, edges[0], radii[0])
    assert np.array_equal(off.cpu().numpy(), want_off)
    assert idx.shape[0] == want_idx.shape[0] and np.array_equal(idx.cpu().numpy(), want_idx)
    # and the fused kernel's population is the length of those lists
    f = multiscale.process_gpu(dev, dev, edges[:1], radii[:1])
    assert np.array_equal(f[:, 0].cpu().numpy(), np.diff(want_off).astype(np.float64))


def test_forest_on_gpu_features():
    from sklearn.ensemble import RandomForestClassifier
    pts, labels = synth.scene_cloud(30000, extent=10.0, n_poles=8, n_spheres=3, seed=81)
    feats = multiscale.process_single_core(pts, pts, [0.1, 0.2], [0.3, 0.6])
    clf = RandomForestClassifier(n_estimators=12, max_depth=9, random_state=0).fit(
        feats[:20000], labels[:20000])
    model = classification.ForestModel.from_sklearn(clf)
    assert np.abs(model.predict_proba(feats[20000:]) - clf.predict_proba(feats[20000:])).max() < 1e-14
    assert np.array_equal(model.predict(feats[20000:]), clf.predict(feats[20000:]))


# ---- BASELINE configurations at full size: size-independent properties ----------------------------

def test_config1_full_size_checksum_against_oracle():
    # config 1: 100k uniform points, one scale - small enough for the vectorised oracle
    pts, _, edges, radii = synth.make_config("c1_uniform_100k")
    got = multiscale.process_single_core(pts, pts, edges, radii)
    want = oracle.process_fast(pts, pts, edges, radii)
    assert_features_close(got, want, pts)
    assert len(oracle.Lattice(pts, edges[0]).unique_addresses(pts)) == 51997


def test_config2_full_size_properties():
    # config 2: 1M-point scene, 3 scales.  properties that need no oracle at this size:
    #  (a) permutation equivariance: shuffling the rows shuffles the output rows, bit for bit
    #  (b) query-subset consistency: a query subset against the same search cloud gives the same rows
    #  (c) ranges: population in [1, W^3], 1 >= l1 >= l2 >= 0, l1 + l2 <= 1, l1 >= 1/3
    #  (d) a 20k-row sample agrees with the oracle
    pts, _, edges, radii = synth.make_config("c2_scene_1m")
    dev = torch.from_numpy(pts).cuda()
    full = multiscale.process_gpu(dev, dev, edges, radii)
    perm = torch.randperm(len(pts), generator=torch.Generator().manual_seed(5)).cuda()
    shuffled = dev[perm].contiguous()
    again = multiscale.process_gpu(shuffled, shuffled, edges, radii)
    assert torch.equal(again, full[perm])
    sub = perm[:20000]
    part = multiscale.process_gpu(dev[sub].contiguous(), dev, edges, radii)
    assert torch.equal(part, full[sub])
    f = full.cpu().numpy()
    for s in range(len(edges)):
        n, l1, l2 = f[:, 4 * s], f[:, 4 * s + 2], f[:, 4 * s + 3]
        assert n.min() >= 1 and n.max() <= 343
        ok = n >= 2
        assert np.all(l1[ok] >= 1.0 / 3.0 - 1e-12) and np.all(l1[ok] <= 1.0 + 1e-12)
        assert np.all(l2[ok] <= l1[ok] + 1e-12) and np.all(l2[ok] >= -1e-12)
        assert np.all(l1[ok] + l2[ok] <= 1.0 + 1e-12)
    rows = sub.cpu().numpy()
    want = oracle.process_fast(pts[rows], pts, edges, radii)
    assert_features_close(f[rows], want, pts)
    # (e) ALL rows against the plain-C oracle (real voxel centres, two-pass covariance, Jacobi)
    assert_features_close(f, oracle.process_c(pts, pts, edges, radii), pts)


# ---- multi-GPU path pieces on one GPU ---------------------------------------------------------------

def test_halo_kernels_against_numpy():
    from nimrud_amd import parallel
    pts, _ = synth.scene_cloud(50000, extent=20.0, n_poles=10, n_spheres=4, seed=91)
    wide = np.concatenate((pts, np.zeros((len(pts), 2))), axis=1)          # strided rows
    cloud = torch.from_numpy(wide).cuda()
    be = parallel.HipBackend()
    boxes = np.array([[0, 0, -1, 8, 8, 7], [6, 6, -1, 14, 21, 7], [-5, -5, -5, 30, 30, 30],
                      [100, 100, 100, 101, 101, 101]], dtype=np.float64)
    dboxes = torch.from_numpy(boxes).cuda()
    skip = 2
    counts = be.halo_count(cloud, dboxes, skip).cpu().numpy()
    masks = [np.all((pts >= b[:3]) & (pts <= b[3:]), axis=1) for b in boxes]
    want = [int(m.sum()) if j != skip else 0 for j, m in enumerate(masks)]
    assert list(counts) == want and want[0] > 0 and want[1] > 0 and want[3] == 0
    offsets = torch.from_numpy(np.concatenate(([0], np.cumsum(want)[:-1]))).cuda()
    packed = be.halo_pack(cloud, dboxes, skip, offsets, int(sum(want))).cpu().numpy()
    off = 0
    for j, m in enumerate(masks):
        if j == skip:
            continue
        seg = packed[off:off + want[j]]
        off += want[j]
        # order within a destination is unspecified: compare as sorted row sets
        a = seg[np.lexsort(seg.T[::-1])]
        b = pts[m][np.lexsort(pts[m].T[::-1])]
        assert np.array_equal(a, b)
    out = torch.empty((len(pts), 3), dtype=torch.float64, device="cuda")
    be.copy_xyz(cloud, out)
    assert np.array_equal(out.cpu().numpy(), pts)


def test_cellset_kernels_against_numpy():
    # the coarse cell sets of the Morton-tile halos: the device grid, the dilation and the bit packing
    # against the numpy mirror the CPU tests use (tests/test_parallel_cpu.py), and count / pack by cell set
    from nimrud_amd import parallel
    from test_parallel_cpu import NumpyBackend
    pts, _ = synth.scene_cloud(60000, extent=30.0, n_poles=20, n_spheres=6, seed=99)
    parts = parallel.partition_by_morton(pts, 3, 0.4)
    margin = parallel.halo_margin([0.1, 0.2, 0.4], [0.3, 0.6, 1.2])
    glob = np.concatenate((pts.min(0), pts.max(0)))
    dglob = torch.from_numpy(glob).cuda()
    be, nb = parallel.HipBackend(), NumpyBackend()
    sets = []
    for part in parts:
        tile = np.ascontiguousarray(pts[part])
        got = be.cellset(torch.from_numpy(tile).cuda(), dglob, margin).cpu().numpy()
        want = nb.cellset(torch.from_numpy(tile), torch.from_numpy(glob), margin).numpy()
        assert np.array_equal(got, want)
        assert 0 < np.unpackbits(got.view(np.uint8)).sum() < 32 * parallel.CELLSET_WORDS
        sets.append(want)
    dsets = torch.from_numpy(np.stack(sets)).cuda()
    tile0 = np.ascontiguousarray(pts[parts[0]])
    dest = (dglob, margin, dsets)
    counts = be.halo_count(torch.from_numpy(tile0).cuda(), dest, 0).cpu().numpy()
    hdest = (torch.from_numpy(glob), margin, torch.from_numpy(np.stack(sets)))
    want_counts = nb.halo_count(torch.from_numpy(tile0), hdest, 0).numpy()
    assert np.array_equal(counts, want_counts) and counts[0] == 0 and counts[1:].sum() > 0
    offsets = np.concatenate(([0], np.cumsum(counts)[:-1]))
    packed = be.halo_pack(torch.from_numpy(tile0).cuda(), dest, 0, torch.from_numpy(offsets).cuda(),
                          int(counts.sum())).cpu().numpy()
    want_rows = nb.halo_pack(torch.from_numpy(tile0), hdest, 0, torch.from_numpy(offsets),
                             int(counts.sum())).numpy()
    for j in range(1, 3):
        a = packed[offsets[j]:offsets[j] + counts[j]]
        b = want_rows[offsets[j]:offsets[j] + counts[j]]
        assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])


@pytest.mark.parametrize("halo", ["cells", "boxes"])
def test_halo_exchange_through_rccl_one_rank(halo):
    # nm_halo_exchange on real RCCL: a one-rank communicator created by the library itself, the rank its
    # own neighbour (NM_HALO_INCLUDE_SELF), so the all-gathers, the pack, and a grouped ncclSend/ncclRecv
    # of the whole tile run on hardware, on the stream the kernels use, into the buffer behind the tile.
    # the search cloud then is [tile | tile]: same voxels, so the features must be bit-identical.
    from nimrud_amd import parallel
    pts, _ = synth.scene_cloud(50000, extent=16.0, n_poles=10, n_spheres=4, seed=103)
    dev = torch.from_numpy(pts).cuda()
    comm = parallel.RcclComm(rank=0, world=1)
    try:
        plan = parallel.TilePlan(dev, [0.1, 0.2, 0.4], [0.3, 0.6, 1.2], comm=comm, halo=halo)
        plan.include_self = True
        plan.static = True          # (set before the first step, as bench.py does: the first step makes the plan)
        out = parallel.process_tile(plan)
        torch.cuda.synchronize()
        assert plan.halo_received == len(pts) and plan.halo_sent == len(pts)
        halo_rows = plan._buffer[len(pts):2 * len(pts)].cpu().numpy()
        assert np.array_equal(halo_rows[np.lexsort(halo_rows.T[::-1])], pts[np.lexsort(pts.T[::-1])])
        want = multiscale.process_gpu(dev, dev, [0.1, 0.2, 0.4], [0.3, 0.6, 1.2])
        assert torch.equal(out, want)
        # steps on a static cloud keep the plan (NM_HALO_REUSE_PLAN): the rows are packed and exchanged again,
        # the all-gathers and the host synchronisation that learns the sizes are not repeated
        import ctypes
        rt = _device_runtime()
        syncs, exchanges = ctypes.c_int64(0), ctypes.c_int64(0)
        rt.check(rt.lib.nm_halo_stats(rt.ctx, ctypes.byref(syncs), ctypes.byref(exchanges)))
        s0, e0 = syncs.value, exchanges.value
        plan._buffer[len(pts):].zero_()
        for _ in range(3):
            out_again = parallel.process_tile(plan)
            plan._buffer[len(pts):2 * len(pts)].add_(0.0)      # (touch: the rows are really there again)
        torch.cuda.synchronize()
        rt.check(rt.lib.nm_halo_stats(rt.ctx, ctypes.byref(syncs), ctypes.byref(exchanges)))
        assert exchanges.value - e0 == 3 and syncs.value - s0 == 0
        assert plan.halo_received == len(pts) and torch.equal(out_again, want)
        halo_rows = plan._buffer[len(pts):2 * len(pts)].cpu().numpy()
        assert np.array_equal(halo_rows[np.lexsort(halo_rows.T[::-1])], pts[np.lexsort(pts.T[::-1])])
        plan.static = False
        # a second step reuses the buffers; without the self-neighbour nothing is exchanged
        plan.include_self = False
        out2 = parallel.process_tile(plan)
        assert plan.halo_received == 0 and torch.equal(out2, want)
    finally:
        comm.close()


def test_halo_exchange_with_an_empty_tile_and_with_an_unwell_rank():
    """nm_halo_exchange must never leave a rank alone in a collective: an EMPTY tile takes part (empty box,
    empty cell set, nothing sent or received), and a rank that alone is unwell (here: a sticky lattice report
    of an earlier call) goes through the all-gathers and every rank returns its status afterwards.  one-rank
    RCCL communicator: the collectives and the status word run on hardware."""
    import ctypes
    from nimrud_amd import _ffi, parallel, device as nm_device
    rt = _device_runtime()
    comm = parallel.RcclComm(rank=0, world=1)
    try:
        empty = torch.empty((0, 3), dtype=torch.float64, device="cuda")
        plan = parallel.TilePlan(empty, [0.1, 0.2], [0.3, 0.6], comm=comm, halo="cells")
        plan.include_self = True
        out = parallel.process_tile(plan)
        torch.cuda.synchronize()
        assert out.shape == (0, 8) and plan.halo_received == 0 and plan.halo_sent == 0
        # an unwell rank: leave a lattice report in the context, then enter the exchange without looking
        pts = synth.uniform_cloud(4000, extent=2.0, seed=77)
        dev = torch.from_numpy(pts).cuda()
        multiscale.process_gpu(dev, dev, [1e-9], [3e-9])
        torch.cuda.synchronize()
        work = torch.empty(int(rt.lib.nm_halo_workspace_bytes(8000, 1)), dtype=torch.uint8, device="cuda")
        recv = torch.empty((8000, 3), dtype=torch.float64, device="cuda")
        glob = torch.empty(6, dtype=torch.float64, device="cuda")
        sent, received = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = rt.lib.nm_halo_exchange(rt.ctx, comm.handle, 1, 0, nm_device.ptr(dev), 4000, 3, 0.5,
                                     parallel.HALO_CELLS | parallel.HALO_INCLUDE_SELF, nm_device.ptr(recv), 8000,
                                     ctypes.byref(received), ctypes.byref(sent), nm_device.ptr(glob),
                                     nm_device.ptr(work), work.numel(), rt.stream())
        assert rc == _ffi.NM_ERR_LATTICE         # after the collectives, not before them
        with pytest.raises(ValueError, match="too small"):
            rt.check(rc)
        # reported once; the context and the communicator carry on
        plan = parallel.TilePlan(dev, [0.1], [0.3], comm=comm, halo="cells")
        out = parallel.process_tile(plan)
        assert torch.equal(out, multiscale.process_gpu(dev, dev, [0.1], [0.3]))
    finally:
        comm.close()


def test_prefix_query_mode_matches_separate_clouds():
    # queries = leading rows of the search buffer (what a tile + halo looks like)
    pts, _ = synth.scene_cloud(40000, extent=12.0, n_poles=8, n_spheres=3, seed=93)
    search = torch.from_numpy(pts).cuda()
    n_query = 25000
    from nimrud_amd import parallel
    be = parallel.HipBackend()
    out = torch.empty((n_query, 8), dtype=torch.float64, device="cuda")
    info = torch.zeros((2, 4), dtype=torch.int64, device="cuda")
    bounds = torch.from_numpy(np.concatenate((pts.min(0), pts.max(0)))).cuda()
    be.features(search, n_query, bounds, [0.1, 0.2], [0.3, 0.6], out, info)
    want = multiscale.process_gpu(search[:n_query].clone(), search, [0.1, 0.2], [0.3, 0.6])
    assert torch.equal(out, want)
    ref = oracle.process_fast(pts[:n_query], pts, [0.1, 0.2], [0.3, 0.6])
    assert_features_close(out.cpu().numpy(), ref, pts)


def test_single_rank_tile_plan_on_gpu():
    from nimrud_amd import parallel
    pts, _ = synth.scene_cloud(30000, extent=10.0, n_poles=6, n_spheres=2, seed=95)
    plan = parallel.TilePlan(torch.from_numpy(pts).cuda(), [0.1, 0.2], [0.3, 0.6])
    out = parallel.process_tile(plan)
    want = multiscale.process_gpu(torch.from_numpy(pts).cuda(), torch.from_numpy(pts).cuda(),
                                  [0.1, 0.2], [0.3, 0.6])
    assert torch.equal(out, want)


def _two_rank_worker(rank, world, port, points, parts, results):
    import os
    import torch.distributed as dist
    from nimrud_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tile = torch.from_numpy(np.ascontiguousarray(points[parts[rank]])).cuda()
        plan = parallel.TilePlan(tile, [0.1, 0.2, 0.4], [0.3, 0.6, 1.2])
        out = parallel.process_tile(plan)
        results[rank] = (out.cpu().numpy(), plan.halo_received)
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu_hip_backend():
    # both ranks run the HIP data path on cuda:0; the collectives go over gloo (staged through host
    # memory) because RCCL needs one GPU per rank.  result must equal the single-GPU run.
    import socket
    import torch.multiprocessing as mp
    from nimrud_amd import parallel
    points, _ = synth.scene_cloud(60000, extent=30.0, n_poles=20, n_spheres=6, seed=97)
    parts = parallel.partition_tiles(points, 2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_two_rank_worker, args=(2, port, points, parts, results), nprocs=2, join=True)
    dev = torch.from_numpy(points).cuda()
    whole = multiscale.process_gpu(dev, dev, [0.1, 0.2, 0.4], [0.3, 0.6, 1.2]).cpu().numpy()
    for rank in range(2):
        out, received = results[rank]
        assert np.array_equal(out, whole[parts[rank]])       # bit-identical
        assert 0 < received < len(points) - len(parts[rank])


# ---- the one-sort ladder against the per-scale path ---------------------------------------------------

def test_ladder_call_is_bit_identical_to_per_scale_calls():
    pts, _ = synth.scene_cloud(150000, extent=25.0, n_poles=30, n_spheres=8, seed=101)
    edges = [0.40, 0.05, 0.10, 0.80, 0.20]               # caller order, finest not first
    radii = [1.20, 0.15, 0.30, 2.40, 0.60]
    dev = torch.from_numpy(pts).cuda()
    a, ia = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
    b, ib = multiscale.process_gpu(dev, dev, edges, radii, return_info=True, per_scale=True)
    assert torch.equal(a, b)
    for x, y in zip(ia, ib):
        assert x.voxels == y.voxels and x.degenerate == y.degenerate and x.leaves == y.leaves
    # separate query cloud, partly outside
    rs = np.random.RandomState(102)
    query = torch.from_numpy(rs.rand(20000, 3) * 30.0 - 2.5).cuda()
    c = multiscale.process_gpu(query, dev, edges, radii)
    d = multiscale.process_gpu(query, dev, edges, radii, per_scale=True)
    assert torch.equal(c, d)
    want = oracle.process_fast(query.cpu().numpy(), pts, edges[:2], radii[:2])
    assert_features_close(c.cpu().numpy()[:, :8], want, pts)


def test_covariance_output_against_numpy_cov():
    """nm_set_covariance_output: the ddof=1 covariance numpy.cov gives for the oracle's neighborhoods
    (features.py:43), through the ladder call, the per-scale calls, the generic kernel (r/e = 5.5) and
    the kNN fallback; and its normalised eigenvalues are the features of the same row."""
    pts, _ = synth.scene_cloud(2500, extent=6.0, n_poles=4, n_spheres=2, seed=818)
    pts = pts + np.array([120.0, -45.0, 7.0])
    edges, radii = [0.10, 0.20, 0.1], [0.30, 0.60, 0.55]
    dev = torch.from_numpy(pts).cuda()
    feats, cov = multiscale.process_gpu_covariance(dev, dev, edges, radii)
    feats2, cov2 = multiscale.process_gpu_covariance(dev, dev, edges, radii, per_scale=True)
    assert torch.equal(cov, cov2) and torch.equal(feats, feats2)
    assert torch.equal(feats, multiscale.process_gpu(dev, dev, edges, radii))
    got = cov.cpu().numpy()
    f = feats.cpu().numpy()
    for s, (e, r) in enumerate(zip(edges, radii)):
        want = oracle.one_scale_covariance(pts, pts, e, r)
        c = got[:, 6 * s:6 * s + 6]
        scale = np.abs(want).max()
        assert np.abs(c - want).max() <= 1e-9 * scale
        few = f[:, 4 * s] < 2
        assert np.all(c[few] == 0.0)
        # eigenvalues of the 3x3 it describes, normalised, are columns 2 and 3
        m = np.zeros((len(c), 3, 3))
        iu = np.triu_indices(3)
        m[:, iu[0], iu[1]] = c
        m[:, iu[1], iu[0]] = c
        w = np.linalg.eigvalsh(m)[:, ::-1]
        ok = ~few & (w.sum(1) > 0)
        assert np.abs(w[ok, 0] / w[ok].sum(1) - f[ok, 4 * s + 2]).max() < 1e-9
        assert np.abs(w[ok, 1] / w[ok].sum(1) - f[ok, 4 * s + 3]).max() < 1e-9
    # with the fallback on, sparse rows carry the covariance of their k nearest voxels: still consistent
    rs = np.random.RandomState(819)
    sparse = rs.rand(3000, 3) * 6.0
    sdev = torch.from_numpy(sparse).cuda()
    fk, ck = multiscale.process_gpu_covariance(sdev, sdev, [0.1], [0.3], knn_min=6)
    fk, ck = fk.cpu().numpy(), ck.cpu().numpy()
    m = np.zeros((len(ck), 3, 3))
    iu = np.triu_indices(3)
    m[:, iu[0], iu[1]] = ck
    m[:, iu[1], iu[0]] = ck
    w = np.linalg.eigvalsh(m)[:, ::-1]
    ok = w.sum(1) > 0
    assert ok.sum() > 1000
    assert np.abs(w[ok, 0] / w[ok].sum(1) - fk[ok, 2]).max() < 1e-9
    # the switch is cleared after the call
    assert torch.equal(multiscale.process_gpu(dev, dev, edges, radii), feats)


def test_normal_output_against_numpy_eigh():
    """nm_set_normal_output: the eigenvector of the smallest covariance eigenvalue (plane normal), sign
    convention included, against numpy.linalg.eigh on the oracle's neighborhoods wherever the data
    define it (relative gap between the two smallest eigenvalues above 1e-3); always a unit vector,
    always an eigenvector of the covariance the library itself reports."""
    pts, _ = synth.scene_cloud(2500, extent=6.0, n_poles=4, n_spheres=2, seed=828)
    edges, radii = [0.10, 0.20, 0.1], [0.30, 0.60, 0.55]
    dev = torch.from_numpy(pts).cuda()
    cov = torch.zeros((len(pts), 6 * len(edges)), dtype=torch.float64, device="cuda")
    feats, normals = multiscale.process_gpu_normals(dev, dev, edges, radii, cov_out=cov)
    f2, n2 = multiscale.process_gpu_normals(dev, dev, edges, radii, per_scale=True)
    assert torch.equal(normals, n2) and torch.equal(feats, f2)
    got, f, c = normals.cpu().numpy(), feats.cpu().numpy(), cov.cpu().numpy()
    for s, (e, r) in enumerate(zip(edges, radii)):
        want, gap = oracle.one_scale_normals(pts, pts, e, r)
        v = got[:, 3 * s:3 * s + 3]
        few = f[:, 4 * s] < 3
        assert np.all(v[few] == 0.0)
        assert np.abs(np.linalg.norm(v[~few], axis=1) - 1.0).max() < 1e-12
        clear = ~few & (gap > 1e-3)
        assert clear.sum() > 1000
        # the direction, and the sign wherever the deciding (z) component is not rounding noise
        d = np.minimum(np.abs(v[clear] - want[clear]).max(axis=1), np.abs(v[clear] + want[clear]).max(axis=1))
        assert d.max() < 1e-6
        up = clear & (np.abs(want[:, 2]) > 1e-6)
        assert up.sum() > 500 and np.all(v[up, 2] > 0) and np.abs(v[up] - want[up]).max() < 1e-6
        # an eigenvector of the reported covariance with the smallest eigenvalue
        m = np.zeros((len(v), 3, 3))
        iu = np.triu_indices(3)
        m[:, iu[0], iu[1]] = c[:, 6 * s:6 * s + 6]
        m[:, iu[1], iu[0]] = c[:, 6 * s:6 * s + 6]
        w = np.linalg.eigvalsh(m)
        resid = np.einsum("nij,nj->ni", m, v) - w[:, :1] * v
        scale = np.abs(m).max(axis=(1, 2))
        ok = ~few & (scale > 0)
        assert (np.abs(resid[ok]).max(axis=1) / scale[ok]).max() < 1e-9


def test_vector_field_mean_against_oracle():
    """nm_field_mean: neighborhood means of per-point attributes over the same voxels
    and the same ball as the features; a constant field comes back constant, a coordinate field comes
    back as the neighborhood's voxel-weighted centre."""
    rs = np.random.RandomState(939)
    pts, labels = synth.scene_cloud(6000, extent=8.0, n_poles=5, n_spheres=2, seed=938)
    query = np.concatenate((pts[:1500], rs.rand(300, 3) * 10.0 - 1.0))
    attr = np.stack([np.sin(pts[:, 0]), pts[:, 2] ** 2, labels.astype(np.float64),
                     np.full(len(pts), 3.25), rs.rand(len(pts))], axis=1)
    edges, radii = [0.1, 0.25], [0.3, 1.2]
    got = fields.vector_field_mean(query, pts, attr, edges, radii)
    assert got.shape == (len(query), 10)
    for s, (e, r) in enumerate(zip(edges, radii)):
        want = oracle.one_scale_field_mean(query, pts, attr, e, r)
        assert np.abs(got[:, 5 * s:5 * s + 5] - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
        pop = oracle.one_scale_fast(query, pts, e, r)[:, 0]
        const = got[:, 5 * s + 3]
        assert np.all(const[pop == 0] == 0.0) and np.abs(const[pop > 0] - 3.25).max() < 1e-12
    # one column, given as a vector; query = search on the device
    dev = torch.from_numpy(pts).cuda()
    one = fields.vector_field_mean_gpu(dev, dev, torch.from_numpy(attr[:, 1]).cuda(), [0.1], [0.3])
    assert one.shape == (len(pts), 1)
    assert np.abs(one.cpu().numpy()[:, 0] - oracle.one_scale_field_mean(pts, pts, attr[:, 1], 0.1, 0.3)[:, 0]).max() < 1e-11
    with pytest.raises(ValueError):
        fields.vector_field_mean(query, pts, np.zeros((len(pts), 17)), [0.1], [0.3])
    with pytest.raises(ValueError):
        fields.vector_field_mean(query, pts, attr[:-1], [0.1], [0.3])


def test_ladder_argument_checks():
    """the C ABI refuses what it cannot do, with the reference's exception type where there is one."""
    pts = synth.uniform_cloud(3000, extent=2.0, seed=909)
    dev = torch.from_numpy(pts).cuda()
    with pytest.raises(AssertionError):                       # multiscale.py:32
        multiscale.process_gpu(dev, dev, [0.1, 0.2], [0.3])
    with pytest.raises(ValueError):                           # more scales than one call takes
        multiscale.process_gpu(dev, dev, [0.1] * 33, [0.3] * 33)
    with pytest.raises(ValueError):                           # 64 address bits are not enough
        multiscale.process_single_core(pts, pts, [1e-9], [3e-9])       # geometry.py:59-60
    # the GPU-resident form only enqueues work (the lattices are built on the device): the same failure
    # surfaces at the next synchronisation point, once, and the context carries on
    multiscale.process_gpu(dev, dev, [0.1, 1e-9], [0.3, 3e-9])
    with pytest.raises(ValueError, match="too small"):
        _device_runtime().check_async(wait=True)
    _device_runtime().check_async(wait=True)
    with pytest.raises(ValueError):                           # a cloud without extent on an axis
        flat = pts.copy()
        flat[:, 2] = 1.0                                      # span/e = 1 exactly -> width 0 (geometry.py:74)
        multiscale.process_single_core(flat, flat, [0.25], [0.75])
    with pytest.raises(ValueError):
        multiscale.process_gpu(dev, dev, [0.1], [0.3], cov_out=torch.zeros((3000, 5), dtype=torch.float64,
                                                                           device="cuda"))
    with pytest.raises(ValueError):
        multiscale.process_gpu(dev, dev, [0.1], [0.3], normal_out=torch.zeros((3000, 3), dtype=torch.float32,
                                                                              device="cuda"))
    # row strides are 32-bit on the device: 2^31 elements or more is refused before anything is launched
    import ctypes
    from nimrud_amd import _ffi, device as nm_device
    rt = _device_runtime()
    e1, r1 = (ctypes.c_double * 1)(0.1), (ctypes.c_double * 1)(0.3)
    work = torch.empty(int(rt.lib.nm_ladder_workspace_bytes(3000, 3000, 1)), dtype=torch.uint8, device="cuda")
    out = torch.zeros((3000, 4), dtype=torch.float64, device="cuda")
    rc = rt.lib.nm_ladder_features(rt.ctx, nm_device.ptr(dev), 3000, 3, nm_device.ptr(dev), 3000, 3, e1, r1, 1,
                                   None, nm_device.ptr(out), 1 << 31, None, nm_device.ptr(work), work.numel(),
                                   rt.stream())
    assert rc == _ffi.NM_ERR_INVALID
    assert b"stride" in rt.lib.nm_last_error(rt.ctx)
    assert float(out.abs().sum()) == 0.0
    # and the library is still usable afterwards
    got = multiscale.process_gpu(dev, dev, [0.1], [0.3]).cpu().numpy()
    assert_features_close(got, oracle.process_fast(pts, pts, [0.1], [0.3]), pts)


def test_bounds_pass_matches_numpy():
    # nm_bounds (one atomic set per block) for sizes around the unrolled loop's tail and the grid's width; the
    # ladder's own bounds pass (per-block extrema in the workspace, no atomics) is what every other test runs on
    import ctypes
    from nimrud_amd import device as nm_device
    rt = _device_runtime()
    rs = np.random.RandomState(4711)
    for n in (1, 2, 63, 257, 1024 * 256 - 1, 1024 * 256 * 4 + 3, 1500001):
        pts = rs.randn(n, 5) * np.array([3.0, 50.0, 0.01, 1.0, 1.0]) + np.array([1e5, -2e6, 0.5, 0.0, 0.0])
        dev = torch.from_numpy(pts).cuda()
        out = torch.empty(6, dtype=torch.float64, device="cuda")
        rt.check(rt.lib.nm_bounds(rt.ctx, nm_device.ptr(dev), n, 5, nm_device.ptr(out), rt.stream()))
        got = out.cpu().numpy()
        assert np.array_equal(got[:3], pts[:, :3].min(0)) and np.array_equal(got[3:], pts[:, :3].max(0))
        # a contiguous (n, 3) cloud takes the flat form (16-byte pairs, axes by position): odd and even n, the
        # extrema planted in the first and last rows and in the odd tail; 8 bytes off alignment: the row form
        c3 = np.ascontiguousarray(pts[:, :3])
        c3[0] = c3.min(0) - 1.0
        c3[-1] = c3.max(0) + 1.0
        flat = torch.from_numpy(np.concatenate([[0.0], c3.ravel()])).cuda()
        for ofs in (1, 0):
            src = flat[ofs:] if ofs else torch.from_numpy(c3).cuda()
            rt.check(rt.lib.nm_bounds(rt.ctx, nm_device.ptr(src), n, 3, nm_device.ptr(out), rt.stream()))
            got = out.cpu().numpy()
            assert np.array_equal(got[:3], c3.min(0)) and np.array_equal(got[3:], c3.max(0)), (n, ofs)


def test_device_built_lattices_match_the_hosts():
    # nm_ladder_features builds every lattice on the device (geometry.py:37-64: min - e/2, widths =
    # ceil(log2(span/e))); nm_multiscale_features takes them from the host, where numpy does that arithmetic.
    # same lattices -> bit-identical features, for extents that land exactly on powers of two as well
    # (width = log2 exactly) and at UTM-like offsets.
    import ctypes
    from nimrud_amd import _ffi, device as nm_device
    rs = np.random.RandomState(707)
    rt = _device_runtime()
    cases = []
    for k in range(6):
        cases.append(rs.rand(20000, 3) * rs.uniform(1.0, 40.0, 3) + rs.uniform(-50, 50, 3))
    exact = rs.rand(20000, 3) * np.array([12.7, 6.3, 3.1])
    exact[0] = 0.0
    exact[1] = [12.7, 6.3, 3.1]                       # span/e = 2^7, 2^6, 2^5 at e = 0.1 (+ one cell)
    cases.append(exact)
    cases.append(cases[0] + np.array([4.0e5, 5.1e6, 300.0]))
    for pts in cases:
        pts = np.ascontiguousarray(pts)
        edges, radii = [0.1, 0.2, 0.45], [0.3, 0.6, 0.9]
        dev = torch.from_numpy(pts).cuda()
        got, info = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
        # the host-lattice entry point with numpy's lattices
        lats = (_ffi.NmLattice * 3)()
        for s, e in enumerate(edges):
            mc, _, widths, _ = geometry.lattice_parameters(pts.min(0), pts.max(0), e)
            lat = geometry.make_nm_lattice(mc, e, widths)
            ctypes.memmove(ctypes.byref(lats[s]), ctypes.byref(lat), ctypes.sizeof(_ffi.NmLattice))
        rad = (ctypes.c_double * 3)(*radii)
        want = torch.empty_like(got)
        nbytes = rt.lib.nm_multiscale_workspace_bytes(len(pts), len(pts), lats, 3)
        work = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
        rt.check(rt.lib.nm_multiscale_features(
            rt.ctx, nm_device.ptr(dev), len(pts), 3, nm_device.ptr(dev), len(pts), 3, lats, rad, 3,
            nm_device.ptr(want), 12, None, nm_device.ptr(work), work.numel(), rt.stream()))
        assert torch.equal(got, want)
        for s, e in enumerate(edges):
            lat = oracle.Lattice(pts, e)
            assert info[s].voxels == len(np.unique(lat.coordinate_to_address(pts)))


def test_scale_loop_and_per_scale_launches_are_bit_identical():
    """nm_set_fuse_scales: consecutive scales with one radius/edge ratio run in one launch of the search
    kernel (the wave walks the scales) or in one launch each.  same bits, same counters.  (nm_set_overlap
    is accepted and ignored since ABI 5.)"""
    from nimrud_amd import device
    rt = device.get_runtime()
    pts, _ = synth.scene_cloud(300000, extent=35.0, n_poles=40, n_spheres=10, seed=606)
    # 0.1/0.3 and 0.2/0.6 share a window, 0.4/1.0 (ratio 2.5) does not, 0.8/2.4 again: three launches
    edges, radii = [0.10, 0.20, 0.40, 0.80], [0.30, 0.60, 1.00, 2.40]
    dev = torch.from_numpy(pts).cuda()
    a, ia = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
    try:
        rt.check(rt.lib.nm_set_fuse_scales(rt.ctx, 0))
        rt.check(rt.lib.nm_set_overlap(rt.ctx, 1))
        for _ in range(2):
            b, ib = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
            assert torch.equal(a, b)
            assert [(x.voxels, x.leaves, x.degenerate) for x in ia] == \
                   [(x.voxels, x.leaves, x.degenerate) for x in ib]
    finally:
        rt.check(rt.lib.nm_set_fuse_scales(rt.ctx, 1))
        rt.check(rt.lib.nm_set_overlap(rt.ctx, 0))
    want = oracle.process_fast(pts, pts, edges, radii)
    assert_features_close(a.cpu().numpy(), want, pts)


def test_forest_behind_the_last_scale_matches_the_stand_alone_evaluator():
    # nm_set_forest_output: labels and probabilities out of the search kernel's epilogue against
    # nm_forest_eval on the finished matrix and against the oracle's forest walk; also with the kNN fallback
    # on (rows are rewritten after the last kernel: the library must evaluate the finished matrix) and with a
    # last scale of unusual ratio (generic kernel: no epilogue to carry the forest).
    pts, labels = synth.scene_cloud(80000, extent=16.0, n_poles=14, n_spheres=5, seed=153, five_class=True)
    dev = torch.from_numpy(pts).cuda()
    for edges, radii, kw in (([0.1, 0.2, 0.4], [0.3, 0.6, 1.2], {}),
                             ([0.1, 0.2, 0.4], [0.3, 0.6, 1.2], {"knn_min": 6}),
                             ([0.1, 0.2, 0.1], [0.3, 0.6, 0.62], {}),
                             ([0.25], [0.75], {})):
        feats = multiscale.process_gpu(dev, dev, edges, radii, **kw)
        tr, va = classification.balanced_split(labels, seed=1)
        model, clf = classification.train_forest(feats[torch.from_numpy(tr).cuda()], labels[tr],
                                                 n_estimators=12, max_depth=9, n_jobs=4)
        label, proba, feats2 = classification.classify_cloud(dev, edges, radii, model, want_proba=True, **kw)
        assert torch.equal(feats, feats2)
        label0, proba0, _ = classification.classify_cloud(dev, edges, radii, model, fused=False,
                                                          want_proba=True, **kw)
        assert torch.equal(label, label0) and torch.equal(proba, proba0)
        omodel = classification.ForestModel.flatten_sklearn(clf)
        rows = np.arange(0, len(pts), 17)
        want = oracle.forest_predict_proba(omodel, feats.cpu().numpy()[rows])
        assert np.abs(proba.cpu().numpy()[rows] - want).max() <= 1e-15
        assert np.array_equal(label.cpu().numpy()[rows], want.argmax(1))
    # a forest the epilogue cannot take (more features than 20) falls back to the stand-alone evaluator
    edges6, radii6 = [0.1, 0.15, 0.2, 0.3, 0.4, 0.6], [0.3, 0.45, 0.6, 0.9, 1.2, 1.8]
    feats = multiscale.process_gpu(dev, dev, edges6, radii6)
    model, clf = classification.train_forest(feats[::7], labels[::7], n_estimators=8, max_depth=8, n_jobs=4)
    label, _ = classification.classify_cloud(dev, edges6, radii6, model)
    assert np.array_equal(model.classes[label.cpu().numpy()[::5]], clf.predict(feats.cpu().numpy()[::5]))


def test_plain_c_host_through_the_c_abi(tmp_path):
    """examples/c_abi_demo.c: gcc-compiled C99, device buffers from hipMalloc, lattices built in C from
    nm_bounds - no Python and no torch between the caller and libnimrud_hip.so.  same numbers as the
    Python host, and as the oracle."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "c_abi_demo")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    pts, _ = synth.scene_cloud(60000, extent=12.0, n_poles=10, n_spheres=4, seed=515)
    edges, radii = [0.05, 0.10, 0.20], [0.15, 0.30, 0.60]
    src, dst = tmp_path / "points.f64", tmp_path / "features.f64"
    np.ascontiguousarray(pts, dtype="<f8").tofile(src)
    args = [exe, str(src), str(len(pts)), str(dst)]
    for e, r in zip(edges, radii):
        args += [repr(e), repr(r)]
    via_python = multiscale.process_single_core(pts, pts, edges, radii)
    want = oracle.process_c(pts, pts, edges, radii)
    for device_lattice in ("0", "1"):      # lattices built in C from nm_bounds / on the device by the library
        env = dict(os.environ, NM_DEMO_DEVICE_LATTICE=device_lattice)
        done = subprocess.run(args, capture_output=True, text=True, timeout=120, env=env)
        assert done.returncode == 0, done.stderr
        assert ("device" if device_lattice == "1" else "host") in done.stdout
        got = np.fromfile(dst, dtype="<f8").reshape(len(pts), 4 * len(edges))
        assert np.array_equal(got, via_python)
        assert_features_close(got, want, pts)
    # what VoxelFilter raises on the host comes back through nm_check
    bad = subprocess.run([exe, str(src), str(len(pts)), str(dst), "1e-9", "3e-9"], capture_output=True,
                         text=True, timeout=120, env=dict(os.environ, NM_DEMO_DEVICE_LATTICE="1"))
    assert bad.returncode == 3 and "too small" in bad.stderr


# ---- edge cases --------------------------------------------------------------------------------------

def test_empty_and_tiny_inputs():
    search = synth.uniform_cloud(500, extent=2.0, seed=111)
    out = multiscale.process_single_core(np.zeros((0, 3)), search, [0.2, 0.4], [0.6, 1.2])
    assert out.shape == (0, 8)
    out = multiscale.process_single_core(search, search, [], [])
    assert out.shape == (500, 0)
    two = np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]])
    got = multiscale.process_single_core(two, two, [0.5], [2.0])
    want = oracle.process_fast(two, two, [0.5], [2.0])
    assert_features_close(got, want, two)
    with pytest.raises(ValueError):
        multiscale.process_single_core(two[:1], two[:1], [0.5], [2.0])     # geometry.py:34


@pytest.mark.parametrize("ratio", [0.3, 0.49, 6.0, 8.4])
def test_generic_kernel_radius_ratios(ratio):
    # W = 1 (only the home voxel can qualify), W = 13 and W = 17: the generic kernel
    pts = synth.uniform_cloud(3000, extent=2.0, seed=113)
    e = 0.1
    got = multiscale.process_single_core(pts, pts, [e], [ratio * e])
    want = oracle.process_fast(pts, pts, [e], [ratio * e])
    assert_features_close(got, want, pts)


def test_duplicates_and_points_on_cell_boundaries():
    rs = np.random.RandomState(115)
    e = 0.25
    base = rs.randint(0, 12, size=(4000, 3)).astype(np.float64) * e      # exactly on multiples of e
    pts = np.concatenate((base, base[:1500], base[:700] + e / 2, rs.rand(500, 3) * 3.0), axis=0)
    for r in (0.75, 0.5, 1.0):
        got = multiscale.process_single_core(pts, pts, [e], [r])
        want = oracle.process_fast(pts, pts, [e], [r])
        assert_features_close(got, want, pts)
    off, idx = multiscale.neighbor_lists(pts[:500], pts, e, 0.75)
    voxels = oracle.Lattice(pts, e).unique_voxels(pts)
    w_off, w_idx = oracle.neighbors_to_csr(oracle.ball_neighbors_kdtree(pts[:500], voxels, 0.75))
    assert np.array_equal(off, w_off) and np.array_equal(idx, w_idx)


def test_far_from_the_origin_the_deviation_is_the_references_own_cancellation():
    """the randomised sweep's largest eigen-feature deviations (4e-9 to 8e-9, inside the 1e-5 |b| + 1e-9 contract)
    all sit on clouds millions of metres from the origin.  they are not the kernel's: it forms the covariance from
    exact integer moments of cell offsets, while the reference (and the oracle that restates it) subtracts a mean
    from coordinates of magnitude 1e7 in fp64 (numpy.cov, features.py:43).  against the covariance of the exact
    integer cell offsets - the same neighborhoods, no cancellation - the kernel is good to 1e-13; numpy.cov on
    the raw centres is off by 1e-9 and more."""
    rs = np.random.RandomState(87)
    pts = synth.uniform_cloud(12000, extent=3.0, seed=87) * 1.0000001 + np.array([8.9e6, -3.1e6, 4.0e5])
    e, r = 0.0655109, 0.17083
    dev = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    got = multiscale.process_gpu(dev, dev, [e], [r]).cpu().numpy()
    lat = oracle.Lattice(pts, e)
    voxels = lat.unique_voxels(pts)
    rows = rs.choice(len(pts), 400, replace=False)
    lists = oracle.ball_neighbors_kdtree(pts[rows], voxels, r)
    worst_gpu = worst_numpy = 0.0
    for row, nb in zip(rows, lists):
        centres = voxels[nb]
        assert got[row, 0] == len(centres)
        if len(centres) < 3:
            continue
        cells = np.round((centres - centres[0]) / e)
        assert np.abs((centres - centres[0]) / e - cells).max() < 1e-3
        w = np.linalg.eigvalsh(np.cov(cells, rowvar=False))
        w = w / w.sum()
        v = np.linalg.eigvalsh(np.cov(centres, rowvar=False))
        v = v / v.sum()
        worst_gpu = max(worst_gpu, abs(got[row, 2] - w[2]), abs(got[row, 3] - w[1]))
        worst_numpy = max(worst_numpy, abs(v[2] - w[2]), abs(v[1] - w[1]))
    print("far from the origin: kernel vs exact %.3g, numpy.cov on raw centres vs exact %.3g" % (worst_gpu, worst_numpy))
    assert worst_gpu <= 1e-13
    assert worst_numpy <= 1e-7          # (what the contract's absolute term is there for)


def test_huge_coordinates_disable_window_pruning():
    # 16 ulp of the largest coordinate exceeds 1e-4 cell: the kernel must test every candidate
    pts = synth.uniform_cloud(4000, extent=1.0, seed=117) + np.array([3.0e9, -2.0e9, 1.0e9])
    e, r = 0.05, 0.15
    got = multiscale.process_single_core(pts, pts, [e], [r])
    want = oracle.process_fast(pts, pts, [e], [r])
    assert np.array_equal(got[:, 0], want[:, 0])
    assert_features_close(got, want, pts)


def test_wide_window_many_rows():
    # W = 9 on a thin slab: long boxes, row cap and superblock table limits of the staging
    rs = np.random.RandomState(119)
    pts = np.concatenate((rs.rand(20000, 1) * 40.0, rs.rand(20000, 1) * 40.0, rs.rand(20000, 1) * 0.3),
                         axis=1)
    e = 0.1
    got, info = multiscale.process_gpu(torch.from_numpy(pts).cuda(), torch.from_numpy(pts).cuda(),
                                       [e], [4.0 * e], return_info=True)
    want = oracle.process_fast(pts, pts, [e], [4.0 * e])
    assert_features_close(got.cpu().numpy(), want, pts)


# ---- k-nearest-voxel fallback (config 4; build-defined, pinned by the build's own oracle) --------------

@pytest.mark.parametrize("per_scale", [False, True])
def test_knn_fallback_against_oracle(per_scale):
    # a dense core plus a sparse halo of stragglers whose radius neighborhoods are nearly empty
    rs = np.random.RandomState(131)
    core, _ = synth.scene_cloud(20000, extent=6.0, n_poles=4, n_spheres=2, seed=132)
    sparse = rs.rand(1500, 3) * np.array([12.0, 12.0, 4.0]) - np.array([3.0, 3.0, 0.5])
    pts = np.concatenate((core, sparse), axis=0)
    e, r, k = 0.1, 0.3, 8
    dev = torch.from_numpy(pts).cuda()
    got = multiscale.process_gpu(dev, dev, [e], [r], knn_min=k, knn_radius_factor=4.0,
                                 per_scale=per_scale).cpu().numpy()
    want = oracle.one_scale_knn(pts, pts, e, r, k, radius_factor=4.0)
    plain = oracle.one_scale_fast(pts, pts, e, r)
    touched = plain[:, 0] < k
    assert touched.sum() > 500 and (~touched).sum() > 5000
    assert np.array_equal(got[:, 0], want[:, 0])                 # population column is untouched
    assert_features_close(got, want, pts)
    assert np.abs(want[touched] - plain[touched]).max() > 1e-3   # the fallback really changed rows
    # and switching it off again restores the plain result
    off = multiscale.process_gpu(dev, dev, [e], [r]).cpu().numpy()
    assert_features_close(off, plain, pts)


def test_knn_fallback_behind_the_generic_kernel():
    """r/e = 5.5 takes the generic search kernel, which does not write the sparse bits itself (k_knn_mark
    does): same contract, smaller cloud (the oracle's kNN is a cKDTree.query per sparse row)."""
    rs = np.random.RandomState(133)
    core = synth.uniform_cloud(6000, extent=2.0, seed=134)
    sparse = rs.rand(600, 3) * 8.0 - 3.0
    pts = np.concatenate((core, sparse), axis=0)
    e, r, k = 0.1, 0.55, 10
    dev = torch.from_numpy(pts).cuda()
    got = multiscale.process_gpu(dev, dev, [e], [r], knn_min=k, knn_radius_factor=2.0).cpu().numpy()
    want = oracle.one_scale_knn(pts, pts, e, r, k, radius_factor=2.0)
    plain = oracle.one_scale_fast(pts, pts, e, r)
    touched = plain[:, 0] < k
    assert touched.sum() > 100
    assert np.array_equal(got[:, 0], want[:, 0])
    assert_features_close(got, want, pts)


def test_descriptors():
    pts, _ = synth.scene_cloud(20000, extent=8.0, n_poles=5, n_spheres=2, seed=141)
    feats = multiscale.process_single_core(pts, pts, [0.1, 0.2], [0.3, 0.6])
    feats[:5, 2:4] = 0.0                                           # undefined rows
    got = features.descriptors(feats)
    assert got.shape == (len(pts), 6)
    for s in range(2):
        l1, l2 = feats[:, 4 * s + 2], feats[:, 4 * s + 3]
        l3 = np.maximum(1.0 - l1 - l2, 0.0)
        ok = l1 > 0
        want = np.zeros((len(pts), 3))
        want[ok] = np.stack(((l1 - l2)[ok] / l1[ok], (l2 - l3)[ok] / l1[ok], l3[ok] / l1[ok]), axis=1)
        assert np.abs(got[:, 3 * s:3 * s + 3] - want).max() < 1e-14
    assert np.all(got[:5, :3] == 0.0)
    # ground points are planar, pole points linear
    _, labels = synth.scene_cloud(20000, extent=8.0, n_poles=5, n_spheres=2, seed=141)
    d = features.descriptors(multiscale.process_single_core(pts, pts, [0.2], [0.6]))
    assert np.median(d[labels == 0, 1]) > 0.6 and np.median(d[labels == 1, 0]) > 0.5


def test_config3_full_size_properties():
    # config 3 (the benchmark workload): 10M points x 5 scales.  at this size: range invariants on every
    # row, voxel counts against numpy.unique on the host, and an oracle comparison on a spatial crop
    # (the oracle gets the crop plus its halo and the GLOBAL lattice extrema).
    pts, _, edges, radii = synth.make_config("c3_scene_10m")
    dev = torch.from_numpy(pts).cuda()
    full, info = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
    f = full.cpu().numpy()
    for s, e in enumerate(edges):
        n, l1, l2 = f[:, 4 * s], f[:, 4 * s + 2], f[:, 4 * s + 3]
        assert n.min() >= 1 and n.max() <= 123          # lattice points in a ball of radius 3
        ok = n >= 2
        assert np.all(l1[ok] >= 1.0 / 3.0 - 1e-12) and np.all(l1[ok] + l2[ok] <= 1.0 + 1e-12)
        assert np.all(l2[ok] <= l1[ok] + 1e-12) and np.all(l2[ok] >= -1e-12)
        assert np.all(f[:, 4 * s + 1] <= 3.0 * e + e)   # centroid lies within the ball (plus slack)
        lat = oracle.Lattice(pts, e)
        assert info[s].voxels == len(np.unique(lat.coordinate_to_address(pts)))
    lo, hi = pts.min(0), pts.max(0)
    centre = np.array([95.0, 95.0, 1.0])
    inner = np.all(np.abs(pts[:, :2] - centre[:2]) <= 3.0, axis=1)
    outer = np.all(np.abs(pts[:, :2] - centre[:2]) <= 3.0 + 3.2, axis=1)
    rows = np.nonzero(inner)[0]
    assert len(rows) > 5000
    want = oracle.process_fast(pts[rows], pts[outer], edges, radii, bounds=(lo, hi))
    assert_features_close(f[rows], want, pts)
    # every one of the 5e7 point-scales against the plain-C oracle: populations bit-exact, features
    # within the contract
    for s, (e, r) in enumerate(zip(edges, radii)):
        want_s = oracle.one_scale_c(pts, pts, e, r)
        assert_features_close(f[:, 4 * s:4 * s + 4], want_s, pts)


def test_reference_ladder_shape_one_edge_three_radii():
    """the reference's own ladders: one voxel edge, several radii (point_clouds.py:29-35: voxel 0.05, scales
    0.15 / 0.20 / 0.25).  the three scales share one lattice, hence ONE occupancy index (built once, ScaleDev::
    shared), and run on the W = 7, 9 and 11 instances of the table kernel.  every row of every scale against the
    plain-C oracle: populations bit-exact, features within the contract; and against the same scales computed one
    call each (own index each): bit-identical."""
    cfg = synth.CONFIGS["ref_ladder_10m"]
    pts, _, edges, radii = synth.make_config("ref_ladder_10m", n=1_500_000)
    assert edges == cfg["edges"] and len(set(edges)) == 1
    dev = torch.from_numpy(pts).cuda()
    full, info = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
    f = full.cpu().numpy()
    m = len(np.unique(oracle.Lattice(pts, edges[0]).coordinate_to_address(pts)))
    assert [i.voxels for i in info] == [m, m, m]
    for s, (e, r) in enumerate(zip(edges, radii)):
        want_s = oracle.one_scale_c(pts, pts, e, r)
        assert np.array_equal(f[:, 4 * s], want_s[:, 0])
        assert_features_close(f[:, 4 * s:4 * s + 4], want_s, pts)
        alone = multiscale.process_gpu(dev, dev, [e], [r])
        assert torch.equal(alone, full[:, 4 * s:4 * s + 4])
    # mixed ladders: equal edges need not be adjacent, and a finer scale may follow
    edges2, radii2 = [0.1, 0.05, 0.1, 0.05], [0.3, 0.2, 0.5, 0.15]
    sub = dev[:300000].contiguous()
    mixed, info2 = multiscale.process_gpu(sub, sub, edges2, radii2, return_info=True)
    assert info2[0].voxels == info2[2].voxels and info2[1].voxels == info2[3].voxels
    for s, (e, r) in enumerate(zip(edges2, radii2)):
        assert torch.equal(multiscale.process_gpu(sub, sub, [e], [r]), mixed[:, 4 * s:4 * s + 4])
    host = sub.cpu().numpy()
    assert_features_close(mixed.cpu().numpy(), oracle.process_c(host, host, edges2, radii2), host)


def test_index_timeout_is_sticky_and_never_silent(tmp_path):
    # a diagnostic build of the library (make -C nimrud_amd/csrc diag) whose index builder behaves as if
    # its bounded wait for a leaf number had run out in the first block.  the host API must refuse to hand
    # back the incomplete features, and the context must keep failing until the error is cleared.
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(repo, "nimrud_amd", "diag", "libnimrud_hip_timeout.so")
    assert os.path.exists(lib), "diagnostic build missing: make -C nimrud_amd/csrc diag"
    script = tmp_path / "timeout_probe.py"
    script.write_text("""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from nimrud_amd import synth, _ffi, device
from nimrud_amd.minimal import multiscale
pts = synth.uniform_cloud(20000, extent=4.0, seed=5)
try:
    multiscale.process_single_core(pts, pts, [0.1, 0.2], [0.3, 0.6])
    print("RESULT silent")
    raise SystemExit(0)
except _ffi.NimrudHipError as err:
    print("RESULT raised:", err)
rt = device.get_runtime()
dev = torch.from_numpy(pts).cuda()
try:
    multiscale.process_gpu(dev, dev, [0.1], [0.3])
    print("RESULT second call went through")
except _ffi.NimrudHipError:
    print("RESULT sticky")
rt.clear_error()
out = multiscale.process_gpu(dev, dev, [0.1], [0.3])      # enqueued: the context is usable again
torch.cuda.synchronize()
try:
    rt.check_async(wait=True)
    print("RESULT no failure after clear")
except _ffi.NimrudHipError:
    print("RESULT failed again after clear")
""" % repo)
    env = dict(os.environ, NIMRUD_HIP_LIBRARY=lib)
    run = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True,
                         timeout=300)
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("RESULT")]
    assert run.returncode == 0, run.stderr[-2000:]
    assert lines[0].startswith("RESULT raised:") and "timed out" in lines[0], run.stdout
    assert lines[1] == "RESULT sticky", run.stdout
    assert lines[2] == "RESULT failed again after clear", run.stdout     # the build fails every time


def test_context_for_another_device_leaves_the_current_device_alone():
    # nm_create used to hipSetDevice() and never restore it.  on a one-GPU box the only checkable part is
    # that creating, using and destroying a context changes nothing the process can see.
    import ctypes
    from nimrud_amd import _ffi
    lib = _ffi.load()
    before = torch.cuda.current_device()
    ctx = ctypes.c_void_p()
    assert lib.nm_create(ctypes.byref(ctx), torch.cuda.device_count() - 1) == 0
    assert torch.cuda.current_device() == before
    assert lib.nm_check(ctx, 1) == 0
    lib.nm_destroy(ctx)
    assert torch.cuda.current_device() == before
    assert lib.nm_create(ctypes.byref(ctx), torch.cuda.device_count()) != 0     # no such device


def test_out_tensor_is_validated():
    pts = synth.uniform_cloud(3000, extent=2.0, seed=171)
    dev = torch.from_numpy(pts).cuda()
    good = torch.empty((3000, 8), dtype=torch.float64, device="cuda")
    assert multiscale.process_gpu(dev, dev, [0.2, 0.4], [0.6, 1.2], out=good) is good
    for bad in (torch.empty((3000, 8), dtype=torch.float32, device="cuda"),
                torch.empty((3000, 8), dtype=torch.float64),
                torch.empty((2999, 8), dtype=torch.float64, device="cuda"),
                torch.empty((3000, 7), dtype=torch.float64, device="cuda"),
                torch.empty((8, 3000), dtype=torch.float64, device="cuda").t()):
        with pytest.raises(ValueError):
            multiscale.process_gpu(dev, dev, [0.2, 0.4], [0.6, 1.2], out=bad)
    with pytest.raises(ValueError):
        features.neighborhood_features(pts[:10], np.array([0, 4, 12]), pts[:2])
    with pytest.raises(ValueError):
        features.neighborhood_features(pts[:10], np.array([0, 6, 4]), pts[:2])


def test_config5_forest_fixture(golden):
    # the config 5 classifier on the fixture's own inputs: probabilities to 1e-15 of sklearn's, labels equal
    g = golden("g6_forest_c5.npz")
    model = classification.ForestModel.from_arrays(g)
    proba = model.predict_proba(g["x"])
    assert np.abs(proba - g["proba"]).max() <= 1e-15
    assert np.array_equal(model.predict(g["x"]), g["label"])
    # a feature matrix with extra columns and a row stride, as a slice of a wider tensor
    wide = torch.zeros((len(g["x"]), 27), dtype=torch.float64, device="cuda")
    wide[:, :20] = torch.from_numpy(g["x"]).cuda()
    assert np.array_equal(model.predict(wide[:, :20]).cpu().numpy(), g["label"])


def test_config5_full_size_end_to_end(golden):
    # BASELINE config 5 as stated: 10 M points x 5 scales with the forest evaluated behind the last
    # scale.  the fixture's evaluation rows tie the features of the full run to the oracle's (and through
    # them sklearn's labels to ours); fused and stand-alone evaluation must agree on every row.
    g = golden("g6_forest_c5.npz")
    pts, labels, edges, radii = synth.make_config("c5_scene_10m_rf")
    assert np.array_equal(np.asarray(edges), g["edges"]) and np.array_equal(np.asarray(radii), g["radii"])
    model = classification.ForestModel.from_arrays(g)
    dev = torch.from_numpy(pts).cuda()
    label, feats = classification.classify_cloud(dev, edges, radii, model)
    rows = g["eval_rows"]
    f_eval = feats[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert_features_close(f_eval, g["x"], pts)
    got = label.cpu().numpy()
    assert np.array_equal(got[rows], g["truth"]) or (got[rows] == g["truth"]).mean() > 0.95
    # same inputs -> same labels: the oracle's forest on the GPU's features of those rows
    omodel = {k: g[k] for k in ("left", "right", "feature", "threshold", "value", "roots", "classes")}
    assert np.array_equal(model.classes[got[rows]], oracle.forest_predict(omodel, f_eval))
    # sklearn's labels on the oracle's features: a row can only differ where a feature rounds across a
    # split threshold after the fp32 cast
    assert (model.classes[got[rows]] != g["label"]).sum() <= 2
    # the forest ran inside the search kernel, behind each row's last scale.  stand-alone evaluation of the
    # finished matrix (nm_forest_eval): every one of the 1e7 labels identical
    label2 = model.predict(feats)
    assert torch.equal(torch.as_tensor(model.classes, device="cuda")[label.to(torch.int64)], label2)
    proba = model.predict_proba(feats[:100000])
    assert torch.equal(proba.argmax(1).to(torch.int32), label[:100000])
    label3, _ = classification.classify_cloud(dev, edges, radii, model, fused=False)
    assert torch.equal(label, label3)
    assert (got == labels).mean() > 0.95


def test_fused_moments_against_golden_neighbor_lists(golden):
    # the fused kernel's neighbor SET, pinned bit-exactly: population, centroid and covariance of the first
    # 256 queries per scale recomputed in exact integer arithmetic from the REFERENCE's neighbor lists
    # (tests/golden/g*_: s%d_nbr_index captured from cKDTree.query_ball_tree) must reproduce what the
    # kernel derives from its own integer moments - the covariance output is n*S2 - S1*S1^T scaled, so a
    # single wrong neighbor changes it far beyond the 1e-12 compared here.
    for name in PIPELINE_FIXTURES:
        g = golden(name)
        pts = g["points"]
        edges, radii = list(g["edges"]), list(g["radii"])
        dev = torch.from_numpy(pts).cuda()
        feats, cov = multiscale.process_gpu_covariance(dev, dev, edges, radii)
        feats, cov = feats.cpu().numpy(), cov.cpu().numpy()
        for s, e in enumerate(edges):
            lat = oracle.Lattice(pts, e)
            addr = g["s%d_addresses" % s]
            cells = lat.address_to_cells(addr).astype(np.int64)       # (M, 3) integer lattice sites
            off, idx = g["s%d_nbr_offsets" % s], g["s%d_nbr_index" % s]
            for q in range(len(off) - 1):
                nb = cells[idx[off[q]:off[q + 1]]]
                n = len(nb)
                assert feats[q, 4 * s] == n
                if n < 2:
                    continue
                s1 = nb.sum(0)
                s2 = nb.T @ nb
                scatter = n * s2 - np.outer(s1, s1)                   # exact integers
                want = scatter[np.triu_indices(3)] * (e * e) / (n * (n - 1.0))
                got = cov[q, 6 * s:6 * s + 6]
                assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (name, s, q)


def test_config4_lidar_power_law_with_knn_fallback():
    # config 4 at reduced size (2M of 50M points; same generator, so the far field is 25x sparser than
    # at full size and the kNN fallback has plenty to do): power-law density, 5 scales, k_min = 8.
    # checked: range invariants everywhere, an oracle comparison on a dense crop near the scanner and
    # on a sparse crop far out (the oracle gets crop + halo and the global lattice extrema).
    pts, _, edges, radii = synth.make_config("c4_lidar_50m", n=2_000_000)
    k = synth.CONFIGS["c4_lidar_50m"]["knn_min"]
    dev = torch.from_numpy(pts).cuda()
    plain, info = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
    full = multiscale.process_gpu(dev, dev, edges, radii, knn_min=k, knn_radius_factor=3.0)
    p, f = plain.cpu().numpy(), full.cpu().numpy()
    assert np.array_equal(p[:, ::4], f[:, ::4])
    sparse_rows = p[:, 0] < k
    assert sparse_rows.mean() > 0.02                       # the far field really is sparse
    dense_rows = ~sparse_rows
    assert np.array_equal(p[dense_rows, :4], f[dense_rows, :4])
    for s in range(len(edges)):
        n, l1, l2 = f[:, 4 * s], f[:, 4 * s + 2], f[:, 4 * s + 3]
        assert n.min() >= 1 and n.max() <= 123
        assert np.all(l1 <= 1.0 + 1e-12) and np.all(l2 <= l1 + 1e-12) and np.all(l2 >= -1e-12)
    lo, hi = pts.min(0), pts.max(0)
    for centre, half in (((3.0, 0.0), 1.5), ((90.0, 40.0), 12.0)):
        d = np.abs(pts[:, :2] - np.asarray(centre))
        inner = np.all(d <= half, axis=1)
        outer = np.all(d <= half + 3.0 * 2.4 + 1.0, axis=1)
        rows = np.nonzero(inner)[0][:6000]
        assert len(rows) > 200
        for s in (0, 2, 4):
            lat_bounds = (lo, hi)
            want = oracle.one_scale_fast(pts[rows], pts[outer], edges[s], radii[s], bounds=lat_bounds)
            assert_features_close(p[rows, 4 * s:4 * s + 4], want, pts)


def test_config4_10m_every_row_against_the_c_oracle():
    # config 4 at 10 M of its 50 M points (the full size: the test below): power-law density, 5 scales.  every one of the 5e7 point-scales against the plain-C
    # oracle - populations bit-exact, features within the contract - and the kNN fallback (k_min = 8; it has no
    # reference counterpart: parity unpinned by the reference, pinned by the build's oracle) against
    # oracle.one_scale_knn on a sparse crop far from the scanner.
    pts, _, edges, radii = synth.make_config("c4_lidar_50m", n=10_000_000)
    k = synth.CONFIGS["c4_lidar_50m"]["knn_min"]
    dev = torch.from_numpy(pts).cuda()
    plain = multiscale.process_gpu(dev, dev, edges, radii).cpu().numpy()
    for s, (e, r) in enumerate(zip(edges, radii)):
        want_s = oracle.one_scale_c(pts, pts, e, r)
        assert_features_close(plain[:, 4 * s:4 * s + 4], want_s, pts)
    full = multiscale.process_gpu(dev, dev, edges, radii, knn_min=k, knn_radius_factor=3.0).cpu().numpy()
    assert np.array_equal(plain[:, ::4], full[:, ::4])
    dense = plain[:, 0] >= k
    assert np.array_equal(plain[dense, :4], full[dense, :4]) and (~dense).mean() > 0.01
    lo, hi = pts.min(0), pts.max(0)
    d = np.abs(pts[:, :2] - np.array([100.0, 60.0]))
    inner = np.all(d <= 6.0, axis=1)
    outer = np.all(d <= 6.0 + 3.0 * 2.4 + 1.0, axis=1)
    rows = np.nonzero(inner)[0][:4000]
    assert len(rows) > 300
    touched = 0
    for s in (0, 1, 3):
        # the oracle gets the crop plus a halo wider than the fallback's reach (3 r) and the GLOBAL extrema,
        # so that its lattice is the cloud's
        want = oracle.one_scale_knn(pts[rows], pts[outer], edges[s], radii[s], k, 3.0, bounds=(lo, hi))
        assert_features_close(full[rows, 4 * s:4 * s + 4], want, pts)
        touched += int((want[:, 0] < k).sum())
    assert touched > 200


def test_config4_full_size_50m_every_row_against_the_c_oracle():
    """config 4 at its stated size: 50 M points (power-law density), 5 scales, two of them on the hash form of the
    occupancy index, 79 GB of workspace.  every one of the 2.5e8 point-scales against the plain-C oracle:
    populations bit-exact, features within the contract.  (about 75 s on the GPU box: 10 s of cloud, 3 s of GPU
    including the 8 GB download, 60 s of oracle on the box's cpu share.)"""
    pts, _, edges, radii = synth.make_config("c4_lidar_50m")
    assert len(pts) == 50_000_000
    dev = torch.from_numpy(pts).cuda()
    got, info = multiscale.process_gpu(dev, dev, edges, radii, return_info=True)
    got = got.cpu().numpy()
    del dev
    _device_runtime().release_workspace()
    torch.cuda.empty_cache()
    lo, hi = pts.min(0), pts.max(0)
    for s, (e, r) in enumerate(zip(edges, radii)):
        want, m = oracle.one_scale_c(pts, pts, e, r, bounds=(lo, hi), return_voxel_count=True)
        assert info[s].voxels == m
        assert_features_close(got[:, 4 * s:4 * s + 4], want, pts)
        del want


def test_classify_cloud_end_to_end():
    # config 5 in miniature: features -> balanced split -> sklearn fit -> GPU forest -> confusion
    pts, labels = synth.scene_cloud(60000, extent=14.0, n_poles=12, n_spheres=4, seed=151)
    dev = torch.from_numpy(pts).cuda()
    edges, radii = [0.1, 0.2, 0.4], [0.3, 0.6, 1.2]
    feats = multiscale.process_gpu(dev, dev, edges, radii)
    tr, va = classification.balanced_split(labels, seed=0)
    model, clf = classification.train_forest(feats[torch.from_numpy(tr).cuda()], labels[tr],
                                             n_estimators=16, max_depth=10, n_jobs=4)
    pred, feats2 = classification.classify_cloud(dev, edges, radii, model)
    assert torch.equal(feats, feats2)
    got = model.classes[pred.cpu().numpy()]
    assert np.array_equal(got[va], clf.predict(feats[torch.from_numpy(va).cuda()].cpu().numpy()))
    assert np.array_equal(model.predict(feats).cpu().numpy(), got)
    conf = classification.confusion_matrix(pred[torch.from_numpy(va).cuda()],
                                           torch.from_numpy(labels[va]).cuda(), n_classes=3)
    assert conf.sum() == len(va) and np.trace(conf) / conf.sum() > 0.9
    # keyword arguments of process_gpu pass through; return_info comes back as the last element, on both paths
    for fused in (True, False):
        pred3, feats3, info = classification.classify_cloud(dev, edges, radii, model, fused=fused,
                                                            return_info=True)
        assert torch.equal(pred3, pred) and torch.equal(feats3, feats) and len(info) == 3
        assert [i.voxels for i in info] == [len(oracle.Lattice(pts, e).unique_addresses(pts)) for e in edges]


def test_unusable_lattice_never_looks_like_features_and_never_poisons_the_context():
    """process_gpu only enqueues: a lattice the device cannot address is reported at the next synchronisation
    point.  until then the columns of that scale hold NaN (labels -1 behind a fused
    forest), never uninitialised memory; and whichever call meets the report first - nm_check or the entry of
    the next library call - raises the reference's ValueError (geometry.py:59-60) ONCE and leaves the context
    usable."""
    pts = synth.uniform_cloud(5000, extent=2.0, seed=4242)
    dev = torch.from_numpy(pts).cuda()
    out = torch.full((5000, 8), 7.0, dtype=torch.float64, device="cuda")
    multiscale.process_gpu(dev, dev, [0.1, 1e-9], [0.3, 3e-9], out=out)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    assert np.isnan(host[:, 4:]).all()
    assert_features_close(host[:, :4], oracle.process_fast(pts, pts, [0.1], [0.3]), pts)
    # no check_async: the next call into the library meets the report at its entry
    with pytest.raises(ValueError, match="too small"):
        multiscale.process_gpu(dev, dev, [0.1], [0.3])
    got = multiscale.process_gpu(dev, dev, [0.1], [0.3]).cpu().numpy()       # once: the context carries on
    assert_features_close(got, host[:, :4], pts)
    _device_runtime().check_async(wait=True)


def test_verbose_mode_prints_like_the_reference(capsys):
    # multiscale.py:47-65: per-scale and total timing lines
    pts = synth.uniform_cloud(3000, extent=2.0, seed=161)
    quiet = multiscale.process_single_core(pts, pts, [0.2, 0.4], [0.6, 1.2])
    loud = multiscale.process_single_core(pts, pts, [0.2, 0.4], [0.6, 1.2], verbose=True)
    assert np.array_equal(quiet, loud)
    text = capsys.readouterr().out
    assert text.count("this scale took") == 2 and "final rate of" in text
    assert "querying 3000 points against a search space of" in text


def test_flexcloud_on_the_device():
    # the container with device-resident assets, through the pipeline conveniences
    from nimrud_amd.utils import point_clouds
    pts, labels = synth.scene_cloud(30000, extent=10.0, n_poles=6, n_spheres=2, seed=171)
    pts = pts + np.array([4.0e5, 5.1e6, 300.0])                      # UTM-like: the corner shift helps
    cloud = point_clouds.FlexCloud(pts, device="cuda")
    feats = cloud.add_multiscale_features([0.1, 0.2], [0.3, 0.6])
    assert feats.is_cuda and feats.shape == (30000, 8)
    assert cloud.assets["geometry_mso"]["meta"] == {"voxel": [0.1, 0.2], "scales": [0.3, 0.6]}
    want = oracle.process_fast(pts - pts[0], pts - pts[0], [0.1, 0.2], [0.3, 0.6])
    assert_features_close(feats.cpu().numpy(), want, pts - pts[0])
    cloud.add_asset(labels[:20000], np.arange(20000), "known_label")
    idx, block = cloud.intersection(["geometry_mso", "known_label"])
    assert idx.is_cuda and block.shape == (20000, 9)
    model, clf = classification.train_forest(block[:, :8], block[:, 8].to(torch.int64), n_estimators=8,
                                             max_depth=8, n_jobs=4)
    predicted = cloud.add_labels_from(model, "geometry_mso")
    assert (predicted.cpu().numpy()[20000:] == labels[20000:]).mean() > 0.9
    host = point_clouds.FlexCloud(pts)
    assert np.array_equal(host.take(), pts + 0.0)


# ---- the spatial order (nm_order.hip: the radix sort of our own) ------------------------------------------------

def _compact_zorder_key(cells, widths, key_bits=30):
    """numpy restatement of nm_order_key: bit b of every axis that has a bit b, lowest bits first (the
    order of a 3-way Morton interleave with the always-zero bits squeezed out), cut to `key_bits` from the top"""
    key = np.zeros(len(cells), dtype=np.uint64)
    at = 0
    for b in range(int(max(widths))):
        for a in range(3):
            if b < widths[a]:
                key |= ((cells[:, a].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(at)
                at += 1
    total = int(sum(widths))
    return (key >> np.uint64(max(total - key_bits, 0))).astype(np.uint32)


@pytest.mark.parametrize("n,edge,kind", [(1, 0.1, "uniform"), (300, 0.1, "uniform"), (8192, 0.05, "uniform"),
                                         (8193, 0.05, "scene"), (250_000, 0.25, "uniform"),
                                         (2_000_003, 0.05, "scene"), (4_500_000, 0.05, "scene")])
def test_spatial_order_is_a_sorted_permutation(n, edge, kind):
    """nm_spatial_order: the permutation is one, the coordinates are carried along bit for bit, the keys the
    library reports are the compact Z-order keys of the cells (restated in numpy above) and they come out
    ascending.  the sort's top digit is exact by construction; inside it the order rests on the lane order of
    LDS atomics (nm_order.hip): the test reports the inversions it finds and tolerates a per-mille of them."""
    import ctypes
    from nimrud_amd import device as nm_device
    rt = _device_runtime()
    if kind == "uniform":
        pts = np.random.RandomState(n).rand(n, 3) * np.array([10.0, 7.0, 3.0]) + np.array([-2.0, 40.0, 1.0])
    else:
        pts, _ = synth.scene_cloud(n, extent=30.0 if n < 100000 else 80.0, n_poles=20, n_spheres=6, seed=n % 97)
    pts = np.ascontiguousarray(pts)
    lo = pts.min(0) if n > 1 else pts[0] - 1.0
    hi = pts.max(0) if n > 1 else pts[0] + 1.0
    mc, _, widths, _ = geometry.lattice_parameters(lo, hi, edge)
    lat = geometry.make_nm_lattice(mc, edge, widths)
    dev = torch.from_numpy(pts).cuda()
    order = torch.empty(n, dtype=torch.int32, device="cuda")
    sxyz = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    nbytes = rt.lib.nm_spatial_order_workspace_bytes(n)
    work = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
    rt.check(rt.lib.nm_spatial_order(rt.ctx, nm_device.ptr(dev), n, 3, ctypes.byref(lat), nm_device.ptr(order),
                                     nm_device.ptr(sxyz), nm_device.ptr(keys), nm_device.ptr(work), work.numel(),
                                     rt.stream()))
    torch.cuda.synchronize()
    order = order.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert np.array_equal(np.sort(order), np.arange(n))
    assert np.array_equal(sxyz.cpu().numpy(), pts[order])
    got = keys.cpu().numpy().view(np.uint32)
    cells = np.floor((pts[order] - mc) / edge).astype(np.int64)
    # (clouds of up to 2^22 points keep 20 key bits and are sorted in two passes, larger ones 30 and three:
    # nm_order_plan)
    key_bits = 20 if n <= (1 << 22) else 30
    want = _compact_zorder_key(cells, widths, key_bits)
    # the library takes the cell from a multiplication by fl(1/e): a point within rounding of a cell face may
    # carry its neighbour's key
    assert np.mean(got != want) < 1e-4
    total = min(int(sum(widths)), key_bits)
    passes = 2 if total <= 20 else 3               # a short key is sorted in two passes
    bpp = max((total + passes - 1) // passes, 1)
    top = got >> np.uint32((passes - 1) * bpp)
    assert np.all(np.diff(top.astype(np.int64)) >= 0)
    inversions = int(np.sum(np.diff(got.astype(np.int64)) < 0))
    print("spatial order: n = %d, key bits %d, %d adjacent inversions" % (n, total, inversions))
    assert inversions <= max(n // 1000, 0)
