"""
CPU tests (no GPU): the C-ABI library loads and exports every symbol include/nimrud_hip.h declares,
the ctypes table matches the header, and the host-side logic (lattice parameters, batcher, synthetic
clouds, forest flattening, the loud failure without a GPU) behaves.
"""

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO
from nimrud_amd import _ffi, synth
from nimrud_amd.utils import generic, geometry
from oracle import nimrud_oracle as oracle

HEADER = os.path.join(REPO, "include", "nimrud_hip.h")


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_ffi.LIBRARY_PATH)
    names = header_functions()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), "libnimrud_hip.so does not export %s" % name


def test_binding_table_matches_header():
    assert sorted(_ffi.SIGNATURES) == header_functions()
    lib = _ffi.load()
    assert lib.nm_abi_version() == _ffi.ABI_VERSION


def test_struct_layouts_match_header():
    # struct nm_lattice {double[3]; double; int32[3]; int32[2]} -> 32 + 20 (+4 pad) = 56 bytes
    assert ctypes.sizeof(_ffi.NmLattice) == 56
    assert _ffi.NmLattice.edge.offset == 24 and _ffi.NmLattice.widths.offset == 32
    assert _ffi.NmLattice.shifts.offset == 44
    assert ctypes.sizeof(_ffi.NmForest) == 10 * 8 + 6 * 4
    assert _ffi.NmForest.d_packed8.offset == 9 * 8 + 6 * 4


def test_workspace_queries_need_no_gpu():
    lib = _ffi.load()
    lat = geometry.make_nm_lattice([0.0, 0.0, 0.0], 0.1, [10, 10, 6])
    small = lib.nm_scale_workspace_bytes(1000, 1000, ctypes.byref(lat))
    big = lib.nm_scale_workspace_bytes(100000, 100000, ctypes.byref(lat))
    assert 0 < small < big
    assert lib.nm_voxelize_workspace_bytes(1000) < lib.nm_voxelize_workspace_bytes(1000000)
    # the ladder's workspace depends on the point counts and the number of scales only
    assert 0 < lib.nm_ladder_workspace_bytes(1000, 1000, 2) < lib.nm_ladder_workspace_bytes(1000, 1000, 5)
    assert lib.nm_ladder_workspace_bytes(1000, 1000, 33) == 0
    assert lib.nm_halo_workspace_bytes(0, 8) < lib.nm_halo_workspace_bytes(100000, 8)


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from nimrud_amd.minimal import multiscale
    pts = synth.uniform_cloud(100)
    with pytest.raises(RuntimeError):
        multiscale.process_single_core(pts, pts, [0.25], [0.75])
    with pytest.raises(RuntimeError):
        geometry.VoxelFilter(pts, 0.25)


def test_length_mismatch_is_assertion_error():
    # multiscale.py:32-33
    from nimrud_amd.minimal import multiscale
    pts = synth.uniform_cloud(100)
    with pytest.raises(AssertionError):
        multiscale.process_single_core(pts, pts, [0.25, 0.5], [0.75])


@pytest.mark.parametrize("name", ["g1_uniform.npz", "g2_scene.npz", "g3_offset.npz"])
def test_lattice_parameters_match_reference(golden, name):
    g = golden(name)
    pts = g["points"]
    for s, e in enumerate(g["edges"]):
        mc, xc, widths, shifts = geometry.lattice_parameters(pts.min(0), pts.max(0), e)
        assert np.array_equal(mc, g["s%d_min_corner" % s])
        assert np.array_equal(widths, g["s%d_widths" % s])
        assert np.array_equal(shifts, g["s%d_shifts" % s])
        lat = oracle.Lattice(pts, e)
        assert np.array_equal(xc, lat.maximum_corner)


def test_lattice_parameters_known_answers():
    # geometry_tests.py:84-138
    mc, xc, widths, shifts = geometry.lattice_parameters([0, 0, 0], [100, 100, 100], 0.001)
    assert list(widths) == [17, 17, 17] and list(shifts) == [17, 34]
    with pytest.raises(ValueError):
        geometry.lattice_parameters([0, 0, 0], [100, 100, 100], 0.00001)
    with pytest.raises(ValueError):
        geometry.lattice_parameters([0, 0], [100, 100], 0.00000001)


def test_batcher():
    # generic.py:8-26
    arr = np.arange(10)
    assert [list(c) for c in generic.batcher(arr, 4)] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    assert list(generic.batcher([1, 2, 3], 2)) == [[1, 2], [3]]
    assert list(generic.batcher(iter(range(5)), 2)) == [[0, 1], [2, 3], [4]]
    assert list(generic.batcher(iter(()), 2)) == []
    assert list(generic.batcher((1, 2, 3), 2)) == [[1, 2], [3]]       # not sliced: like the reference
    import torch
    t = torch.arange(5)
    assert [c.tolist() for c in generic.batcher(t, 2)] == [[0, 1], [2, 3], [4]]
    assert next(iter(generic.batcher(t, 2))).data_ptr() == t.data_ptr()    # views, no copies


def test_synthetic_clouds_are_deterministic_and_fp32_representable():
    a, la = synth.scene_cloud(5000, extent=5.0, n_poles=3, n_spheres=2, seed=3)
    b, lb = synth.scene_cloud(5000, extent=5.0, n_poles=3, n_spheres=2, seed=3)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    assert np.array_equal(a, a.astype(np.float32).astype(np.float64))
    assert set(np.unique(la)) == {0, 1, 2}
    pts, _, edges, radii = synth.make_config("c1_uniform_100k", n=1000)
    assert pts.shape == (1000, 3) and edges == [0.25] and radii == [0.75]
    order = synth.morton_sort(a, 0.8)
    assert sorted(order) == list(range(len(a)))


def test_forest_flattening_matches_sklearn(golden):
    from sklearn.ensemble import RandomForestClassifier
    from nimrud_amd.minimal.classification import ForestModel
    rs = np.random.RandomState(0)
    x = rs.rand(400, 6)
    y = (x[:, 0] + x[:, 3] > 1).astype(int) + (x[:, 5] > 0.8)
    clf = RandomForestClassifier(n_estimators=5, max_depth=6, random_state=0).fit(x, y)
    model = ForestModel.flatten_sklearn(clf)
    xe = rs.rand(300, 6)
    assert np.abs(oracle.forest_predict_proba(model, xe) - clf.predict_proba(xe)).max() < 1e-15
    assert np.array_equal(oracle.forest_predict(model, xe), clf.predict(xe))


def test_training_glue_host_side():
    # apc.py:895-917 balanced sampling, ml.py:521-552 mc_confusion
    from nimrud_amd.minimal import classification
    labels = np.array([0] * 50 + [1] * 21 + [2] * 200)
    tr, va = classification.balanced_split(labels, seed=1)
    assert len(tr) == len(va) == 3 * 10 and not set(tr) & set(va)
    for c in range(3):
        assert (labels[tr] == c).sum() == 10 and (labels[va] == c).sum() == 10
    lies = np.array([0, 1, 1, 2, 2, 2, 0])
    truth = np.array([0, 1, 2, 2, 2, 0, 0])
    conf = classification.confusion_matrix(lies, truth)
    want = np.zeros((3, 3))
    for row in range(3):
        for col in range(3):
            want[row, col] = ((lies == row) * (truth == col)).sum()
    assert np.array_equal(conf, want)


def test_forest_node_packing_is_exact_on_the_config5_fixture():
    """host logic, no GPU: the 8-byte node records of the fused classifier (ForestModel.pack_nodes8) - fp32
    thresholds rounded DOWN, compared in fp32 - walk to the same leaves as sklearn's fp32-cast features against
    fp64 thresholds: probabilities of the 4 096 fixture rows to 1e-15, labels equal."""
    import os
    from nimrud_amd.minimal.classification import ForestModel
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_forest_c5.npz"))
    rec, leaf, roots = ForestModel.pack_nodes(g["left"], g["right"], g["feature"], g["threshold"], g["value"],
                                              g["roots"])
    p8 = ForestModel.pack_nodes8(rec, 20)
    assert p8 is not None and p8.dtype.itemsize == 8
    # the rounding is downwards and tight
    internal = rec["left"] >= 0
    t32, t64 = p8["threshold"][internal].astype(np.float64), rec["threshold"][internal]
    assert np.all(t32 <= t64) and np.all(np.nextafter(p8["threshold"][internal], np.float32(np.inf)) > t64)
    x = g["x"].astype(np.float32)
    n = len(x)
    proba = np.zeros((n, leaf.shape[1]))
    rows = np.arange(n)
    packed, thr = p8["packed"], p8["threshold"]
    for r in roots:
        node = np.full(n, r, dtype=np.int64)
        while True:
            pk = packed[node]
            active = (pk >> 31) == 0
            if not active.any():
                break
            f = ((pk[active] >> 8) & 31).astype(np.int64)
            left = (pk[active] >> 13).astype(np.int64)
            go_left = x[rows[active], f] <= thr[node[active]]          # fp32 against fp32
            node[active] = left + np.where(go_left, 0, 1)
        proba += leaf[((packed[node] >> 13) & 0x3FFFF).astype(np.int64)]
    proba /= len(roots)
    assert np.abs(proba - g["proba"]).max() <= 1e-15
    assert np.array_equal(proba.argmax(1), g["label"])
    assert ForestModel.pack_nodes8(rec, 33) is None             # more features than the 5-bit field holds


def test_halo_plan_from_count_matrix():
    """nm_halo_plan_from_matrix: the host half of nm_halo_exchange's step 3 (offsets, totals, the verdict every
    rank must share), on synthetic count matrices for 2 and 3 ranks - ordering of the receive offsets against
    the senders' segments, one rank short of room, one rank unwell.  no GPU, no communicator."""
    import ctypes
    import numpy as np
    from nimrud_amd import _ffi
    lib = _ffi.load()

    def plan(matrix, rank):
        m = np.ascontiguousarray(matrix, dtype=np.int64)
        world = m.shape[0]
        send_off = (ctypes.c_int64 * world)()
        recv_off = (ctypes.c_int64 * world)()
        sent, recv, culprit = ctypes.c_int64(-1), ctypes.c_int64(-1), ctypes.c_int32(-7)
        rc = lib.nm_halo_plan_from_matrix(m.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), world, rank,
                                          send_off, recv_off, ctypes.byref(sent), ctypes.byref(recv),
                                          ctypes.byref(culprit))
        return rc, list(send_off), list(recv_off), sent.value, recv.value, culprit.value

    rs = np.random.RandomState(5)
    for world in (2, 3):
        pairs = rs.randint(0, 50, size=(world, world))
        np.fill_diagonal(pairs, 0)
        cap = np.full((world, 2), 1000)
        matrix = np.hstack([pairs, cap, np.zeros((world, 1), dtype=np.int64)])
        for rank in range(world):
            rc, send_off, recv_off, sent, recv, culprit = plan(matrix, rank)
            assert rc == _ffi.NM_OK and culprit == -1
            assert send_off == list(np.concatenate([[0], np.cumsum(pairs[rank])[:-1]]))
            assert recv_off == list(np.concatenate([[0], np.cumsum(pairs[:, rank])[:-1]]))     # in rank order
            assert sent == pairs[rank].sum() and recv == pairs[:, rank].sum()
        # exactly one rank is short of room (receive side): EVERY rank gets the same verdict and culprit,
        # and still learns its own totals (that is what it grows its buffers by)
        short = world - 1
        tight = matrix.copy()
        tight[short, world + 1] = pairs[:, short].sum() - 1
        verdicts = [plan(tight, rank) for rank in range(world)]
        assert all(v[0] == _ffi.NM_ERR_WORKSPACE and v[5] == short for v in verdicts)
        assert [v[3] for v in verdicts] == list(pairs.sum(1)) and [v[4] for v in verdicts] == list(pairs.sum(0))
        # send side
        tight = matrix.copy()
        tight[0, world] = pairs[0].sum() - 1
        assert all(plan(tight, rank)[0] == _ffi.NM_ERR_WORKSPACE and plan(tight, rank)[5] == 0
                   for rank in range(world))
        # one rank unwell (a sticky failure of an earlier call): every rank returns ITS status, before any
        # capacity question
        sick = matrix.copy()
        sick[1, world + 2] = _ffi.NM_ERR_HIP
        sick[0, world] = 0
        assert all(plan(sick, rank)[0] == _ffi.NM_ERR_HIP and plan(sick, rank)[5] == 1 for rank in range(world))
    # an empty tile: zero row and zero column
    matrix = np.array([[0, 0, 0, 10, 10, 0], [0, 0, 5, 10, 10, 0], [0, 3, 0, 10, 10, 0]])
    rc, send_off, recv_off, sent, recv, _ = plan(matrix, 0)
    assert rc == _ffi.NM_OK and sent == 0 and recv == 0
    assert lib.nm_halo_plan_from_matrix(None, 2, 0, None, None, None, None, None) == _ffi.NM_ERR_INVALID
