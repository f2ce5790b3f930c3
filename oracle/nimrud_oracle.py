"""
CPU ORACLE for the nimrud multiscale neighborhood-feature hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; `nimrud_amd/` never does.

It restates, in numpy/scipy, the algorithm of the reference `nimrud/minimal` path (all citations are
relative to the reference checkout, `/root/reference/`):

  * lattice definition + 64-bit packed voxel address     nimrud/utils/geometry.py:23-79, 103-116
  * unique occupied voxels and their centres             nimrud/utils/geometry.py:120-154
  * radius search of voxel centres around query points   nimrud/minimal/multiscale.py:87-103
  * population / centroid distance / PCA eigen-features  nimrud/minimal/features.py:14-57
  * per-scale assembly and the scale loop                nimrud/minimal/multiscale.py:27-123
  * random-forest evaluation (classifier slot)           nimrud/prototypes/apc.py:1463,1022,1034

Third-party arithmetic on the reference path that is NOT in /root/reference (versions are unpinned by
the reference; these are the versions in the authoring container, which are the de-facto pin):
  scipy 1.15.3  `scipy.spatial.cKDTree.query_ball_tree`  (inclusive Euclidean ball, fp64)
  numpy 2.2.6   `numpy.cov` (ddof=1), `numpy.unique`, `numpy.linalg.eigvalsh` (LAPACK dsyevd)
  sklearn 1.7.2 `RandomForestClassifier.predict_proba`   (mean of per-tree leaf distributions)

PINNING: the reference has no tests for multiscale.py/features.py.  This oracle is pinned (a) by the
reference's own VoxelFilter known-answer tests (nimrud/utils/tests/geometry_tests.py:84-279,
restated in tests/test_oracle.py) and (b) by golden vectors captured from the imported reference in
the authoring container (tests/golden/make_golden.py -> tests/golden/*.npz).

Degenerate neighborhoods: the reference documents "all undefined features are represented by zeros"
(multiscale.py:4-5) but with numpy 2.x `np.cov` raises FloatingPointError for k<2 before the
try-block of features.pca (features.py:43 is outside the try at :45).  The oracle implements the
documented intent (zeros) and exposes `strict=True` to reproduce the raise.
"""

import numpy as np
from scipy.spatial import cKDTree

MAX_ADDRESS_LENGTH = 64          # geometry.py:12
LEAFSIZE = 300                   # multiscale.py:18
QUERY_CHUNK_SIZE = 1000          # multiscale.py:21


# --------------------------------------------------------------------------------------------------
# lattice (VoxelFilter) arithmetic
# --------------------------------------------------------------------------------------------------

class Lattice(object):
    """bounding lattice of a cloud: restates VoxelFilter.__init__/_calculate_shift/_calculate_masks
    (geometry.py:23-79)."""

    def __init__(self, points, edge_length, bounds=None):
        """`bounds` = (per-axis min, per-axis max) overrides the extrema of `points`: the multi-rank
        tests build the lattice of the WHOLE cloud while holding only a tile of it."""
        points = np.asarray(points)
        if points.ndim != 2:                                   # geometry.py:30
            raise ValueError("wrong point cloud array shape")
        if points.shape[1] not in (2, 3):                      # geometry.py:32
            raise ValueError("only 2D and 3D spaces supported")
        if points.shape[0] < 2:                                # geometry.py:34
            raise ValueError("need at least 2 points to define a voxel grid")
        self.edge_length = edge_length
        lo, hi = (points.min(0), points.max(0)) if bounds is None else \
            (np.asarray(bounds[0], dtype=np.float64), np.asarray(bounds[1], dtype=np.float64))
        self.minimum_corner = lo - edge_length / 2             # geometry.py:37
        self.maximum_corner = hi + edge_length / 2             # geometry.py:38
        span = self.maximum_corner - self.minimum_corner       # geometry.py:55
        widths = np.ceil(np.log2(span / edge_length))          # geometry.py:56
        if widths.sum() > MAX_ADDRESS_LENGTH:                  # geometry.py:59
            raise ValueError("edge length is too small to address this space")
        self.widths = widths.astype(np.int64)
        self.shifts = np.cumsum(widths)[:-1].astype(np.int64)  # geometry.py:62
        if np.any(self.widths <= 0):
            # geometry.py:74 builds int("0b" + "1"*width, 2), which is a ValueError for width 0
            raise ValueError("zero-width axis: cloud has no extent beyond one voxel on an axis")
        masks = [(1 << int(w)) - 1 for w in self.widths]       # geometry.py:74
        for num, s in enumerate(self.shifts):                  # geometry.py:76-77
            masks[num + 1] = masks[num + 1] << int(s)
        self.masks = masks

    def check_in_bounds(self, points):
        """geometry.py:83-99"""
        pts = np.atleast_2d(points)
        if pts.ndim != 2:
            raise ValueError("wrong array shape")
        if pts.shape[1] != self.shifts.size + 1:
            raise ValueError("wrong number of spatial dimensions")
        if np.any(pts.min(0) < self.minimum_corner) or np.any(pts.max(0) > self.maximum_corner):
            raise ValueError("some points fall outside filter bounding region")
        return pts

    def cell_coordinates(self, points):
        """integer lattice coordinates: floor((p - min_corner) / e), true division (geometry.py:108)"""
        pts = self.check_in_bounds(points)
        return np.floor((pts - self.minimum_corner) / self.edge_length).astype(np.int64)

    def coordinate_to_address(self, points):
        """x + (y << s0) + (z << s1)   (geometry.py:111-115)"""
        cells = self.cell_coordinates(points)
        addr = cells[:, 0].copy()
        for col, s in enumerate(self.shifts):
            addr += cells[:, col + 1] << s
        return addr

    def address_to_cells(self, addresses):
        """mask and shift back (geometry.py:129-134)"""
        addresses = np.atleast_1d(np.asarray(addresses, dtype=np.int64))
        cols = []
        for num, m in enumerate(self.masks):
            c = addresses & np.int64(m)
            if num > 0:
                c = c >> self.shifts[num - 1]
            cols.append(c)
        return np.stack(cols, axis=1)

    def address_to_coordinate(self, addresses):
        """voxel centre = cell*e + min_corner + e*0.5, evaluated left to right (geometry.py:137)"""
        cells = self.address_to_cells(addresses)
        return cells * self.edge_length + self.minimum_corner + self.edge_length * 0.5

    def unique_addresses(self, points):
        """sorted distinct addresses (geometry.py:148-150); their order defines the voxel index"""
        return np.unique(self.coordinate_to_address(points))

    def unique_voxels(self, points):
        """geometry.py:142-154"""
        return self.address_to_coordinate(self.unique_addresses(points))


# --------------------------------------------------------------------------------------------------
# neighbor search
# --------------------------------------------------------------------------------------------------

def ball_neighbors_kdtree(query_xyz, voxel_xyz, radius):
    """per query point, the sorted indices of voxel centres with Euclidean distance <= radius.
    Same call sequence as the reference: a tree over the search voxels, a tree per 1000-point query
    chunk, dual-tree query_ball_tree (multiscale.py:87,94,100,103).  Inner order in scipy is
    unspecified, so lists are returned sorted."""
    search_tree = cKDTree(voxel_xyz, leafsize=LEAFSIZE)
    out = []
    for start in range(0, len(query_xyz), QUERY_CHUNK_SIZE):
        chunk = query_xyz[start:start + QUERY_CHUNK_SIZE]
        chunk_tree = cKDTree(chunk, leafsize=LEAFSIZE)
        for idx in chunk_tree.query_ball_tree(search_tree, radius):
            out.append(np.sort(np.asarray(idx, dtype=np.int64)))
    return out


def ball_neighbors_bruteforce(query_xyz, voxel_xyz, radius):
    """the predicate the tree evaluates at its leaves, spelled out: ((dx*dx + dy*dy) + dz*dz) <= r*r
    in fp64 with no fused multiply-add (scipy ckdtree's p=2 path compares squared distances).
    O(Nq*M): small cases only.  Pins the inclusive boundary and the operation order."""
    r2 = np.float64(radius) * np.float64(radius)
    out = []
    for q in query_xyz:
        d = q[None, :] - voxel_xyz
        s = d[:, 0] * d[:, 0]
        for c in range(1, d.shape[1]):
            s = s + d[:, c] * d[:, c]
        out.append(np.nonzero(s <= r2)[0].astype(np.int64))
    return out


def neighbors_to_csr(neighbor_lists):
    counts = np.array([len(n) for n in neighbor_lists], dtype=np.int64)
    offsets = np.concatenate(([0], np.cumsum(counts)))
    flat = np.concatenate(neighbor_lists) if len(neighbor_lists) and offsets[-1] else \
        np.zeros(0, dtype=np.int64)
    return offsets, flat.astype(np.int64)


# --------------------------------------------------------------------------------------------------
# per-neighborhood operators (features.py)
# --------------------------------------------------------------------------------------------------

def population(neighborhood_points):
    """features.py:32-36"""
    return int(np.atleast_2d(neighborhood_points).shape[0]) if np.size(neighborhood_points) else 0


def centroid(query_point, neighborhood_points):
    """|| q - mean(neighborhood) ||_2, 0 for an empty neighborhood (features.py:21-29)"""
    if population(neighborhood_points) == 0:
        return 0.0
    return float(np.linalg.norm(query_point - neighborhood_points.mean(0)))


def pca(neighborhood_points, strict=False):
    """two largest eigenvalues of the ddof=1 covariance, normalised by the eigenvalue sum
    (features.py:39-57).  k<2 -> zeros (documented intent) or FloatingPointError when strict."""
    k = population(neighborhood_points)
    if k < 2:
        if strict:
            raise FloatingPointError("covariance undefined for fewer than 2 points")
        return np.zeros(2)
    cov = np.cov(neighborhood_points, rowvar=False)            # features.py:43
    eig = np.linalg.eigvalsh(cov)                              # ascending, features.py:46
    eig = eig / eig.sum()                                      # features.py:55
    return eig[:0:-1]                                          # [largest, middle], features.py:57


# --------------------------------------------------------------------------------------------------
# pipeline
# --------------------------------------------------------------------------------------------------

def one_scale(query_cloud, search_cloud, edge_length, radius, strict=False, return_neighbors=False):
    """(Nq,4) block [population, centroid distance, l1/sum, l2/sum] for one scale, faithful to the
    reference's structure: per-neighborhood numpy operators (multiscale.py:70-123)."""
    query_xyz = np.asarray(query_cloud, dtype=np.float64)[:, :3]
    search_xyz = np.asarray(search_cloud, dtype=np.float64)[:, :3]
    lattice = Lattice(search_xyz, edge_length)
    voxels = lattice.unique_voxels(search_xyz)
    nbrs = ball_neighbors_kdtree(query_xyz, voxels, radius)
    out = np.zeros((len(query_xyz), 4))
    for i, (q, idx) in enumerate(zip(query_xyz, nbrs)):
        nb = voxels.take(idx, axis=0)                          # features.py:14-18
        out[i, 0] = population(nb)
        out[i, 1] = centroid(q, nb)
        out[i, 2:] = pca(nb, strict=strict)
    if return_neighbors:
        return out, nbrs
    return out


def one_scale_covariance(query_cloud, search_cloud, edge_length, radius):
    """(Nq,6): upper triangle [xx, xy, xz, yy, yz, zz] of numpy.cov(neighborhood, rowvar=False) - the
    matrix features.pca forms at features.py:43 before taking its eigenvalues; zeros where the
    neighborhood has fewer than 2 voxels.  the reference does not return it (only the two normalised
    eigenvalues, features.py:57): this restates an intermediate of a pinned function."""
    query_xyz = np.asarray(query_cloud, dtype=np.float64)[:, :3]
    search_xyz = np.asarray(search_cloud, dtype=np.float64)[:, :3]
    lattice = Lattice(search_xyz, edge_length)
    voxels = lattice.unique_voxels(search_xyz)
    nbrs = ball_neighbors_kdtree(query_xyz, voxels, radius)
    out = np.zeros((len(query_xyz), 6))
    iu = np.triu_indices(3)
    for i, idx in enumerate(nbrs):
        if len(idx) >= 2:
            out[i] = np.cov(voxels.take(idx, axis=0), rowvar=False)[iu]
    return out


def one_scale_normals(query_cloud, search_cloud, edge_length, radius):
    """(Nq,3) unit eigenvector of the smallest eigenvalue of numpy.cov(neighborhood) (numpy.linalg.eigh),
    last non-zero component in x, y, z order made positive; zeros below 3 voxels.  also returns (Nq,)
    the gap (l2 - l3) / (l1 + l2 + l3): where it is tiny the direction is not defined by the data."""
    query_xyz = np.asarray(query_cloud, dtype=np.float64)[:, :3]
    search_xyz = np.asarray(search_cloud, dtype=np.float64)[:, :3]
    lattice = Lattice(search_xyz, edge_length)
    voxels = lattice.unique_voxels(search_xyz)
    nbrs = ball_neighbors_kdtree(query_xyz, voxels, radius)
    out = np.zeros((len(query_xyz), 3))
    gap = np.zeros(len(query_xyz))
    for i, idx in enumerate(nbrs):
        if len(idx) < 3:
            continue
        w, v = np.linalg.eigh(np.cov(voxels.take(idx, axis=0), rowvar=False))
        n = v[:, 0]
        nz = np.flatnonzero(np.abs(n) > 0)
        if len(nz) and n[nz[-1]] < 0:
            n = -n
        out[i] = n
        gap[i] = (w[1] - w[0]) / w.sum() if w.sum() > 0 else 0.0
    return out, gap


def one_scale_field_mean(query_cloud, search_cloud, attributes, edge_length, radius):
    """(Nq, D): mean over the voxels within `radius` of each query point of the voxel attribute, a voxel's
    attribute being the mean of `attributes` over the search points in it; zeros where no voxel is in
    reach.  the definition of nimrud_amd.minimal.fields (legacy V_MSO,
    prototypes/mso.py:12-173, is fp32 and partition dependent and cannot serve as a pin)."""
    query_xyz = np.asarray(query_cloud, dtype=np.float64)[:, :3]
    search_xyz = np.asarray(search_cloud, dtype=np.float64)[:, :3]
    attr = np.asarray(attributes, dtype=np.float64).reshape(len(search_xyz), -1)
    lattice = Lattice(search_xyz, edge_length)
    addresses = lattice.coordinate_to_address(search_xyz)
    unique, inverse = np.unique(addresses, return_inverse=True)
    counts = np.bincount(inverse, minlength=len(unique)).astype(np.float64)
    vmean = np.stack([np.bincount(inverse, weights=attr[:, d], minlength=len(unique)) / counts
                      for d in range(attr.shape[1])], axis=1)
    voxels = lattice.address_to_coordinate(unique)
    nbrs = ball_neighbors_kdtree(query_xyz, voxels, radius)
    out = np.zeros((len(query_xyz), attr.shape[1]))
    for i, idx in enumerate(nbrs):
        if len(idx):
            out[i] = vmean.take(idx, axis=0).mean(axis=0)
    return out


def process(query_cloud, search_cloud, edge_lengths, radii, strict=False):
    """(Nq, 4*S): per-scale blocks concatenated column-wise in caller order (multiscale.py:27-67)"""
    assert len(edge_lengths) == len(radii), \
        "edge_lengths and radii should be equal-length sequences."
    return np.concatenate(
        [one_scale(query_cloud, search_cloud, e, r, strict=strict)
         for e, r in zip(edge_lengths, radii)], axis=1)


def one_scale_fast(query_cloud, search_cloud, edge_length, radius, workers=1, bounds=None):
    """same numbers as `one_scale` but vectorised (bulk query_ball_point, segmented sums, stacked
    eigvalsh) so that 1e5-1e6-point parity cases finish in seconds.  Covariance is formed from
    mean-centred neighbor coordinates exactly like np.cov; summation order differs (<=1e-12)."""
    query_xyz = np.ascontiguousarray(np.asarray(query_cloud, dtype=np.float64)[:, :3])
    search_xyz = np.asarray(search_cloud, dtype=np.float64)[:, :3]
    lattice = Lattice(search_xyz, edge_length, bounds=bounds)
    voxels = lattice.unique_voxels(search_xyz)
    tree = cKDTree(voxels, leafsize=LEAFSIZE)
    out = np.zeros((len(query_xyz), 4))
    step = 200000
    for start in range(0, len(query_xyz), step):
        q = query_xyz[start:start + step]
        lists = tree.query_ball_point(q, radius, workers=workers)
        counts = np.fromiter((len(x) for x in lists), dtype=np.int64, count=len(lists))
        blk = out[start:start + step]
        blk[:, 0] = counts
        nz = counts > 0
        if not nz.any():
            continue
        flat = np.fromiter((j for x in lists for j in x), dtype=np.int64, count=int(counts.sum()))
        owner = np.repeat(np.arange(len(q)), counts)
        pts = voxels[flat]
        sums = np.zeros((len(q), 3))
        np.add.at(sums, owner, pts)
        mean = np.zeros_like(sums)
        mean[nz] = sums[nz] / counts[nz, None]
        blk[nz, 1] = np.linalg.norm(q[nz] - mean[nz], axis=1)
        dev = pts - mean[owner]
        outer = dev[:, :, None] * dev[:, None, :]
        cov = np.zeros((len(q), 3, 3))
        np.add.at(cov, owner, outer)
        ok = counts >= 2
        cov[ok] /= (counts[ok] - 1)[:, None, None]
        eig = np.linalg.eigvalsh(cov[ok])
        eig = eig / eig.sum(1)[:, None]
        blk[ok, 2] = eig[:, 2]
        blk[ok, 3] = eig[:, 1]
    return out


def process_fast(query_cloud, search_cloud, edge_lengths, radii, workers=1, bounds=None):
    assert len(edge_lengths) == len(radii)
    return np.concatenate(
        [one_scale_fast(query_cloud, search_cloud, e, r, workers=workers, bounds=bounds)
         for e, r in zip(edge_lengths, radii)], axis=1)


def one_scale_knn(query_cloud, search_cloud, edge_length, radius, k_min, radius_factor=3.0, bounds=None):
    """BUILD-DEFINED extension (BASELINE config 4), no reference counterpart - parity for it is pinned
    by this function only.  like one_scale_fast, but a query whose radius neighborhood holds fewer than
    k_min voxels is re-evaluated on its k_min nearest voxel centres (cKDTree.query on the same voxel
    set) within radius_factor*radius; column 0 keeps the radius population."""
    out = one_scale_fast(query_cloud, search_cloud, edge_length, radius, bounds=bounds)
    query_xyz = np.asarray(query_cloud, dtype=np.float64)[:, :3]
    search_xyz = np.asarray(search_cloud, dtype=np.float64)[:, :3]
    voxels = Lattice(search_xyz, edge_length, bounds=bounds).unique_voxels(search_xyz)
    tree = cKDTree(voxels, leafsize=LEAFSIZE)
    sparse = np.nonzero(out[:, 0] < k_min)[0]
    if len(sparse) == 0:
        return out
    k = min(k_min, len(voxels))
    dist, idx = tree.query(query_xyz[sparse], k=k)
    dist, idx = dist.reshape(len(sparse), -1), idx.reshape(len(sparse), -1)
    limit = radius * radius_factor
    for row, d, i in zip(sparse, dist, idx):
        nb = voxels[i[d <= limit]]
        out[row, 1] = centroid(query_xyz[row], nb)
        out[row, 2:] = pca(nb)
    return out


# --------------------------------------------------------------------------------------------------
# the plain-C restatement (oracle/lattice_oracle.c): all rows of full-size configurations in seconds
# --------------------------------------------------------------------------------------------------

_C_ORACLE = None


def _c_oracle():
    global _C_ORACLE
    if _C_ORACLE is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblattice_oracle.so")
        if not os.path.exists(path):
            raise ImportError("oracle/liblattice_oracle.so is not built: make -C oracle")
        lib = ctypes.CDLL(path)
        lib.nm_oracle_scale.restype = ctypes.c_long
        lib.nm_oracle_scale.argtypes = [
            ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
            ctypes.c_long, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
            ctypes.c_void_p, ctypes.c_int]
        lib.nm_oracle_neighbors.restype = ctypes.c_long
        lib.nm_oracle_neighbors.argtypes = [
            ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
            ctypes.c_long, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _C_ORACLE = lib
    return _C_ORACLE


def neighbor_lists_c(query_cloud, search_cloud, edge_length, radius, threads=0, bounds=None):
    """the neighbor index lists of multiscale.py:103 for ALL queries, as CSR (offsets int64[Nq + 1], index
    int64[total]): oracle/lattice_oracle.c enumerates, per query, the addresses of the occupied voxels within
    the radius (hash set of addresses, fp64 inclusive test, ascending address order); an address becomes the
    reference's search-voxel index by its rank in the sorted unique address array (np.unique,
    geometry.py:150).  fast enough for a million queries; pinned against the golden neighbor lists captured
    from the reference (tests/test_oracle.py)."""
    query = np.ascontiguousarray(np.asarray(query_cloud, dtype=np.float64))
    search = np.ascontiguousarray(np.asarray(search_cloud, dtype=np.float64))
    lattice = Lattice(search[:, :3], edge_length, bounds=bounds)
    mc = np.ascontiguousarray(lattice.minimum_corner, dtype=np.float64)
    widths = np.ascontiguousarray(lattice.widths, dtype=np.int32)
    nq = query.shape[0]
    counts = np.zeros(nq, dtype=np.int32)
    lib = _c_oracle()
    args = (query.ctypes.data, nq, query.shape[1], search.ctypes.data, search.shape[0], search.shape[1],
            mc.ctypes.data, float(edge_length), widths.ctypes.data, float(radius))
    if lib.nm_oracle_neighbors(*args, counts.ctypes.data, None, None, int(threads)) < 0:
        raise MemoryError("lattice_oracle: allocation failed")
    offsets = np.zeros(nq + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    addr = np.zeros(max(int(offsets[-1]), 1), dtype=np.uint64)
    if lib.nm_oracle_neighbors(*args, None, offsets.ctypes.data, addr.ctypes.data, int(threads)) < 0:
        raise MemoryError("lattice_oracle: allocation failed")
    addr = addr[:int(offsets[-1])]
    unique = lattice.unique_addresses(search[:, :3]).astype(np.uint64)
    index = np.searchsorted(unique, addr).astype(np.int64)
    assert np.array_equal(unique[index], addr)
    return offsets, index


def one_scale_c(query_cloud, search_cloud, edge_length, radius, threads=0, bounds=None,
                return_voxel_count=False):
    """one scale through oracle/lattice_oracle.c: gathers the real voxel centres of every neighborhood
    and computes mean / ddof=1 covariance / Jacobi eigenvalues in fp64 (no kd-tree, no numpy.cov, no
    LAPACK, no integer moments).  threads = 0 uses all cores OpenMP sees."""
    query = np.ascontiguousarray(np.asarray(query_cloud, dtype=np.float64))
    search = np.ascontiguousarray(np.asarray(search_cloud, dtype=np.float64))
    lattice = Lattice(search[:, :3], edge_length, bounds=bounds)
    mc = np.ascontiguousarray(lattice.minimum_corner, dtype=np.float64)
    widths = np.ascontiguousarray(lattice.widths, dtype=np.int32)
    out = np.zeros((query.shape[0], 4))
    m = _c_oracle().nm_oracle_scale(
        query.ctypes.data, query.shape[0], query.shape[1], search.ctypes.data, search.shape[0],
        search.shape[1], mc.ctypes.data, float(edge_length), widths.ctypes.data, float(radius),
        out.ctypes.data, int(threads))
    if m < 0:
        raise MemoryError("lattice_oracle: allocation failed")
    return (out, int(m)) if return_voxel_count else out


def process_c(query_cloud, search_cloud, edge_lengths, radii, threads=0, bounds=None):
    assert len(edge_lengths) == len(radii)
    return np.concatenate([one_scale_c(query_cloud, search_cloud, e, r, threads=threads, bounds=bounds)
                           for e, r in zip(edge_lengths, radii)], axis=1)


# --------------------------------------------------------------------------------------------------
# classifier slot: random forest evaluation
# --------------------------------------------------------------------------------------------------

def forest_predict_proba(model, features):
    """mean over trees of the leaf class distribution; the reference's classifier slot is
    sklearn's RandomForestClassifier (apc.py:1463) applied with predict_proba (apc.py:1034).
    sklearn casts X to float32 and sends a sample left when x[feature] <= threshold (threshold f64).
    `model` is the flat export described in nimrud_amd/minimal/classification.py:
      left, right (int32, -1 at leaves), feature (int32), threshold (f64), value (nodes, C) f64
      normalised per node, roots (int32 index of each tree's root)."""
    x = np.asarray(features).astype(np.float32).astype(np.float64)
    n = x.shape[0]
    n_classes = model["value"].shape[1]
    proba = np.zeros((n, n_classes))
    rows = np.arange(n)
    for root in model["roots"]:
        node = np.full(n, root, dtype=np.int64)
        active = model["left"][node] >= 0
        while active.any():
            cur = node[active]
            go_left = x[rows[active], model["feature"][cur]] <= model["threshold"][cur]
            node[active] = np.where(go_left, model["left"][cur], model["right"][cur])
            active = model["left"][node] >= 0
        proba += model["value"][node]
    return proba / len(model["roots"])


def forest_predict(model, features):
    return model["classes"][np.argmax(forest_predict_proba(model, features), axis=1)]
