"""
multiscale operator processing pipeline on the MI355X: the drop-in for nimrud/minimal/multiscale.py.

features are generated for points in the query cloud, using geometry from the search cloud.  for each
scale: voxel-filter the search cloud at `edge_length`, find every occupied voxel centre within
`radius` of each query point, and emit [population, distance to the neighborhood centroid, largest
and middle eigenvalue of the covariance normalised by the eigenvalue sum] (multiscale.py:1-6,70-123).
undefined features (neighborhoods of fewer than two voxels) are zeros, as the reference documents
(multiscale.py:4-5); pass strict=True to get the FloatingPointError the reference actually raises
from numpy.cov in that case.

same names and signatures as the reference:
    process_single_core(query_cloud, search_cloud, edge_lengths, radii, verbose=False)  -> (Nq, 4*S)
    one_scale_single_core(query_cloud, search_cloud, edge_length, radius, verbose=False) -> (Nq, 4)
("single core" is kept for compatibility: the work runs on one GPU.)  additive, for data that already
lives in HBM: process_gpu / one_scale_gpu take and return torch GPU tensors.

each scale is one call into libnimrud_hip.so (nm_scale_features): cell keys -> radix sort -> sparse
occupancy index -> fused search/moments/eigen kernel.  see DESIGN.md.
"""

import ctypes
import time

import numpy as np
import torch

from nimrud_amd import device as _device
from nimrud_amd.utils import geometry

# kept for API compatibility with the reference (multiscale.py:18-24).  the GPU path has no kd-tree
# and no host-side chunk loop, so LEAFSIZE and QUERY_CHUNK_SIZE have no effect here.
LEAFSIZE = 300
QUERY_CHUNK_SIZE = 1000
VERBOSITY_INTERVAL = 100


class ScaleInfo(object):
    """what one scale reported: voxels = M (occupied search voxels), degenerate = neighborhoods with
    population < 2, extra_passes = search-kernel passes beyond one per wave, leaves = index leaves."""

    def __init__(self, values):
        self.voxels, self.degenerate, self.extra_passes, self.leaves = (int(v) for v in values)

    def __repr__(self):
        return "ScaleInfo(voxels=%d, degenerate=%d, extra_passes=%d, leaves=%d)" % (
            self.voxels, self.degenerate, self.extra_passes, self.leaves)


def _scale_into(rt, query, search, shared, lo, hi, edge_length, radius, out_view, info_view):
    """enqueue one scale: features of `query` against `search` into out_view (Nq,4 strided)."""
    vf = geometry.VoxelFilter.from_bounds(lo, hi, edge_length, device=rt.device)
    lat = vf.nm_lattice
    nq, ns = query.shape[0], search.shape[0]
    nbytes = rt.lib.nm_scale_workspace_bytes(nq, ns, ctypes.byref(lat))
    work = rt.workspace(nbytes)
    qt = search if shared else query
    rt.check(rt.lib.nm_scale_features(
        rt.ctx, _device.ptr(qt), nq, _device.row_stride(qt),
        _device.ptr(search), ns, _device.row_stride(search),
        ctypes.byref(lat), float(radius),
        _device.ptr(out_view), int(out_view.stride(0)), _device.ptr(info_view),
        _device.ptr(work), work.numel(), rt.stream()))


def _ladder_into(rt, query, search, shared, bounds, edge_lengths, radii, out, info, knn_min=0,
                 knn_radius_factor=3.0):
    """enqueue the whole ladder in one library call (nm_ladder_features): the cloud is sorted once, every
    scale's lattice is built on the device from the search cloud's extrema and every index from that
    order.  nothing visits the host: the call can be queued behind other work.
    bounds: None (the library measures the search cloud) or a (6,) fp64 device tensor {min xyz, max xyz}
    - the global extrema of a multi-GPU job."""
    # the fallback switch is context state in the C ABI: always set it, so no call inherits another's
    rt.check(rt.lib.nm_set_knn_fallback(rt.ctx, int(knn_min), float(knn_radius_factor)))
    n_scales = len(edge_lengths)
    edg = (ctypes.c_double * n_scales)(*[float(e) for e in edge_lengths])
    rad = (ctypes.c_double * n_scales)(*[float(r) for r in radii])
    nq, ns = query.shape[0], search.shape[0]
    nbytes = rt.lib.nm_ladder_workspace_bytes_for(nq, ns, edg, n_scales)     # equal edges share an index
    if nbytes == 0:
        raise ValueError("bad ladder arguments (at most 32 scales per call)")
    work = rt.workspace(nbytes)
    qt = search if shared else query
    if bounds is not None and not (isinstance(bounds, torch.Tensor) and bounds.is_cuda and
                                   bounds.dtype == torch.float64 and bounds.numel() == 6
                                   and bounds.is_contiguous()):
        raise ValueError("bounds must be a contiguous (6,) fp64 tensor on the device")
    rt.check(rt.lib.nm_ladder_features(
        rt.ctx, _device.ptr(qt), nq, _device.row_stride(qt),
        _device.ptr(search), ns, _device.row_stride(search), edg, rad, n_scales,
        _device.ptr(bounds), _device.ptr(out), int(out.stride(0)), _device.ptr(info),
        _device.ptr(work), work.numel(), rt.stream()))


def _report_scale(rt, verbose, nq, info, s, this_edge, this_radius, inner_start):
    """the reference's per-scale progress lines (multiscale.py:47-65)."""
    if not verbose:
        return
    torch.cuda.synchronize(rt.device)
    inner = time.perf_counter() - inner_start
    print("querying {} points against a search space of {} voxels".format(nq, int(info[s, 0])))
    print("using a voxel edge length of {} and radius of {}".format(this_edge, this_radius))
    print("this scale took {}s".format(np.around(inner, 6)))
    print("one scale rate of {} points per second".format(np.around(nq / inner, 3)))
    print("===================================")


def process_gpu(query_cloud, search_cloud, edge_lengths, radii, verbose=False, strict=False,
                return_info=False, out=None, per_scale=False, knn_min=0, knn_radius_factor=3.0,
                cov_out=None, normal_out=None):
    """process_single_core for clouds resident in HBM: torch GPU tensors in, (Nq, 4*S) fp64 GPU tensor
    out.  nothing crosses PCIe, and nothing waits for the device: the lattices are built on the GPU from
    the cloud's extrema (nm_ladder_features), so the call only enqueues work.  what the reference's
    VoxelFilter raises synchronously for an unusable edge length (geometry.py:59-60) therefore surfaces
    at the next synchronisation point: strict / return_info (which read 4 counters per scale back),
    process_single_core, the next call into the library, or `nimrud_amd.device.get_runtime().check_async()`.
    the ladder normally runs as ONE library call that sorts the cloud once for all scales;
    verbose=True or per_scale=True runs one self-contained call per scale instead (same numbers).
    knn_min > 0 switches on the k-nearest-voxel fallback (an extension the reference does not have,
    BASELINE config 4): neighborhoods with fewer than knn_min voxels take their centroid and eigen
    features from the knn_min nearest voxels within knn_radius_factor * radius.
    cov_out, a (Nq, 6*S) fp64 GPU tensor, additionally receives per scale the upper triangle
    [xx, xy, xz, yy, yz, zz] of the neighborhood covariance whose eigenvalues the features are
    (see process_gpu_covariance).  normal_out, (Nq, 3*S), receives the unit
    eigenvector of its smallest eigenvalue - the surface normal, pointing up (process_gpu_normals)."""
    assert len(edge_lengths) == len(radii), \
        "edge_lengths and radii should be equal-length sequences."
    shared = query_cloud is search_cloud
    rt, search = _device.as_cloud(search_cloud)
    query = search if shared else _device.as_cloud(query_cloud, rt.device)[1]
    if search.shape[1] < 3 or query.shape[1] < 3:
        raise ValueError("only 3D spaces supported by the multiscale pipeline")
    if search.shape[0] < 2:
        raise ValueError("need at least 2 points to define a voxel grid")
    n_scales = len(edge_lengths)
    nq = query.shape[0]
    if out is None:
        out = torch.empty((nq, 4 * n_scales), dtype=torch.float64, device=rt.device)
    elif not (isinstance(out, torch.Tensor) and out.dtype == torch.float64 and out.device == rt.device
              and out.ndim == 2 and out.shape[0] == nq and out.shape[1] >= 4 * n_scales
              and (out.shape[1] == 0 or out.stride(1) == 1)
              and (nq <= 1 or out.stride(0) >= 4 * n_scales)):
        # the kernels write 4*S doubles per row at the row pitch of `out`: anything else would put
        # 8-byte stores outside the allocation
        raise ValueError("out must be a (Nq, >= 4*S) fp64 tensor on the clouds' device with unit column "
                         "stride")
    # the per-scale counters (voxels, degenerate neighborhoods, extra passes, leaves) cost a pass over the
    # index: only when somebody will look at them
    want_info = bool(strict or return_info or verbose)
    info = torch.zeros((max(n_scales, 1), 4), dtype=torch.int64, device=rt.device) if want_info else None
    if n_scales == 0:
        return (out, []) if return_info else out

    if cov_out is not None:
        if not (isinstance(cov_out, torch.Tensor) and cov_out.dtype == torch.float64 and
                cov_out.device == out.device and cov_out.ndim == 2 and cov_out.shape[0] == nq and
                cov_out.shape[1] >= 6 * n_scales and cov_out.stride(1) == 1):
            raise ValueError("cov_out must be a (Nq, >= 6*S) fp64 tensor on the clouds' device")
    if normal_out is not None:
        if not (isinstance(normal_out, torch.Tensor) and normal_out.dtype == torch.float64 and
                normal_out.device == out.device and normal_out.ndim == 2 and normal_out.shape[0] == nq
                and normal_out.shape[1] >= 3 * n_scales and normal_out.stride(1) == 1):
            raise ValueError("normal_out must be a (Nq, >= 3*S) fp64 tensor on the clouds' device")
    outer_start = time.perf_counter()
    rt.check(rt.lib.nm_set_knn_fallback(rt.ctx, int(knn_min), float(knn_radius_factor)))
    lo = hi = None
    if verbose or per_scale:
        lo, hi = _device.cloud_bounds(rt, search)       # the per-scale form builds its lattices here

    def covariance_columns(first_scale):
        # context state of the C ABI, like the fallback switch: set for this call, cleared after it
        if cov_out is not None:
            rt.check(rt.lib.nm_set_covariance_output(
                rt.ctx, ctypes.c_void_p(cov_out.data_ptr() + 48 * first_scale), int(cov_out.stride(0))))
        if normal_out is not None:
            rt.check(rt.lib.nm_set_normal_output(
                rt.ctx, ctypes.c_void_p(normal_out.data_ptr() + 24 * first_scale),
                int(normal_out.stride(0))))

    try:
        if not (verbose or per_scale):
            covariance_columns(0)
            _ladder_into(rt, query, search, shared, None, edge_lengths, radii, out, info,
                         knn_min=knn_min, knn_radius_factor=knn_radius_factor)
            edge_lengths_loop = []
        else:
            edge_lengths_loop = list(zip(edge_lengths, radii))
        for s, (this_edge, this_radius) in enumerate(edge_lengths_loop):
            inner_start = time.perf_counter()
            covariance_columns(s)
            _scale_into(rt, query, search, shared, lo, hi, this_edge, this_radius,
                        out[:, 4 * s:4 * s + 4], info[s] if info is not None else None)
            _report_scale(rt, verbose, nq, info, s, this_edge, this_radius, inner_start)
    finally:
        if cov_out is not None:
            rt.check(rt.lib.nm_set_covariance_output(rt.ctx, None, 0))
        if normal_out is not None:
            rt.check(rt.lib.nm_set_normal_output(rt.ctx, None, 0))
    if verbose:
        torch.cuda.synchronize(rt.device)
        outer = time.perf_counter() - outer_start
        print("calculating all scales took {}s".format(np.around(outer, 6)))
        print("final rate of {} points per second".format(np.around(nq / outer, 3)))

    if strict or return_info:
        host = info.cpu().numpy()
        rt.check_async(wait=True)        # an unusable lattice, an index that timed out or overflowed
        if strict and host[:n_scales, 1].any():
            raise FloatingPointError(
                "%d neighborhoods have fewer than 2 voxels; their covariance is undefined "
                "(the reference raises here from numpy.cov, features.py:43)"
                % int(host[:n_scales, 1].sum()))
        if return_info:
            return out, [ScaleInfo(row) for row in host[:n_scales]]
    return out


def process_gpu_covariance(query_cloud, search_cloud, edge_lengths, radii, **kwargs):
    """(features (Nq, 4*S), covariances (Nq, 6*S)): per scale the four features and the upper triangle
    [xx, xy, xz, yy, yz, zz] of the ddof=1 covariance of the neighborhood's voxel centres - the matrix
    features.pca builds with numpy.cov (features.py:43).  zeros where fewer than 2 voxels."""
    rt, search = _device.as_cloud(search_cloud)
    nq = search.shape[0] if query_cloud is search_cloud else _device.as_cloud(query_cloud, rt.device)[1].shape[0]
    cov = torch.zeros((nq, 6 * len(edge_lengths)), dtype=torch.float64, device=rt.device)
    feats = process_gpu(query_cloud, search_cloud, edge_lengths, radii, cov_out=cov, **kwargs)
    return feats, cov


def process_gpu_normals(query_cloud, search_cloud, edge_lengths, radii, **kwargs):
    """(features (Nq, 4*S), normals (Nq, 3*S)): per scale the unit eigenvector of the smallest eigenvalue
    of the neighborhood covariance, last non-zero component (x, y, z order) positive; zeros where fewer
    than 3 voxels."""
    rt, search = _device.as_cloud(search_cloud)
    nq = search.shape[0] if query_cloud is search_cloud else _device.as_cloud(query_cloud, rt.device)[1].shape[0]
    normals = torch.zeros((nq, 3 * len(edge_lengths)), dtype=torch.float64, device=rt.device)
    feats = process_gpu(query_cloud, search_cloud, edge_lengths, radii, normal_out=normals, **kwargs)
    return feats, normals


def one_scale_gpu(query_cloud, search_cloud, edge_length, radius, verbose=False, strict=False):
    """(Nq,4) GPU tensor for one analysis scale."""
    return process_gpu(query_cloud, search_cloud, [edge_length], [radius], verbose=verbose,
                       strict=strict)


def process_single_core(query_cloud, search_cloud, edge_lengths, radii, verbose=False, strict=False,
                        knn_min=0, knn_radius_factor=3.0):
    """compute features at multiple scales.  returns an array of feature vectors aligned with the
    query cloud: numpy (Nq, 4*S) fp64, scale blocks in caller order (multiscale.py:27-67).
    knn_min / knn_radius_factor: see process_gpu (extension, off by default)."""
    assert len(edge_lengths) == len(radii), \
        "edge_lengths and radii should be equal-length sequences."
    shared = query_cloud is search_cloud
    rt, search = _device.as_cloud(search_cloud)
    query = search if shared else _device.as_cloud(query_cloud, rt.device)[1]
    result = process_gpu(query, search, edge_lengths, radii, verbose=verbose, strict=strict,
                         knn_min=knn_min, knn_radius_factor=knn_radius_factor)
    host = result.cpu().numpy()
    # the copy has synchronised the stream: anything a kernel of this call reported is in by now.  an
    # incomplete result is never returned silently.
    rt.check_async(wait=True)
    return host


def one_scale_single_core(query_cloud, search_cloud, edge_length, radius, verbose=False,
                          strict=False):
    """generate a 4d feature vector representing one analysis scale (multiscale.py:70-123)."""
    return process_single_core(query_cloud, search_cloud, [edge_length], [radius], verbose=verbose,
                               strict=strict)


def neighbor_lists(query_cloud, search_cloud, edge_length, radius):
    """the neighbor index lists of multiscale.py:103 as CSR (offsets int64[Nq+1], index int64[total]):
    for every query point the ascending positions, in the sorted unique voxel array of
    VoxelFilter.unique_voxels(search_cloud), of the voxels within `radius`.  inspection / parity mode:
    it searches by address and is much slower than the fused feature path."""
    shared = query_cloud is search_cloud
    as_torch = isinstance(query_cloud, torch.Tensor)
    rt, search = _device.as_cloud(search_cloud)
    query = search if shared else _device.as_cloud(query_cloud, rt.device)[1]
    vf = geometry.VoxelFilter(search[:, :3], edge_length, device=rt.device)
    addr = vf.unique_addresses(search[:, :3])
    nq = query.shape[0]
    counts = torch.empty(nq, dtype=torch.int32, device=rt.device)
    lat = vf.nm_lattice
    null = ctypes.c_void_p(0)
    rt.check(rt.lib.nm_scale_neighbors(rt.ctx, _device.ptr(query), nq, _device.row_stride(query),
                                       _device.ptr(addr), addr.shape[0], ctypes.byref(lat),
                                       float(radius), _device.ptr(counts), null, null, rt.stream()))
    offsets = torch.zeros(nq + 1, dtype=torch.int64, device=rt.device)
    torch.cumsum(counts, 0, out=offsets[1:])
    total = int(offsets[-1])
    index = torch.empty(max(total, 1), dtype=torch.int64, device=rt.device)
    rt.check(rt.lib.nm_scale_neighbors(rt.ctx, _device.ptr(query), nq, _device.row_stride(query),
                                       _device.ptr(addr), addr.shape[0], ctypes.byref(lat),
                                       float(radius), null, _device.ptr(offsets), _device.ptr(index),
                                       rt.stream()))
    index = index[:total]
    if as_torch:
        return offsets, index
    return offsets.cpu().numpy(), index.cpu().numpy()
