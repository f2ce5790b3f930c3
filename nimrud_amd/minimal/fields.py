"""
vector-field multiscale operator: neighborhood means of arbitrary per-point attributes (SURVEY.md section 8f,
rank 4).  the reference's current path (nimrud/minimal) has no such operator; its legacy pycuda generation
does (prototypes/mso.py:12-173, V_MSO: voxelize the search space, carry the attribute field over to the
voxels, average over the voxels within each radius).  this is that operator on the lattice of
nimrud/minimal: same voxel filter, same voxel centres, same inclusive ball as the feature operator
(multiscale.py:76-103), a voxel's attribute being the mean of the attributes of the search points in it.
"""

import ctypes

import numpy as np
import torch

from nimrud_amd import device as _device
from nimrud_amd.utils import geometry

MAX_DIMS = 16


def vector_field_mean_gpu(query_cloud, search_cloud, attributes, edge_lengths, radii):
    """(Nq, D*S) fp64 GPU tensor: for every scale s the D-vector mean of the voxel attributes over the
    voxels within radii[s] of each query point (zeros where there are none), scales side by side in caller
    order like the feature matrix.  clouds and attributes are torch GPU tensors (or anything
    device.as_cloud takes); attributes is (Ns, D) or (Ns,), 1 <= D <= 16."""
    assert len(edge_lengths) == len(radii), \
        "edge_lengths and radii should be equal-length sequences."
    shared = query_cloud is search_cloud
    rt, search = _device.as_cloud(search_cloud)
    query = search if shared else _device.as_cloud(query_cloud, rt.device)[1]
    if search.shape[1] < 3 or query.shape[1] < 3:
        raise ValueError("only 3D spaces supported by the multiscale pipeline")
    if search.shape[0] < 2:
        raise ValueError("need at least 2 points to define a voxel grid")
    attr = torch.as_tensor(attributes).to(device=rt.device, dtype=torch.float64)
    if attr.ndim == 1:
        attr = attr.reshape(-1, 1)
    if attr.ndim != 2 or attr.shape[0] != search.shape[0]:
        raise ValueError("attributes must have one row per search point")
    dims = int(attr.shape[1])
    if not 1 <= dims <= MAX_DIMS:
        raise ValueError("between 1 and %d attribute columns per call" % MAX_DIMS)
    attr = attr.contiguous()
    nq, ns, n_scales = query.shape[0], search.shape[0], len(edge_lengths)
    out = torch.zeros((nq, dims * n_scales), dtype=torch.float64, device=rt.device)
    if n_scales == 0 or nq == 0:
        return out
    lo, hi = _device.cloud_bounds(rt, search)
    for s, (e, r) in enumerate(zip(edge_lengths, radii)):
        lat = geometry.VoxelFilter.from_bounds(lo, hi, e, device=rt.device).nm_lattice
        nbytes = rt.lib.nm_field_workspace_bytes(nq, ns, ctypes.byref(lat), dims)
        work = rt.workspace(nbytes)
        rt.check(rt.lib.nm_field_mean(
            rt.ctx, _device.ptr(query), nq, _device.row_stride(query),
            _device.ptr(search), ns, _device.row_stride(search),
            _device.ptr(attr), int(attr.stride(0)), dims, ctypes.byref(lat), float(r),
            ctypes.c_void_p(out.data_ptr() + 8 * dims * s), int(out.stride(0)),
            _device.ptr(work), work.numel(), rt.stream()))
    return out


def vector_field_mean(query_cloud, search_cloud, attributes, edge_lengths, radii):
    """numpy in, numpy out (the clouds and the attribute table cross PCIe)."""
    return vector_field_mean_gpu(np.asarray(query_cloud), np.asarray(search_cloud),
                                 np.asarray(attributes), edge_lengths, radii).cpu().numpy()
