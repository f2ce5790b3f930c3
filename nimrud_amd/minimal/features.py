"""
functions to be mapped over point neighborhoods: the drop-in for nimrud/minimal/features.py.

the fused pipeline (multiscale.py) never materialises neighborhoods; these operators exist for code
that calls them directly, as the reference's own pipeline does (multiscale.py:106-116).  each takes
numpy arrays (or torch GPU tensors) like the reference and evaluates on the GPU through
nm_neighborhood_features; `neighborhood_features` is the batched form (CSR) to use for real work.

unlike the reference this module does not call np.seterr(invalid="raise") at import (features.py:10):
that is process-global state a library should not touch.
"""

import numpy as np
import torch

from nimrud_amd import device as _device


def take(neighborhood_idx, search_space_cloud):
    """return an array of points from the search space (features.py:14-18)."""
    if isinstance(search_space_cloud, torch.Tensor):
        idx = torch.as_tensor(neighborhood_idx, dtype=torch.int64, device=search_space_cloud.device)
        return search_space_cloud.index_select(0, idx)
    return np.asarray(search_space_cloud).take(neighborhood_idx, axis=0)


def neighborhood_features(points, offsets, query_points):
    """batched operators: neighborhood b is points[offsets[b]:offsets[b+1]] (rows of 3 coordinates),
    query_points[b] its query point.  returns (B,4): population, centroid distance, two largest
    eigenvalues of the covariance normalised by their sum (zeros where undefined)."""
    as_torch = isinstance(points, torch.Tensor)
    rt, pts = _device.as_cloud(points)
    pts = pts[:, :3].contiguous()
    qry = _device.as_cloud(query_points, rt.device)[1][:, :3].contiguous()
    off = torch.as_tensor(np.asarray(offsets) if not isinstance(offsets, torch.Tensor) else offsets,
                          dtype=torch.int64).to(rt.device).contiguous()
    nb = off.shape[0] - 1
    if off.ndim != 1 or nb < 0:
        raise ValueError("offsets must be a 1-D array of B+1 row numbers")
    if qry.shape[0] != nb:
        raise ValueError("one query point per neighborhood expected")
    if nb > 0:
        # the kernel trusts the offsets: they must be monotone row numbers inside `points`
        lo, hi, steps = int(off[0]), int(off[-1]), off[1:] - off[:-1]
        if lo < 0 or hi > pts.shape[0] or bool((steps < 0).any()):
            raise ValueError("offsets must be non-decreasing and lie in [0, len(points)]")
    out = torch.empty((nb, 4), dtype=torch.float64, device=rt.device)
    rt.check(rt.lib.nm_neighborhood_features(rt.ctx, _device.ptr(pts), _device.ptr(off),
                                             _device.ptr(qry), nb, _device.ptr(out), 4,
                                             rt.stream()))
    return out if as_torch else out.cpu().numpy()


def _single(query_point, neighborhood_points):
    nb = np.atleast_2d(np.asarray(neighborhood_points, dtype=np.float64))
    if nb.size == 0:
        nb = np.zeros((0, 3))
    q = np.zeros((1, 3)) if query_point is None else \
        np.asarray(query_point, dtype=np.float64).reshape(1, -1)[:, :3]
    if nb.shape[0] == 0:
        return np.zeros(4)
    pts = np.concatenate((nb[:, :3], np.zeros((1, 3))), axis=0)   # as_cloud wants >= 1 row
    return neighborhood_features(pts, np.array([0, nb.shape[0]]), q)[0]


def centroid(query_point, neighborhood_points):
    """distance between the query point and the mean of its neighborhood; 0 when the neighborhood is
    empty (features.py:21-29).
    one neighborhood per call means one upload, one launch and one download per call - milliseconds, like
    `pca` below.  mapping these over neighborhoods the way the reference does (multiscale.py:109-116) is
    only sensible for a handful; `neighborhood_features` is the batched form and
    `multiscale.process_gpu` the real path."""
    return float(_single(query_point, neighborhood_points)[1])


def population(neighborhood_points):
    """count the points in the neighborhood (features.py:32-36)."""
    nb = np.asarray(neighborhood_points)
    return int(np.atleast_2d(nb).shape[0]) if nb.size else 0


def pca(neighborhood_points, strict=False):
    """the normalized variance of the first two principal components of the neighborhood
    (features.py:39-57): [largest, middle] eigenvalue of the ddof=1 covariance over the eigenvalue sum.
    fewer than two points -> zeros (the documented value, multiscale.py:4-5), or FloatingPointError
    with strict=True (what numpy.cov makes the reference do).
    scalar convenience form (one launch per call): see the note on `centroid`."""
    if population(neighborhood_points) < 2:
        if strict:
            raise FloatingPointError("covariance undefined for fewer than 2 points")
        return np.zeros(2)
    return _single(None, neighborhood_points)[2:4].copy()


def descriptors(feature_matrix):
    """linearity, planarity, scatter per scale from a (N, 4*S) feature matrix of the multiscale pipeline:
    (l1-l2)/l1, (l2-l3)/l1, l3/l1 with l3 = 1 - l1 - l2.  returns (N, 3*S); undefined rows are zeros.
    an addition to the reference, whose minimal path stops at (l1, l2) (features.py:57)."""
    as_torch = isinstance(feature_matrix, torch.Tensor)
    rt, feat = _device.as_cloud(feature_matrix)
    n, cols = feat.shape
    if cols % 4:
        raise ValueError("feature matrix must have 4 columns per scale")
    n_scales = cols // 4
    out = torch.empty((n, 3 * n_scales), dtype=torch.float64, device=rt.device)
    rt.check(rt.lib.nm_descriptors(rt.ctx, _device.ptr(feat), n, n_scales, _device.row_stride(feat),
                                   _device.ptr(out), 3 * n_scales, rt.stream()))
    return out if as_torch else out.cpu().numpy()
