"""
voxel filter on the MI355X: the host-side mirror of nimrud/utils/geometry.py `VoxelFilter` (:16-154).

same constructor, attributes (`minimum_corner`, `maximum_corner`, `edge_length`, `shifts`, `widths`,
`masks`) and methods (`coordinate_to_address`, `address_to_coordinate`, `unique_voxels`) as the
reference, same ValueError conditions.  the per-point work (bounds, cell addresses, sort + unique,
centres) runs in libnimrud_hip.so; only the handful of scalar lattice parameters are computed here.
numpy arrays in -> numpy arrays out; torch GPU tensors in -> torch GPU tensors out (no PCIe).

nested partitioning (geometry.py:203-505) is unfinished in the reference and out of scope; its idea
(query tile + buffered search tile) lives on as the multi-GPU halo in nimrud_amd/parallel.py.
"""

import ctypes

import numpy as np
import torch

from nimrud_amd import _ffi
from nimrud_amd import device as _device

MAX_ADDRESS_LENGTH = 64


def lattice_parameters(minimum, maximum, edge_length):
    """(min_corner, max_corner, widths, shifts) from the cloud's per-axis extrema: the scalar part of
    VoxelFilter.__init__ / _calculate_shift (geometry.py:37-38, 55-64)."""
    minimum = np.asarray(minimum, dtype=np.float64)
    maximum = np.asarray(maximum, dtype=np.float64)
    half = edge_length / 2
    min_corner = minimum - half
    max_corner = maximum + half
    widths = np.ceil(np.log2((max_corner - min_corner) / edge_length))
    if widths.sum() > MAX_ADDRESS_LENGTH:
        raise ValueError("edge length is too small to address this space")
    shifts = np.cumsum(widths)[:-1]
    return min_corner, max_corner, widths.astype(np.int64), shifts.astype(np.int64)


def make_nm_lattice(min_corner, edge_length, widths):
    """struct nm_lattice for the C ABI.  a 2-D lattice is passed as 3-D with one z cell."""
    lat = _ffi.NmLattice()
    dims = len(widths)
    w = [int(x) for x in widths] + [1] * (3 - dims)
    mc = [float(x) for x in min_corner] + [-float(edge_length) / 2] * (3 - dims)
    for a in range(3):
        lat.min_corner[a] = mc[a]
        lat.widths[a] = w[a]
    lat.edge = float(edge_length)
    lat.shifts[0] = w[0]
    lat.shifts[1] = w[0] + w[1]
    return lat


class VoxelFilter(object):
    """cubic grid of `edge_length` enclosing a 2-D or 3-D cloud; converts coordinates to packed 64-bit
    grid addresses and back (geometry.py:16-22)."""

    def __init__(self, points, edge_length, device=None):
        ndim = points.ndim if hasattr(points, "ndim") else np.asarray(points).ndim
        shape = tuple(points.shape) if hasattr(points, "shape") else np.asarray(points).shape
        if ndim != 2:
            raise ValueError("wrong point cloud array shape")
        if shape[1] not in (2, 3):
            raise ValueError("only 2D and 3D spaces supported")
        if shape[0] < 2:
            raise ValueError("need at least 2 points to define a voxel grid")
        self._dims = shape[1]
        self._rt, cloud = _device.as_cloud(_pad3(points), device)
        lo, hi = _device.cloud_bounds(self._rt, cloud)
        self._init_from_bounds(lo[:self._dims], hi[:self._dims], edge_length)

    @classmethod
    def from_bounds(cls, minimum, maximum, edge_length, device=None):
        """a filter for a cloud whose per-axis extrema are already known (one bounds pass serves every
        scale of the ladder, and every rank of a multi-GPU job uses the global extrema)."""
        self = cls.__new__(cls)
        self._dims = len(minimum)
        self._rt = _device.get_runtime(device)
        self._init_from_bounds(np.asarray(minimum), np.asarray(maximum), edge_length)
        return self

    def _init_from_bounds(self, lo, hi, edge_length):
        self.edge_length = edge_length
        self.minimum_corner, self.maximum_corner, self.widths, self.shifts = \
            lattice_parameters(lo, hi, edge_length)
        if np.any(self.widths < 1):
            # the reference fails here too: int("0b" + "1"*0, 2) at geometry.py:74
            raise ValueError("cloud has no extent beyond one voxel on some axis")
        masks = [(1 << int(w)) - 1 for w in self.widths]
        for num, shift in enumerate(self.shifts):
            masks[num + 1] = masks[num + 1] << int(shift)
        self.masks = masks
        self._lat = make_nm_lattice(self.minimum_corner, edge_length, self.widths)

    # ----------------------------------------------------------------------------------------------

    @property
    def nm_lattice(self):
        return self._lat

    def _check_in_bounds(self, points):
        """geometry.py:83-99: shape checks, then min_corner <= points <= max_corner."""
        if isinstance(points, torch.Tensor):
            if points.ndim == 1:
                points = points.reshape(1, -1)
        else:
            points = np.atleast_2d(points)
        if points.ndim != 2:
            raise ValueError("wrong array shape")
        if points.shape[1] != self.shifts.size + 1:
            raise ValueError("wrong number of spatial dimensions")
        rt, cloud = _device.as_cloud(_pad3(points), self._rt.device)
        lo, hi = _device.cloud_bounds(rt, cloud)
        d = self._dims
        if np.any(lo[:d] < self.minimum_corner) or np.any(hi[:d] > self.maximum_corner):
            raise ValueError("some points fall outside filter bounding region")
        return cloud

    def coordinate_to_address(self, points):
        """real-world coordinates -> integer voxel addresses, one per point (geometry.py:103-116)."""
        as_torch = isinstance(points, torch.Tensor)
        cloud = self._check_in_bounds(points)
        rt = self._rt
        n = cloud.shape[0]
        addr = torch.empty(n, dtype=torch.int64, device=rt.device)
        rt.check(rt.lib.nm_coordinate_to_address(
            rt.ctx, _device.ptr(cloud), n, _device.row_stride(cloud), ctypes.byref(self._lat),
            _device.ptr(addr), ctypes.c_void_p(0), rt.stream()))
        return addr if as_torch else addr.cpu().numpy()

    def address_to_coordinate(self, addresses):
        """integer addresses -> voxel centre coordinates (geometry.py:120-138)."""
        as_torch = isinstance(addresses, torch.Tensor)
        rt = self._rt
        if as_torch:
            addr = addresses.to(device=rt.device, dtype=torch.int64).reshape(-1).contiguous()
        else:
            addr = torch.from_numpy(
                np.ascontiguousarray(np.atleast_1d(addresses), dtype=np.int64)).to(rt.device)
        m = addr.shape[0]
        out = torch.empty((m, 3), dtype=torch.float64, device=rt.device)
        rt.check(rt.lib.nm_address_to_coordinate(rt.ctx, _device.ptr(addr), m,
                                                 ctypes.byref(self._lat), _device.ptr(out),
                                                 rt.stream()))
        out = out[:, :self._dims]
        return out if as_torch else out.cpu().numpy()

    def unique_addresses(self, points):
        """sorted distinct addresses of the occupied voxels: coordinate_to_address + numpy.unique
        (geometry.py:148-150).  position in this array is the reference's search-voxel index."""
        as_torch = isinstance(points, torch.Tensor)
        cloud = self._check_in_bounds(points)
        addr = self._unique_addresses_device(cloud)
        return addr if as_torch else addr.cpu().numpy()

    def _unique_addresses_device(self, cloud):
        rt = self._rt
        n = cloud.shape[0]
        out = torch.empty(n, dtype=torch.int64, device=rt.device)
        count = torch.zeros(2, dtype=torch.int64, device=rt.device)
        nbytes = rt.lib.nm_voxelize_workspace_bytes(n)
        work = rt.workspace(nbytes)
        rt.check(rt.lib.nm_voxelize(rt.ctx, _device.ptr(cloud), n, _device.row_stride(cloud),
                                    ctypes.byref(self._lat), _device.ptr(out), _device.ptr(count),
                                    _device.ptr(work), work.numel(), rt.stream()))
        m, oob = (int(v) for v in count.cpu())
        if oob:
            raise ValueError("some points fall outside filter bounding region")
        return out[:m]

    def unique_voxels(self, points):
        """unique centre coordinates of all grid cells that contain a point (geometry.py:142-154)."""
        as_torch = isinstance(points, torch.Tensor)
        cloud = self._check_in_bounds(points)
        centres = self.address_to_coordinate(self._unique_addresses_device(cloud))
        return centres if as_torch else centres.cpu().numpy()

    def find_neighbors(self, address):
        raise NameError("find_neighbors not implemented yet")          # geometry.py:158-164

    def find_facing_neighbors(self, address):
        raise NameError("find_facing_neighbors not implemented yet")   # geometry.py:166-172


def _pad3(points):
    """2-D clouds ride through the 3-D kernels with a zero z column."""
    if isinstance(points, torch.Tensor):
        if points.shape[1] == 2:
            return torch.cat((points, torch.zeros_like(points[:, :1])), dim=1)
        return points
    points = np.asarray(points)
    if points.shape[1] == 2:
        return np.concatenate((points, np.zeros((points.shape[0], 1), dtype=points.dtype)), axis=1)
    return points
