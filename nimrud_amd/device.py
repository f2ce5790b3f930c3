"""
device plumbing: one nm_ctx per GPU, a grow-only scratch buffer, host<->HBM movement of clouds.
PyTorch-ROCm is used for exactly three things here: device memory (tensors as buffers), streams, and
(in parallel.py) torch.distributed.  all arithmetic happens in libnimrud_hip.so.
"""

import ctypes
import threading

import numpy as np
import torch

from nimrud_amd import _ffi

_runtimes = {}
_lock = threading.Lock()


class Runtime(object):
    """per-device state: the library handle, its nm_ctx and a scratch buffer in HBM."""

    def __init__(self, index):
        self.lib = _ffi.load()
        self.index = index
        self.device = torch.device("cuda", index)
        ctx = ctypes.c_void_p()
        rc = self.lib.nm_create(ctypes.byref(ctx), index)
        if rc != _ffi.NM_OK:
            raise _ffi.NimrudHipError("nm_create(device=%d) failed with status %d" % (index, rc))
        self.ctx = ctx
        self._work = None

    def check(self, rc):
        if rc == _ffi.NM_ERR_LATTICE:
            # a lattice a kernel of an EARLIER call could not address, met at the entry of this one (NM_ENTER),
            # or this call's own: either way a property of one call's arguments, not of the context - raise it
            # once as the ValueError of VoxelFilter.__init__ (geometry.py:59-60) and leave the context usable
            msg = self.lib.nm_last_error(self.ctx)
            msg = msg.decode("utf-8", "replace") if msg else "lattice cannot be addressed"
            self.lib.nm_clear_error(self.ctx)
            raise ValueError(msg)
        _ffi.check(self.lib, self.ctx, rc)

    def check_async(self, wait=True):
        """raise what a kernel of an earlier call reported (nm_check): an occupancy index that timed out
        or overflowed, a lattice built on the device that cannot be addressed.  wait=True waits for the
        status snapshot of the last call - callers that have just synchronised pay nothing."""
        self.check(self.lib.nm_check(self.ctx, 1 if wait else 0))

    def clear_error(self):
        self.check(self.lib.nm_clear_error(self.ctx))

    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, nbytes):
        """a byte buffer of at least `nbytes` (grow-only; reused across scales and calls)"""
        if self._work is None or self._work.numel() < nbytes:
            self._work = None
            self._work = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._work

    def release_workspace(self):
        self._work = None


def get_runtime(device=None):
    """the Runtime of a GPU.  there is no CPU fallback: without a visible GPU this raises."""
    if not torch.cuda.is_available():
        raise RuntimeError("nimrud_amd needs an AMD GPU (MI355X / gfx950); no device is visible "
                           "and there is no CPU fallback")
    if device is None:
        index = torch.cuda.current_device()
    elif isinstance(device, int):
        index = device
    else:
        device = torch.device(device)
        index = device.index if device.index is not None else torch.cuda.current_device()
    with _lock:
        rt = _runtimes.get(index)
        if rt is None:
            rt = Runtime(index)
            _runtimes[index] = rt
    return rt


def ptr(tensor):
    return ctypes.c_void_p(tensor.data_ptr()) if tensor is not None else ctypes.c_void_p(0)


def as_cloud(points, device=None):
    """a cloud as an fp64 row-major tensor in HBM.  numpy arrays are uploaded; torch tensors already
    on the GPU are used in place when they are fp64 with unit column stride.
    all point clouds are 2-D, one row per point, geometry in the first three columns
    (minimal/README.md:38-40).  the reference computes in numpy's default fp64; fp32 input is
    promoted here so that lattice cells match (caveat i)."""
    if isinstance(points, torch.Tensor):
        t = points
        if t.ndim != 2:
            raise ValueError("wrong point cloud array shape")
        rt = get_runtime(t.device if t.is_cuda else device)
        if (not t.is_cuda) or t.dtype != torch.float64 or (t.shape[0] > 1 and t.stride(1) != 1) \
                or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
            t = t.to(device=rt.device, dtype=torch.float64).contiguous()
        return rt, t
    arr = np.asarray(points)
    if arr.ndim != 2:
        raise ValueError("wrong point cloud array shape")
    rt = get_runtime(device)
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    return rt, torch.from_numpy(arr).to(rt.device)


def row_stride(t):
    return int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])


def cloud_bounds_device(rt, t):
    """6 doubles {min xyz, max xyz} of the first three columns, left on the device."""
    out = torch.empty(6, dtype=torch.float64, device=rt.device)
    rt.check(rt.lib.nm_bounds(rt.ctx, ptr(t), t.shape[0], row_stride(t), ptr(out), rt.stream()))
    return out


def cloud_bounds(rt, t):
    """(min xyz, max xyz) of the first three (or two) columns, computed on the device."""
    mm = cloud_bounds_device(rt, t).cpu().numpy()
    return mm[:3].copy(), mm[3:].copy()
