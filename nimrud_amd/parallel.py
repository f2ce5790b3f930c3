"""
multi-GPU execution of the multiscale pipeline: one process per GPU, one spatial tile per rank, halo
exchange over RCCL, then the single-GPU scale loop.

the reference is single-process; what it does have is the idea - its legacy partitioners pair each
query tile with a search tile grown by the largest scale (prototypes/mso.py:892-927,
prototypes/apc.py:399-428,595, utils/geometry.py:203-253).  here:

  1. every rank holds one tile of the cloud (its rows are its query points): a Morton-contiguous run
     of the cloud (partition_by_morton; what BASELINE's north_star names) or any other subset.
  2. all-gather (6 doubles per rank): every tile's bounding box.  their per-axis min / max are the
     GLOBAL extrema, so every rank builds the same lattice as a single-process run on the whole cloud
     (geometry.py:37: min_corner = min - e/2).
  3. who needs what: rank j needs the points of other tiles that lie within
     margin = max_s(radius_s + sqrt(3)/2 * edge_s) of one of ITS points - a voxel centre within radius of
     one of j's query points can only be occupied by points that close.  decided per destination either
     by j's bounding box grown by the margin (halo="boxes") or by j's set of occupied coarse cells dilated
     by the margin (halo="cells", the default: Morton runs are L-shaped, their boxes cover far more than
     they do; the cell sets are one 256 KB all-gather).
  4. all-gather of the pair counts, ONE host synchronisation, then an all-to-all-v of 24-byte rows
     (grouped ncclSend / ncclRecv: each pair of neighbouring tiles talks over its own xGMI link; nothing
     is reduced).
  5. the scale loop on [own tile | received halo] with the global lattice; queries are the leading
     rows of that buffer, so it is sorted and indexed once.

features are bit-identical to a single-GPU run over the whole cloud (integer moments of identical
voxel sets); they stay on the owning rank, rows aligned with its tile.

two transports:
  * RcclComm - the product: steps 2-4 are ONE call into libnimrud_hip.so (nm_halo_exchange) on a
    communicator the library created itself; torch.distributed only carries the 128-byte unique id to
    the other ranks (any backend, gloo is enough).
  * torch.distributed collectives driven from here (exchange_halo) - the same protocol step by step.
    this is what the CPU tests run over gloo with a numpy backend in place of the kernels, and what lets
    two ranks share one GPU in a rehearsal (RCCL needs one GPU per rank).
"""

import ctypes
import math

import numpy as np
import torch
import torch.distributed as dist

from nimrud_amd import device as _device

CELLSET_WORDS = 65536           # NM_HALO_CELLSET_WORDS: 2^21 coarse cells, one bit each
HALO_BOXES, HALO_CELLS, HALO_INCLUDE_SELF, HALO_REUSE_PLAN = 0, 1, 4, 8


def halo_margin(edge_lengths, radii):
    """distance beyond a tile from which search points can still matter."""
    return max(r + 0.5 * math.sqrt(3.0) * e * (1.0 + 1e-12) for e, r in zip(edge_lengths, radii))


def coarse_grid(global_minmax, margin):
    """the coarse grid of the cell-set halos, as csrc/nm_halo.hip's nm_coarse_grid computes it: cubic
    cells of edge margin/4 over the global box unless that needs more than 2^21 cells.  returns
    (lo (3,), inverse cell edge, dims (3,), dilation D in cells).  host mirror for the numpy test backend
    and for sizing estimates; the kernels evaluate the same expressions on the device."""
    g = np.asarray(global_minmax, dtype=np.float64)
    ext = np.maximum(g[3:] - g[:3], 0.0)
    e = margin * 0.25 if margin > 0.0 else 1.0
    for _ in range(400):
        if np.prod(np.floor(ext / e) + 1.0) <= float(1 << 21):
            break
        e *= 1.25
    inv_e = 1.0 / e
    dims = (np.floor(ext / e) + 1.0).astype(np.int64)
    dilation = int(math.floor(margin * inv_e * (1.0 + 1e-12))) + 1
    return g[:3].copy(), inv_e, dims, dilation


class RcclComm(object):
    """an RCCL communicator owned by libnimrud_hip.so (nm_comm_create), one rank per process and GPU.
    `broadcast` carries the 128-byte unique id from rank 0 to the others: by default
    torch.distributed.broadcast_object_list on the default group (gloo is enough) - the only thing the
    multi-GPU path uses torch.distributed for."""

    def __init__(self, rank=None, world=None, device=None, broadcast=None):
        from nimrud_amd import _ffi
        self.rt = _device.get_runtime(device)
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world = int(rank), int(world)
        uid = ctypes.create_string_buffer(_ffi.NM_COMM_ID_BYTES)
        failed = 0
        if self.rank == 0:
            failed = self.rt.lib.nm_comm_unique_id(uid)
        if self.world > 1:
            # rank 0 broadcasts in any case - the id, or the word that it has none - so that the other ranks
            # never wait for a rank that has raised
            message = bytes(uid.raw) if failed == _ffi.NM_OK else b"FAILED %d" % failed
            if broadcast is None:
                box = [message]
                dist.broadcast_object_list(box, src=0)
                payload = box[0]
            else:
                payload = broadcast(message)
            if payload.startswith(b"FAILED"):
                raise _ffi.NimrudHipError("rank 0: nm_comm_unique_id failed with status %s"
                                          % payload[7:].decode("ascii", "replace"))
            uid = ctypes.create_string_buffer(payload, _ffi.NM_COMM_ID_BYTES)
        elif failed != _ffi.NM_OK:
            raise _ffi.NimrudHipError("nm_comm_unique_id failed with status %d" % failed)
        handle = ctypes.c_void_p()
        self.rt.check(self.rt.lib.nm_comm_create(self.rt.ctx, self.world, self.rank, uid,
                                                 ctypes.byref(handle)))
        self.handle = handle

    def close(self):
        if self.handle:
            self.rt.lib.nm_comm_destroy(self.rt.ctx, self.handle)
            self.handle = ctypes.c_void_p()


class HipBackend(object):
    """the data-path operations of a tile, on the GPU through the C ABI."""

    def __init__(self, device=None):
        self.rt = _device.get_runtime(device)
        self.device = self.rt.device

    def bounds(self, cloud):
        return _device.cloud_bounds_device(self.rt, cloud)

    def cellset(self, cloud, global_minmax, margin):
        """this tile's dilated coarse cell set: int32 (CELLSET_WORDS,) on the device."""
        rt = self.rt
        out = torch.empty(CELLSET_WORDS, dtype=torch.int32, device=self.device)
        nbytes = rt.lib.nm_halo_cellset_workspace_bytes()
        work = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        rt.check(rt.lib.nm_halo_cellset(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                        _device.row_stride(cloud), _device.ptr(global_minmax),
                                        float(margin), _device.ptr(out), _device.ptr(work),
                                        work.numel(), rt.stream()))
        return out

    def halo_count(self, cloud, dest, skip):
        """rows of `cloud` per destination.  dest = (n,6) grown boxes, or (global_minmax, margin,
        cellsets (n, CELLSET_WORDS))."""
        rt = self.rt
        if isinstance(dest, tuple):
            glob, margin, sets = dest
            counts = torch.empty(sets.shape[0], dtype=torch.int64, device=self.device)
            rt.check(rt.lib.nm_halo_count_cells(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                                _device.row_stride(cloud), _device.ptr(glob),
                                                float(margin), _device.ptr(sets), sets.shape[0], skip,
                                                _device.ptr(counts), rt.stream()))
            return counts
        counts = torch.empty(dest.shape[0], dtype=torch.int64, device=self.device)
        rt.check(rt.lib.nm_halo_count(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                      _device.row_stride(cloud), _device.ptr(dest), dest.shape[0],
                                      skip, _device.ptr(counts), rt.stream()))
        return counts

    def halo_pack(self, cloud, dest, skip, offsets, total):
        rt = self.rt
        out = torch.empty((max(total, 1), 3), dtype=torch.float64, device=self.device)
        n_dest = dest[2].shape[0] if isinstance(dest, tuple) else dest.shape[0]
        cursor = torch.empty(n_dest, dtype=torch.int64, device=self.device)
        if isinstance(dest, tuple):
            glob, margin, sets = dest
            rt.check(rt.lib.nm_halo_pack_cells(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                               _device.row_stride(cloud), _device.ptr(glob),
                                               float(margin), _device.ptr(sets), n_dest, skip,
                                               _device.ptr(offsets), _device.ptr(cursor),
                                               _device.ptr(out), rt.stream()))
        else:
            rt.check(rt.lib.nm_halo_pack(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                         _device.row_stride(cloud), _device.ptr(dest), n_dest,
                                         skip, _device.ptr(offsets), _device.ptr(cursor),
                                         _device.ptr(out), rt.stream()))
        return out[:total]

    def copy_xyz(self, cloud, out):
        rt = self.rt
        rt.check(rt.lib.nm_copy_xyz(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                    _device.row_stride(cloud), _device.ptr(out), rt.stream()))

    def features(self, search, n_query, bounds, edge_lengths, radii, out, info):
        """the scale ladder (one library call): queries are the first n_query rows of `search`;
        `bounds` = the global extrema, 6 doubles on the device (they never visit the host)."""
        from nimrud_amd.minimal import multiscale
        multiscale._ladder_into(self.rt, search[:n_query], search, True, bounds, edge_lengths, radii,
                                out, info)


class TilePlan(object):
    """one rank's share of a multi-GPU job: `cloud` is this rank's tile, (N, >=3) fp64 on this rank's
    device (a numpy array is uploaded).  edge_lengths / radii as in process_single_core.
    comm: an RcclComm -> the exchange is one nm_halo_exchange call; None -> torch.distributed
    collectives on `group` (or a single rank).  halo: "cells" (default) or "boxes"."""

    def __init__(self, cloud, edge_lengths, radii, group=None, backend=None, comm=None, halo="cells"):
        assert len(edge_lengths) == len(radii), \
            "edge_lengths and radii should be equal-length sequences."
        if halo not in ("cells", "boxes"):
            raise ValueError("halo must be 'cells' or 'boxes'")
        self.backend = backend if backend is not None else HipBackend(
            cloud.device if isinstance(cloud, torch.Tensor) and cloud.is_cuda else None)
        if isinstance(self.backend, HipBackend):
            cloud = _device.as_cloud(cloud, self.backend.device)[1]
        self.cloud = cloud
        self.edge_lengths = list(edge_lengths)
        self.radii = list(radii)
        self.group = group
        self.comm = comm
        self.halo = halo
        if comm is not None:
            self.rank, self.world = comm.rank, comm.world
        else:
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.margin = halo_margin(self.edge_lengths, self.radii) if self.edge_lengths else 1.0
        # RCCL moves device buffers directly.  a gloo group (CPU rehearsals on a one-GPU box) cannot,
        # so device tensors are staged through host memory for the collectives only.
        self.stage_on_host = bool(comm is None and dist.is_initialized()
                                  and dist.get_backend(group) == "gloo"
                                  and isinstance(cloud, torch.Tensor) and cloud.is_cuda)
        self._info = None
        self.want_info = False          # per-scale counters (a pass over the index): only on request
        self._search_points = cloud.shape[0]
        self.halo_sent = 0
        self.halo_received = 0
        self.include_self = False       # testing aid for one-rank communicators (NM_HALO_INCLUDE_SELF)
        # static = True: the caller promises that NO rank's tile changes between steps (every rank sets it or
        # none).  the RCCL exchange then keeps the plan of its first step - boxes, cell sets, pair counts - and
        # later steps pack and exchange the halo rows again without the all-gathers and without the host
        # synchronisation that learns the sizes (NM_HALO_REUSE_PLAN): a step enqueues and returns
        self.static = False
        self._planned = False
        # the tile's coordinates live at the front of a (N + slack, 3) buffer; halo rows are received
        # straight behind them, so a step never copies the tile
        self._buffer = None
        self._work = None
        self._send_rows = max(1024, cloud.shape[0] // 8)
        self._ensure_buffer(max(1024, cloud.shape[0] // 8))

    def _ensure_buffer(self, halo_rows):
        n = self.cloud.shape[0]
        if self._buffer is not None and self._buffer.shape[0] >= n + halo_rows:
            return
        self._buffer = torch.empty((n + halo_rows + halo_rows // 4, 3), dtype=torch.float64,
                                   device=self.cloud.device)
        self.backend.copy_xyz(self.cloud, self._buffer[:n])

    def last_info(self):
        """ScaleInfo per scale of the last process_tile call made with plan.want_info = True"""
        from nimrud_amd.minimal.multiscale import ScaleInfo
        if self._info is None:
            raise RuntimeError("set plan.want_info = True before the process_tile call to be inspected")
        host = self._info.cpu().numpy()
        return [ScaleInfo(row) for row in host[:len(self.edge_lengths)]]

    def search_points(self):
        return self._search_points


def exchange_halo_rccl(plan):
    """steps 2-4 as ONE library call (nm_halo_exchange).  returns (global extrema: 6 doubles on the
    device, H); the H received halo rows sit in plan._buffer[N:N+H].  the call synchronises the stream
    once, inside the library, to learn the sizes; when a buffer of some rank is too small every rank
    learns so before anything is sent, grows its own and all call again."""
    from nimrud_amd import _ffi
    rt = plan.backend.rt
    cloud, n = plan.cloud, plan.cloud.shape[0]
    glob = torch.empty(6, dtype=torch.float64, device=cloud.device)
    mode = (HALO_CELLS if plan.halo == "cells" else HALO_BOXES) | \
        (HALO_INCLUDE_SELF if plan.include_self else 0)
    sent, received = ctypes.c_int64(0), ctypes.c_int64(0)
    if plan.static and plan._planned and plan._work is not None and \
            plan._planned == (mode, n, plan._work.data_ptr(), plan._buffer.data_ptr()):
        capacity = plan._buffer.shape[0] - n
        rt.check(rt.lib.nm_halo_exchange(
            rt.ctx, plan.comm.handle, plan.world, plan.rank, _device.ptr(cloud), n,
            _device.row_stride(cloud), float(plan.margin), mode | HALO_REUSE_PLAN,
            ctypes.c_void_p(plan._buffer.data_ptr() + 24 * n), capacity,
            ctypes.byref(received), ctypes.byref(sent), _device.ptr(glob),
            _device.ptr(plan._work), plan._work.numel(), rt.stream()))
        plan.halo_sent, plan.halo_received = int(sent.value), int(received.value)
        return glob, int(received.value)
    plan._planned = False
    for _ in range(8):
        nbytes = rt.lib.nm_halo_workspace_bytes(int(plan._send_rows), plan.world)
        if plan._work is None or plan._work.numel() < nbytes:
            plan._work = torch.empty(int(nbytes), dtype=torch.uint8, device=cloud.device)
        capacity = plan._buffer.shape[0] - n
        rc = rt.lib.nm_halo_exchange(
            rt.ctx, plan.comm.handle, plan.world, plan.rank, _device.ptr(cloud), n,
            _device.row_stride(cloud), float(plan.margin), mode,
            ctypes.c_void_p(plan._buffer.data_ptr() + 24 * n), capacity,
            ctypes.byref(received), ctypes.byref(sent), _device.ptr(glob),
            _device.ptr(plan._work), plan._work.numel(), rt.stream())
        if rc != _ffi.NM_ERR_WORKSPACE:
            rt.check(rc)
            plan.halo_sent, plan.halo_received = int(sent.value), int(received.value)
            plan._planned = (mode, n, plan._work.data_ptr(), plan._buffer.data_ptr())
            return glob, int(received.value)
        # some rank (maybe this one) is short of room: every rank is here; grow what this one lacks
        if sent.value > plan._send_rows:
            plan._send_rows = int(sent.value) + int(sent.value) // 4
        if received.value > capacity:
            plan._ensure_buffer(int(received.value))
    raise _ffi.NimrudHipError("halo exchange buffers did not settle")


def exchange_halo(plan):
    """steps 2-4 over torch.distributed collectives (gloo rehearsals, CPU tests, single rank).
    returns (global extrema as a (6,) tensor on the cloud's device, H)."""
    be, cloud, group = plan.backend, plan.cloud, plan.group
    dev = cloud.device
    local = be.bounds(cloud)                               # (6,) lo xyz, hi xyz on the device
    if plan.world == 1:
        return local, 0
    cdev = torch.device("cpu") if plan.stage_on_host else dev    # where the collectives run
    local_c = local.to(cdev).contiguous()
    # one all-gather serves both purposes: every tile's box, and (their min / max) the global extrema
    gathered = torch.empty((plan.world, 6), dtype=torch.float64, device=cdev)
    try:
        dist.all_gather_into_tensor(gathered.reshape(-1), local_c, group=group)
    except (RuntimeError, NotImplementedError, AttributeError):
        box_list = [torch.empty(6, dtype=torch.float64, device=cdev) for _ in range(plan.world)]
        dist.all_gather(box_list, local_c, group=group)
        gathered = torch.stack(box_list)
    glob = torch.cat((gathered[:, :3].min(dim=0).values, gathered[:, 3:].max(dim=0).values)).to(dev)
    if plan.halo == "cells":
        own = be.cellset(cloud, glob, plan.margin).to(cdev).contiguous()
        set_list = [torch.empty_like(own) for _ in range(plan.world)]
        dist.all_gather(set_list, own, group=group)
        dest = (glob, plan.margin, torch.stack(set_list).to(dev).contiguous())
    else:
        boxes = gathered.to(dev).clone()
        boxes[:, :3] -= plan.margin
        boxes[:, 3:] += plan.margin
        dest = boxes
    send_counts = be.halo_count(cloud, dest, plan.rank).to(cdev)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    # the one read-back: the pair counts
    host = torch.cat((send_counts, recv_counts)).cpu()
    w = plan.world
    send_list = [int(v) for v in host[:w]]
    recv_list = [int(v) for v in host[w:2 * w]]
    offsets = torch.zeros(w, dtype=torch.int64)
    offsets[1:] = torch.cumsum(torch.tensor(send_list[:-1], dtype=torch.int64), 0)
    packed = be.halo_pack(cloud, dest, plan.rank, offsets.to(dev), sum(send_list)).to(cdev)
    n, h = cloud.shape[0], sum(recv_list)
    plan._ensure_buffer(h)
    if plan.stage_on_host:
        recv = torch.empty((h, 3), dtype=torch.float64, device=cdev)
    else:
        recv = plan._buffer[n:n + h]                       # the halo lands behind the tile
    dist.all_to_all_single(recv.reshape(-1), packed.reshape(-1),
                           [3 * c for c in recv_list], [3 * c for c in send_list], group=group)
    if plan.stage_on_host:
        plan._buffer[n:n + h] = recv.to(dev)
    plan.halo_sent, plan.halo_received = sum(send_list), h
    return glob, h


def process_tile(plan, out=None):
    """features of this rank's tile, (N, 4*S) fp64 on this rank's device, rows aligned with the
    tile.  collective: every rank of the group must call it."""
    be, cloud = plan.backend, plan.cloud
    n = cloud.shape[0]
    n_scales = len(plan.edge_lengths)
    if plan.comm is not None:
        bounds, n_halo = exchange_halo_rccl(plan)
    else:
        bounds, n_halo = exchange_halo(plan)
    search = plan._buffer[:n + n_halo]
    plan._search_points = search.shape[0]
    if out is None:
        out = torch.empty((n, 4 * n_scales), dtype=torch.float64, device=cloud.device)
    info = torch.zeros((max(n_scales, 1), 4), dtype=torch.int64, device=cloud.device) \
        if plan.want_info else None
    if n > 0:      # an empty tile took part in the exchange (the others may need nothing from it) and has no rows
        be.features(search, n, bounds, plan.edge_lengths, plan.radii, out, info)
    plan._info = info
    return out


def process_multi_gpu(tile_cloud, edge_lengths, radii, group=None, comm=None, halo="cells"):
    """convenience wrapper: this rank's tile in (numpy or torch), this rank's features out (same kind).
    one rank per GPU; pass an RcclComm (the product path), or initialise torch.distributed and let the
    collectives run on `group`."""
    as_torch = isinstance(tile_cloud, torch.Tensor)
    plan = TilePlan(tile_cloud, edge_lengths, radii, group=group, comm=comm, halo=halo)
    out = process_tile(plan)
    if as_torch:
        return out
    host = out.cpu().numpy()
    plan.backend.rt.check_async(wait=True)
    return host


def partition_tiles(points, world):
    """split a host cloud into `world` spatially compact tiles of (nearly) equal point count by
    recursive median bisection along the longest axis.  the tiles' bounding boxes are disjoint.
    returns a list of sorted index arrays."""
    xyz = np.asarray(points)[:, :3]

    def split(idx, parts):
        if parts == 1:
            return [np.sort(idx)]
        left_parts = parts // 2
        sub = xyz[idx]
        axis = int(np.argmax(sub.max(0) - sub.min(0)))
        k = int(round(len(idx) * left_parts / parts))
        order = np.argsort(sub[:, axis], kind="stable")
        return split(idx[order[:k]], left_parts) + split(idx[order[k:]], parts - left_parts)

    return split(np.arange(len(xyz)), world)


def partition_by_morton(points, world, edge_length):
    """split a host cloud into `world` Morton-contiguous runs of (nearly) equal point count (how a
    Morton-ordered archive is cut into files, and how BASELINE's north_star shards the cloud).  Morton
    runs can be L-shaped: use them with halo="cells" (the default), their bounding boxes are much larger
    than they are.  returns a list of index arrays."""
    from nimrud_amd import synth
    order = synth.morton_sort(np.asarray(points)[:, :3], edge_length)
    return [np.sort(chunk) for chunk in np.array_split(order, world)]
