"""
multi-GPU execution of the multiscale pipeline: one process per GPU, one spatial tile per rank, halo
exchange over RCCL (torch.distributed backend "nccl" on ROCm) and then the single-GPU scale loop.

the reference is single-process; what it does have is the idea - its legacy partitioners pair each
query tile with a search tile grown by the largest scale (prototypes/mso.py:892-927,
prototypes/apc.py:399-428,595, utils/geometry.py:203-253).  here:

  1. every rank holds one tile of the cloud (its rows are its query points).
  2. all-gather (6 doubles per rank): every tile's bounding box.  their per-axis min / max are the
     GLOBAL extrema, so every rank builds the same lattice as a single-process run on the whole cloud
     (geometry.py:37: min_corner = min - e/2).
  3. all-to-all (one int64 per pair): how many halo rows each pair will exchange.
  4. all-to-all-v: each rank sends rank j the points of its tile that lie within
     margin = max_s(radius_s + sqrt(3)/2 * edge_s) of j's box.  a voxel centre within radius of one of
     j's query points can only be occupied by points that close, so after the exchange every voxel j
     can see is occupied on j exactly when it is occupied in the global cloud.  point-to-point volume:
     each pair talks over its own xGMI link; nothing is reduced.
  5. the scale loop on [own tile | received halo] with the global lattice; queries are the leading
     rows of that buffer, so it is sorted and indexed once per scale.

features are bit-identical to a single-GPU run over the whole cloud (integer moments of identical
voxel sets); they stay on the owning rank, rows aligned with its tile.

`backend` is the seam the CPU tests use: HipBackend (the product) drives libnimrud_hip.so; the tests
substitute a numpy backend to exercise the collectives over gloo on machines without a GPU.
"""

import ctypes
import math

import numpy as np
import torch
import torch.distributed as dist

from nimrud_amd import device as _device


def halo_margin(edge_lengths, radii):
    """distance beyond a tile's bounding box from which search points can still matter."""
    return max(r + 0.5 * math.sqrt(3.0) * e * (1.0 + 1e-12) for e, r in zip(edge_lengths, radii))


class HipBackend(object):
    """the data-path operations of a tile, on the GPU through the C ABI."""

    def __init__(self, device=None):
        self.rt = _device.get_runtime(device)
        self.device = self.rt.device

    def bounds(self, cloud):
        return _device.cloud_bounds_device(self.rt, cloud)

    def halo_count(self, cloud, boxes, skip):
        rt = self.rt
        counts = torch.empty(boxes.shape[0], dtype=torch.int64, device=self.device)
        rt.check(rt.lib.nm_halo_count(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                      _device.row_stride(cloud), _device.ptr(boxes), boxes.shape[0],
                                      skip, _device.ptr(counts), rt.stream()))
        return counts

    def halo_pack(self, cloud, boxes, skip, offsets, total):
        rt = self.rt
        out = torch.empty((max(total, 1), 3), dtype=torch.float64, device=self.device)
        cursor = torch.empty(boxes.shape[0], dtype=torch.int64, device=self.device)
        rt.check(rt.lib.nm_halo_pack(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                     _device.row_stride(cloud), _device.ptr(boxes), boxes.shape[0],
                                     skip, _device.ptr(offsets), _device.ptr(cursor),
                                     _device.ptr(out), rt.stream()))
        return out[:total]

    def copy_xyz(self, cloud, out):
        rt = self.rt
        rt.check(rt.lib.nm_copy_xyz(rt.ctx, _device.ptr(cloud), cloud.shape[0],
                                    _device.row_stride(cloud), _device.ptr(out), rt.stream()))

    def features(self, search, n_query, lo, hi, edge_lengths, radii, out, info):
        """the scale ladder (one library call): queries are the first n_query rows of `search`."""
        from nimrud_amd.minimal import multiscale
        multiscale._ladder_into(self.rt, search[:n_query], search, True, lo, hi, edge_lengths, radii,
                                out, info)


class TilePlan(object):
    """one rank's share of a multi-GPU job: `cloud` is this rank's tile, (N, >=3) fp64 on this rank's
    device (a numpy array is uploaded).  edge_lengths / radii as in process_single_core."""

    def __init__(self, cloud, edge_lengths, radii, group=None, backend=None):
        assert len(edge_lengths) == len(radii), \
            "edge_lengths and radii should be equal-length sequences."
        self.backend = backend if backend is not None else HipBackend(
            cloud.device if isinstance(cloud, torch.Tensor) and cloud.is_cuda else None)
        if isinstance(self.backend, HipBackend):
            cloud = _device.as_cloud(cloud, self.backend.device)[1]
        self.cloud = cloud
        self.edge_lengths = list(edge_lengths)
        self.radii = list(radii)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.margin = halo_margin(self.edge_lengths, self.radii) if self.edge_lengths else 0.0
        # RCCL moves device buffers directly.  a gloo group (CPU rehearsals on a one-GPU box) cannot,
        # so device tensors are staged through host memory for the collectives only.
        self.stage_on_host = bool(dist.is_initialized() and dist.get_backend(group) == "gloo"
                                  and isinstance(cloud, torch.Tensor) and cloud.is_cuda)
        self._info = None
        self._search_points = cloud.shape[0]
        self.halo_sent = 0
        self.halo_received = 0
        # the tile's coordinates live at the front of a (N + slack, 3) buffer; halo rows are received
        # straight behind them, so a step never copies the tile
        self._buffer = None
        self._ensure_buffer(max(1024, cloud.shape[0] // 8))

    def _ensure_buffer(self, halo_rows):
        n = self.cloud.shape[0]
        if self._buffer is not None and self._buffer.shape[0] >= n + halo_rows:
            return
        self._buffer = torch.empty((n + halo_rows + halo_rows // 4, 3), dtype=torch.float64,
                                   device=self.cloud.device)
        self.backend.copy_xyz(self.cloud, self._buffer[:n])

    def last_info(self):
        from nimrud_amd.minimal.multiscale import ScaleInfo
        host = self._info.cpu().numpy()
        return [ScaleInfo(row) for row in host[:len(self.edge_lengths)]]

    def search_points(self):
        return self._search_points


def exchange_halo(plan):
    """steps 2-4: returns (global lo, global hi, H); the H received halo rows sit in
    plan._buffer[N:N+H].  one host synchronisation: the split sizes of the all-to-all-v and the global
    extrema are read back together."""
    be, cloud, group = plan.backend, plan.cloud, plan.group
    dev = cloud.device
    local = be.bounds(cloud)                               # (6,) lo xyz, hi xyz on the device
    if plan.world == 1:
        mm = local.cpu().numpy()
        return mm[:3], mm[3:], 0
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
    glo = gathered[:, :3].min(dim=0).values
    ghi = gathered[:, 3:].max(dim=0).values
    boxes = gathered.to(dev).clone()
    boxes[:, :3] -= plan.margin
    boxes[:, 3:] += plan.margin
    send_counts = be.halo_count(cloud, boxes, plan.rank).to(cdev)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    # the one read-back: counts (as doubles, exact below 2^53) next to the global extrema
    host = torch.cat((send_counts.to(torch.float64), recv_counts.to(torch.float64), glo, ghi)).cpu()
    w = plan.world
    send_list = [int(v) for v in host[:w]]
    recv_list = [int(v) for v in host[w:2 * w]]
    glo_h, ghi_h = host[2 * w:2 * w + 3].numpy().copy(), host[2 * w + 3:].numpy().copy()
    offsets = torch.zeros(w, dtype=torch.int64)
    offsets[1:] = torch.cumsum(torch.tensor(send_list[:-1], dtype=torch.int64), 0)
    packed = be.halo_pack(cloud, boxes, plan.rank, offsets.to(dev), sum(send_list)).to(cdev)
    n, h = cloud.shape[0], sum(recv_list)
    plan._ensure_buffer(h)
    if plan.stage_on_host:
        recv = torch.empty((h, 3), dtype=torch.float64, device=cdev)
    else:
        recv = plan._buffer[n:n + h]                       # RCCL writes the halo behind the tile
    dist.all_to_all_single(recv.reshape(-1), packed.reshape(-1),
                           [3 * c for c in recv_list], [3 * c for c in send_list], group=group)
    if plan.stage_on_host:
        plan._buffer[n:n + h] = recv.to(dev)
    plan.halo_sent, plan.halo_received = sum(send_list), h
    return glo_h, ghi_h, h


def process_tile(plan, out=None):
    """features of this rank's tile, (N, 4*S) fp64 on this rank's device, rows aligned with the
    tile.  collective: every rank of the group must call it."""
    be, cloud = plan.backend, plan.cloud
    n = cloud.shape[0]
    n_scales = len(plan.edge_lengths)
    lo, hi, n_halo = exchange_halo(plan)
    search = plan._buffer[:n + n_halo]
    plan._search_points = search.shape[0]
    if out is None:
        out = torch.empty((n, 4 * n_scales), dtype=torch.float64, device=cloud.device)
    info = torch.zeros((max(n_scales, 1), 4), dtype=torch.int64, device=cloud.device)
    be.features(search, n, lo, hi, plan.edge_lengths, plan.radii, out, info)
    plan._info = info
    return out


def process_multi_gpu(tile_cloud, edge_lengths, radii, group=None):
    """convenience wrapper: this rank's tile in (numpy or torch), this rank's features out (same kind).
    torch.distributed must be initialised (backend "nccl") with one rank per GPU."""
    as_torch = isinstance(tile_cloud, torch.Tensor)
    plan = TilePlan(tile_cloud, edge_lengths, radii, group=group)
    out = process_tile(plan)
    return out if as_torch else out.cpu().numpy()


def partition_tiles(points, world):
    """split a host cloud into `world` spatially compact tiles of (nearly) equal point count by
    recursive median bisection along the longest axis.  the tiles' bounding boxes are disjoint, which
    keeps the box-based halo small.  returns a list of sorted index arrays."""
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
    Morton-ordered archive is cut into files).  correct with process_tile, but Morton runs can be
    L-shaped, so their bounding boxes - and with them the box-based halos - can be much larger than
    those of partition_tiles.  returns a list of index arrays."""
    from nimrud_amd import synth
    order = synth.morton_sort(np.asarray(points)[:, :3], edge_length)
    return [np.sort(chunk) for chunk in np.array_split(order, world)]
