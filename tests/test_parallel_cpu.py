"""
CPU tests of the multi-GPU path (nimrud_amd/parallel.py) with world_size 2 over gloo: the global
bounds all-reduce, the box all-gather, the count exchange and the all-to-all-v of halo rows are the
product code; the per-tile data-path operations (which are HIP kernels in production) are replaced by
a numpy backend, and the per-tile features by the oracle with the GLOBAL lattice.  the assembled
result must equal a single-process oracle run over the whole cloud.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nimrud_amd import parallel, synth
from oracle import nimrud_oracle as oracle

EDGES, RADII = [0.1, 0.2, 0.4], [0.3, 0.6, 1.2]


class NumpyBackend(object):
    """test double for parallel.HipBackend (CPU tensors, numpy arithmetic, oracle features)."""

    def bounds(self, cloud):
        x = cloud.numpy()[:, :3]
        return torch.from_numpy(np.concatenate((x.min(0), x.max(0))))

    @staticmethod
    def _cells(x, glob, margin):
        # the grid of csrc/nm_halo.hip (parallel.coarse_grid is its host mirror)
        lo, inv_e, dims, dilation = parallel.coarse_grid(glob, margin)
        c = np.floor((x - lo) * inv_e)
        c = np.minimum(np.maximum(c, 0), dims - 1).astype(np.int64)
        return (c[:, 2] * dims[1] + c[:, 1]) * dims[0] + c[:, 0], dims, dilation

    def cellset(self, cloud, global_minmax, margin):
        """occupied coarse cells, dilated by D cells per axis, one bit per cell"""
        x = cloud.numpy()[:, :3]
        flat, dims, dilation = self._cells(x, global_minmax.numpy(), margin)
        grid = np.zeros((dims[2], dims[1], dims[0]), dtype=bool)
        grid.reshape(-1)[flat] = True
        for axis in range(3):
            acc = grid.copy()
            for d in range(1, dilation + 1):
                src = [slice(None)] * 3
                dst = [slice(None)] * 3
                src[axis], dst[axis] = slice(d, None), slice(None, -d)
                if grid.shape[axis] > d:
                    acc[tuple(dst)] |= grid[tuple(src)]
                    acc[tuple(src)] |= grid[tuple(dst)]
            grid = acc
        bits = np.zeros(parallel.CELLSET_WORDS * 32, dtype=bool)
        bits[:grid.size] = grid.reshape(-1)
        words = np.packbits(bits.reshape(-1, 32), axis=1, bitorder="little").view("<u4").reshape(-1)
        return torch.from_numpy(words.view(np.int32).copy())

    def _masks(self, cloud, dest, skip):
        x = cloud.numpy()[:, :3]
        if isinstance(dest, tuple):
            glob, margin, sets = dest
            flat, _, _ = self._cells(x, glob.numpy(), margin)
            words = sets.numpy().view(np.uint32)
            masks = [((words[j][flat >> 5] >> (flat & 31).astype(np.uint32)) & 1).astype(bool)
                     if j != skip else np.zeros(len(x), dtype=bool) for j in range(len(words))]
            return x, masks
        b = dest.numpy()
        masks = [np.all((x >= b[j, :3]) & (x <= b[j, 3:]), axis=1) if j != skip
                 else np.zeros(len(x), dtype=bool) for j in range(len(b))]
        return x, masks

    def halo_count(self, cloud, dest, skip):
        _, masks = self._masks(cloud, dest, skip)
        return torch.tensor([int(m.sum()) for m in masks], dtype=torch.int64)

    def halo_pack(self, cloud, dest, skip, offsets, total):
        x, masks = self._masks(cloud, dest, skip)
        rows = [x[m] for m in masks]
        assert [len(r) for r in rows[:-1]] == list(np.diff(offsets.numpy()))
        return torch.from_numpy(np.concatenate(rows, axis=0).reshape(-1, 3).copy())

    def copy_xyz(self, cloud, out):
        out.copy_(cloud[:, :3])

    def features(self, search, n_query, bounds, edge_lengths, radii, out, info):
        s = search.numpy()
        lo, hi = bounds.numpy()[:3], bounds.numpy()[3:]
        out.copy_(torch.from_numpy(oracle.process_fast(s[:n_query], s, edge_lengths, radii,
                                                       bounds=(lo, hi))))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, points, parts, results, halo="cells"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tile = torch.from_numpy(np.ascontiguousarray(points[parts[rank]]))
        plan = parallel.TilePlan(tile, EDGES, RADII, backend=NumpyBackend(), halo=halo)
        out = parallel.process_tile(plan)
        results[rank] = (out.numpy().copy(), plan.halo_sent, plan.halo_received,
                         plan.search_points())
    finally:
        dist.destroy_process_group()


def _run_world(points, parts, halo):
    world = len(parts)
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), points, parts, results, halo), nprocs=world,
             join=True)
    return results


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,halo", [(2, "cells"), (3, "cells"), (2, "boxes")])
def test_halo_exchange_matches_single_process(world, halo):
    # world 3 on a long strip: ranks 0 and 2 are not neighbours and exchange nothing
    extent = 24.0 if world == 2 else 40.0
    points, _ = synth.scene_cloud(16000, extent=extent, n_poles=12, n_spheres=4, seed=13)
    if world == 3:
        points = points[points[:, 1] < 8.0]
    parts = parallel.partition_tiles(points, world)
    assert sum(len(p) for p in parts) == len(points)
    results = _run_world(points, parts, halo)
    whole = oracle.process_fast(points, points, EDGES, RADII)
    sent = recv = 0
    for rank in range(world):
        out, s, r, n_search = results[rank]
        want = whole[parts[rank]]
        assert np.array_equal(out[:, ::4], want[:, ::4]), "population differs on rank %d" % rank
        assert np.abs(out - want).max() <= 1e-10
        assert n_search == len(parts[rank]) + r
        assert 0 < r < len(points) - len(parts[rank])      # a real halo, but not the whole cloud
        sent += s
        recv += r
    assert sent == recv


@pytest.mark.timeout(600)
def test_morton_tiles_get_surface_sized_halos_with_cell_sets():
    # Morton-contiguous tiles (what north_star shards by) are L-shaped: their bounding boxes overlap the
    # neighbours almost entirely, so box halos ship most of the cloud.  cell-set halos must stay within
    # 1.5x of what compact bisection tiles need on the same cloud, and the features must still be those
    # of a single-process run, bit for bit in the populations.
    points, _ = synth.scene_cloud(40000, extent=48.0, n_poles=30, n_spheres=10, seed=17)
    world = 3                                          # 3 runs of a Z curve: two of them are L-shaped
    morton = parallel.partition_by_morton(points, world, 0.4)
    compact = parallel.partition_tiles(points, world)
    whole = oracle.process_fast(points, points, EDGES, RADII)
    received = {}
    for name, parts, halo in (("morton-cells", morton, "cells"), ("compact-cells", compact, "cells"),
                              ("morton-boxes", morton, "boxes")):
        results = _run_world(points, parts, halo)
        total = 0
        for rank in range(world):
            out, _, r, n_search = results[rank]
            want = whole[parts[rank]]
            assert np.array_equal(out[:, ::4], want[:, ::4]), (name, rank)
            assert np.abs(out - want).max() <= 1e-10
            assert n_search == len(parts[rank]) + r
            total += r
        received[name] = total
    assert received["morton-cells"] <= 1.5 * received["compact-cells"], received
    assert received["morton-cells"] < received["morton-boxes"], received


def test_coarse_grid_mirror():
    # the host mirror of nm_coarse_grid: cell edge margin/4 when that fits 2^21 cells, coarser otherwise
    lo, inv_e, dims, dilation = parallel.coarse_grid([0, 0, 0, 190, 190, 6], 3.1)
    assert abs(1.0 / inv_e - 3.1 / 4) < 1e-12 and dilation == 5 and np.prod(dims) <= 1 << 21
    lo, inv_e, dims, dilation = parallel.coarse_grid([0, 0, 0, 3000, 3000, 300], 0.5)
    assert 1.0 / inv_e > 0.5 / 4 and np.prod(dims) <= 1 << 21 and dilation == int(0.5 * inv_e) + 1
    lo, inv_e, dims, dilation = parallel.coarse_grid([1, 2, 3, 1, 2, 3], 1.0)       # a single point
    assert list(dims) == [1, 1, 1]


def test_margin_and_partition_helpers():
    m = parallel.halo_margin([0.1, 0.8], [0.3, 2.4])
    assert m >= 2.4 + 0.8 * np.sqrt(3) / 2 and m < 2.4 + 0.8
    pts = synth.uniform_cloud(1000, seed=3)
    for parts in (parallel.partition_by_morton(pts, 3, 0.5), parallel.partition_tiles(pts, 3)):
        assert sorted(np.concatenate(parts)) == list(range(1000))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # bisection tiles have disjoint bounding boxes along the split axis
    a, b = parallel.partition_tiles(pts, 2)
    axis = int(np.argmax(pts.max(0) - pts.min(0)))
    assert pts[a][:, axis].max() <= pts[b][:, axis].min()


def test_single_rank_plan_needs_no_process_group():
    pts = torch.from_numpy(synth.uniform_cloud(3000, extent=3.0, seed=5))
    plan = parallel.TilePlan(pts, [0.25], [0.75], backend=NumpyBackend())
    out = parallel.process_tile(plan).numpy()
    want = oracle.process_fast(pts.numpy(), pts.numpy(), [0.25], [0.75])
    assert np.array_equal(out[:, 0], want[:, 0]) and np.abs(out - want).max() <= 1e-10
    assert plan.halo_received == 0
