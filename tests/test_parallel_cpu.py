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

    def _masks(self, cloud, boxes, skip):
        x = cloud.numpy()[:, :3]
        b = boxes.numpy()
        masks = [np.all((x >= b[j, :3]) & (x <= b[j, 3:]), axis=1) if j != skip
                 else np.zeros(len(x), dtype=bool) for j in range(len(b))]
        return x, masks

    def halo_count(self, cloud, boxes, skip):
        _, masks = self._masks(cloud, boxes, skip)
        return torch.tensor([int(m.sum()) for m in masks], dtype=torch.int64)

    def halo_pack(self, cloud, boxes, skip, offsets, total):
        x, masks = self._masks(cloud, boxes, skip)
        rows = [x[m] for m in masks]
        assert [len(r) for r in rows[:-1]] == list(np.diff(offsets.numpy()))
        return torch.from_numpy(np.concatenate(rows, axis=0).reshape(-1, 3).copy())

    def copy_xyz(self, cloud, out):
        out.copy_(cloud[:, :3])

    def features(self, search, n_query, lo, hi, edge_lengths, radii, out, info):
        s = search.numpy()
        out.copy_(torch.from_numpy(oracle.process_fast(s[:n_query], s, edge_lengths, radii,
                                                       bounds=(lo, hi))))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, points, parts, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tile = torch.from_numpy(np.ascontiguousarray(points[parts[rank]]))
        plan = parallel.TilePlan(tile, EDGES, RADII, backend=NumpyBackend())
        out = parallel.process_tile(plan)
        results[rank] = (out.numpy().copy(), plan.halo_sent, plan.halo_received,
                         plan.search_points())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_matches_single_process(world):
    # world 3 on a long strip: ranks 0 and 2 are not neighbours and exchange nothing
    extent = 24.0 if world == 2 else 40.0
    points, _ = synth.scene_cloud(16000, extent=extent, n_poles=12, n_spheres=4, seed=13)
    if world == 3:
        points = points[points[:, 1] < 8.0]
    parts = parallel.partition_tiles(points, world)
    assert sum(len(p) for p in parts) == len(points)
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), points, parts, results), nprocs=world, join=True)
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
