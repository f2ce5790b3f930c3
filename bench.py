"""
bench.py - the BASELINE.json metric on MI355X: point-scale feature ops/s of the multiscale
neighborhood-feature path (10M points x 5 scales), with the achieved-HBM roofline figure of the
dominant kernel and the reference CPU path timed beside it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

a "step" is one pass of the hot path over one batch of synthetic input: all scales of
process_single_core's scale loop (cell keys -> sort -> occupancy index -> fused search/moments/eigen
kernel, per scale) over a cloud that is already resident in HBM.  with N > 1 every rank owns one
Morton-contiguous spatial tile of the same size (weak scaling) and a step begins with the halo
exchange over RCCL.  rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
TRAFFIC_FILE = os.path.join(REPO, "profiles", "r1_final_traffic.json")


def measured_traffic(points_per_gpu):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (tools/pmc_summary.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE
    doubled as the gfx950 note in MI355X_MICROARCH.md prescribes and as calibrated on k_cell_keys_only).
    only valid for the configuration it was collected on; None otherwise."""
    if points_per_gpu != 10_000_000 or not os.path.exists(TRAFFIC_FILE):
        return None
    try:
        data = json.load(open(TRAFFIC_FILE))
        for name, rec in data["kernels"].items():
            if name.startswith("k_scale_features<7"):
                return rec.get("hbm_bytes_per_launch_mean")
    except Exception:   # noqa: BLE001
        return None
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3_scene_10m")
    ap.add_argument("--points", type=int, default=None, help="points per GPU (default: the config's)")
    ap.add_argument("--overlap", type=int, default=None,
                    help="nm_set_overlap value (0 = sequential stages, 1 = default pipelining)")
    ap.add_argument("--cpu-sample", type=int, default=150000,
                    help="points of the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def cpu_baseline(points, edges, radii, sample):
    """the oracle's faithful restatement of nimrud/minimal (chunked kd-tree query + per-neighborhood
    numpy operators), one core, on a Morton-contiguous slice of the same cloud."""
    from oracle import nimrud_oracle as oracle
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:   # noqa: BLE001
        limiter = None
    lo = max(0, len(points) // 2 - sample // 2)
    tile = np.ascontiguousarray(points[lo:lo + sample])
    t0 = time.perf_counter()
    oracle.process(tile, tile, edges, radii)
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    return {
        "value": len(tile) * len(edges) / dt,
        "unit": "point-scales/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d-point Morton-contiguous slice of the same cloud as query and search, %d scales, "
                  "oracle.process (scipy cKDTree.query_ball_tree in 1000-point chunks + per-"
                  "neighborhood numpy cov/eigvalsh), %.1f s, host has %d cpus"
                  % (len(tile), len(edges), dt, os.cpu_count() or 0),
    }


def _cpu_chunk(args):
    chunk, tile, edges, radii = args
    from oracle import nimrud_oracle as oracle
    return oracle.process(chunk, tile, edges, radii).shape[0]


def cpu_baseline_multicore(points, edges, radii, sample, workers):
    """the same restatement over a process pool, 1000-point query chunks per task - the parallelisation
    the reference itself suggests (nimrud/minimal/multiscale.py:92-93).  every task voxel-filters and
    indexes the search tile again, exactly what mapping one_scale_single_core over chunks would do."""
    import multiprocessing as mp
    lo = max(0, len(points) // 2 - sample // 2)
    tile = np.ascontiguousarray(points[lo:lo + sample])
    step = max(1000, sample // (workers * 4) // 1000 * 1000)
    tasks = [(tile[i:i + step], tile, edges, radii) for i in range(0, len(tile), step)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        done = sum(pool.map(_cpu_chunk, tasks))
    dt = time.perf_counter() - t0
    return {"value": done * len(edges) / dt, "unit": "point-scales/s", "cores": workers,
            "kind": "port", "sample": "same %d-point slice, %d query chunks over %d processes, %.1f s"
                                      % (len(tile), len(tasks), workers, dt)}


def cpu_lattice_c(points, edges, radii, sample):
    """a stronger CPU point than the reference's structure: the oracle's plain-C restatement (hash set of
    occupied voxels, lattice enumeration, OpenMP over queries) on a larger slice of the same cloud."""
    from oracle import nimrud_oracle as oracle
    threads = min(16, os.cpu_count() or 1)
    lo = max(0, len(points) // 2 - sample // 2)
    tile = np.ascontiguousarray(points[lo:lo + sample])
    t0 = time.perf_counter()
    oracle.process_c(tile, tile, edges, radii, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": len(tile) * len(edges) / dt, "unit": "point-scales/s", "cores": threads,
            "kind": "port", "sample": "oracle/lattice_oracle.c (gcc -O2 -fopenmp) on a %d-point slice, "
                                      "%d scales, %.1f s" % (len(tile), len(edges), dt)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from nimrud_amd import synth, device as nm_device
    from nimrud_amd.minimal import multiscale

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, world))
    # one rank per GPU.  NIMRUD_BENCH_BACKEND=gloo lets several ranks share a GPU to rehearse the
    # multi-rank code path on a one-GPU box (collectives staged through host memory; not a benchmark).
    backend = os.environ.get("NIMRUD_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- synthetic input: one tile per rank, tiles side by side along x ---------------------------
    cfg = synth.CONFIGS[args.workload]
    n_points = args.points or cfg["n"]
    points, _, edges, radii = synth.make_config(args.workload, n=n_points, seed_offset=rank)
    if world > 1:
        # tiles abut along x (ground planes tile seamlessly; spheres and poles near a seam reach into
        # the neighbour), so every seam carries a real halo of width max(radius + 0.87 edge)
        extent = cfg.get("extent", 0.0) * (np.sqrt(n_points / cfg["n"]) if cfg["kind"] == "scene"
                                           else (n_points / cfg["n"]) ** (1.0 / 3.0))
        points[:, 0] += rank * float(extent)
    cloud = torch.from_numpy(points).to(dev)
    n_scales = len(edges)
    rt = nm_device.get_runtime(dev)
    if args.overlap is not None:
        rt.check(rt.lib.nm_set_overlap(rt.ctx, args.overlap))

    if world > 1:
        from nimrud_amd import parallel
        plan = parallel.TilePlan(cloud, edges, radii)

        def step():
            return parallel.process_tile(plan)
    else:
        out = torch.empty((cloud.shape[0], 4 * n_scales), dtype=torch.float64, device=dev)

        def step():
            return multiscale.process_gpu(cloud, cloud, edges, radii, out=out)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    rt.lib.nm_profile_begin(rt.ctx)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    import ctypes
    ms = (ctypes.c_double * 4)()
    launches = ctypes.c_int64(0)
    rt.check(rt.lib.nm_profile_end(rt.ctx, ctypes.byref(ms), ctypes.byref(launches)))

    # occupied voxels per scale (for the algorithmic byte count), outside the timed region
    _, info = multiscale.process_gpu(cloud, cloud, edges, radii, return_info=True) \
        if world == 1 else (None, plan.last_info())
    voxels = [i.voxels for i in info]
    n_local_search = cloud.shape[0] if world == 1 else plan.search_points()

    if world > 1:
        cdev = dev if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([cloud.shape[0], plan.halo_received], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_points, total_halo = int(tot[0].item()), int(tot[1].item())
    else:
        total_points, total_halo = cloud.shape[0], 0

    # what a plain device-to-device copy reaches on this box ("report fraction of both nominal
    # and measured-copy bandwidth"); outside the timed region
    copy_gbps = None
    if rank == 0:
        src = torch.empty(1 << 27, dtype=torch.float64, device=dev)      # 1 GiB
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(dev)
        copy_gbps = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        point_scales = total_points * n_scales * args.steps
        value = point_scales / elapsed
        # dominant kernel = k_scale_features<7>: per launch it reads the query coordinates (24 B) and
        # the occupied-voxel set (8 B per voxel as addresses) and writes 4 fp64 features (32 B).
        nq = cloud.shape[0]
        alg_bytes = float(np.mean([56.0 * nq + 8.0 * m for m in voxels]))
        k_ms = ms[2] / max(launches.value, 1)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        record = {
            "metric": "point-scale feature ops/sec",
            "value": value,
            "unit": "point-scales/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d points/GPU plane+pole+sphere scene, %d scales e=%s r=3e, "
                            "query cloud = search cloud, rows in Morton order of the coarsest cell"
                            % (args.workload, nq, n_scales, edges),
                "points_per_gpu": nq,
                "scales": n_scales,
                "parallelism": "tiles%d" % world,
                "search_points_incl_halo": int(n_local_search),
                "halo_points_exchanged_per_step": int(total_halo),
                "collectives": "none" if world == 1 else
                               "all-gather(6 f64/rank) + all-to-all(counts) + all-to-all-v(halo rows) per step",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_scale_features<7>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "measured_copy_GBps": copy_gbps,
                "frac_of_measured_copy": achieved / copy_gbps if copy_gbps else None,
                "traffic": measured_traffic(nq) if world == 1 else None,
                "traffic_source": "profiles/r1_final_traffic.json (rocprofv3 --pmc, separate passes)",
                "alg_bytes_per_launch": alg_bytes,
                "kernel_ms_avg": k_ms,
                "note": "nominal bound is HBM, the measured one is vector-ALU issue: SQ_ACTIVE_INST_VALU "
                        "covers 91 % of the kernel's duration on every SIMD (profiles/r1_final_traffic.json); "
                        "PMC traffic is 1.09x the algorithmic bytes, nothing is re-read from HBM",
            },
            "stage_ms_per_step": {
                "cell_keys_and_sort": ms[0] / args.steps,
                "index_build": ms[1] / args.steps,
                "search_feature_kernel": ms[2] / args.steps,
            },
            "voxels_per_scale": voxels,
        }
        if world == 1 and args.cpu_sample > 0:
            record["cpu_baseline"] = cpu_baseline(points, edges, radii, args.cpu_sample)
            workers = min(32, os.cpu_count() or 1)
            if workers > 1:
                try:
                    record["cpu_baseline_multicore"] = cpu_baseline_multicore(
                        points, edges, radii, args.cpu_sample, workers)
                except Exception as err:   # noqa: BLE001 - a reported extra, never fatal
                    record["cpu_baseline_multicore"] = {"error": str(err)[:200]}
            try:
                record["cpu_lattice_c"] = cpu_lattice_c(points, edges, radii, 2_000_000)
            except Exception as err:       # noqa: BLE001
                record["cpu_lattice_c"] = {"error": str(err)[:200]}
        else:
            record["cpu_baseline"] = None
        print(json.dumps(record))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
