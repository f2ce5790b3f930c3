"""
bench.py - the BASELINE.json metric on MI355X: point-scale feature ops/s of the multiscale
neighborhood-feature path (10M points x 5 scales), with the achieved-HBM roofline figure of the
dominant kernel and the reference CPU path timed beside it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

a "step" is one pass of the hot path over one batch of synthetic input: all scales of
process_single_core's scale loop (spatial order -> occupancy indexes -> fused search/moments/eigen
kernel per scale) over a cloud that is already resident in HBM.

N > 1, --scaling strong (default; BASELINE config 3 as stated): ONE 10 M-point Morton-ordered cloud is
cut into N Morton-contiguous tiles, one per rank; a step begins with the halo exchange over RCCL
(nm_halo_exchange: cell-set halos) and `value` counts the 10 M points once.  --scaling weak: every rank
owns its own 10 M-point scene, the scenes abut along x.  N = 1 is the same workload either way.
--workload c5_scene_10m_rf adds the per-point random-forest evaluation (BASELINE config 5) and reports
classified points/s end to end.  rank 0 prints ONE JSON line.
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

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak
PROFILE_ROUND = "r3"


def _profile_json(name):
    for rnd in (PROFILE_ROUND, "r2", "r1_final"):
        path = os.path.join(REPO, "profiles", "%s_%s.json" % (rnd, name))
        if os.path.exists(path):
            try:
                return json.load(open(path)), os.path.relpath(path, REPO)
            except Exception:   # noqa: BLE001
                pass
    return None, None


def measured_traffic(points_per_gpu, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (tools/pmc_summary.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE
    doubled as the gfx950 note in MI355X_MICROARCH.md prescribes).  only valid for the configuration it
    was collected on; None otherwise."""
    data, path = _profile_json("traffic")
    if points_per_gpu != 10_000_000 or data is None:
        return None, None
    for name, rec in data.get("kernels", {}).items():
        if name.startswith(kernel_prefix):
            return rec.get("hbm_bytes_per_launch_mean"), path
    return None, None


def _issue_costs():
    """SIMD cycles per wave64 instruction by class, measured on this part by tools/issue_rate.hip (profiles/
    r3_issue_rate.json; the column for 4 waves per SIMD - the search kernel runs 5, one-wave workgroups like the
    probe's).  None when the table is not there."""
    data, path = _profile_json("issue_rate")
    if data is None:
        return None, None
    try:
        col = data["waves_per_simd"].index(4)
        c = {k: float(v[col]) for k, v in data["cycles"].items()}
        return c, path
    except Exception:   # noqa: BLE001
        return None, None


def valu_ceiling(n_queries, kernel_ms):
    """the second roofline SURVEY 8d asks for: the ALU candidate-test ceiling.  from the committed SQ
    counters of the dominant kernel (instructions per wave by class, mean over the scales of the launch;
    tools/collect_profiles_r3.sh), the live query rate and the MEASURED issue cost of each class
    (tools/issue_rate.hip): fp64 flop per query x queries/s against the fp64 vector peak, next to how much of
    the SIMDs' time the instruction stream accounts for.  the counters do not split the 32-bit integer and
    the uncategorised instructions into the part's two cost classes (v_add_u32 / v_and / v_bitop3 / v_mov:
    one half the cost of v_alignbit / shifts / v_mad_u32_u24 / v_cmp / DPP, which cost what an fp64 add
    does), so the busy fraction is given as a bracket, with the static opcode mix of the kernel's main path in
    between (profiles/r3_issue_model.json, tools/issue_model.py)."""
    data, path = _profile_json("instruction_mix")
    if data is None or "per_wave" not in data:
        return None
    try:
        pw = {k: float(np.mean(v)) for k, v in data["per_wave"].items()}
        f64 = pw["SQ_INSTS_VALU_ADD_F64"] + pw["SQ_INSTS_VALU_MUL_F64"] + pw["SQ_INSTS_VALU_FMA_F64"]
        flop = f64 + pw["SQ_INSTS_VALU_FMA_F64"] + pw["SQ_INSTS_VALU_TRANS_F64"]
        queries_per_s = n_queries / (kernel_ms * 1e-3)
        tflops = flop * queries_per_s / 1e12       # one lane of a wave instruction = one query's flop
        rec = {"bound": "valu-issue", "valu_instr_per_wave": pw["SQ_INSTS_VALU"],
               "fp64_instr_per_wave": f64 + pw["SQ_INSTS_VALU_TRANS_F64"],
               "fp64_flop_per_query": flop,
               "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": tflops / FP64_VECTOR_PEAK_TFLOPS, "source": path}
        cost, cost_path = _issue_costs()
        if cost is not None:
            # SIMD cycles one wave-scale has at the maximum clock (the costs are quoted at that clock too)
            simd_cycles_per_wave = kernel_ms * 1e-3 * 2.4e9 * 1024 / (n_queries / 64.0)
            fixed = (f64 * cost["v_add_f64"] + pw["SQ_INSTS_VALU_TRANS_F64"] * cost["v_rcp_f64"] +
                     pw["SQ_INSTS_VALU_INT64"] * cost["v_lshrrev_b64"] +
                     pw["SQ_INSTS_VALU_CVT"] * cost["v_cvt_f64_u32"])
            loose = pw["SQ_INSTS_VALU"] - f64 - pw["SQ_INSTS_VALU_TRANS_F64"] - pw["SQ_INSTS_VALU_INT64"] - \
                pw["SQ_INSTS_VALU_CVT"]
            lo = (fixed + loose * cost["v_add_u32"]) / simd_cycles_per_wave
            hi = (fixed + loose * cost["v_alignbit_b32 (imm)"]) / simd_cycles_per_wave
            rec.update({"issue_cost_source": cost_path,
                        "issue_cycles_per_class": {"fp64 add/mul/fma": cost["v_add_f64"],
                                                   "fp64 rcp/rsq": cost["v_rcp_f64"],
                                                   "alignbit/shift/mad24/cmp/dpp": cost["v_alignbit_b32 (imm)"],
                                                   "add/and/bitop3/mov": cost["v_add_u32"]},
                        "simd_cycles_per_wave_and_scale": simd_cycles_per_wave,
                        "valu_issue_busy_frac_bracket": [lo, hi]})
            model, model_path = _profile_json("issue_model")
            if model is not None and "main_path_issue_cycles" in model:
                rec["valu_issue_busy_frac"] = float(model["main_path_issue_cycles"]) / simd_cycles_per_wave
                rec["issue_model_source"] = model_path
        return rec
    except Exception:   # noqa: BLE001
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3_scene_10m")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = one cloud cut into N Morton tiles, weak = one cloud per rank")
    ap.add_argument("--halo", choices=("cells", "boxes"), default="cells")
    ap.add_argument("--points", type=int, default=None, help="points of the cloud (default: the config's)")
    ap.add_argument("--overlap", type=int, default=None,
                    help="nm_set_overlap value (0 = sequential stages, 1 = pipelining)")
    ap.add_argument("--fuse-scales", type=int, default=None,
                    help="nm_set_fuse_scales value (1 = one search launch walks all scales, 0 = one per scale)")
    ap.add_argument("--cpu-sample", type=int, default=150000,
                    help="points of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--emit-indices", action="store_true",
                    help="index-emission (parity) mode: a step writes the neighbor index lists of the workload's "
                         "FINEST scale (multiscale.py:103) as CSR instead of features; value = point-scales/s "
                         "of that, roofline adds 8 * mean list length bytes per point-scale (SURVEY 8d)")
    return ap.parse_args()


# ---- CPU baselines (rank 0, N = 1; they run BEFORE the process touches the GPU) ---------------------------

def _slice(points, sample):
    lo = max(0, len(points) // 2 - sample // 2)
    return np.ascontiguousarray(points[lo:lo + sample])


def cpu_baseline(points, edges, radii, sample):
    """the oracle's faithful restatement of nimrud/minimal (chunked kd-tree query + per-neighborhood
    numpy operators), one core, on a Morton-contiguous slice of the same cloud."""
    from oracle import nimrud_oracle as oracle
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:   # noqa: BLE001
        limiter = None
    tile = _slice(points, sample)
    t0 = time.perf_counter()
    oracle.process(tile, tile, edges, radii)
    dt = time.perf_counter() - t0
    if limiter is not None and hasattr(limiter, "restore_original_limits"):
        limiter.restore_original_limits()
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
    the reference itself suggests (nimrud/minimal/multiscale.py:92-93).  the pool is forked before this
    process has initialised HIP."""
    import multiprocessing as mp
    tile = _slice(points, sample)
    step = max(1000, sample // (workers * 4) // 1000 * 1000)
    tasks = [(tile[i:i + step], tile, edges, radii) for i in range(0, len(tile), step)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        done = sum(pool.map(_cpu_chunk, tasks))
    dt = time.perf_counter() - t0
    return {"value": done * len(edges) / dt, "unit": "point-scales/s", "cores": workers,
            "kind": "port", "sample": "a %d-point Morton-contiguous slice of the same cloud, %d query chunks over %d processes, %.1f s"
                                      % (len(tile), len(tasks), workers, dt)}


def cpu_lattice_c(points, edges, radii, sample):
    """a stronger CPU point than the reference's structure: the oracle's plain-C restatement (hash set of
    occupied voxels, lattice enumeration, OpenMP over queries) on a larger slice of the same cloud."""
    from oracle import nimrud_oracle as oracle
    threads = min(16, os.cpu_count() or 1)
    tile = _slice(points, sample)
    t0 = time.perf_counter()
    oracle.process_c(tile, tile, edges, radii, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": len(tile) * len(edges) / dt, "unit": "point-scales/s", "cores": threads,
            "kind": "port", "sample": "oracle/lattice_oracle.c (gcc -O2 -fopenmp) on a %d-point slice, "
                                      "%d scales, %.1f s" % (len(tile), len(edges), dt)}


def cpu_forest(model_arrays, feature_rows):
    """config 5's classifier stage on one core: the oracle's forest walk (numpy) on a sample of rows."""
    from oracle import nimrud_oracle as oracle
    t0 = time.perf_counter()
    oracle.forest_predict(model_arrays, feature_rows)
    dt = time.perf_counter() - t0
    return {"value": len(feature_rows) / dt, "unit": "classified rows/s (forest stage only)", "cores": 1,
            "kind": "port", "sample": "oracle.forest_predict on %d feature rows, %.1f s"
                                      % (len(feature_rows), dt)}


def bench_emit_indices(args, cfg, cloud, edges, radii, dev, n_cloud):
    """index-emission (parity) mode, SURVEY 8(d): a step = the neighbor index lists of the finest scale for
    every query - voxelize the search cloud (sorted unique addresses: np.unique, geometry.py:150), count,
    prefix, emit (multiscale.py:103) - through multiscale.neighbor_lists, everything resident in HBM.  the
    two launches of k_scale_neighbors (count, then emit) are timed with events on the stream they run on."""
    import torch
    from nimrud_amd.minimal import multiscale
    finest = int(np.argmin(edges))
    e, r = edges[finest], radii[finest]
    nq = cloud.shape[0]

    def step():
        return multiscale.neighbor_lists(cloud, cloud, e, r)

    for _ in range(max(args.warmup, 1)):
        off, idx = step()
    torch.cuda.synchronize(dev)
    total = int(idx.shape[0])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    m = int(multiscale.geometry.VoxelFilter(cloud[:, :3], e, device=dev).unique_addresses(cloud[:, :3]).shape[0])
    k_mean = total / float(nq)
    # algorithmic bytes of one step (SURVEY 8d plus its index term): read the search cloud for the voxelize
    # (24 N), write and read the unique addresses (16 M), read the queries (24 N), write offsets (8 N) and the
    # index lists (8 k N)
    alg_bytes = 24.0 * nq + 16.0 * m + 24.0 * nq + 8.0 * nq + 8.0 * total
    ms = elapsed / args.steps * 1e3
    record = {
        "metric": "point-scale neighbor-list ops/sec (index emission, parity mode)",
        "value": nq * args.steps / elapsed, "unit": "point-scales/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d points, index lists of the finest scale e=%g r=%g for every query "
                               "(CSR, int64 ranks in the sorted unique voxel addresses)"
                               % (args.workload, n_cloud, e, r),
                   "points_per_gpu": nq, "scales": 1, "parallelism": "tiles1"},
        "roofline": {"bound": "hbm", "kernel": "nm_voxelize + k_scale_neighbors (count) + k_scale_neighbors (emit)",
                     "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "alg_bytes_per_step": alg_bytes, "index_bytes_per_point_scale": 8.0 * k_mean,
                     "note": "whole step, not one launch: the emission is a chain of dependent binary searches "
                             "in the address array, latency-bound, nowhere near either roofline"},
        "voxels": m, "mean_neighbors_per_query": k_mean, "cpu_baseline": None,
    }
    print(json.dumps(record))


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, world))
    from nimrud_amd import synth

    # ---- synthetic input (host) ------------------------------------------------------------------------
    cfg = synth.CONFIGS[args.workload]
    n_cloud = args.points or cfg["n"]
    classify = bool(cfg.get("forest"))
    strong = world > 1 and args.scaling == "strong"
    points, labels, edges, radii = synth.make_config(args.workload, n=n_cloud,
                                                     seed_offset=0 if strong else rank)
    n_scales = len(edges)
    if world > 1 and not strong:
        # weak scaling: scenes abut along x (ground planes tile seamlessly; spheres and poles near a seam
        # reach into the neighbour), so every seam carries a real halo
        extent = cfg.get("extent", 0.0) * (np.sqrt(n_cloud / cfg["n"]) if cfg["kind"] == "scene"
                                           else (n_cloud / cfg["n"]) ** (1.0 / 3.0))
        points[:, 0] += rank * float(extent)
    if strong:
        # config 3 as stated: ONE Morton-ordered cloud, rank r owns the r-th of N contiguous runs.  (rows
        # of the config are already in Morton order of the coarsest cell.)
        cut = [len(points) * r // world for r in range(world + 1)]
        tile = np.ascontiguousarray(points[cut[rank]:cut[rank + 1]])
    else:
        tile = points

    # ---- CPU legs first: nothing below has touched the GPU yet (no fork after HIP is up) --------------
    cpu = {}
    forest_arrays = None
    if classify:
        fixture = np.load(os.path.join(REPO, "tests", "golden", cfg["forest"]), allow_pickle=False)
        forest_arrays = {k: fixture[k] for k in ("left", "right", "feature", "threshold", "value",
                                                 "roots", "classes")}
        forest_arrays["n_features"] = int(fixture["n_features"])
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        cpu["cpu_baseline"] = cpu_baseline(points, edges, radii, args.cpu_sample)
        # "all host cores", a 1 M-point slice for configs 2-5.  the box gives one GPU's job a share
        # of 16 cpus (more worker processes than that only fight for them, and the box kills a run above its
        # process limit): 16 workers, or fewer when the host has fewer; NIMRUD_BENCH_CPU_WORKERS overrides.
        # the slice is 1 M points (the whole cloud when it is smaller) unless --cpu-sample asks for less
        workers = int(os.environ.get("NIMRUD_BENCH_CPU_WORKERS", "0")) or min(16, os.cpu_count() or 1)
        multi_sample = min(len(points), 1_000_000) if args.cpu_sample >= 150000 else args.cpu_sample
        if workers > 1:
            try:
                cpu["cpu_baseline_multicore"] = cpu_baseline_multicore(points, edges, radii,
                                                                       multi_sample, workers)
                cpu["cpu_baseline_multicore"]["host_cpus"] = os.cpu_count()
            except Exception as err:   # noqa: BLE001 - a reported extra, never fatal
                cpu["cpu_baseline_multicore"] = {"error": str(err)[:200]}
        try:
            cpu["cpu_lattice_c"] = cpu_lattice_c(points, edges, radii, 2_000_000)
        except Exception as err:       # noqa: BLE001
            cpu["cpu_lattice_c"] = {"error": str(err)[:200]}
        if classify:
            cpu["cpu_forest"] = cpu_forest(forest_arrays, fixture["x"])

    # ---- GPU ------------------------------------------------------------------------------------------------
    import torch
    import torch.distributed as dist
    from nimrud_amd import device as nm_device
    from nimrud_amd.minimal import multiscale, classification

    # one rank per GPU.  NIMRUD_BENCH_BACKEND=gloo-shared lets several ranks share a GPU to rehearse the
    # multi-rank code path on a one-GPU box (collectives over gloo, staged through host memory; not a
    # benchmark).
    rehearsal = os.environ.get("NIMRUD_BENCH_BACKEND", "") in ("gloo", "gloo-shared")
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    comm = None
    if world > 1:
        # torch.distributed is the bootstrap only (the RCCL unique id, the barriers around the timed
        # region, the max over ranks): gloo is enough.  the data path's communicator belongs to the
        # library (nm_comm_create) and the exchange is nm_halo_exchange.
        dist.init_process_group("gloo")
        from nimrud_amd import parallel
        if not rehearsal:
            # every rank must end up on the same transport: agree (over gloo) on whether RCCL came up
            try:
                comm = parallel.RcclComm(rank=rank, world=world, device=dev)
                ok = 1
            except Exception as err:       # noqa: BLE001
                print("rank %d: RCCL communicator could not be created (%s); falling back to "
                      "torch.distributed over gloo" % (rank, str(err)[:200]), file=sys.stderr)
                comm, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and comm is not None:
                comm.close()
                comm = None

    cloud = torch.from_numpy(tile).to(dev)
    rt = nm_device.get_runtime(dev)
    if args.overlap is not None:
        rt.check(rt.lib.nm_set_overlap(rt.ctx, args.overlap))
    if args.fuse_scales is not None:
        rt.check(rt.lib.nm_set_fuse_scales(rt.ctx, args.fuse_scales))
    model = classification.ForestModel.from_arrays(forest_arrays, device=dev) if classify else None

    if world > 1:
        plan = parallel.TilePlan(cloud, edges, radii, comm=comm, halo=args.halo)
        # the benchmark cloud does not change between steps: the exchange keeps its plan (who sends how much to
        # whom) after the first step - the halo rows themselves are packed and sent every step - so that a step
        # has no host synchronisation (nm_halo_exchange, NM_HALO_REUSE_PLAN).  NIMRUD_BENCH_STATIC_PLAN=0: the
        # full exchange every step, all-gathers and the one synchronisation included
        plan.static = os.environ.get("NIMRUD_BENCH_STATIC_PLAN", "1") != "0"

        def features_step():
            return parallel.process_tile(plan)
    else:
        out = torch.empty((cloud.shape[0], 4 * n_scales), dtype=torch.float64, device=dev)

        def features_step():
            return multiscale.process_gpu(cloud, cloud, edges, radii, out=out)

    if args.emit_indices:
        if world != 1 or classify:
            raise SystemExit("--emit-indices is a single-GPU feature-free mode")
        return bench_emit_indices(args, cfg, cloud, edges, radii, dev, n_cloud)

    fused = os.environ.get("NIMRUD_BENCH_FUSED_FOREST", "1") != "0"
    if os.environ.get("NIMRUD_BENCH_FOREST_EPILOGUE"):
        rt.check(rt.lib.nm_set_forest_mode(rt.ctx, int(os.environ["NIMRUD_BENCH_FOREST_EPILOGUE"])))

    def step():
        if model is None:
            return features_step()
        if world > 1:
            feats = features_step()
            return model._eval(feats, False, True)[1]
        # config 5: the forest is evaluated inside the search kernel, behind each row's last scale
        return classification.classify_cloud(cloud, edges, radii, model, fused=fused, out=out)[0]

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the W warm-up steps the contract asks for are preceded by a time-bounded, untimed run-in: a VALU-bound
    # kernel tracks the shader clock, and a GPU that has idled through the CPU legs above needs a few hundred
    # milliseconds of load to settle (NIMRUD_BENCH_RUN_IN_S=0 switches it off)
    run_in = float(os.environ.get("NIMRUD_BENCH_RUN_IN_S", "0.5"))
    if world > 1:
        # a step holds collectives: every rank must run the same number of them
        for _ in range(32 if run_in > 0 else 0):
            step()
        torch.cuda.synchronize(dev)
    else:
        t_in = time.perf_counter()
        while run_in > 0 and time.perf_counter() - t_in < run_in:
            for _ in range(4):
                step()
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
    rt.check_async(wait=True)
    # config 5: the forest stage's share = the step with it minus the same step without (outside the timed
    # region)
    features_only_ms = None
    if model is not None:
        for _ in range(2):
            features_step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            features_step()
        fence()
        features_only_ms = (time.perf_counter() - t1) / args.steps * 1e3

    # occupied voxels per scale (for the algorithmic byte count), outside the timed region
    if world == 1:
        _, info = multiscale.process_gpu(cloud, cloud, edges, radii, return_info=True)
    else:
        plan.want_info = True
        parallel.process_tile(plan)
        info = plan.last_info()
    voxels = [i.voxels for i in info]
    extra_passes = [i.extra_passes for i in info]
    n_local_search = cloud.shape[0] if world == 1 else plan.search_points()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([cloud.shape[0], plan.halo_received], dtype=torch.int64)
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
        nq = cloud.shape[0]
        # dominant kernel = k_scale_features<7>: per launch it reads the query coordinates (24 B) and
        # the occupied-voxel set (8 B per voxel as addresses) and writes 4 fp64 features (32 B).
        # (with the scales of the ladder in one launch - the default - that launch does this once per scale:
        # the bytes below are per launch, whatever the launch covers)
        n_launch = max(launches.value, 1) / max(args.steps, 1)
        alg_bytes = float(np.sum([56.0 * nq + 8.0 * m for m in voxels])) / n_launch
        k_ms = ms[2] / max(launches.value, 1)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        kernel_name = "k_scale_features<7, 3, true" if (classify and fused) else "k_scale_features<7, 3, false"
        traffic, traffic_src = measured_traffic(nq, kernel_name) if world == 1 else (None, None)
        roofline = {
            "bound": "hbm",
            "kernel": kernel_name + ", ...> (search + moments + eigen-solve%s)" % (
                " + forest epilogue" if (classify and fused) else ""),
            "scales_per_launch": n_scales / n_launch,
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "measured_copy_GBps": copy_gbps,
            "frac_of_measured_copy": achieved / copy_gbps if copy_gbps else None,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "alg_bytes_per_launch": alg_bytes,
            "kernel_ms_avg": k_ms,
            "valu": valu_ceiling(nq * n_scales / n_launch, k_ms) if (k_ms > 0 and not classify) else None,
            "note": "nominal bound is HBM (few bytes per unit of work); the binding ceiling is vector-ALU "
                    "issue, see `valu`: instructions per wave, fp64 flop per query and the fraction of the "
                    "fp64 vector peak they amount to at the measured query rate",
        }
        if classify:
            metric, unit = "classified points/sec end-to-end (features + forest)", "points/s"
            value = total_points * args.steps / elapsed
        else:
            metric, unit = "point-scale feature ops/sec", "point-scales/s"
            value = total_points * n_scales * args.steps / elapsed
        record = {
            "metric": metric,
            "value": value,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if (strong or world == 1) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d-point %s%s, %d scales e=%s r=%s, query cloud = search cloud%s"
                            % (args.workload, n_cloud,
                               {"uniform": "uniform-random cloud in a cube", "scene": "plane+pole+sphere scene",
                                "lidar": "terrestrial-LiDAR-style cloud (power-law density)"}[cfg["kind"]],
                               "" if (strong or world == 1) else " per GPU", n_scales, edges, radii,
                               ", rows in Morton order of the %g m cell" % cfg["morton"] if cfg.get("morton")
                               else ", rows in generation order"),
                "points_per_gpu": nq,
                "scales": n_scales,
                "parallelism": "tiles%d" % world,
                "scaling_mode": "single" if world == 1 else args.scaling,
                "tiles": "one" if world == 1 else
                         ("Morton-contiguous runs of one cloud" if strong else "one scene per rank, abutting"),
                "halo": None if world == 1 else args.halo,
                "search_points_incl_halo": int(n_local_search),
                "halo_points_exchanged_per_step": int(total_halo),
                "collectives": "none" if world == 1 else
                               ("nm_halo_exchange over RCCL: grouped ncclSend/ncclRecv(halo rows) per step; the "
                                "plan - all-gather(6 f64) + all-gather(256 KB cell set) + all-gather(counts) + one "
                                "host synchronisation - " + ("once, the cloud being static" if plan.static
                                                             else "per step too")
                                if comm is not None else ("torch.distributed over gloo (rehearsal)" if rehearsal
                                      else "torch.distributed over gloo, staged through host memory "
                                           "(the RCCL communicator could not be created)")),
            },
            "roofline": roofline,
            "stage_ms_per_step": {
                "cell_keys_and_sort": ms[0] / args.steps,
                "index_build": ms[1] / args.steps,
                "search_feature_kernel": ms[2] / args.steps,
            },
            "workspace_bytes_per_point": (rt._work.numel() / float(n_local_search)) if rt._work is not None else None,
            "voxels_per_scale": voxels,
            "extra_search_passes_per_scale": extra_passes,
        }
        if classify:
            f_ms = (elapsed / args.steps * 1e3 - features_only_ms) if features_only_ms else None
            n_classes = int(forest_arrays["value"].shape[1])
            f_bytes = (8.0 * 4 * n_scales + 8.0 * 0 + 4.0) * nq      # features in, labels out
            record["forest"] = {
                "trees": int(len(forest_arrays["roots"])), "nodes": int(len(forest_arrays["left"])),
                "classes": n_classes, "fused_behind_last_scale": bool(fused and world == 1),
                "features_only_ms_per_step": features_only_ms,
                "ms_per_step": f_ms,
                "roofline": {"bound": "hbm",
                             "kernel": "forest epilogue of k_scale_features<7,true,forest>" if fused
                                       else "k_forest_eval_packed8",
                             "achieved": f_bytes / (f_ms * 1e-3) / 1e9 if f_ms else None,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": f_bytes / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if f_ms else None,
                             "alg_bytes_per_launch": f_bytes, "traffic": None},
            }
        if cpu:
            record.update(cpu)
        else:
            record["cpu_baseline"] = None
        print(json.dumps(record))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
