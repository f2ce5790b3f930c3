"""
summarise rocprofv3 output of bench.py into profiles/: per-kernel stats from a --kernel-trace --stats run
and per-launch HBM traffic of the dominant kernel from separate --pmc FETCH_SIZE / WRITE_SIZE passes.

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced
streaming read; WRITE_SIZE is exact; both are in KiB.  calibrated on our own access pattern in the same
runs: k_bounds reads exactly 24 B per point with 8-byte loads at a 24-byte stride and reports 12 B per
point; the key kernels write exactly 8 (or 12) B per point and report exactly that.

usage: python tools/pmc_summary.py <trace_dir> <fetch_dir> <write_dir> <sq_dir...> <tag>
"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_counters(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    trace_dir, fetch_dir, write_dir = sys.argv[1:4]
    sq_dirs, tag = sys.argv[4:-1], sys.argv[-1]
    out = {"tag": tag, "units": "bytes per launch; FETCH_SIZE KiB x 1024 x 2, WRITE_SIZE KiB x 1024"}
    stats = glob.glob(os.path.join(trace_dir, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        dst = os.path.join(REPO, "profiles", "%s_kernel_stats.csv" % tag)
        with open(stats[0]) as src, open(dst, "w") as o:
            o.write(src.read())
    per = {}
    for name, d, factor in (("fetch", fetch_dir, 2048.0), ("write", write_dir, 1024.0)):
        agg = collections.defaultdict(list)
        for r in read_counters(d):
            agg[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * factor))
        for k, v in agg.items():
            v.sort()
            per.setdefault(k, {})[name] = [b for _, b in v]
    kernels = {}
    for k, v in per.items():
        short = k.split("(")[0].replace("void ", "")
        f, w = v.get("fetch", []), v.get("write", [])
        kernels[short] = {
            "launches": max(len(f), len(w)),
            "fetch_bytes_per_launch_mean": sum(f) / len(f) if f else None,
            "write_bytes_per_launch_mean": sum(w) / len(w) if w else None,
        }
        if "k_scale_features" in short:
            n = min(len(f), len(w))
            kernels[short]["hbm_bytes_per_launch"] = [f[i] + w[i] for i in range(n)]
            kernels[short]["hbm_bytes_per_launch_mean"] = sum(f[-5:] + w[-5:]) / 5.0 if n >= 5 else None
    out["kernels"] = kernels
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sq_dirs:
        for r in read_counters(d):
            if "k_scale_features" in r["Kernel_Name"]:
                sq[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    out["k_scale_features_sq_counters_last5_launches_mean"] = {
        c: sum(v[0] for _, v in sorted(byd.items())[-5:]) / 5.0 for c, byd in sq.items()}
    dst = os.path.join(REPO, "profiles", "%s_traffic.json" % tag)
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
