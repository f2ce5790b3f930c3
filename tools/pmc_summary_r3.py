"""
summarise the output of tools/collect_profiles_r3.sh into profiles/r3_*:
  r3_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats per configuration
  r3_traffic.json             HBM bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE passes
                              (config 3, and the search kernel with the forest epilogue of config 5)
  r3_instruction_mix.json     SQ counters of the search kernel per wave AND SCALE (one launch walks the five
                              scales of the benchmark ladder: totals / waves / 5)
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced
streaming read; WRITE_SIZE is exact; both are in KiB.  (calibration: r1's profiles/README.md.)

usage: python tools/pmc_summary_r2.py gpurun_out/<dir>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SCALES = 5


def read_counters(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    out_dir = sys.argv[1]
    for d in sorted(glob.glob(os.path.join(out_dir, "trace_*"))):
        if not os.path.isdir(d):
            continue
        cfg = os.path.basename(d)[len("trace_"):]
        stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(REPO, "profiles", "r3_%s_kernel_stats.csv" % cfg))
    for name in ("bench_c3", "bench_c5", "bench_c1", "bench_c2", "bench_c3_perscale", "bench_ref_ladder",
                 "bench_c3_tile_1250k", "bench_emit_indices_c2", "bench_rehearsal_2rank", "bench_rehearsal_3rank",
                 "issue_rate"):
        src = os.path.join(out_dir, name + ".json")
        if os.path.exists(src) and os.path.getsize(src) > 0:
            shutil.copy(src, os.path.join(REPO, "profiles", "r3_%s.json" % name))
    if os.path.exists(os.path.join(out_dir, "issue_rate.txt")):
        shutil.copy(os.path.join(out_dir, "issue_rate.txt"), os.path.join(REPO, "profiles", "r3_issue_rate.txt"))
    traffic = {"tag": "r3", "units": "bytes per launch; FETCH_SIZE KiB x 1024 x 2, WRITE_SIZE KiB x 1024",
               "kernels": {}}
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
        cfg = os.path.basename(d).split("_")[1]
        for r in read_counters(d):
            k = short(r["Kernel_Name"])
            c = r["Counter_Name"]
            v = float(r["Counter_Value"])
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                key = k if cfg == "c3" else "%s [config 5 step]" % k
                if cfg == "c5" and "k_scale_features" not in k:
                    continue
                rec = traffic["kernels"].setdefault(key, {"fetch": [], "write": []})
                rec["fetch" if c == "FETCH_SIZE" else "write"].append(v * (2048.0 if c == "FETCH_SIZE" else 1024.0))
            elif cfg == "c3" and "k_scale_features<7" in k:
                sq[c][int(r["Dispatch_Id"])].append(v)
    for k, rec in traffic["kernels"].items():
        f, w = rec.pop("fetch"), rec.pop("write")
        rec["launches"] = max(len(f), len(w))
        rec["fetch_bytes_per_launch_mean"] = sum(f) / len(f) if f else None
        rec["write_bytes_per_launch_mean"] = sum(w) / len(w) if w else None
        if "k_scale_features" in k and f and w:
            # the last launch of each pass is the timed step's
            rec["hbm_bytes_per_launch_mean"] = f[-1] + w[-1]
    json.dump(traffic, open(os.path.join(REPO, "profiles", "r3_traffic.json"), "w"), indent=1)
    per_wave = {}
    waves = None
    last = {c: sorted(byd.items())[-1][1][0] for c, byd in sq.items()}      # the timed step's launch
    waves = last.get("SQ_WAVES")
    if waves:
        for c, v in last.items():
            per_wave[c] = [v / waves / N_SCALES]
        per_wave["SQ_WAVES"] = [1.0]
    mix = {"what": "per-wave, per-scale counters of k_scale_features<7, 3, false, true> - ONE launch walks the "
                   "five scales of the measured bench step, %s waves; totals divided by waves and by 5; "
                   "rocprofv3 --pmc, separate passes of `bench.py --steps 1 --warmup 1` "
                   "(tools/collect_profiles_r3.sh)" % (int(waves) if waves else "?"),
           "per_wave": per_wave}
    json.dump(mix, open(os.path.join(REPO, "profiles", "r3_instruction_mix.json"), "w"), indent=1)
    print("wrote profiles/r3_traffic.json, r3_instruction_mix.json")
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_ADD_F64",
              "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"):
        if c in per_wave:
            print("  %-28s %.1f" % (c, per_wave[c][0]))
    for k, rec in traffic["kernels"].items():
        if "k_scale_features" in k or "k_index_fused" in k:
            print("  %s: fetch %.4g write %.4g" % (k[:70], rec["fetch_bytes_per_launch_mean"] or 0,
                                                  rec["write_bytes_per_launch_mean"] or 0))


if __name__ == "__main__":
    main()
