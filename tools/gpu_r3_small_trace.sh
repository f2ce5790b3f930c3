# kernel traces of the small workloads (fixed per-step cost): bash tools/gpu_r3_small_trace.sh <outdir>
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O
run() {  # name args...
  local name=$1; shift
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$name -o trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --cpu-sample 0 > $GRAFT_REPO_ROOT/$O/$name.json 2>/dev/null)
  find $O/prof_$name -name "*_kernel_trace.csv" -delete
  python3 - $O $name <<'PY'
import csv, sys, json
O, name = sys.argv[1], sys.argv[2]
d = json.loads(open("%s/%s.json" % (O, name)).read().strip().splitlines()[-1])
steps = d["steps"] + d["warmup"]
print(name, "ms/step %.4f" % d["ms_per_step"], d["stage_ms_per_step"])
rows = list(csv.DictReader(open("%s/prof_%s/trace_kernel_stats.csv" % (O, name))))
tot = 0.0
for r in rows:
    calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    if calls >= d["steps"]:
        per_step = calls / float(calls // d["steps"] * d["steps"]) if False else None
    print("   %-46s calls %5d avg %8.1f us" % (r["Name"][:46], calls, avg))
PY
}
run c1 --workload c1_uniform_100k --steps 100 --warmup 10
run p1250k --points 1250000 --steps 50 --warmup 5
