# round 3: smoke + A/B of library variants on one box + kernel trace of the in-tree library
#   bash tools/gpu_r3_ab.sh <outdir> <variant>...      (libs in build_abl/lib_<variant>.so)
#   SMALL=1: also the small workloads (config 1, config 2, a 1.25 M-point tile of config 3) per variant
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
line() {   # file, label
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st = d["stage_ms_per_step"]
    print("%-22s ms/step %.4f  search %.4f index %.4f order %.4f  extra passes %s" % (
        sys.argv[2], d["ms_per_step"], st["search_feature_kernel"], st["index_build"], st["cell_keys_and_sort"],
        d.get("extra_search_passes_per_scale")))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
for V in "$@"; do
  export NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 > $O/c3_${V}_$rep.json 2> $O/c3_${V}_$rep.err || echo "c3 $V failed"
  line $O/c3_${V}_$rep.json "$V rep$rep c3"
  if [ -n "$SMALL" ] && [ $rep = 1 ]; then
    timeout -k 10 200 python bench.py --workload c1_uniform_100k --steps 200 --warmup 20 --cpu-sample 0 > $O/c1_${V}.json 2> $O/c1_${V}.err || echo "c1 $V failed"
    line $O/c1_${V}.json "$V c1"
    timeout -k 10 200 python bench.py --workload c2_scene_1m --steps 100 --warmup 10 --cpu-sample 0 > $O/c2_${V}.json 2> $O/c2_${V}.err || echo "c2 $V failed"
    line $O/c2_${V}.json "$V c2"
    timeout -k 10 200 python bench.py --points 1250000 --steps 100 --warmup 10 --cpu-sample 0 > $O/p1250k_${V}.json 2> $O/p1250k_${V}.err || echo "p1250k $V failed"
    line $O/p1250k_${V}.json "$V c3@1.25M"
  fi
done
done
unset NIMRUD_HIP_LIBRARY
if [ -z "$NOPROF" ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err
cd $GRAFT_REPO_ROOT
find $O/prof -name "*_kernel_trace.csv" -delete
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -22 $O/kernel_stats.csv | cut -c1-150
fi
