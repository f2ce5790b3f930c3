# A/B of library variants on one box: bash tools/gpu_ab.sh <outdir> <variant>...   (libs in build_abl/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for V in "$@"; do
  export NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > $O/c3_${V}_$rep.json 2> $O/c3_${V}_$rep.err || echo "c3 $V failed"
  timeout -k 10 200 python bench.py --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 0 > $O/c5_${V}_$rep.json 2> $O/c5_${V}_$rep.err || echo "c5 $V failed"
  python - <<PY
import json
for w in ("c3","c5"):
    try:
        d=json.loads(open("$O/%s_${V}_$rep.json"%w).read().strip().splitlines()[-1])
        print("$V rep$rep", w, "ms/step %.3f"%d["ms_per_step"], "search %.3f"%d["stage_ms_per_step"]["search_feature_kernel"], "index %.3f"%d["stage_ms_per_step"]["index_build"], "order %.3f"%d["stage_ms_per_step"]["cell_keys_and_sort"], "forest", d.get("forest",{}).get("ms_per_step"), "extra passes", d.get("extra_search_passes_per_scale"))
    except Exception as e:
        print("$V", w, "ERR", e)
PY
done
done
