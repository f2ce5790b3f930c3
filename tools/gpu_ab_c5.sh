# A/B of library variants on config 5 (and the forest tests): bash tools/gpu_ab_c5.sh <outdir> <variant>...
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for V in "$@"; do
  export NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so
  if [ $rep = 1 ]; then timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "forest or config5_forest or classify" > $O/pytest_$V.log 2>&1; tail -1 $O/pytest_$V.log; fi
  timeout -k 10 200 python bench.py --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 0 > $O/c5_${V}_$rep.json 2> $O/c5_${V}_$rep.err || echo "c5 $V failed"
  python - <<PY
import json
d=json.loads(open("$O/c5_${V}_$rep.json").read().strip().splitlines()[-1])
print("$V rep$rep c5 ms/step %.3f"%d["ms_per_step"], "search %.3f"%d["stage_ms_per_step"]["search_feature_kernel"], "forest %.3f"%d["forest"]["ms_per_step"], "features only %.3f"%d["forest"]["features_only_ms_per_step"])
PY
done
done
