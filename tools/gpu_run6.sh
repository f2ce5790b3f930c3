cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "forest or config5 or classify" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do for M in 0 1; do
NIMRUD_BENCH_FOREST_EPILOGUE=$M timeout -k 10 200 python bench.py --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 0 > $O/c5_m${M}_$rep.json 2> $O/c5_m${M}_$rep.err || echo "c5 failed"
python - <<PY
import json
d=json.loads(open("$O/c5_m${M}_$rep.json").read().strip().splitlines()[-1])
print("c5 mode$M rep$rep ms/step %.3f"%d["ms_per_step"], "search %.3f"%d["stage_ms_per_step"]["search_feature_kernel"], "forest", d["forest"]["ms_per_step"], "frac", d["forest"]["roofline"]["frac"])
PY
done; done
