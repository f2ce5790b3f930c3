cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2h}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "pytest exit 0" $O/pytest.log || exit 1
timeout -k 10 300 python tests/fuzz_parity.py 300 > $O/fuzz.log 2>&1; echo "fuzz exit $?"; tail -3 $O/fuzz.log
for rep in 1 2; do
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > $O/c3_$rep.json 2> $O/c3_$rep.err || echo "c3 failed"
python - <<PY
import json
d=json.loads(open("$O/c3_$rep.json").read().strip().splitlines()[-1])
print("c3 rep$rep ms/step %.3f"%d["ms_per_step"], "search %.3f"%d["stage_ms_per_step"]["search_feature_kernel"], "index %.3f"%d["stage_ms_per_step"]["index_build"], "order %.3f"%d["stage_ms_per_step"]["cell_keys_and_sort"])
PY
done
