# what the driver does at round end (smoke, pytest -m gpu, default bench) plus two fuzz runs: bash tools/gpu_final_r3.sh <outdir>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3_last}; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "Memory access fault" $O/pytest.log && exit 1
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default: ms/step %.3f value %.4g frac %.3f busy %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["valu"].get("valu_issue_busy_frac")))
PY
timeout -k 10 600 python tests/fuzz_parity.py 600 31415 > $O/fuzz_a.log 2>&1; echo "fuzz a exit $?"; tail -2 $O/fuzz_a.log
timeout -k 10 600 python tests/fuzz_parity.py 600 27182 > $O/fuzz_b.log 2>&1; echo "fuzz b exit $?"; tail -2 $O/fuzz_b.log
