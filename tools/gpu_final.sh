cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_final8; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 exit $?"
timeout -k 10 400 python bench.py --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 30000 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit $?"
NIMRUD_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --points 4000000 --steps 3 --warmup 1 > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank_rehearsal.err; echo "rehearsal exit $?"
python - <<PY
import json
for f in ("bench_default","bench_c3","bench_c5","bench_2rank_rehearsal"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step %.3f"%d["ms_per_step"], "value %.4g"%d["value"], d["unit"], "frac %.3f"%d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], (d["roofline"].get("valu") or {}).get("valu_instr_per_wave"), d["config"].get("halo_points_exchanged_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 600 python tests/fuzz_parity.py 600 31415 > $O/fuzz_a.log 2>&1; echo "fuzz a exit $?"; tail -1 $O/fuzz_a.log
timeout -k 10 600 python tests/fuzz_parity.py 600 27182 > $O/fuzz_b.log 2>&1; echo "fuzz b exit $?"; tail -1 $O/fuzz_b.log
