set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
grep -q "pytest exit 0" $O/pytest.log || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > $O/bench_c3_fused.json 2> $O/bench_c3_fused.err; echo "bench exit $?"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --fuse-scales 0 > $O/bench_c3_perscale.json 2> $O/bench_c3_perscale.err; echo "bench exit $?"
timeout -k 10 300 python bench.py --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 0 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit $?"
NIMRUD_BENCH_FUSED_FOREST=0 timeout -k 10 300 python bench.py --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 0 > $O/bench_c5_unfused.json 2> $O/bench_c5_unfused.err; echo "bench c5 unfused exit $?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5_scene_10m_rf --steps 5 --warmup 2 --cpu-sample 0 > $GRAFT_REPO_ROOT/$O/trace_c5.json 2> $GRAFT_REPO_ROOT/$O/trace_c5.err; echo "trace exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --fuse-scales 0 > $GRAFT_REPO_ROOT/$O/trace_c3.json 2> $GRAFT_REPO_ROOT/$O/trace_c3.err; echo "trace exit $?"
