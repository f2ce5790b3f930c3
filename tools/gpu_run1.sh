set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench_c3.json 2> gpurun_out/r2a/bench_c3.err; echo "bench exit $?"
timeout -k 10 300 python bench.py --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r2a/bench_c5.json 2> gpurun_out/r2a/bench_c5.err; echo "bench c5 exit $?"
NIMRUD_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --points 2000000 --steps 3 --warmup 1 > gpurun_out/r2a/bench_2rank_rehearsal.json 2> gpurun_out/r2a/bench_2rank_rehearsal.err; echo "rehearsal exit $?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/trace_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5_scene_10m_rf --steps 5 --warmup 2 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r2a/trace_c5.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2a/trace_c5.err; echo "trace exit $?"
