#!/bin/bash
# run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_profiles_r3.sh <outdir-under-gpurun_out>
# clean bench lines (configs 1, 2, 3, 5, the reference's ladder shape, index emission, a 1.25 M-point tile),
# kernel-trace runs, one rocprofv3 --pmc run per counter set for config 3 (counters are collected in their own
# runs, never together with a trace), the issue-rate table, and two gloo rehearsals of the multi-rank step.
# summarise afterwards with tools/pmc_summary_r3.py.
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r3_final}
rm -rf "$OUT"; mkdir -p "$OUT"
b() {  # name args...
  local name=$1; shift
  timeout -k 10 400 python3 "$R/bench.py" "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || echo "bench $name failed"
}
b c3 --steps 20 --warmup 3
b c5 --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 30000
b c1 --workload c1_uniform_100k --steps 200 --warmup 20 --cpu-sample 0
b c2 --workload c2_scene_1m --steps 100 --warmup 10 --cpu-sample 0
b ref_ladder --workload ref_ladder_10m --steps 10 --warmup 3 --cpu-sample 0
b c3_perscale --steps 10 --warmup 3 --cpu-sample 0 --fuse-scales 0
b c3_tile_1250k --points 1250000 --steps 100 --warmup 10 --cpu-sample 0
b emit_indices_c2 --workload c2_scene_1m --emit-indices --steps 5 --warmup 2 --cpu-sample 0
echo "benches done"
"$R/build_abl/issue_rate" "$OUT/issue_rate.json" > "$OUT/issue_rate.txt" 2>&1 || echo "issue rate failed"
cd /tmp && export TMPDIR=/tmp
trace() {   # name, command...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- "$@" > "$OUT/trace_$name.out" 2> "$OUT/trace_$name.err" || echo "trace $name failed"
  # gpurun copies back at most 64 MiB: the per-dispatch traces are large and only the stats are summarised
  find "$OUT/trace_$name" -name "*_kernel_trace.csv" -delete
  echo "trace $name done"
}
trace c3 python3 "$R/bench.py" --steps 20 --warmup 3 --cpu-sample 0
trace c5 python3 "$R/bench.py" --workload c5_scene_10m_rf --steps 5 --warmup 2 --cpu-sample 0
trace c1 python3 "$R/bench.py" --workload c1_uniform_100k --steps 50 --warmup 5 --cpu-sample 0
trace c2 python3 "$R/bench.py" --workload c2_scene_1m --steps 20 --warmup 3 --cpu-sample 0
trace ref_ladder python3 "$R/bench.py" --workload ref_ladder_10m --steps 5 --warmup 2 --cpu-sample 0
trace c3_tile_1250k python3 "$R/bench.py" --points 1250000 --steps 50 --warmup 5 --cpu-sample 0
trace emit_indices_c2 python3 "$R/bench.py" --workload c2_scene_1m --emit-indices --steps 3 --warmup 1 --cpu-sample 0
trace c4 python3 "$R/tools/config4_timing.py" 20000000
timeout -k 10 500 python3 "$R/tools/config4_timing.py" > "$OUT/c4_full_50m.log" 2>&1 || echo "c4 full failed"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1)); D=$OUT/pmc_c3_$i; mkdir -p "$D"
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 1 --warmup 1 --cpu-sample 0 > "$D/bench.json" 2> "$D/err.log" || echo "pmc pass failed: $P"
  echo "pmc c3 $i done"
done
cd "$R"
# the multi-rank step rehearsed on one GPU (HIP kernels, gloo transport staged through the host; not a benchmark)
for N in 2 3; do
  NIMRUD_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29517 + N)) bench.py --gpus $N --points 4000000 --steps 3 --warmup 1 > "$OUT/bench_rehearsal_${N}rank.json" 2> "$OUT/bench_rehearsal_${N}rank.err" || echo "rehearsal $N failed"
done
ls "$OUT"
