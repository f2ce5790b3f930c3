#!/bin/bash
# run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_profiles_r2.sh <outdir-under-gpurun_out>
# clean bench lines (configs 1, 2, 3, 5), kernel-trace runs (configs 1-5 and the field operator), and one
# rocprofv3 --pmc run per counter set for config 3 and the forest step of config 5 (counters are collected in
# their own runs, never together with a trace).  summarise afterwards with tools/pmc_summary_r2.py.
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r2_final}
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 python3 "$R/bench.py" --steps 10 --warmup 3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err" || echo "bench c3 failed"
timeout -k 10 300 python3 "$R/bench.py" --workload c5_scene_10m_rf --steps 10 --warmup 3 --cpu-sample 30000 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" || echo "bench c5 failed"
timeout -k 10 300 python3 "$R/bench.py" --workload c1_uniform_100k --steps 50 --warmup 10 --cpu-sample 0 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err" || echo "bench c1 failed"
timeout -k 10 300 python3 "$R/bench.py" --workload c2_scene_1m --steps 20 --warmup 5 --cpu-sample 0 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err" || echo "bench c2 failed"
timeout -k 10 300 python3 "$R/bench.py" --steps 10 --warmup 3 --cpu-sample 0 --fuse-scales 0 > "$OUT/bench_c3_perscale.json" 2> "$OUT/bench_c3_perscale.err" || echo "bench c3 per-scale failed"
echo "benches done"
cd /tmp && export TMPDIR=/tmp
trace() {   # name, command...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- "$@" > "$OUT/trace_$name.out" 2> "$OUT/trace_$name.err" || echo "trace $name failed"
  # gpurun copies back at most 64 MiB: the per-dispatch traces are large and only the stats are summarised
  find "$OUT/trace_$name" -name "*_kernel_trace.csv" -delete
  echo "trace $name done"
}
trace c3 python3 "$R/bench.py" --steps 5 --warmup 2 --cpu-sample 0
trace c5 python3 "$R/bench.py" --workload c5_scene_10m_rf --steps 5 --warmup 2 --cpu-sample 0
trace c1 python3 "$R/bench.py" --workload c1_uniform_100k --steps 20 --warmup 5 --cpu-sample 0
trace c2 python3 "$R/bench.py" --workload c2_scene_1m --steps 10 --warmup 3 --cpu-sample 0
trace c4 python3 "$R/tools/config4_timing.py" 20000000
trace field python3 "$R/tools/field_timing.py"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1)); D=$OUT/pmc_c3_$i; mkdir -p "$D"
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 1 --warmup 1 --cpu-sample 0 > "$D/bench.json" 2> "$D/err.log" || echo "pmc pass failed: $P"
  echo "pmc c3 $i done"
done
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); D=$OUT/pmc_c5_$i; mkdir -p "$D"
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$D" -- python3 "$R/bench.py" --workload c5_scene_10m_rf --steps 1 --warmup 1 --cpu-sample 0 > "$D/bench.json" 2> "$D/err.log" || echo "pmc pass failed: $P"
  echo "pmc c5 $i done"
done
ls "$OUT"
