#!/bin/bash
# run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_profiles.sh <outdir-under-gpurun_out>
# one clean bench line, one kernel-trace run, and one rocprofv3 --pmc run per counter set (counters are
# collected in their own runs, never together with a trace).  summarise afterwards with tools/pmc_summary.py.
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-final}
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 python3 "$R/bench.py" --steps 10 --warmup 3 > "$OUT/bench_clean.json" 2> "$OUT/bench_clean.err" || echo "bench failed"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 5 --warmup 2 --cpu-sample 0 > "$OUT/trace.json" 2> "$OUT/trace.err" || echo "trace failed"
for P in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  D=$OUT/${P:0:12}
  D=${D// /_}
  mkdir -p "$D"
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 1 --warmup 1 --cpu-sample 0 > "$D/bench.json" 2> "$D/err.log" || echo "pmc pass failed: $P"
  echo "done $P"
done
ls "$OUT"
