# per-scale durations (kernel trace, one launch per scale) and SQ counters of the search kernel for library
# variants:  bash tools/gpu_pmc_variant.sh <outdir> <variant>...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  export NIMRUD_HIP_LIBRARY=$R/build_abl/lib_$V.so
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$V -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --fuse-scales 0 > $O/trace_$V.json 2> $O/trace_$V.err || echo "trace $V failed"
  i=0
  for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_BRANCH"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $O/pmc_${V}_$i -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --fuse-scales 0 > $O/pmc_${V}_$i.json 2> $O/pmc_${V}_$i.err || echo "pmc $V $i failed"
  done
  echo "$V done"
done
python3 $R/tools/pmc_variant_summary.py $O "$@"
