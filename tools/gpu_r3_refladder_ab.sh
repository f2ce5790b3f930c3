cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3l
for V in slab11 slab9; do
export NIMRUD_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_abl/lib_$V.so
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3l/prof_$V -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload ref_ladder_10m --steps 5 --warmup 2 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r3l/ref_$V.json 2>/dev/null)
find gpurun_out/r3l/prof_$V -name "*_kernel_trace.csv" -delete
python3 -c "
import csv
for r in list(csv.DictReader(open('gpurun_out/r3l/prof_$V/trace_kernel_stats.csv')))[:3]:
    print('$V', r['Name'][:45], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))"
done
