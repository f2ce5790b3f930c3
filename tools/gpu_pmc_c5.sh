# SQ counters of the config 5 step (forest behind the last scale), separate passes; run on the GPU box
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2e}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1)); D=$O/pmc$i; mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5_scene_10m_rf --steps 1 --warmup 1 --cpu-sample 0 > $D/bench.json 2> $D/err.log || echo "pmc pass failed: $P"
  echo "done $P"
done
python3 - <<PY
import csv,glob,collections,json
out={}
for d in sorted(glob.glob("$O/pmc*")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_scale_features" not in k and "k_index_fused" not in k: continue
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,c in acc.items():
            for name,v in c.items():
                out.setdefault(k,{})[name]=sum(v)/len(v)
json.dump(out,open("$O/pmc_summary.json","w"),indent=1)
for k,c in out.items():
    print(k)
    for n,v in sorted(c.items()): print("   %-32s %.4g"%(n,v))
PY
