# SQ counters of the forest launches (mode 0: rows in spatial order; mode 2: the same with the trees' top in LDS)
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 2; do
  for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"; do
    D=$GRAFT_REPO_ROOT/$O/pmc_m${m}_$(echo $P | cut -c1-12 | tr ' ' '_'); mkdir -p $D
    NIMRUD_BENCH_FOREST_EPILOGUE=$m timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5_scene_10m_rf --steps 1 --warmup 1 --cpu-sample 0 > $D/bench.json 2> $D/err.log || echo "pmc failed m=$m $P"
  done
done
cd $GRAFT_REPO_ROOT
python3 - $O <<'PY'
import csv, glob, sys, json, collections
O = sys.argv[1]
out = {}
for m in ("0", "2"):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("%s/pmc_m%s_*/**/*counter_collection.csv" % (O, m), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_forest" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    disp = max(n.values()) if n else 0
    # per dispatch (the run launches the forest kernel a few times), then per wave
    launches = {k: n[k] for k in n}
    waves = agg.get("SQ_WAVES", 0.0) / max(n.get("SQ_WAVES", 1), 1)
    out["mode_%s" % m] = {k: agg[k] / n[k] / max(waves, 1.0) for k in agg}
    out["mode_%s" % m]["waves_per_launch"] = waves
json.dump(out, open("%s/forest_sq_counters.json" % O, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
