# SQ counters and HBM bytes of EVERY kernel of one bench step (config 3 unless arguments are given):
#   bash tools/pmc_all_kernels.sh <outdir> [bench args...]      -> gpurun_out/<outdir>/pmc_all.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $O/pmc_$i -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 "$@" > $O/pmc_$i.json 2> $O/pmc_$i.err || echo "pmc $i failed"
done
python3 - $O <<'PY' > $O/pmc_all.txt
import collections, csv, glob, os, sys
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    # the LAST dispatch of each kernel name (the measured step, not the warm-up)
    last = {}
    for r in rows:
        last[r["Kernel_Name"]] = max(last.get(r["Kernel_Name"], -1), int(r["Dispatch_Id"]))
    for r in rows:
        if int(r["Dispatch_Id"]) == last[r["Kernel_Name"]]:
            tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    c = tot[k]
    print(k[:100])
    print("   " + " ".join("%s=%.4g" % (n.replace("SQ_", ""), c[n]) for n in sorted(c)))
# PMC_MATCH=<kernel name prefix> PMC_LAST=<n>: the last n dispatches of that kernel one by one (a ladder run scale by
# scale with --fuse-scales 0 launches the search kernel once per scale under one name)
match, last_n = os.environ.get("PMC_MATCH"), int(os.environ.get("PMC_LAST", "0"))
if match and last_n:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(match)]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last_n:]
        for r in rows:
            if int(r["Dispatch_Id"]) in ids:
                per[ids.index(int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    for i in sorted(per):
        print("dispatch %d of the last %d of %s" % (i, last_n, match))
        print("   " + " ".join("%s=%.4g" % (n.replace("SQ_", ""), per[i][n]) for n in sorted(per[i])))
PY
find $O -name "*counter_collection.csv" -delete
cat $O/pmc_all.txt
