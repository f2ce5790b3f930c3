"""summarise tools/gpu_pmc_variant.sh: per-scale duration and per-wave counters of the search kernel."""
import collections, csv, glob, os, sys
out = sys.argv[1]
for v in sys.argv[2:]:
    rows = []
    for f in glob.glob(os.path.join(out, "trace_" + v, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    ks = [r for r in rows if r["Kernel_Name"].startswith("void k_scale_features<")]
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ks]
    print(v, "search launches", len(d), "last step per scale (us):", ["%.0f" % x for x in d[-5:]])
    for i in (1, 2):
        rows = []
        for f in glob.glob(os.path.join(out, "pmc_%s_%d" % (v, i), "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        order = {}
        for r in rows:
            if not r["Kernel_Name"].startswith("void k_scale_features<"):
                continue
            order.setdefault(r["Dispatch_Id"], len(order))
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        ids = sorted(per, key=lambda k: int(k))[-5:]
        for k in ids:
            c = per[k]
            print("  ", v, "dispatch", k, " ".join("%s=%.0f" % (n.replace("SQ_", ""), c[n] / 156250.0) for n in sorted(c)))
