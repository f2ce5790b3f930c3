"""print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace csv:
   python tools/step_timeline.py gpurun_out/prof_x"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_bounds_init')]
lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, len(rows))
t0 = int(rows[lo]['Start_Timestamp'])
tot = 0.0
for r in rows[lo:hi]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  {r['Kernel_Name'][:56]}")
print("sum of kernel time %.1f us, span %.1f us" % (tot, (int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3))
