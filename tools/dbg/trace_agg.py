"""aggregate a rocprofv3 kernel_trace.csv by (kernel, grid): calls, mean us.   usage: trace_agg.py DIR [name filter]"""
import csv, glob, os, sys, collections
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("tsc::", "").replace("void ", "")
    if flt and flt not in name:
        continue
    key = (name[:40], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0, 1e9, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
for (name, grid), (n, tot, lo, hi) in agg.items():
    print(f"{name:40s} grid {grid:8d} calls {n:5d} mean {tot / n:9.1f} us  min {lo:9.1f} max {hi:9.1f}")
