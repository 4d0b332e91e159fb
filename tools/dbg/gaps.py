"""timeline of one step from a rocprofv3 kernel_trace.csv: kernel, duration, gap to the previous kernel's end.  usage: gaps.py DIR [step index]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# steps begin at k_transform_describe / k_transform (first kernel of the front half)
starts = [i for i, r in enumerate(rows) if "k_init_run" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
lo = starts[which]
hi = starts[which + 1] if which + 1 < len(starts) else len(rows)
# back up to the front half of this step
while lo > 0 and "k_export_run" not in rows[lo - 1]["Kernel_Name"] and lo > starts[which] - 12:
    lo -= 1
prev_end = None
tot_k = tot_gap = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("tsc::", "").replace("void ", "")[:34]
    print(f"{name:34s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} dur {(e - s) / 1e3:8.2f} us  gap {gap:7.2f} us")
    tot_k += (e - s) / 1e3
    tot_gap += max(gap, 0.0) if prev_end else 0.0
    prev_end = max(e, prev_end or 0)
print("kernels", round(tot_k, 1), "gaps", round(tot_gap, 1), "span", round((prev_end - int(rows[lo]["Start_Timestamp"])) / 1e3, 1))
