"""Timeline of one step's kernels from a rocprofv3 --kernel-trace run (tools/kstats.sh OUTDIR): start / end in us relative to the
step's first kernel, queue, name.   usage: python3 tools/timeline.py gpurun_out/OUTDIR [n_kernels]"""
import csv
import glob
import os
import sys

src = sys.argv[1]
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = max(glob.glob(src + "/trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
firsts = [i for i, r in enumerate(rows) if "k_clash" in r["Kernel_Name"]]
i0 = firsts[len(firsts) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - 2):i0 + n_show]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{a:9.1f} {b:9.1f}  {b - a:7.1f} us  queue {r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
