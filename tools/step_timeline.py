"""Per-dispatch timeline of ONE step from a rocprofv3 kernel trace: python tools/step_timeline.py TRACE_DIR [step_from_end]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
short = lambda n: ("k_rmsd_sieve_mm" if "sieve_mm" in n else n.split("(")[0].replace("void ", "").replace("tsc::", ""))[:40]
idx = [i for i, n in enumerate(names) if "k_init_run" in n]
s = idx[-back]
st = s
while st > 0 and "k_export_run" not in names[st - 1]:
    st -= 1
t0 = int(rows[st]["Start_Timestamp"])
for i in range(st, len(rows)):
    r = rows[i]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{short(names[i]):42s} start {(a - t0) / 1e3:8.1f} dur {(b - a) / 1e3:7.1f} grid {r.get('Grid_Size_X')}x{r.get('Grid_Size_Y')}")
    if "k_export_run" in names[i] and i > s:
        break
