#!/bin/bash
# Runs on the GPU box (through gpurun): per-kernel durations of a bench command (rocprofv3 --kernel-trace --stats), printed per step.
# usage: tools/kstats.sh OUTDIR [bench.py arguments]        (outputs under gpurun_out/OUTDIR)
set -e
P=gpurun_out/$1; shift
mkdir -p "$P"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-side-leg $* > "$P/bench.json" 2> "$P/trace.err"
python3 - "$P" <<'PY'
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)   # (the newest run of this directory)
rows = list(csv.DictReader(open(f)))
# steps really executed under the profiler: warm-up + the timed loop + bench.py's events-off and detail loops -- counted, not assumed:
# k_init_run runs exactly once per prune run, i.e. once per step
steps = max([int(r["Calls"]) for r in rows if r["Name"].startswith("tsc::k_init_run") or "k_init_run" in r["Name"]] or [12])
print("steps under the profiler:", steps)
tot = 0
for r in rows:
    tot += int(r["TotalDurationNs"])
    print(f'{r["Name"][:70]:70s} calls/step {int(r["Calls"]) / steps:6.1f}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  per step {int(r["TotalDurationNs"]) / steps / 1e3:8.1f} us')
print("sum of kernel time per step (us):", tot / steps / 1e3)
PY
