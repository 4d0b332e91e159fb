#!/bin/bash
# Runs on the GPU box (through gpurun): a kernel trace and a plain run of the bench command, the figures an A/B of one kernel edit looks at.
# usage: tools/quick_ab.sh NAME [bench.py arguments]        (outputs under gpurun_out/NAME)
set -e
P=gpurun_out/$1; shift
mkdir -p "$P"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-side-leg $* > "$P/b_trace.json" 2> "$P/trace.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu --no-side-leg $* > "$P/b.json" 2> "$P/b.err"
python3 - "$P" <<'PY'
import csv, glob, json, sys
P = sys.argv[1]
d = json.loads(open(P + "/b.json").read().strip().split("\n")[-1])
print("ms_per_step", round(d["ms_per_step"], 4), "events_off", round(d["events_off"]["ms_per_step"], 4), "front", round(d["roofline_front"]["ms"], 4),
      "stages", {k: round(v, 4) for k, v in d["stage_ms_per_step"].items()})
f = glob.glob(P + "/trace/*/*kernel_stats.csv")[0]
tot = 0
for r in csv.DictReader(open(f)):
    tot += int(r["TotalDurationNs"])
    print(f'{r["Name"].split("(")[0][-44:]:44s} {int(r["Calls"]) / 13:5.1f}/step {float(r["AverageNs"]) / 1e3:8.2f} us')
print("kernel sum per step (13 steps)", round(tot / 13 / 1e3, 1), "us")
PY
