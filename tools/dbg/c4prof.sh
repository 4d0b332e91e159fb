#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/tr_c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_c4 -- python3 bench.py --config C4 --steps 4 --warmup 1 --no-cpu --no-side-leg --pass-timing 0 > /dev/null 2>&1
f=$(find /tmp/tr_c4 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max([int(r["Calls"]) for r in rows if "k_init_run" in r["Name"]] or [1])
tot = 0
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    tot += int(r["TotalDurationNs"])
    print(f'{r["Name"][:58]:58s} calls/step {int(r["Calls"])/steps:5.1f} avg {float(r["AverageNs"])/1e3:8.2f} us per step {int(r["TotalDurationNs"])/steps/1e3:8.1f}')
print("sum per step", tot/steps/1e3, "steps", steps)
PY
