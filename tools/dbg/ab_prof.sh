#!/bin/bash
export TMPDIR=/tmp
for d in ab_r2 .; do
  (cd $d && rm -rf /tmp/tr_$$ && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$$ -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-side-leg --pass-timing 0 > /dev/null 2>&1
   f=$(find /tmp/tr_$$ -name "*kernel_stats.csv" | head -1)
   echo "== $d"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max([int(r["Calls"]) for r in rows if "k_init_run" in r["Name"]] or [12])
tot = 0
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    tot += int(r["TotalDurationNs"])
    print(f'{r["Name"][:60]:60s} calls/step {int(r["Calls"])/steps:5.1f} avg {float(r["AverageNs"])/1e3:7.2f} us per step {int(r["TotalDurationNs"])/steps/1e3:7.1f}')
print("sum per step", tot/steps/1e3, "steps", steps)
PY
  )
done
