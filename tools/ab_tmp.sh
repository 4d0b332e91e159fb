timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests_r5m.log 2>&1; tail -2 gpurun_out/gpu_tests_r5m.log
for c in C4 C5 C5chain C3; do timeout -k 10 200 python3 bench.py --config $c --no-cpu --no-side-leg --no-selftest > gpurun_out/x.json 2>/dev/null; python3 - "$c" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().split('\n')[-1]);print(sys.argv[1], round(d['ms_per_step'],4), round(d["events_off"]["ms_per_step"],4) if d.get("events_off") else None, d["config"].get("parity_vs_recorded_oracle"))
PY
done
