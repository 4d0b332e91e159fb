timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core or prune_golden or prune_c2 or prune_edge or non_finite or random_small or prune_large_golden or culled or full_size" > gpurun_out/t_wl.log 2>&1; tail -2 gpurun_out/t_wl.log
for i in 1 2; do timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-side-leg > gpurun_out/x.json 2>/dev/null; python3 - <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().split('\n')[-1]);print("C3", round(d['ms_per_step'],4), round(d["events_off"]["ms_per_step"],4), d["config"].get("parity_vs_recorded_oracle"), [(p["k"], p["tile_ms"], round(p["ms"]-p["tile_ms"],4)) for p in d["passes"] if p["tile_ms"]>0])
PY
done
timeout -k 10 200 python3 bench.py --config C4 --steps 6 --warmup 2 --no-cpu --no-side-leg --no-selftest > gpurun_out/x.json 2>/dev/null; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/x.json').read().strip().split('\n')[-1]);print("C4", round(d['ms_per_step'],4), d["config"].get("parity_vs_recorded_oracle"), [(p["k"], p["tile_ms"]) for p in d["passes"]])
PY
