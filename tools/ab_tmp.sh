timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "culled or partitioned or full_size or prune_large_golden or sharded" > gpurun_out/t_wl.log 2>&1; tail -2 gpurun_out/t_wl.log
for i in 1 2; do timeout -k 10 200 python3 bench.py --config C4 --steps 8 --warmup 2 --no-cpu --no-side-leg --no-selftest > gpurun_out/x.json 2>/dev/null; python3 - <<'PY'
import json,sys
d=json.loads(open('gpurun_out/x.json').read().strip().split('\n')[-1]);print("C4", round(d['ms_per_step'],4), d["config"].get("parity_vs_recorded_oracle"), [(p["k"], p["tile_ms"]) for p in d["passes"] if p["tile_ms"]>0.3])
PY
done
