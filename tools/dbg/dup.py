import os, subprocess, sys, json
# per-pass kernel time with the stage-1 gathers doubled (TSC_DUP=1) against the same library without (TSC_DUP=0)
for cfg in ("C3", "C4"):
    for dup in ("0", "1"):
        env = dict(os.environ, TSCODE_AMD_LIB=os.path.abspath("ab_libs/dup.so"), TSCODE_AMD_LAX="1", TSC_DUP=dup)
        out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--no-cpu", "--no-side-leg"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(cfg, "dup", dup, "ms", round(d["ms_per_step"], 4), "off", round(d["events_off"]["ms_per_step"], 4), "parity", d["config"]["parity_vs_recorded_oracle"],
                  [(p["k"], p["tile_ms"]) for p in d["passes"] if p["tile_ms"] > 0.06])
        except Exception as e:
            print(cfg, dup, "FAILED", out.stderr[-400:])
