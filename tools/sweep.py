"""Option sweep on the GPU box: every combination is benched in ONE process (alternating, several rounds) so that box-to-box
and run-to-run drift cancels.  usage: python tools/sweep.py [config] ; prints mean events-off ms per step for each combination."""
import itertools, json, statistics, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tscode_amd import get_engine
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
ens = make_config(cfg)
pipe = DevicePipeline(ens, device_index=0, mode=0)
eng = pipe.engine
eng.set_option("pass_timing", 0)
grid = {"seg_cols": [0, 256, 512, 1024], "drain_min": [16, 32, 64], "local_max_chunk": [128, 256, 512]}
if cfg != "C3":
    grid = {"seg_cols": [0, 1024, 2048, 4096], "drain_min": [32, 64], "local_max_chunk": [256]}
defaults = {"seg_cols": 0, "drain_min": 64, "local_max_chunk": 256}
if len(sys.argv) > 2:      # a grid of one's own: python tools/sweep.py C3 '{"early_basis": [0, 1], "sieve_cpl": [1, 2, 4]}' (first values = defaults)
    grid = json.loads(sys.argv[2])
    defaults = {k: v[0] for k, v in grid.items()}
combos = [dict(zip(grid, v)) for v in itertools.product(*grid.values())]
steps = 20 if cfg == "C3" else 4
res = {i: [] for i in range(len(combos))}
ref = None
for rnd in range(4 if cfg == "C3" else 2):
    for i, c in enumerate(combos):
        for k, v in c.items():
            eng.set_option(k, v)
        pipe.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = pipe.step()
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / steps * 1e3)
        if ref is None:
            ref = (r["n_pass"], r["n_keep"])
        assert (r["n_pass"], r["n_keep"]) == ref
for k, v in defaults.items():
    eng.set_option(k, v)
rows = sorted(((statistics.mean(v), statistics.pstdev(v), combos[i]) for i, v in res.items()), key=lambda x: x[0])
for m, sd, c in rows:
    print(f"{m:8.4f} +- {sd:6.4f}  {c}{'   <- defaults' if c == defaults else ''}")
