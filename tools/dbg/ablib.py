"""A/B of library builds in alternating subprocesses: python tools/dbg/ablib.py lib1.so lib2.so ...  (C3 pipeline, events off)"""
import os, subprocess, sys, statistics
libs = sys.argv[1:]
code = r'''
import sys, time
sys.path.insert(0, ".")
import torch
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
cfg = sys.argv[1]
ens = make_config(cfg)
pipe = DevicePipeline(ens, device_index=0, mode=0)
pipe.set_option("pass_timing", 0)
steps = 60 if cfg == "C3" else 6
for _ in range(5): pipe.step()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps): r = pipe.step()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps * 1e3)
print(best, r["n_keep"])
'''
cfg = os.environ.get("AB_CFG", "C3")
res = {l: [] for l in libs}
for rnd in range(4):
    for l in libs:
        env = dict(os.environ, TSCODE_AMD_LIB=os.path.abspath(l), TSCODE_AMD_LAX="1")
        out = subprocess.run([sys.executable, "-c", code, cfg], env=env, capture_output=True, text=True)
        try:
            res[l].append(float(out.stdout.split()[0]))
        except Exception:
            print(l, "FAILED", out.stderr[-500:])
for l, v in res.items():
    print(f"{l}: mean {statistics.mean(v):.4f} min {min(v):.4f}  {['%.4f' % x for x in v]}")
