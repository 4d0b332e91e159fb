import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
import tscode_amd
from tscode_amd.synthetic import make_ensemble, quat_to_mat
from tscode_amd.pipeline import DevicePipeline
def run(ens, algo, steps=5):
    pipe = DevicePipeline(ens, device_index=0, mode=0)
    pipe.set_option("prune_algo", algo)
    res = pipe.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): res = pipe.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    st = res["stats"]
    return dt, res["n_pass"], res["n_keep"], sum(s["pairs_screened"] for s in st), sum(s["pairs_computed"] for s in st), sum(s["candidates"] for s in st), sum(s["pairs_evaluated"] for s in st)
for name, kw in (("C3", dict(sigma_rot_deg=1.0, sigma_t=0.03, seed=1003)), ("C3hard", dict(sigma_rot_deg=6.0, sigma_t=0.12, seed=7003))):
    ens = make_ensemble(100000, (25, 25), children=10, shell=(4.0, 9.0), **kw)
    for algo in (2, 1):
        print(name, "algo", algo, run(ens, algo), flush=True)
