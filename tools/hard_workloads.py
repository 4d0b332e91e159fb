#!/usr/bin/env python
"""Workloads on which the prune's descriptor sieve has the least to offer, beside the favourable C3:
  spread        100 000 x 50 atoms like C3, but the children scatter AROUND the threshold (6 degrees, 0.12 A): many pairs the quartic
                tests cannot decide
  unscreenable  100 000 x 30 heavy atoms whose rotation-invariant descriptors all coincide (synthetic.make_unscreenable): the screen
                drops nothing, every pair a pass looks at reaches H; reference-exact mode (the cache ends most rows early) and
                cache-free mode (nothing does)
Per leg: ms per step with data resident in HBM for the automatic kernel choice, the descriptor sieve and the register-tiled
all-pairs kernel; pairs screened, H formed, exact-path candidates; the CPU oracle on a bounded sample of the same input.
usage (GPU box): python tools/hard_workloads.py [--n 100000] [--cpu-sample 12000] > profiles/r03_hard_workloads.json"""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tscode_amd
from tscode_amd.engine import Engine
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_ensemble, make_unscreenable

ALGOS = {"auto": 0, "sieve": 2, "tile": 1}


def summarize(stats):
    sc, fm = sum(s["pairs_screened"] for s in stats), sum(s["pairs_computed"] for s in stats)
    return {"passes": len(stats), "pairs_screened": sc, "H_formed": fm, "screen_pass_rate": (fm / sc) if sc else None,
            "exact_path": sum(s["candidates"] for s in stats), "reference_pair_evaluations": sum(s["pairs_evaluated"] for s in stats),
            "kernels": sorted({s["algo"] for s in stats})}


def pipeline_leg(ens, mode, steps):
    out = {}
    for name, algo in ALGOS.items():
        pipe = DevicePipeline(ens, device_index=0, mode=mode)
        pipe.set_option("prune_algo", algo)
        res = pipe.step()
        torch.cuda.synchronize()
        gc.collect(); gc.freeze()       # (a generation-2 collection costs about 40 ms: bench.py, timed_loop)
        t0 = time.perf_counter()
        for _ in range(steps):
            res = pipe.step()
        torch.cuda.synchronize()
        out[name] = {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "n_pass_clash": int(res["n_pass"]), "n_survivors": int(res["n_keep"]),
                     **summarize(res["stats"])}
        del pipe
    return out


def prune_leg(heavy, mode, steps):
    out = {}
    dev = torch.device("cuda:0")
    d_heavy = torch.from_numpy(heavy).to(dev)
    mask = torch.empty(len(heavy), dtype=torch.uint8, device=dev)
    for name, algo in ALGOS.items():
        eng = Engine(0)
        eng.set_option("prune_algo", algo)
        stats = eng.prune_heavy_dev(d_heavy, len(heavy), heavy.shape[1], 0.5, mode, mask)
        gc.collect(); gc.freeze()
        t0 = time.perf_counter()
        for _ in range(steps):
            stats = eng.prune_heavy_dev(d_heavy, len(heavy), heavy.shape[1], 0.5, mode, mask)
        out[name] = {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "n_survivors": int(mask.sum()), **summarize(stats)}
        del eng
    return out


def cpu_prune(heavy, mode):
    import oracle
    oracle.build()
    oracle.set_num_threads(max(1, min(oracle.num_threads(), len(os.sched_getaffinity(0)), 32)))
    t0 = time.perf_counter()
    ref = oracle.prune_heavy(heavy, 0.5, mode=mode)
    dt = time.perf_counter() - t0
    return {"seconds": dt, "structures": len(heavy), "structures_per_s": len(heavy) / dt, "n_survivors": int(ref["mask"].sum()),
            "pair_evaluations": int(sum(s["pairs_evaluated"] for s in ref["stats"])), "threads": oracle.num_threads(),
            "kind": "port (oracle, chunk-parallel like the reference's prange)"}


def measure(n=100_000, n_cpu=12_000, steps=5):
    """n_cpu = 0: no CPU leg."""
    out = {"what": __doc__.split("usage")[0].strip(), "n": n, "device": torch.cuda.get_device_name(0), "legs": {}}
    spread = dict(children=10, sigma_rot_deg=6.0, sigma_t=0.12, shell=(4.0, 9.0))
    ens = make_ensemble(n, (25, 25), 7003, **spread)
    leg = {"workload": f"{n} conformers x 50 atoms (30 heavy), children 6 degrees / 0.12 A around their parent, seed 7003; embed + clash + prune, mode 0",
           "n": n, "gpu": pipeline_leg(ens, 0, steps)}
    del ens
    if n_cpu:
        import oracle
        sample = make_ensemble(n_cpu, (25, 25), 7003, **spread)
        poses = sample.poses()
        poses = poses[oracle.compenetration_mask(poses, sample.ids, 1.5, 0)]
        leg["cpu"] = {**cpu_prune(np.ascontiguousarray(poses[:, sample.atomnos != 1]), 0), "sample": f"prune only, the same generator at {n_cpu} conformers"}
    out["legs"]["spread"] = leg
    heavy = make_unscreenable(n)
    for mode in (0, 1):
        leg = {"workload": f"{n} structures x 30 heavy atoms with coinciding descriptors (synthetic.make_unscreenable, seed 99); prune only, mode {mode}",
               "n": n, "gpu": prune_leg(heavy, mode, steps)}
        if n_cpu:
            leg["cpu"] = {**cpu_prune(make_unscreenable(n_cpu), mode), "sample": f"the same generator at {n_cpu} structures"}
        out["legs"][f"unscreenable_mode{mode}"] = leg
    return out


def main():
    n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 100_000
    n_cpu = int(sys.argv[sys.argv.index("--cpu-sample") + 1]) if "--cpu-sample" in sys.argv else 12_000
    print(json.dumps(measure(n, n_cpu), indent=1))


if __name__ == "__main__":
    main()
