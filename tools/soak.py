"""Soak: many consecutive steps of the resident pipeline, every step's survivor mask compared with the first (and with the
recorded oracle result): catches ordering races between the streams / tickets that a handful of test steps would miss."""
import hashlib, json, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
cfg, steps = (sys.argv[1] if len(sys.argv) > 1 else "C3"), int(sys.argv[2]) if len(sys.argv) > 2 else 300
ens = make_config(cfg)
exp = json.load(open("tests/golden/expected_full.json")).get(f"{cfg}:{ens.n_poses}:mode0")
pipe = DevicePipeline(ens, device_index=0, mode=0)
bad, first, t0 = 0, None, time.time()
for i in range(steps):
    res = pipe.step()
    d = hashlib.sha256(np.packbits(pipe.h_keep[:res["n_pass"]].numpy().astype(bool)).tobytes()).hexdigest()[:16]
    key = (res["n_pass"], res["n_keep"], d, tuple(s["pairs_evaluated"] for s in res["stats"]))
    if first is None:
        first = key
        if exp:
            assert (exp["n_pass"], exp["n_keep"], exp["keep_sha256_16"]) == key[:3], key[:3]
    bad += key != first
print(f"{cfg}: {steps} steps, {bad} differ from the first (which equals the recorded oracle result: {bool(exp)}), {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
