"""Soak: many consecutive steps of the resident pipeline, every step's survivor mask compared with the first (and with the
recorded oracle result): catches ordering races between the streams / tickets that a handful of test steps would miss."""
import hashlib, json, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
cfg, steps = (sys.argv[1] if len(sys.argv) > 1 else "C3"), int(sys.argv[2]) if len(sys.argv) > 2 else 300
args = [a for a in sys.argv[3:] if a.startswith("--")]
sys.argv = sys.argv[:3] + [a for a in sys.argv[3:] if not a.startswith("--")]
if "--dropin" in args:      # the host-array drop-in call instead of the resident pipeline: prune_conformers_rmsd on the config's survivors of the clash check
    import oracle
    import tscode_amd
    ens = make_config(cfg)
    poses = ens.poses()
    poses = poses[tscode_amd.compenetration_mask(poses, ens.ids, 1.5, 0)]
    first, bad, t0 = None, 0, time.time()
    for i in range(steps):
        _, mask = tscode_amd.prune_conformers_rmsd(poses, ens.atomnos, 0.5)
        d = hashlib.sha256(np.packbits(mask).tobytes()).hexdigest()[:16]
        first = first or d
        bad += d != first
    e = json.load(open("tests/golden/expected_full.json")).get(f"{cfg}:{ens.n_poses}:mode0")
    print(f"{cfg} drop-in: {steps} steps, {bad} differ from the first; first equals the recorded oracle mask: {bool(e) and e['keep_sha256_16'] == first}, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)
ens = make_config(cfg)
mode = 1 if "--mode1" in args else 0          # (--mode1: the cache-free prune)
exp = json.load(open("tests/golden/expected_full.json")).get(f"{cfg}:{ens.n_poses}:mode{mode}")
pipe = DevicePipeline(ens, device_index=0, mode=mode)
for opt in sys.argv[3:]:                                  # library tunables name=value
    from tscode_amd import get_engine
    pipe.set_option(opt.split("=")[0], float(opt.split("=")[1]))
kinds = {}
bad, first, t0 = 0, None, time.time()
for i in range(steps):
    res = pipe.step()
    d = hashlib.sha256(np.packbits(pipe.h_keep[:res["n_pass"]].numpy().astype(bool)).tobytes()).hexdigest()[:16]
    key = (res["n_pass"], res["n_keep"], d, tuple(s["pairs_evaluated"] for s in res["stats"]))
    if first is None:
        first = key
        if exp:
            assert (exp["n_pass"], exp["n_keep"], exp["keep_sha256_16"]) == key[:3], key[:3]
    if key != first:
        bad += 1
        what = ("n_pass" if key[0] != first[0] else "") + (" n_keep" if key[1] != first[1] else "") + (" mask" if key[2] != first[2] else "") + \
               (" evals" if key[3] != first[3] else "")
        kinds[what] = kinds.get(what, 0) + 1
        if bad <= 3:
            print("step", i, what, key[:2], first[:2], [a - b for a, b in zip(key[3], first[3])] if len(key[3]) == len(first[3]) else (len(key[3]), len(first[3])), flush=True)
print(kinds)
print(f"{cfg}: {steps} steps, {bad} differ from the first (which equals the recorded oracle result: {bool(exp)}), {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
