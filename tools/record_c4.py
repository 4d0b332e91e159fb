"""One-off: the CPU oracle's result for C4 (1M x 50), prune only (no margin run: that doubles an 80-minute job).
Writes gpurun_out/c4_expected.json; merged into tests/golden/expected_full.json by hand when it finishes."""
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tscode_amd.synthetic import make_config
oracle.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
ens = make_config("C4")
poses = oracle.transform_batch(ens.frag_coords, ens.conf_idx, ens.rot, ens.pos)
cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
heavy = np.ascontiguousarray(poses[cm][:, ens.atomnos != 1])
clash_margin = oracle.clash_margin(poses, ens.ids, 1.5)
del poses
print("clash done", int(cm.sum()), clash_margin, flush=True)
t0 = time.time()
res = oracle.prune_heavy(heavy, 0.5, mode=0, row_parallel=True)
out = {"C4:%d:mode0" % ens.n_poses: {
    "n_pass": int(cm.sum()), "n_keep": int(res["mask"].sum()),
    "clash_sha256_16": hashlib.sha256(np.packbits(cm).tobytes()).hexdigest()[:16],
    "keep_sha256_16": hashlib.sha256(np.packbits(res["mask"]).tobytes()).hexdigest()[:16],
    "passes": [{"k": s["k"], "active_after": s["n_active_after"], "pairs_evaluated": s["pairs_evaluated"]} for s in res["stats"]],
    "margins": {"rmsd": None, "maxdev": None, "clash": clash_margin},
    "oracle_seconds": time.time() - t0}}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/c4_expected.json", "w"), indent=1, sort_keys=True)
print("done", out, flush=True)
