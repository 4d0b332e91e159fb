"""One-off: the CPU oracle's result for BASELINE config 5 AS A CHAIN at full size -- csearch rotations of fragment 0 (20 000 angle
sets x 8 torsions, the walk-back loop of torsion_module.py:487-498 included) -> the kept candidates are that fragment's conformers ->
500 000 trimolecular poses -> clash mask -> prune (mode 0) -- with exactly the inputs bench.py --config C5chain builds.
Writes gpurun_out/c5chain_expected.json; merged into tests/golden/expected_full.json when it finishes (about ten minutes on 6 cores)."""
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tscode_amd.synthetic import make_config
from tscode_amd.pipeline import CsearchChain

CHAIN_CANDIDATES, CHAIN_TORSIONS, TORSION_SEED, ANGLE_SEED, DRAW_SEED, THRESH = 20000, 8, 5, 6, 7, 1.4     # bench.py, run_leg
oracle.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
ens = make_config("C5")
n0 = ens.frag_coords[0].shape[1]
torsions, masks = CsearchChain.chain_torsions(n0, CHAIN_TORSIONS, seed=TORSION_SEED)
angles = np.random.default_rng(ANGLE_SEED).choice(np.array([0, 0, 60, 120, 180, 240, 300, 25]), size=(CHAIN_CANDIDATES, CHAIN_TORSIONS)).astype(np.int32)
t0 = time.time()
out, rb, margin = oracle.csearch_rotate(ens.frag_coords[0][0], torsions, masks, angles, THRESH, 0, return_margin=True)
confs = out[rb != 0]                                                   # torsion_module.py:505 (n_out = every candidate)
print("csearch", len(confs), "kept of", len(out), "margin", margin, f"{time.time() - t0:.0f}s", flush=True)
draw = np.random.default_rng(DRAW_SEED).integers(0, 2 ** 30, size=ens.n_poses)
ci = ens.conf_idx.copy()
ci[:, 0] = draw % len(confs)
poses = oracle.transform_batch([confs] + [f for f in ens.frag_coords[1:]], ci, ens.rot, ens.pos)
cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
clash_margin = oracle.clash_margin(poses, ens.ids, 1.5)
heavy = np.ascontiguousarray(poses[cm][:, ens.atomnos != 1])
del poses
print("clash done", int(cm.sum()), clash_margin, flush=True)
t0 = time.time()
res = oracle.prune_heavy(heavy, 0.5, mode=0, row_parallel=True)
rec = {"C5chain:%d:mode0" % ens.n_poses: {
    "n_conformers": int(len(confs)), "csearch_margin": float(margin),
    "n_pass": int(cm.sum()), "n_keep": int(res["mask"].sum()),
    "clash_sha256_16": hashlib.sha256(np.packbits(cm).tobytes()).hexdigest()[:16],
    "keep_sha256_16": hashlib.sha256(np.packbits(res["mask"]).tobytes()).hexdigest()[:16],
    "passes": [{"k": s["k"], "active_after": s["n_active_after"], "pairs_evaluated": s["pairs_evaluated"]} for s in res["stats"]],
    "margins": {"rmsd": None, "maxdev": None, "clash": clash_margin},
    "oracle_seconds": time.time() - t0}}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rec, open("gpurun_out/c5chain_expected.json", "w"), indent=1, sort_keys=True)
print("done", rec, flush=True)
