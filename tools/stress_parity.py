"""Randomised parity stress: ensembles whose children are spread around the threshold (many pairs undecided by the quartic
tests, long walks of the exact path), GPU prune vs the oracle: masks, per-pass counts, pair-evaluation counts, guard band."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import oracle
from tscode_amd import get_engine
from tscode_amd.synthetic import make_ensemble
eng = get_engine(0)
bad = 0
for seed, (n, apf, children, srot, st) in enumerate([(20000, (25, 25), 10, 6.0, 0.12), (20000, (15, 15), 6, 10.0, 0.2), (12000, (40, 30), 20, 4.0, 0.15),
                                                     (30000, (25, 25), 3, 8.0, 0.25), (8000, (70, 70, 60), 8, 2.0, 0.1)]):
    ens = make_ensemble(n, apf, 7000 + seed, children=children, sigma_rot_deg=srot, sigma_t=st, shell=(4.0, 9.0) if len(apf) == 2 else (8.0, 15.0))
    poses = ens.poses()
    poses = poses[oracle.compenetration_mask(poses, ens.ids, 1.5, 0)]
    heavy = np.ascontiguousarray(poses[:, ens.atomnos != 1])
    for mode in (0, 1):
        mr, mm = oracle.prune_margins(heavy, 0.5, mode)
        t0 = time.time(); ref = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True); tc = time.time() - t0
        # (algo, chunk-local kernel, float32 stage 1 forced, every pass culled)
        routes = ((0, 1, 0, 0), (2, 0, 0, 0), (1, 1, 0, 0), (2, 0, 1, 0), (2, 0, 1, 1)) if heavy.shape[1] <= 32 else ((0, 1, 0, 0), (2, 0, 0, 0), (2, 0, 1, 0), (2, 0, 1, 1))
        for algo, local, f32, culled in routes:
            eng.set_option("prune_algo", algo); eng.set_option("local_pass", local)
            eng.set_option("stage1_f32", 2 if f32 else 1)
            eng.set_option("cull", 2 if culled else 1); eng.set_option("cull_min_pairs", 0 if culled else 2.0e9)
            mask, stats = eng.prune_heavy(heavy, 0.5, mode)
            ok = np.array_equal(mask, ref["mask"]) and [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
            bad += not ok
            print(f"seed {seed} N={len(heavy)} h={heavy.shape[1]} mode {mode} algo {algo} local {local} f32 {f32} culled {culled}: {'OK ' if ok else 'MISMATCH'} survivors {mask.sum()} "
                  f"(ref {ref['mask'].sum()}), exact-path pairs {sum(s['candidates'] for s in stats)}, margins rmsd {mr:.1e} maxdev {mm:.1e}, oracle {tc:.1f}s", flush=True)
eng.set_option("prune_algo", 0); eng.set_option("local_pass", 1); eng.set_option("stage1_f32", 1); eng.set_option("cull", 1); eng.set_option("cull_min_pairs", 2.0e9)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
