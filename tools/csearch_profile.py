"""What bounds k_csearch_rotate (tscode/torsion_module.py:463-500)?   Runs on the GPU box.

The workload of tools/next_rows_bench.py (200 atoms, 8 torsions, 100 000 candidates, thresh 1.4), device-resident, timed with HIP
events, three ways:
  * as it is;
  * with thresh = 0.05 A: no rotation ever clashes, so every non-zero angle costs exactly one rotation + one torsion_comp_check and the
    walk-back loop of :487-498 never runs -- the difference is what the walk-back costs;
  * the number of (rotation + check) rounds a candidate really takes, counted on a sample by replaying the reference's loop on the
    host with the oracle's torsion_comp_check (the checker; outside the timed part).
tools/csearch_profile.sh adds the SQ counters of the same kernel (rocprofv3 --pmc, separate passes) and merges them in.
usage: python tools/csearch_profile.py [--quick]     (prints one JSON object)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tscode_amd                                   # noqa: E402
from tscode_amd.synthetic import make_config        # noqa: E402

quick = "--quick" in sys.argv
rng = np.random.default_rng(7)
eng = tscode_amd.get_engine(0)
dev = torch.device("cuda:0")
ens = make_config("C5", 4)
coords = ens.poses()[0]
na, n0 = len(coords), ens.frag_coords[0].shape[1]
centres = rng.choice(np.arange(2, n0 - 3), size=8, replace=False)
torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres], dtype=np.int32)
masks = np.zeros((8, na), dtype=np.uint8)
for t, c in enumerate(centres):
    masks[t, c + 1:n0] = 1
M = 20_000 if quick else 100_000
angles = rng.choice(np.array([0, 0, 60, 120, 180, -60, 25]), size=(M, 8)).astype(np.int32)

d_coords, d_tors, d_masks, d_angles = (torch.from_numpy(x).to(dev) for x in (coords, torsions, masks, angles))
d_out = torch.empty((M, na, 3), dtype=torch.float64, device=dev)
d_rb = torch.empty(M, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream(dev)
eng.set_stream(stream.cuda_stream)


def run(thresh):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(2 if quick else 4):
        with torch.cuda.stream(stream):
            a.record(stream)
            eng.csearch_rotate_dev(d_coords, na, d_tors, d_masks, 8, d_angles, M, thresh, 0, d_out, d_rb)
            b.record(stream)
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


ms_real = run(1.4)
rb_real = d_rb.cpu().numpy().copy()
ms_free = run(0.05)
out = {"workload": f"{na} atoms ({n0} in the fragment that turns), 8 torsions, {M} candidates, thresh 1.4, max_clashes 0; angles drawn from (0, 0, 60, 120, 180, -60, 25)",
       "moved_atoms_per_torsion": [int(m.sum()) for m in masks], "kernel_ms": ms_real, "kernel_ms_without_walk_back (thresh 0.05)": ms_free,
       "candidates_per_s": M / ms_real * 1e3, "algorithmic_bytes": M * (na * 24 + 32 + 4), "hbm_GBs": M * (na * 24 + 36) / ms_real / 1e6,
       "rotated_bonds_mean": float(rb_real.mean())}

# rounds per candidate, replayed on the host for a sample (the reference's loop, the oracle's check)
import oracle                                       # noqa: E402  (checker only)
ns = 100 if quick else 400
rounds, nonzero, clashing = [], 0, 0
for m in range(ns):
    c = coords.copy()
    n_round = 0
    for t in range(8):
        ang = int(angles[m, t])
        if ang == 0:
            continue
        nonzero += 1
        c = oracle.rotate_dihedral(c, torsions[t], ang, masks[t])
        n_round += 1
        if not oracle.torsion_comp_check(c, torsions[t], masks[t], 1.4, 0):
            clashing += 1
            for _ in range(ang // 5):
                c = oracle.rotate_dihedral(c, torsions[t], -5, masks[t])
                n_round += 1
                if oracle.torsion_comp_check(c, torsions[t], masks[t], 1.4, 0):
                    break
    rounds.append(n_round)
rounds = np.array(rounds)
pairs_per_round = float(np.mean([int(m.sum()) * (na - int(m.sum()) - 2) for m in masks]))
out.update({"sample": ns, "rounds_per_candidate_mean": float(rounds.mean()), "rounds_per_candidate_max": int(rounds.max()),
            "rounds_per_candidate_without_walk_back": nonzero / ns, "first_rotations_that_clash_frac": clashing / max(nonzero, 1),
            "atom_pairs_per_round_mean": pairs_per_round,
            "ns_per_round_per_wavefront_equivalent": ms_real * 1e6 / (M * rounds.mean()),
            "pair_distances_per_s": M * rounds.mean() * pairs_per_round / ms_real * 1e3})
print(json.dumps(out))
