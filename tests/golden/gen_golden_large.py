"""G16 / G17: the reference's own prune_conformers_rmsd on ensembles large enough for the FINE passes of its schedule.

BUILD CONTAINER ONLY (.gpurunignore lists this file).  `prune_conformers_rmsd` runs a pass with k chunks only while
20 k < n_active (rmsd_pruning.py:186-192), so the fixtures of G3 (N <= 2 500) let the reference reach k <= 100.  The
headline workload (C3) runs k = 2000, 1000, 500, 200 as well, and the cache-key collisions between a fine and a coarse
pass whose chunk starts coincide (rmsd_pruning.py:65-67) only exist there.  Here:

* G16a  N = 40 023: N // k = 20, 40, 80, 200, 400, 800 for k = 2000 ... 50 -- every chunk start of a coarse pass is a
        chunk start of every finer one: the regime where a cached (first, delta) stops nearly every row of the next passes;
* G16b  N = 41 999: N // k = 20, 41, 83, 209, 419, 839 -- starts rarely coincide, every pass does real work;
* G17   N = 104 999: k = 5000 runs, the largest k the reference can run at all (N > 200 000 fails).

N is divisible by no k of the schedule (the last chunk takes a remainder in every pass).  Children of a parent are placed
by the LOCAL shuffle of tscode_amd.synthetic.make_ensemble (local_spread): every pass finds duplicates.

The reference's top-level function itself drives the passes; its `_similarity_mask_rmsd_group` is wrapped (not replaced)
to record each pass's k, output mask and computed pairs.  Un-JITted it costs ~45 us per pair evaluation: minutes (G16) to
about 40 minutes (G17).  The inputs are NOT stored: a fixture holds the generator's parameters and the sha256 of the heavy-atom
array they produce; the tests regenerate the array and compare the digest before anything else.

Usage:  python -B tests/golden/gen_golden_large.py [G16a G16b G17]
"""

from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _reference as R                      # noqa: E402

R.install_standins(full=False)
import tscode.rmsd_pruning as ref_rp        # noqa: E402

from tscode_amd.synthetic import make_ensemble   # noqa: E402

CASES = {
    # name: (N, atoms per fragment, children, local_spread, thr, seed)
    "G16a": (40023, (8, 7), 16, 5000.0, 0.5, 9316),
    "G16b": (41999, (8, 7), 16, 5000.0, 0.5, 9317),
    "G17": (104999, (8, 7), 40, 20000.0, 0.5, 9318),
}


def run_case(name):
    n, apf, ch, spread, thr, seed = CASES[name]
    ens = make_ensemble(n, apf, seed=seed, children=ch, local_spread=spread)
    structures = ens.poses()
    heavy = np.ascontiguousarray(structures[:, ens.atomnos != 1])
    digest = hashlib.sha256(heavy.tobytes()).hexdigest()
    print(f"{name}: N={n} h={heavy.shape[1]} sha256(heavy)={digest[:16]}", flush=True)

    trace = {"ks": [], "masks": [], "nkeys": [], "pairs": [], "t": []}
    group = ref_rp._similarity_mask_rmsd_group

    def traced_group(structs, in_mask, cache, k, rmsd_thr):
        t0 = time.time()
        out_mask, pairs = group(structs, in_mask, cache=cache, k=k, rmsd_thr=rmsd_thr)
        trace["ks"].append(int(k))
        trace["masks"].append(np.packbits(out_mask))
        trace["pairs"].extend(pairs)
        trace["nkeys"].append(len(cache) - 1 + len(pairs))        # the cache as prune_conformers_rmsd extends it (:204)
        trace["t"].append(time.time() - t0)
        print(f"  k={int(k):5d}  active {int(np.count_nonzero(in_mask))} -> {int(np.count_nonzero(out_mask))}  "
              f"pairs cached {len(pairs)}  {time.time() - t0:.1f} s", flush=True)
        return out_mask, pairs

    ref_rp._similarity_mask_rmsd_group = traced_group
    try:
        t0 = time.time()
        pruned, mask = ref_rp.prune_conformers_rmsd(structures, ens.atomnos, rmsd_thr=thr)
    finally:
        ref_rp._similarity_mask_rmsd_group = group
    assert np.array_equal(pruned, structures[mask])
    assert np.array_equal(np.packbits(mask), trace["masks"][-1])
    keys = np.array(sorted(set(trace["pairs"])), dtype=np.int32).reshape(-1, 2)
    path = os.path.join(HERE, f"{name}_prune_large.npz")
    np.savez_compressed(
        path, n=n, atoms_per_frag=np.array(apf), children=ch, local_spread=spread, thr=thr, seed=seed,
        n_heavy=heavy.shape[1], heavy_sha256=np.frombuffer(bytes.fromhex(digest), dtype=np.uint8),
        mask_bits=np.packbits(mask), ks=np.array(trace["ks"]), pass_mask_bits=np.array(trace["masks"]),
        pass_nkeys=np.array(trace["nkeys"]), keys=keys, pass_seconds=np.array(trace["t"]))
    print(f"  survivors {int(mask.sum())}  passes {trace['ks']}  keys {len(keys)}  {time.time() - t0:.0f} s  "
          f"-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


if __name__ == "__main__":
    for name in (sys.argv[1:] or list(CASES)):
        run_case(name)
