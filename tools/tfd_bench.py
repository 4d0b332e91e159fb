"""prune_conformers_tfd on the GPU (host arrays in, mask out) next to the oracle's pair search on the host cores."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import oracle, tscode_amd
rng = np.random.default_rng(3)
quads = np.array([[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 5], [4, 5, 6, 7], [6, 7, 8, 9], [0, 4, 8, 11]])
for n_par in (2000, 10000):
    parents = rng.normal(size=(n_par, 12, 3)) * 2
    s = (parents[:, None] + rng.normal(size=(n_par, 5, 12, 3)) * 0.02).reshape(-1, 12, 3)
    s = np.ascontiguousarray(s[rng.permutation(len(s))])
    tscode_amd.prune_conformers_tfd(s[:500], quads)
    t0 = time.perf_counter(); _, mask = tscode_amd.prune_conformers_tfd(s, quads); dt = time.perf_counter() - t0
    tf = oracle.torsion_fingerprints(s, quads)
    t0 = time.perf_counter(); oracle.tfd_first_similar(tf, len(s), 1, len(s), 10); dc = time.perf_counter() - t0
    print(f"N = {len(s)}: GPU prune_conformers_tfd {dt * 1e3:.1f} ms ({mask.sum()} survive); oracle k=1 pair search alone {dc * 1e3:.0f} ms", flush=True)
