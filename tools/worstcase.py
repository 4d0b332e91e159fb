#!/usr/bin/env python
"""Worst case for the descriptor sieve (nothing can be screened out): time prune_algo 1 (register-tiled) against 2 (sieve)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tscode_amd
from tscode_amd.synthetic import quat_to_mat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(99)
A, B = rng.normal(size=(15, 3)) * 3, rng.normal(size=(15, 3)) * 3
n_par = n // 10
rots = quat_to_mat(rng.normal(size=(n_par, 4)))
which = rng.integers(0, n_par, size=n)
jitter = quat_to_mat(np.concatenate([np.ones((n, 1)), rng.normal(size=(n, 3)) * 0.004], axis=1))
Bs = np.einsum("nij,njk,ak->nai", rots[which], jitter, B)
heavy = np.empty((n, 30, 3))
ia = list(range(0, 8)) + list(range(15, 22))        # pairs (a, a+15) stay inside one body
ib = list(range(8, 15)) + list(range(22, 30))
heavy[:, ia] = A
heavy[:, ib] = Bs
heavy = np.ascontiguousarray(heavy)
eng = tscode_amd.get_engine(0)
res = {}
for algo in (1, 2):
    eng.set_option("prune_algo", algo)
    eng.prune_heavy(heavy, 0.5, 0)
    t0 = time.perf_counter()
    mask, stats = eng.prune_heavy(heavy, 0.5, 0)
    dt = time.perf_counter() - t0
    res[algo] = mask
    print(f"algo {algo}: {dt*1e3:.2f} ms (host arrays in/out), pair-kernel ms {sum(s['tile_ms'] for s in stats):.2f}, survivors {mask.sum()}, "
          f"screened {sum(s['pairs_screened'] for s in stats):.3g}, H formed {sum(s['pairs_computed'] for s in stats):.3g}")
assert np.array_equal(res[1], res[2])
