#!/usr/bin/env python
"""Offline (CPU, NumPy): would SPATIAL CULLING in descriptor space pay before the pair kernel's screen?  

For the late passes of C3 and C4 the active structures of one chunk are sorted along a Morton curve through the leading components
of their descriptors; rows are cut into tiles of 16 and columns into tiles of 128 consecutive structures OF THAT ORDER, and a (row
tile, column tile) pair has to be screened only if the bounding boxes of the two tiles lie within the screen's limit h thr^2 in BOTH
feature families (a lower bound of every pair distance between them).  Counted against what the present left-to-right walk visits:
every column tile behind a row until the row's first similar column (rows that are removed stop early; the sorted order cannot keep
that exit, because the reference removes a row for its FIRST similar column in index order, rmsd_pruning.py:75-77, so every
candidate has to be found and the smallest index taken).

The active set of a pass is a proxy: a uniform random subset of the structures that pass the clash check, of the size the recorded
oracle run had entering that pass (tests/golden/expected_full.json); the removed rows' exit points are drawn from the same run's
pair-evaluation counts.  Descriptors are the kernel's (sieve.hpp): 8 principal components of the atom norms and of the half-chain pair
distances / sqrt 2, estimated from a 4096-structure sample.

usage: python tools/culling_study.py [C3 C4] > profiles/r03_culling_study.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tscode_amd.synthetic import make_config

KD, TI, TJ, THR = 8, 16, 128, 0.5


def descriptors(heavy):
    h = heavy.shape[1]
    f0 = np.linalg.norm(heavy, axis=2)
    f1 = np.linalg.norm(heavy[:, :h // 2] - heavy[:, h // 2:2 * (h // 2)], axis=2) / np.sqrt(2.0)
    out = []
    for f in (f0, f1):
        s = f[:: max(1, len(f) // 4096)][:4096]
        mu = s.mean(0)
        w, v = np.linalg.eigh(np.cov((s - mu).T))
        q = v[:, np.argsort(w)[::-1][:KD]]                       # orthonormal columns: |Q^T x| <= |x|
        out.append((f - mu) @ q)
    return out                                                   # two arrays [n, 8]


def morton_order(d0, bits=10, dims=3):
    x = d0[:, :dims]
    q = ((x - x.min(0)) / (np.ptp(x, axis=0) + 1e-12) * ((1 << bits) - 1)).astype(np.uint64)
    code = np.zeros(len(x), dtype=np.uint64)
    for b in range(bits):
        for k in range(dims):
            code |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * dims + k)
    return np.argsort(code, kind="stable")


def box_gap2(lo_r, hi_r, lo_c, hi_c):
    """squared distance between axis-aligned boxes, rows x columns"""
    gap = np.maximum(0.0, np.maximum(lo_c[None] - hi_r[:, None], lo_r[:, None] - hi_c[None]))
    return (gap * gap).sum(-1)


def study_chunk(D, limit, rng, removed_frac, mean_exit_frac):
    n = len(D[0])
    order = morton_order(D[0])
    Ds = [d[order] for d in D]
    n_rt, n_ct = -(-n // TI), -(-n // TJ)
    boxes_r = [(np.array([d[i * TI:(i + 1) * TI].min(0) for i in range(n_rt)]), np.array([d[i * TI:(i + 1) * TI].max(0) for i in range(n_rt)])) for d in Ds]
    boxes_c = [(np.array([d[j * TJ:(j + 1) * TJ].min(0) for j in range(n_ct)]), np.array([d[j * TJ:(j + 1) * TJ].max(0) for j in range(n_ct)])) for d in Ds]
    needed = 0
    for r0 in range(0, n_rt, 512):
        ok = np.ones((min(512, n_rt - r0), n_ct), dtype=bool)
        for f in range(2):
            ok &= box_gap2(boxes_r[f][0][r0:r0 + 512], boxes_r[f][1][r0:r0 + 512], boxes_c[f][0], boxes_c[f][1]) <= limit
        needed += int(ok.sum())
    # In sorted order a pair (i, j) may sit on either side of the diagonal: all row x column tile pairs count, halved for symmetry
    needed_sym = needed / 2.0
    # the present walk, in index order: row r visits the column tiles behind it up to its exit point (a removed row stops after a
    # fraction mean_exit_frac of its range, the others walk to the end of the chunk)
    rows = np.arange(n)
    span = (n - 1 - rows).astype(np.float64)
    removed = rng.random(n) < removed_frac
    visited_cols = np.where(removed, span * mean_exit_frac, span)
    walk_tiles = float((np.ceil(visited_cols / TJ)).reshape(-1)[: n_rt * TI].sum() / TI) if n >= TI else float(np.ceil(visited_cols / TJ).sum() / TI)
    # exact share of pairs the screen itself lets through (sample)
    m = min(n, 3000)
    idx = rng.choice(n, m, replace=False)
    passes = np.ones((m, m), dtype=bool)
    for d in D:
        x = d[idx]
        g = (x * x).sum(1)
        passes &= (g[:, None] + g[None] - 2.0 * x @ x.T) <= limit
    pass_rate = float((passes.sum() - m) / (m * (m - 1)))
    return {"structures": n, "row_tiles": n_rt, "col_tiles": n_ct, "tile_pairs_all": n_rt * n_ct / 2.0, "tile_pairs_within_limit": needed_sym,
            "tile_pairs_present_walk": walk_tiles, "culled_over_walk": needed_sym / max(walk_tiles, 1.0), "screen_pass_rate_of_pairs": pass_rate}


def main():
    cfgs = [a for a in sys.argv[1:] if a.startswith("C")] or ["C3", "C4"]
    exp = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "expected_full.json")))
    out = {"what": __doc__.split("usage")[0].strip(), "tile": [TI, TJ], "configs": {}}
    for cfg in cfgs:
        ens = make_config(cfg)
        rec = exp[f"{cfg}:{ens.n_poses}:mode0"]
        # the clash survivors: the recorded count, drawn uniformly (the clash verdicts themselves need the oracle or the GPU)
        rng = np.random.default_rng(1)
        keep = np.sort(rng.choice(ens.n_poses, rec["n_pass"], replace=False))
        heavy = np.concatenate([ens.poses(lo, min(lo + 100_000, ens.n_poses))[:, ens.atomnos != 1] for lo in range(0, ens.n_poses, 100_000)])[keep]
        D = descriptors(heavy)
        n, h = heavy.shape[0], heavy.shape[1]
        limit = h * THR * THR
        rows = []
        before = n
        for p in rec["passes"]:
            k, after = p["k"], p["active_after"]
            if k in (20, 5, 2, 1):
                active = np.sort(rng.choice(n, before, replace=False))
                cs = n // k
                chunk = active[(active >= 0) & (active < cs)] if k > 1 else active
                removed_frac = (before - after) / max(before, 1)
                # a removed row's exit: pairs evaluated by removed rows / their full ranges, from the recorded evaluation count
                full = before / k * before / 2.0                                   # all pairs of the pass, were nothing to stop
                kept_rows_pairs = (1.0 - removed_frac) * full
                exit_frac = float(np.clip((p["pairs_evaluated"] - kept_rows_pairs) / max(removed_frac * full, 1.0), 0.0, 1.0))
                r = study_chunk([d[chunk] for d in D], limit, rng, removed_frac, exit_frac)
                r.update({"k": k, "active_before": before, "removed_fraction": removed_frac, "removed_rows_exit_at_fraction_of_range": exit_frac})
                rows.append(r)
                print(cfg, r, file=sys.stderr, flush=True)
            before = after
        out["configs"][cfg] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
