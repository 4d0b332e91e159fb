"""Randomised soak of the matrix-core routes of the prune against the oracle.  As a script (not collected by pytest; run it on a GPU box):
`python tests/soak_mm.py [seconds] [seed]`; `tests/test_gpu_parity.py::test_prune_random_medium_ensembles` runs three fixed cases of it.
Medium ensembles -- 1 500 to 40 000 structures, 1 to 64 heavy atoms, tight and loose families, global and local shuffles, three thresholds,
both modes -- through the 64-row walked kernel, the culled kernel, the 16-row kernel and the packed-fp32 one, with stage 1 in float64 and on
the float32 copy, with and without the chunk-local kernel: survivor masks, the passes' k and the reference's pair-evaluation counts equal the
oracle's.  Cases with a pair on a threshold (guard band 1e-7) are skipped, as in the tests."""
import os
import sys
import time

import numpy as np

# (name, sieve_mm, sieve_mm16, cull, cull_min_pairs, stage1_f32, local_pass, fused_apply)
ROUTES = [("mm64-walked", 2, 1, 1, 2.0e9, 1, 0, 1), ("mm64-walked-f32-local", 2, 1, 1, 2.0e9, 2, 1, 1), ("mm64-culled", 2, 1, 2, 0.0, 1, 0, 1),
          ("mm64-culled-f32", 2, 1, 2, 0.0, 2, 1, 1), ("mm16", 1, 1, 1, 2.0e9, 1, 0, 1), ("mm16-f32-apply", 1, 1, 1, 2.0e9, 2, 1, 0),
          ("vector", 0, 0, 1, 2.0e9, 1, 1, 1)]
DEFAULTS = (("prune_algo", 0), ("sieve_mm", 1), ("sieve_mm16", 1), ("cull", 1), ("cull_min_pairs", 2.0e9), ("stage1_f32", 1), ("local_pass", 1), ("fused_apply", 1))


def make_case(rng, sizes=(1500, 2048, 3001, 6000, 8191, 12000, 20001, 40000)):
    """(heavy f64[n, h, 3], thr, description) of one random ensemble."""
    n = int(rng.choice(sizes))
    h = int(rng.choice([1, 2, 3, 5, 8, 9, 15, 16, 17, 30, 31, 32, 33, 48, 64]))
    children = int(rng.choice([1, 2, 5, 10, 40]))
    n_par = max(1, n // children)
    scale, spread = float(rng.choice([1.0, 3.0, 8.0, 30.0])), float(rng.choice([0.005, 0.03, 0.1, 0.3]))
    thr = float(rng.choice([0.25, 0.5, 1.0]))
    base = rng.normal(size=(n_par, h, 3)) * scale + rng.normal(size=(n_par, 1, 3)) * float(rng.choice([0.0, 2.0]))
    parent = rng.integers(0, n_par, size=n)
    if rng.random() < 0.5:       # a local shuffle: siblings near each other, the fine passes find work
        parent = np.sort(parent)[np.argsort(np.arange(n) + rng.normal(size=n) * float(rng.choice([3.0, 30.0, 300.0])), kind="stable")]
    heavy = np.ascontiguousarray(base[parent] + rng.normal(size=(n, h, 3)) * spread)
    return heavy, thr, f"n {n} h {h} children {children} scale {scale} spread {spread} thr {thr}"


def run_routes(eng, heavy, thr, mode, ref):
    """{route: ok} of the prune through every route against the oracle's result `ref`; leaves the library's options at their defaults."""
    out = {}
    try:
        for name, smm, s16, cull, cmp_, f32, local, fused in ROUTES:
            for k, v in (("prune_algo", 2), ("sieve_mm", smm), ("sieve_mm16", s16), ("cull", cull), ("cull_min_pairs", cmp_), ("stage1_f32", f32),
                         ("local_pass", local), ("fused_apply", fused)):
                eng.set_option(k, v)
            mask, stats = eng.prune_heavy(heavy, thr, mode)
            out[name] = bool(np.array_equal(mask, ref["mask"]) and [s["k"] for s in stats] == [s["k"] for s in ref["stats"]]
                             and [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]])
    finally:
        for k, v in DEFAULTS:
            eng.set_option(k, v)
    return out


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import oracle  # (test infrastructure: the checker)
    from tscode_amd import get_engine
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 515)
    eng = get_engine(0)
    t_end, bad, cases, skipped = time.time() + budget, 0, 0, 0
    while time.time() < t_end:
        heavy, thr, what = make_case(rng)
        for mode in (0, 1):
            mr, mm = oracle.prune_margins(heavy, thr, mode)
            if min(mr, mm) < 1e-7:
                skipped += 1
                continue
            t0 = time.time()
            ref = oracle.prune_heavy(heavy, thr, mode=mode, row_parallel=True)
            tc = time.time() - t0
            res = run_routes(eng, heavy, thr, mode, ref)
            bad += sum(not ok for ok in res.values())
            cases += 1
            print(f"{what} mode {mode}: survivors {int(ref['mask'].sum())} oracle {tc:.1f}s | " + " ".join(f"{k} {'ok' if ok else 'MISMATCH'}" for k, ok in res.items()), flush=True)
    print(f"cases {cases} (x {len(ROUTES)} routes), skipped for the guard band {skipped}, mismatches {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
