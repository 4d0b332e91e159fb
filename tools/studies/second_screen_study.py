"""Offline (CPU): of the pairs that pass the 2 x 8-component screen, how many would a test on the FULL feature vectors (30 atom norms,
15 pair distances: 180 B per structure instead of 720 B of coordinates) drop?  And how many a screen with 16 / 24 components?"""
import sys, os, json
import numpy as np
sys.path.insert(0, ".")
from tscode_amd.synthetic import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
rng = np.random.default_rng(3)
ens = make_config(cfg)
idx = np.sort(rng.choice(ens.n_poses, NS * 2, replace=False))
heavy = np.concatenate([ens.poses(lo, min(lo + 100_000, ens.n_poses))[:, ens.atomnos != 1] for lo in range(0, ens.n_poses, 100_000)])[idx][:NS]
h = heavy.shape[1]
thr = 0.5
limit = h * thr * thr
f0 = np.linalg.norm(heavy, axis=2)
f1 = np.linalg.norm(heavy[:, :h // 2] - heavy[:, h // 2:2 * (h // 2)], axis=2) / np.sqrt(2.0)
def pca(f, k):
    mu = f.mean(0); w, v = np.linalg.eigh(np.cov((f - mu).T)); q = v[:, np.argsort(w)[::-1][:k]]
    return ((f - mu) @ q).astype(np.float32), np.sort(w)[::-1]
(d0, w0), (d1, w1) = pca(f0, 8), pca(f1, 8)
print("spectrum fam0", np.round(w0[:12], 3), "sum", round(w0.sum(), 3)); print("spectrum fam1", np.round(w1[:12], 3), "sum", round(w1.sum(), 3))
F0, F1 = f0.astype(np.float32), f1.astype(np.float32)
d0_16, _ = pca(f0, 16); d1_15, _ = pca(f1, 15)
n_pass = n_full = n_16 = 0
tot = 0
B = 1000
def sq(a, b):
    return (a * a).sum(1)[:, None] + (b * b).sum(1)[None] - 2.0 * a @ b.T
pairs = []
for i0 in range(0, NS, B):
    a = slice(i0, i0 + B)
    s0, s1 = sq(d0[a], d0), sq(d1[a], d1)
    ok = (s0 <= limit) & (s1 <= limit)
    ok &= np.arange(NS)[None, :] > np.arange(i0, min(i0 + B, NS))[:, None]
    r, c = np.nonzero(ok)
    pairs.append(np.stack([r + i0, c], 1))
    tot += (NS - i0 - B / 2) * B
pairs = np.concatenate(pairs)
print("pairs", tot, "pass the 2x8 screen", len(pairs), "fraction", len(pairs) / tot)
p, q = pairs[:, 0], pairs[:, 1]
e0 = ((F0[p] - F0[q]) ** 2).sum(1); e1 = ((F1[p] - F1[q]) ** 2).sum(1)
full = (e0 <= limit) & (e1 <= limit)
g0 = ((d0_16[p] - d0_16[q]) ** 2).sum(1); g1 = ((d1_15[p] - d1_15[q]) ** 2).sum(1)
s16 = (g0 <= limit) & (g1 <= limit)
print("survive full features", full.mean(), " survive 16+15 components", s16.mean())
# true rmsd of a sample of the passing pairs (about the origin, Kabsch)
def rmsd(P, Q):
    H = P.T @ Q; U, S, Vt = np.linalg.svd(H); d = np.sign(np.linalg.det(U @ Vt)); S[-1] *= d
    return np.sqrt(max(0.0, ((P * P).sum() + (Q * Q).sum() - 2 * S.sum()) / len(P)))
sel = rng.choice(len(pairs), min(4000, len(pairs)), replace=False)
rm = np.array([rmsd(heavy[p[i]], heavy[q[i]]) for i in sel])
print("true rmsd of passing pairs: quantiles", np.round(np.quantile(rm, [0.01, 0.1, 0.5, 0.9]), 3), "similar", (rm < thr).mean())
print("  of those surviving the full test:", (rm[full[sel]] < thr).mean() if full[sel].any() else None, "n", int(full[sel].sum()))
# a subset Kabsch (first m atoms) as a lower bound on h rmsd^2
for m in (8, 12, 16):
    lb = np.array([rmsd(heavy[p[i]][:m], heavy[q[i]][:m]) ** 2 * m for i in sel[:1500]])
    print(f"subset Kabsch on {m} atoms drops", (lb > limit).mean())
# more pair-distance families over other DISJOINT pairings (each is a bound of its own)
half = h // 2
def fam_pairs(perm):
    return np.linalg.norm(heavy[:, :half] - heavy[:, half + perm], axis=2) / np.sqrt(2.0)
surv = np.ones(len(pairs), bool)
for name, perm in (("reversed", np.arange(half)[::-1]), ("shift 5", (np.arange(half) + 5) % half), ("shift 9", (np.arange(half) + 9) % half),
                   ("within halves", None)):
    if perm is None:
        a = np.arange(0, h - 1, 2); f = np.linalg.norm(heavy[:, a] - heavy[:, a + 1], axis=2) / np.sqrt(2.0)
    else:
        f = fam_pairs(perm)
    for k in (8,):
        d, _ = pca(f, k)
        e = ((d[p] - d[q]) ** 2).sum(1)
        ok = e <= limit
        surv &= ok
        print(f"family '{name}' ({k} comps): survive {ok.mean():.3f}; all extra families so far: {surv.mean():.3f}; truly similar among survivors {(rm[surv[sel]] < thr).mean() if surv[sel].any() else None}")
