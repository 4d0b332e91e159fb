"""Offline (CPU): which TWO feature families make the strongest 2 x 8-component screen?  Candidates: atom norms, and pair distances over
several disjoint pairings of the atoms.  Counts the pairs of a random subset that pass both families' tests."""
import sys, itertools
import numpy as np
sys.path.insert(0, ".")
from tscode_amd.synthetic import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
rng = np.random.default_rng(3)
ens = make_config(cfg)
idx = np.sort(rng.choice(ens.n_poses, NS * 2, replace=False))
heavy = np.concatenate([ens.poses(lo, min(lo + 100_000, ens.n_poses))[:, ens.atomnos != 1] for lo in range(0, ens.n_poses, 100_000)])[idx][:NS]
h = heavy.shape[1]; half = h // 2
limit = h * 0.25
def pca(f, k=8):
    mu = f.mean(0); w, v = np.linalg.eigh(np.cov((f - mu).T)); q = v[:, np.argsort(w)[::-1][:k]]
    return ((f - mu) @ q).astype(np.float32)
def pairs(perm):
    return np.linalg.norm(heavy[:, :half] - heavy[:, half + perm], axis=2) / np.sqrt(2.0)
fams = {"norms": np.linalg.norm(heavy, axis=2), "half": pairs(np.arange(half)), "rev": pairs(np.arange(half)[::-1]),
        "s5": pairs((np.arange(half) + 5) % half), "s9": pairs((np.arange(half) + 9) % half), "s3": pairs((np.arange(half) + 3) % half), "s7": pairs((np.arange(half) + 7) % half), "shalf": pairs((np.arange(half) + half // 2) % half), "sthird": pairs((np.arange(half) + half // 3) % half), "s8": pairs((np.arange(half) + 8) % half),
        "adj": np.linalg.norm(heavy[:, 0:h - 1:2] - heavy[:, 1:h:2], axis=2) / np.sqrt(2.0)}
D = {k: pca(v) for k, v in fams.items()}
def sq(a, b):
    return (a * a).sum(1)[:, None] + (b * b).sum(1)[None] - 2.0 * a @ b.T
B = 1000
passm = {k: [] for k in D}
tot = 0
# per family: boolean pass matrix blocks are too big to keep; count combos on the fly
names = list(D)
combos = [c for c in itertools.combinations(names, 2) if "adj" not in c] + [(a,) for a in names]
cnt = {c: 0 for c in combos}
for i0 in range(0, NS, B):
    a = slice(i0, i0 + B)
    ok = {k: sq(D[k][a], D[k]) <= limit for k in names}
    upper = np.arange(NS)[None, :] > np.arange(i0, min(i0 + B, NS))[:, None]
    tot += int(upper.sum())
    for c in combos:
        m = upper.copy()
        for k in c:
            m &= ok[k]
        cnt[c] += int(m.sum())
print(cfg, "pairs", tot)
for c, n in sorted(cnt.items(), key=lambda kv: kv[1]):
    print(f"{'+'.join(c):20s} pass {n / tot:.6f}")
