"""Offline: screened pairs of a culled pass for other tile shapes / curves.  C4-like chunk (default 100k structures of the survivors)."""
import sys, numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from culling_study import descriptors, box_gap2
from tscode_amd.synthetic import make_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.default_rng(1)
ens = make_config("C4")
keep = np.sort(rng.choice(ens.n_poses, n, replace=False))
heavy = np.concatenate([ens.poses(lo, min(lo + 100_000, ens.n_poses))[:, ens.atomnos != 1][keep[(keep >= lo) & (keep < lo + 100_000)] - lo] for lo in range(0, ens.n_poses, 100_000)])
D = descriptors(heavy)
limit = heavy.shape[1] * 0.25
def morton(cols, bits):
    x = np.stack(cols, 1)
    q = ((x - x.min(0)) / (np.ptp(x, axis=0) + 1e-12) * ((1 << bits) - 1)).astype(np.uint64)
    code = np.zeros(len(x), dtype=np.uint64)
    for b in range(bits):
        for k in range(x.shape[1]):
            code |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * x.shape[1] + k)
    return np.argsort(code, kind="stable")
def count(order, TI, TJ):
    Ds = [d[order] for d in D]
    n_rt, n_ct = -(-n // TI), -(-n // TJ)
    br = [(np.minimum.reduceat(d, np.arange(0, n, TI)), np.maximum.reduceat(d, np.arange(0, n, TI))) for d in Ds]
    bc = [(np.minimum.reduceat(d, np.arange(0, n, TJ)), np.maximum.reduceat(d, np.arange(0, n, TJ))) for d in Ds]
    need = 0
    for r0 in range(0, n_rt, 256):
        ok = np.ones((min(256, n_rt - r0), n_ct), dtype=bool)
        for f in range(2):
            ok &= box_gap2(br[f][0][r0:r0 + 256], br[f][1][r0:r0 + 256], bc[f][0], bc[f][1]) <= limit
        need += int(ok.sum())
    return need / 2.0 * TI * TJ
base = None
curves = {"f0 comps 0-2, 5 bits (device)": morton([D[0][:, 0], D[0][:, 1], D[0][:, 2]], 5),
          "f0 comps 0-2, 8 bits": morton([D[0][:, 0], D[0][:, 1], D[0][:, 2]], 8),
          "f0 0-1 + f1 0-1, 5 bits": morton([D[0][:, 0], D[0][:, 1], D[1][:, 0], D[1][:, 1]], 5),
          "f0 0-2 + f1 0-2, 4 bits": morton([D[0][:, 0], D[0][:, 1], D[0][:, 2], D[1][:, 0], D[1][:, 1], D[1][:, 2]], 4),
          "f1 comps 0-2, 5 bits": morton([D[1][:, 0], D[1][:, 1], D[1][:, 2]], 5),
          "f0 0-3, 4 bits": morton([D[0][:, 0], D[0][:, 1], D[0][:, 2], D[0][:, 3]], 4)}
for name, order in curves.items():
    for TI, TJ in ((16, 128), (16, 64), (8, 64)):
        c = count(order, TI, TJ)
        base = base or c
        print(f"{name:32s} tiles {TI:2d} x {TJ:3d}: screened pairs {c:.3e}  ({c / base:.2f} of the device's)", flush=True)
