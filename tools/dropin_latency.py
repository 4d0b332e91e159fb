"""Latency of the per-item drop-in functions (one pose / one pair per call, as TSCoDe's Python loops call them)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import tscode_amd
rng = np.random.default_rng(1)
pose = rng.normal(size=(50, 3)) * 4
p, q = rng.normal(size=(2, 30, 3)) * 3
many = rng.normal(size=(36, 50, 3)) * 4
mask = np.zeros(50, dtype=bool); mask[30:] = True
class Mol:
    pass
mols = []
for n in (25, 25):
    m = Mol(); m.atomcoords = rng.normal(size=(3, n, 3)); m.rotation = np.eye(3); m.position = rng.normal(size=3); mols.append(m)
cases = {
    "compenetration_check(50 atoms, ids=(25, 25))": lambda: tscode_amd.compenetration_check(pose, np.array([25, 25]), 1.5, 0),
    "rmsd_and_max_numba(30 atoms)": lambda: tscode_amd.rmsd_and_max_numba(p, q),
    "_rmsd_similarity(ref, 36 structures of 50 atoms)": lambda: tscode_amd._rmsd_similarity(pose, many, 1.0),
    "get_embed(2 molecules)": lambda: tscode_amd.get_embed(mols, (0, 1)),
    "all_dists(25 x 25)": lambda: tscode_amd.all_dists(pose[:25], pose[25:]),
    "torsion_comp_check(50 atoms)": lambda: tscode_amd.torsion_comp_check(pose, (10, 28, 29, 31), mask, 1.5),
}
for name, f in cases.items():
    for _ in range(20):
        f()
    t0 = time.perf_counter()
    for _ in range(500):
        f()
    print(f"{name:52s} {(time.perf_counter() - t0) / 500 * 1e6:7.1f} us per call", flush=True)
