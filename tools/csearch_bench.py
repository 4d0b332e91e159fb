"""Candidates/s of the batched conformational-search rotations (host arrays in and out) next to the oracle on the host cores."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import oracle
from tscode_amd import get_engine
from tscode_amd.synthetic import make_config
eng = get_engine(0)
rng = np.random.default_rng(21)
ens = make_config("C5", 4)
coords = ens.poses()[0]
n, n0 = len(coords), ens.frag_coords[0].shape[1]
centres = rng.choice(np.arange(2, n0 - 3), size=8, replace=False)
torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres], dtype=np.int32)
masks = np.zeros((8, n), dtype=np.uint8)
for t, c in enumerate(centres):
    masks[t, c + 1:n0] = 1
for M in (10_000, 100_000):
    angles = rng.choice(np.array([0, 0, 60, 120, 180, -60, 25]), size=(M, 8)).astype(np.int32)
    eng.csearch_rotate(coords, torsions, masks, angles[:100], 1.4, 0)
    t0 = time.perf_counter(); out, rb = eng.csearch_rotate(coords, torsions, masks, angles, 1.4, 0); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); ro, rr = oracle.csearch_rotate(coords, torsions, masks, angles[:5000], 1.4, 0); dc = time.perf_counter() - t0
    print(f"{n} atoms, 8 torsions, {M} candidates: GPU {M / dt:.3g} candidates/s (PCIe-inclusive, {dt * 1e3:.1f} ms), oracle {5000 / dc:.3g}/s; "
          f"equal on the first 5000: {np.array_equal(rb[:5000], rr)}")
