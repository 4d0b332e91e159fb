import sys, time, cProfile, pstats
sys.path.insert(0, ".")
import numpy as np
import tscode_amd
from tscode_amd.synthetic import make_config
rng = np.random.default_rng(11)
ens3 = make_config("C3", 10)
f1, f2 = ens3.frag_coords[0][:1], ens3.frag_coords[1][:1]
n1, n2 = f1.shape[1], f2.shape[1]
conf1 = f1 + rng.normal(size=(20, n1, 3)) * 0.05
conf2 = f2 + rng.normal(size=(25, n2, 3)) * 0.05
mols = []
for cf, ri in ((conf1[:8], [0, 5]), (conf2[:8], [1, 7])):
    piv = []
    for c in range(len(cf)):
        out = cf[c][ri] - cf[c].mean(axis=0)
        out /= np.linalg.norm(out, axis=1, keepdims=True)
        st_ = cf[c][ri[0]] + 2.2 * out[0] + rng.normal(size=(4, 3)) * 0.3
        en_ = cf[c][ri[1]] + 2.2 * out[1] + rng.normal(size=(4, 3)) * 0.3
        piv.append((en_ - st_, 0.5 * (st_ + en_), np.tile(np.array([[ri[0], ri[1]]]), (4, 1))))
    mols.append(dict(coords=cf, reactive_indices=ri, pivots=piv))
cang = np.stack(np.meshgrid(np.linspace(-45, 45, 6), np.linspace(-45, 45, 6)), -1).reshape(-1, 2)
tscode_amd.cyclical_embed_batch(mols, cang, 1.5)
for _ in range(3):
    t = time.perf_counter(); r = tscode_amd.cyclical_embed_batch(mols, cang, 1.5, return_trace=True); print("ms", (time.perf_counter() - t) * 1e3, len(r[0]))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tscode_amd.cyclical_embed_batch(mols, cang, 1.5)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
