"""Latency of one prune_conformers_rmsd call (host arrays in, mask out) on small ensembles, chunk-local kernel on / off."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from tscode_amd import get_engine
eng = get_engine(0)
rng = np.random.default_rng(5)
for N in (240, 1000, 2000, 4000, 8000, 20000):
    h = 30
    base = rng.normal(size=(N // 6 + 1, h, 3)) * 3
    heavy = np.ascontiguousarray((base[:, None] + rng.normal(size=(N // 6 + 1, 6, h, 3)) * 0.03).reshape(-1, h, 3)[rng.permutation((N // 6 + 1) * 6)][:N])
    out = []
    for local, pca in ((1, 6000), (0, 6000), (1, 0)):          # default; chunk-local kernel off; principal axes at every size
        eng.set_option("local_pass", local)
        eng.set_option("pca_min_n", pca)
        for _ in range(3):
            mask, st = eng.prune_heavy(heavy, 0.5, 0)
        t0 = time.perf_counter()
        for _ in range(20):
            mask, st = eng.prune_heavy(heavy, 0.5, 0)
        out.append(((time.perf_counter() - t0) / 20 * 1e3, int(mask.sum()), [s["algo"] for s in st]))
    print(N, "default %.3f ms  no chunk-local kernel %.3f ms  principal axes always %.3f ms" % (out[0][0], out[1][0], out[2][0]),
          out[0][1] == out[1][1] == out[2][1], out[0][2], flush=True)
eng.set_option("local_pass", 1)
eng.set_option("pca_min_n", 6000)
