import sys, time
sys.path.insert(0, ".")
import numpy as np
import tscode_amd
rng = np.random.default_rng(5)
eng = tscode_amd.get_engine(0)
NF, TQ = 100_000, 6
par = rng.uniform(-180, 180, size=(12000, TQ))
tfq = (par[rng.integers(0, len(par), size=NF)] + rng.normal(size=(NF, TQ)) * 0.5).astype(np.float32)
eng.tfd_greedy_filter(tfq[:100])
for _ in range(3):
    t = time.perf_counter(); acc = eng.tfd_greedy_filter(tfq); print("ms", (time.perf_counter() - t) * 1e3, int(acc.sum()))
