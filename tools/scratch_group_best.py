"""Best case for k_pass_group: n = 56000 heavy-atom structures (every chunk of k = 2000 ... 100 aligned: no halos, no long last chunk)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import tscode_amd
from tscode_amd.synthetic import make_config
ens = make_config("C3")
eng = tscode_amd.get_engine(0)
poses = ens.poses()
cm = tscode_amd.compenetration_mask(poses, ens.ids, 1.5, 0)
heavy = np.ascontiguousarray(poses[cm][:, ens.atomnos != 1])
for n in (56000, 57046 if len(heavy) >= 57046 else len(heavy)):
    hv = torch.from_numpy(heavy[:n].copy()).cuda()
    mask = torch.empty(n, dtype=torch.uint8, device="cuda")
    eng.set_option("pass_timing", 2)
    for grp in (1, 0):
        eng.set_option("pass_group", grp)
        ts = []
        for rep in range(6):
            torch.cuda.synchronize(); t = time.perf_counter()
            st = eng.prune_heavy_dev(hv, n, heavy.shape[1], 0.5, 0, mask)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        first = [(s["k"], round(s["gpu_ms"], 4), s["algo"]) for s in st][:6]
        print(n, "pass_group", grp, "total ms", round(min(ts), 4), first, "survivors", int(mask.sum()))
eng.set_option("pass_group", 1); eng.set_option("pass_timing", 0)
