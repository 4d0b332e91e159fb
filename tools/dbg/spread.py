import sys
sys.path.insert(0, ".")
import numpy as np
import tscode_amd
from tscode_amd.synthetic import make_unscreenable, make_config
eng = tscode_amd.get_engine(0)
for name, heavy in (("unscreenable", make_unscreenable(40000)), ("C2", None)):
    if heavy is None:
        ens = make_config("C2", 40000); heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
    mask, stats = eng.prune_heavy(heavy, 0.5, 1)
    print(name, sorted({s["algo"] for s in stats}), int(mask.sum()))
