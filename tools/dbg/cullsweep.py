import sys, time, json
sys.path.insert(0, ".")
import torch
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
cfg = sys.argv[1]
grid = json.loads(sys.argv[2])
ens = make_config(cfg)
pipe = DevicePipeline(ens, device_index=0, mode=0)
pipe.set_option("pass_timing", 0)
import itertools
combos = [dict(zip(grid, v)) for v in itertools.product(*grid.values())]
res = {i: [] for i in range(len(combos))}
for rnd in range(3):
    for i, c in enumerate(combos):
        for k, v in c.items(): pipe.set_option(k, v)
        pipe.step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): r = pipe.step()
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / 5 * 1e3)
for i, c in enumerate(combos):
    print("%8.3f  %s  keep %d" % (min(res[i]), c, r["n_keep"]))
