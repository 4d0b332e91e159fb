import sys, time
sys.path.insert(0, ".")
import torch
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
ens = make_config(cfg)
pipe = DevicePipeline(ens, device_index=0, mode=0)
for cull in (0, 2, 1):
    pipe.set_option("cull", cull)
    pipe.set_option("pass_timing", 2)
    for _ in range(2): res = pipe.step()
    torch.cuda.synchronize()
    print("cull", cull, "n_keep", res["n_keep"])
    for s in res["stats"]:
        if s["k"] <= 100:
            print("   k %5d  ms %7.3f tile_ms %7.3f  screened %.3g  formed %.3g  exact %d  evaluated %.3g" % (s["k"], s["gpu_ms"], s["tile_ms"], s["pairs_screened"], s["pairs_computed"], s["candidates"], s["pairs_evaluated"]))
