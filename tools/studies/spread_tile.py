import sys, time, gc
sys.path.insert(0, "/root/repo")
import torch
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_ensemble
ens = make_ensemble(100_000, (25, 25), seed=7003, children=10, sigma_rot_deg=6.0, sigma_t=0.12)
pipe = DevicePipeline(ens, device_index=0, mode=0)
pipe.set_option("prune_algo", 1)
pipe.set_option("pass_timing", 2)
for _ in range(3): res = pipe.step()
torch.cuda.synchronize()
print([(s["k"], round(s["gpu_ms"],3), round(s["tile_ms"],3)) for s in res["stats"]])
