import os, sys, json, hashlib
sys.path.insert(0, ".")
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
ens = make_config(sys.argv[1])
pipe = DevicePipeline(ens, device_index=0, rank=rank, world=world, mode=0, front="replicate")
if len(sys.argv) > 2: pipe.set_option("cull", int(sys.argv[2]))
for step in range(4):
    res = pipe.step(); torch.cuda.synchronize()
    keep = pipe.h_keep[:res["n_pass"]].numpy().copy()
    d = hashlib.sha256(np.packbits(keep.astype(bool)).tobytes()).hexdigest()[:12]
    print(rank, step, d, res["n_keep"], [(s["k"], s["n_active_after"], s["pairs_evaluated"]) for s in res["stats"] if s["k"] <= 50], flush=True)
dist.barrier(); dist.destroy_process_group()
