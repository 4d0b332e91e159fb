import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tscode_amd.pipeline import CsearchChain, DevicePipeline
from tscode_amd.synthetic import make_config
which = sys.argv[1]
if which == "chain":
    ens = make_config("C5")
    n0 = ens.frag_coords[0].shape[1]
    torsions, tmasks = CsearchChain.chain_torsions(n0, 8, seed=5)
    table = np.random.default_rng(6).choice(np.array([0, 0, 60, 120, 180, 240, 300, 25]), size=(20000, 8)).astype(np.int32)
    pipe = CsearchChain(ens, torsions, tmasks, table, thresh=1.4, device_index=0, mode=0, seed=7)
else:
    import os, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    pipe = DevicePipeline(make_config("C3"), device_index=0, rank=0, world=1, mode=0, force_sharded=True)
pipe.set_option("pass_timing", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ts = []
for i in range(60):
    torch.cuda.synchronize(); t = time.perf_counter(); pipe.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(which, " ".join(f"{t:.1f}" for t in ts))
