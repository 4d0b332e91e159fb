import sys, time, gc, collections, os
sys.path.insert(0, ".")
import numpy as np, torch
import torch.distributed as dist
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
pipe = DevicePipeline(make_config("C3"), device_index=0, rank=0, world=1, mode=0, force_sharded=len(sys.argv) > 1)
gc.collect()
res = [pipe.step() for _ in range(12)]
torch.cuda.synchronize()
gc.set_debug(gc.DEBUG_SAVEALL)
t = time.perf_counter(); n = gc.collect(); dt = time.perf_counter() - t
print("collected", n, "in", round(dt * 1e3, 2), "ms")
print(collections.Counter(type(o).__name__ for o in gc.garbage).most_common(12))
for o in gc.garbage[:400]:
    if type(o).__name__ in ("dict",):
        ks = list(o.keys())[:6]
        print("dict keys", ks); break
fn = [o for o in gc.garbage if type(o).__name__ == "function"][:5]
print([f.__qualname__ for f in fn])
dist.destroy_process_group()
