import sys, time, threading, gc
sys.path.insert(0, ".")
import torch
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
ens = make_config("C3")
_t = [0.0]
def _note(phase, info):
    if phase == "start": _t[0] = time.perf_counter()
    elif info["generation"] == 2 or (time.perf_counter() - _t[0]) > 2e-3: print(f"   gc gen {info['generation']}: {(time.perf_counter() - _t[0]) * 1e3:.2f} ms, collected {info['collected']}", flush=True)
gc.callbacks.append(_note)
D, steps = 3, 10
for rep in range(8):
    pipes = [DevicePipeline(ens, device_index=0, rank=0, world=1, mode=0) for _ in range(D)]
    for p in pipes:
        p.set_option("pass_timing", 0)
        for _ in range(2): p.step()
    times = [[] for _ in range(D)]
    def worker(i):
        for _ in range(steps):
            t = time.perf_counter(); pipes[i].step(); times[i].append((time.perf_counter() - t) * 1e3)
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(D)]
    torch.cuda.synchronize(); gc.collect(); gc.freeze()
    print('   loop begins', flush=True)
    t0 = time.perf_counter()
    for t in threads: t.start()
    for t in threads: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('   loop ends', flush=True)
    print(f"rep {rep}: {dt / (steps * D) * 1e3:.3f} ms/step; slowest step per thread:", [f"{max(x):.1f}@{x.index(max(x))}" for x in times], flush=True)
    del pipes
