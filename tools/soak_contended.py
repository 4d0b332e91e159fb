"""Soak under in-process contention: a second thread keeps the GPU busy with large matrix products on its own torch stream while
the pipeline steps; every step's result must equal the first.  usage: python tools/soak_contended.py [config] [steps] [opt=value ...]"""
import hashlib, sys, threading, time
sys.path.insert(0, '.')
import numpy as np, torch
from tscode_amd import get_engine
from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config
cfg, steps = (sys.argv[1] if len(sys.argv) > 1 else "C2"), int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pipe = DevicePipeline(make_config(cfg), device_index=0, mode=0)
for opt in sys.argv[3:]:
    pipe.set_option(opt.split("=")[0], float(opt.split("=")[1]))
stop = False
def hammer():
    s = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device="cuda")
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(20):
                a = (a @ a).clamp_(-1, 1)
            s.synchronize()
th = threading.Thread(target=hammer); th.start()
first, bad, t0 = None, 0, time.time()
for i in range(steps):
    res = pipe.step()
    key = (res["n_pass"], res["n_keep"], hashlib.sha256(np.packbits(pipe.h_keep[:res["n_pass"]].numpy().astype(bool)).tobytes()).hexdigest()[:16],
           tuple(s["pairs_evaluated"] for s in res["stats"]))
    first = first or key
    bad += key != first
stop = True; th.join()
print(f"{cfg} {sys.argv[3:]}: {steps} contended steps, {bad} differ from the first, {(time.time() - t0) / steps * 1e3:.2f} ms per step")
sys.exit(1 if bad else 0)
