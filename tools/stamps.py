"""Where the wavefronts of one pair-kernel launch spend their time (measurement build only).

Build a second library with -DTSC_DBG_STAMPS next to the product one (it must live under tscode_amd/ to travel to the GPU box):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -DTSC_DBG_STAMPS \
          -o tscode_amd/libtscode_hip_stamps.so tscode_amd/csrc/*.hip
and run   TSCODE_AMD_LIB=$PWD/tscode_amd/libtscode_hip_stamps.so python tools/stamps.py C3 100 [opt=value ...]
(a negative k stamps k_open_rows of that pass instead of the pair kernel)
The kernel stamps the 100 MHz wall clock (after draining its outstanding memory operations: the stamps perturb it a little) at:
0 wavefront started, 1 prologue data arrived, 2 screen done, 3 candidates evaluated, 4 arrived at the tile's counter, 5 tile applied,
6 arrived at the pass's counter, 7 pass closed.  Printed: percentiles of every phase over the wavefronts that went through it, in us.
"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from tscode_amd.pipeline import DevicePipeline
from tscode_amd.synthetic import make_config

cfg, k = sys.argv[1], int(sys.argv[2])
pipe = DevicePipeline(make_config(cfg), 0)
for opt in sys.argv[3:]:
    pipe.set_option(opt.split("=")[0], float(opt.split("=")[1]))
for _ in range(3):
    pipe.step()
pipe.set_option("dbg_stamp_k", k)
pipe.step()
torch.cuda.synchronize()
lib, h = pipe.engine.lib, pipe.engine._h
cap = 1 << 23
buf = np.zeros((cap, 8), dtype=np.uint64)
n = C.c_int64()
lib.tsc_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
assert lib.tsc_debug_stamps(h, buf.ctypes.data, cap, C.byref(n)) == 0
s = buf[: n.value].astype(np.int64)
started = s[:, 0] > 0
t_start = s[started, 0].min()
print(f"{cfg} pass k = {k}: {n.value} wavefronts launched, {int(started.sum())} stamped a start, {int((s[:, 1] > 0).sum())} had work, "
      f"{int((s[:, 5] > 0).sum())} applied a tile")


def pct(x):
    return "  ".join(f"{np.percentile(x, q) / 100.0:7.2f}" for q in (0, 50, 90, 99, 100))


print("phase (us)                                    min     p50     p90     p99     max")
print("wavefront start after the first one        ", pct(s[started, 0] - t_start))
names = ["prologue (start -> data arrived)", "screen", "drains (candidates evaluated)", "-> arrived at the tile counter", "apply the tile",
         "-> arrived at the pass counter", "close the pass"]
if k < 0:      # k_open_rows of pass -k: 0 started, 1 bit copy made and prefix staged, 2 structure of the row found, 3 cache view walked, 4 rank of the stop column, 5 written
    names = ["bit copy, prefix -> LDS", "search + select (the row's structure)", "descriptor requested, cache view walked", "rank of the stop column", "stores"]
for i, nm in enumerate(names, start=1):
    m = (s[:, i] > 0) & (s[:, i - 1] > 0)
    if m.any():
        print(f"{nm:43s}", pct(s[m, i] - s[m, i - 1]), f"  ({int(m.sum())} wavefronts)")
if k < 0:      # (inside the cooperative walk: 6 = the list refilled (last time), 7 = the list tested (last time))
    for a, b, nm in ((2, 6, "row found -> list refilled (last refill)"), (6, 7, "list refilled -> tested (last round)"), (7, 3, "tested -> walk done")):
        m = (s[:, a] > 0) & (s[:, b] > 0)
        if m.any():
            print(f"{nm:43s}", pct(s[m, b] - s[m, a]), f"  ({int(m.sum())} wavefronts)")
for i in (1, 2, 3, 4, 5, 6, 7):
    m = s[:, i] > 0
    if m.any():
        print(f"stamp {i} after the kernel's first wavefront   ", pct(s[m, i] - t_start))
last = s[s > 0].max()
print(f"first start -> last stamp: {(last - t_start) / 100.0:.2f} us")
