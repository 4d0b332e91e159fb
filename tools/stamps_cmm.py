"""Stamps of the culled matrix-core pair kernel (cull_mm.hpp; a -DTSC_DBG_STAMPS build): python tools/stamps_cmm.py C4 2"""
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
for _ in range(2):
    pipe.step()
pipe.set_option("dbg_stamp_k", k)
pipe.step()
torch.cuda.synchronize()
lib, h = pipe.engine.lib, pipe.engine._h
cap = 1 << 24
buf = np.zeros((cap, 8), dtype=np.uint64)
n = C.c_int64()
lib.tsc_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
assert lib.tsc_debug_stamps(h, buf.ctypes.data, cap, C.byref(n)) == 0
s = buf[: n.value].astype(np.int64)
st = s[:, 0] > 0
t0 = s[st, 0].min()
work = s[:, 2] > 0
print(f"{n.value} wavefronts, {int(st.sum())} started, {int((s[:,1]>0).sum())} tested boxes, {int(work.sum())} had tiles")
pct = lambda x: "  ".join(f"{np.percentile(x, q) / 100.0:8.2f}" for q in (0, 50, 90, 99, 100))
print("phase (us)                        min      p50      p90      p99      max")
for a, b, nm in ((0, 1, "start -> boxes tested"), (1, 2, "-> rows' operands"), (2, 3, "screen + full batches"), (3, 4, "rest evaluated")):
    m = (s[:, a] > 0) & (s[:, b] > 0)
    if m.any():
        print(f"{nm:28s}", pct(s[m, b] - s[m, a]), f" ({int(m.sum())})   total {float((s[m, b] - s[m, a]).sum()) / 100.0 / 1e3:.1f} ms*wave")
tl, nd = s[work, 7] // 64, s[work, 7] % 64
print("column tiles visited per item: mean %.2f; (row tile, column tile) pairs needed per item: mean %.2f -> %.2f row tiles per visited tile" % (tl.mean(), nd.mean(), nd.sum() / tl.sum()))
u5, u6 = buf[: n.value, 5][work], buf[: n.value, 6][work]
pos, ce, be, l2 = int((u5 >> np.uint64(32)).sum()), int((u5 & np.uint64(0xffffffff)).sum()), int((u6 >> np.uint64(32)).sum()), int((u6 & np.uint64(0xffffffff)).sum())
if pos + ce + be + l2:   # (only a build that counts them: MEASURED.md section 8 has the numbers of the one that did)
    print("candidates dropped where they are decoded: position %d, the row's stop column %d, the similar column it has %d, level 2 %d" % (pos, ce, be, l2))
raw = buf[: n.value].reshape(-1, 32) if n.value % 4 == 0 else None     # (a workgroup's 32 words: 8 stamps + the step timers of its one wavefront)
if raw is not None:
    m = (raw[:, 12] > 0) & (raw[:, 12] < 4096)
    tt = raw[m][:, 8:12].astype(np.float64).sum(axis=0)
    steps = float(raw[m][:, 12].sum()) if m.any() else 0.0
    if steps > 0:
      print("per step (cycles of s_memtime, mean over %d steps): operands still under way %.0f, MFMAs + sign bits %.0f, candidates extracted %.0f, evaluation batches %.0f" % (steps, tt[0] / steps, tt[1] / steps, tt[2] / steps, tt[3] / steps))
print("last stamp after first start: %.1f us" % ((s[s > 0].max() - t0) / 100.0))
