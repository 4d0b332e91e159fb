import sys
sys.path.insert(0, ".")
import numpy as np, torch
import tscode_amd
from tscode_amd.engine import PruneStepper
from tscode_amd.synthetic import make_config
from tscode_amd.pipeline import partition_bounds
world, min_chunks, n_poses, mode = 2, 4, 12000, 0
eng = tscode_amd.get_engine(0)
ens = make_config("C2", n_poses)
heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
dev = torch.device("cuda:0")
d_heavy = torch.from_numpy(heavy).to(dev)
n, h = heavy.shape[:2]
# single run, pass by pass
st1 = eng.prune_stepper(d_heavy, n, h, 0.5, mode)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
masks = {}
while True:
    k = st1.next_pass()
    if k == 0: break
    st1.pass_local(0, 1); st1.pass_finish()
    st1.copy_mask(keep); eng.synchronize()
    masks[k] = keep.cpu().numpy().copy()
print({k: int(m.sum()) for k, m in masks.items()})
words = PruneStepper.exchange_words(eng.lib, n, mode)
steppers, exch = [], []
for r in range(world):
    st = eng.prune_stepper(d_heavy, n, h, 0.5, mode)
    ex = torch.zeros(words, dtype=torch.int64, device=dev)
    st.set_partition(r, world, min_chunks, ex)
    steppers.append(st); exch.append(ex)
per_pass = n // 64 + 48
prev = np.ones(n, dtype=np.uint8)
while True:
    k = steppers[0].next_pass(); [s.next_pass() for s in steppers[1:]]
    if k == 0 or not steppers[0].pass_partitioned(): break
    for st in steppers: st.pass_range()
    eng.synchronize()
    want = np.flatnonzero((prev == 1) & (masks[k] == 0))
    got_all = []
    for r, e in enumerate(exch):
        bits = np.unpackbits(e[:per_pass - 8].cpu().numpy().view(np.uint8), bitorder="little")
        got = np.flatnonzero(bits)
        c_lo, c_hi, s_lo, s_hi = partition_bounds(n, k, r, world)
        w_r = want[(want >= s_lo) & (want < s_hi)]
        print(f"k={k} rank {r}: chunks [{c_lo},{c_hi}) structures [{s_lo},{s_hi}) removed {len(got)} want {len(w_r)} tail {e[per_pass-8:per_pass].tolist()}")
        miss = np.setdiff1d(w_r, got); extra = np.setdiff1d(got, w_r)
        if len(miss) or len(extra):
            cs = n // k
            print("   missing", miss[:10], "chunks", (miss[:10] // cs), "extra", extra[:10])
        got_all.append(got)
    total = torch.stack([e[:per_pass] for e in exch]).sum(0)
    for e, st in zip(exch, steppers):
        e[:per_pass].copy_(total); torch.cuda.synchronize(); st.pass_merge()
    eng.synchronize()
    steppers[0].copy_mask(keep); eng.synchronize()
    print("  after merge: active", int(keep.sum()), "single run", int(masks[k].sum()))
    prev = masks[k]
# finish the runs and print the records
first = True
while True:
    if not first:
        ks = [s.next_pass() for s in steppers]
        k = ks[0]
    first = False
    if k == 0: break
    off, w = steppers[0].views_range()
    [s.views_range() for s in steppers[1:]]
    if w:
        eng.synchronize()
        total = torch.stack([e[off:off + w] for e in exch]).sum(0)
        for e in exch: e[off:off + w].copy_(total)
        torch.cuda.synchronize()
    for st in steppers:
        st.views_merged(); st.pass_local(0, 1); st.pass_finish()
for r, st in enumerate(steppers):
    print(r, [(s["k"], s["n_active_before"], s["n_active_after"], s["pairs_evaluated"], s["new_keys"]) for s in st.stats()])
print("single", [(s["k"], s["n_active_before"], s["n_active_after"], s["pairs_evaluated"], s["new_keys"]) for s in st1.stats()])
