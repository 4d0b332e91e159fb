import sys
sys.path.insert(0, ".")
import numpy as np, torch
import tscode_amd, oracle
from tscode_amd.synthetic import make_config
eng = tscode_amd.get_engine(0)
eng.set_option("cull_min_pairs", 0); eng.set_option("cull", 2); eng.set_option("local_pass", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ens = make_config("C2", n)
heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
ref = oracle.prune_heavy(heavy, 0.5, mode=0, row_parallel=True)
dev = torch.device("cuda:0")
d_heavy = torch.from_numpy(heavy).to(dev)
for trial in range(3):
    sts = [eng.prune_stepper(d_heavy, len(heavy), heavy.shape[1], 0.5, 0) for _ in range(W)]
    bests = [torch.empty(len(heavy), dtype=torch.int32, device=dev) for _ in range(W)]
    for s, b in zip(sts, bests): s.use_best_buffer(b)
    while True:
        ks = [s.next_pass() for s in sts]
        if ks[0] == 0: break
        for r, s in enumerate(sts): s.pass_local(r, W)
        eng.synchronize()
        m = torch.stack(bests).amin(0)
        for b in bests: b.copy_(m)
        torch.cuda.synchronize()
        for s in sts: s.pass_finish()
    keep = torch.empty(len(heavy), dtype=torch.uint8, device=dev)
    sts[0].copy_mask(keep); eng.synchronize()
    st = sts[0].stats()
    ok = np.array_equal(keep.cpu().numpy().astype(bool), ref["mask"])
    print("trial", trial, "ok", ok, [(a["k"], a["n_active_after"], b["n_active_after"]) for a, b in zip(st, ref["stats"]) if a["n_active_after"] != b["n_active_after"]][:4])
    for s in sts: s.close()
