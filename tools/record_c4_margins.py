"""One-off: guard-band margins of the C4 oracle record (the smallest distance of any evaluated pair to either threshold)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tscode_amd.synthetic import make_config
oracle.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
ens = make_config("C4")
poses = oracle.transform_batch(ens.frag_coords, ens.conf_idx, ens.rot, ens.pos)
cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
heavy = np.ascontiguousarray(poses[cm][:, ens.atomnos != 1])
del poses
t0 = time.time()
mr, mm = oracle.prune_margins(heavy, 0.5, 0)
json.dump({"rmsd": mr, "maxdev": mm, "seconds": time.time() - t0}, open("gpurun_out/c4_margins.json", "w"))
print("done", mr, mm, time.time() - t0, flush=True)
