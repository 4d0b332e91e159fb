"""Records what the CPU oracle (oracle/tsc_oracle.c, pinned by G1-G6) gives on the full-size BASELINE
configs, so that bench.py and the GPU tests can check a full-size run without re-running minutes of CPU
work: counts, per-pass active counts / pair evaluations and a SHA-256 of the packed survivor mask.

    python tests/golden/gen_expected_full.py C2 C3 [--modes 0 1]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from tscode_amd.synthetic import make_config  # noqa: E402

OUT = os.path.join(HERE, "expected_full.json")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    modes = [0, 1]
    if "--modes" in sys.argv:
        modes = [int(m) for m in sys.argv[sys.argv.index("--modes") + 1:]]
        args = [a for a in args if not a.isdigit()]
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for cfg in args:
        ens = make_config(cfg)
        poses = oracle.transform_batch(ens.frag_coords, ens.conf_idx, ens.rot, ens.pos)
        cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
        heavy = np.ascontiguousarray(poses[cm][:, ens.atomnos != 1])
        clash_margin = oracle.clash_margin(poses, ens.ids, 1.5)
        for mode in modes:
            t0 = time.time()
            res = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True)
            mr, mm = oracle.prune_margins(heavy, 0.5, mode)
            key = f"{cfg}:{ens.n_poses}:mode{mode}"
            data[key] = {
                "n_pass": int(cm.sum()), "n_keep": int(res["mask"].sum()),
                "clash_sha256_16": hashlib.sha256(np.packbits(cm).tobytes()).hexdigest()[:16],
                "keep_sha256_16": hashlib.sha256(np.packbits(res["mask"]).tobytes()).hexdigest()[:16],
                "passes": [{"k": s["k"], "active_after": s["n_active_after"], "pairs_evaluated": s["pairs_evaluated"]} for s in res["stats"]],
                "margins": {"rmsd": mr, "maxdev": mm, "clash": clash_margin},
                "oracle_seconds": time.time() - t0,
            }
            print(key, data[key]["n_pass"], data[key]["n_keep"], "margins", mr, mm, clash_margin, f"{time.time() - t0:.0f}s", flush=True)
            json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
