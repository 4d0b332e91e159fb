#!/usr/bin/env python
"""Per-kernel micro-benchmarks of the streaming kernels (K1 transform, K2 clash mask, fused K1+K2, compaction),
timed with HIP events on the engine's stream; prints achieved GB/s against the algorithmic byte counts of
SURVEY.md 8(d).  Usage: python tools/microbench.py [C3|C5] [n_poses]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import tscode_amd
    from tscode_amd import FragmentSet
    from tscode_amd.synthetic import make_config
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else None
    ens = make_config(cfg, n)
    eng = tscode_amd.get_engine(0)
    dev = torch.device("cuda:0")
    eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    fs = FragmentSet(ens.frag_coords)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_frags, d_ci, d_rot, d_pos = t(fs.flat), t(ens.conf_idx), t(ens.rot), t(ens.pos)
    N, na, nm = ens.n_poses, ens.n_atoms, len(ens.frag_coords)
    poses = torch.empty((N, na, 3), dtype=torch.float64, device=dev)
    mask = torch.empty(N, dtype=torch.uint8, device=dev)
    comp = torch.empty((N, na, 3), dtype=torch.float64, device=dev)
    ids = ens.ids.astype(np.int32)
    pair_count = sum(int(ids[a]) * int(ids[b]) for a in range(nm) for b in range(a + 1, nm))

    def timeit(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            eng.timer_begin()
            fn()
            best = min(best, eng.timer_end())
        return best

    rows = []
    ms = timeit(lambda: eng.transform_batch_dev(fs, d_frags, d_ci, d_rot, d_pos, N, poses))
    b = N * (na * 24 + nm * 100)
    rows.append(("K1 k_transform", ms, b, N * na * 18))
    ms = timeit(lambda: eng.clash_mask_dev(poses, N, na, ids, 1.5, 0, mask))
    rows.append(("K2 k_clash<false>", ms, N * na * 24 + N, N * pair_count * 9))
    ms = timeit(lambda: eng.embed_clash_mask_dev(fs, d_frags, d_ci, d_rot, d_pos, N, 1.5, 0, mask))
    rows.append(("K1+K2 k_clash<true>", ms, N * nm * 100 + N, N * (pair_count * 9 + na * 18)))
    kept = [0]

    def compact():
        kept[0] = eng.compact_rows_dev(poses, mask, N, na * 24, comp)
    ms = timeit(compact, reps=5)
    rows.append(("compact_rows (scan + gather, 1 sync)", ms, N + kept[0] * na * 48, 0))
    print(f"{cfg}: {N} poses x {na} atoms, {nm} fragments; {int(mask.sum())} pass the clash check")
    for name, ms, byt, fl in rows:
        print(f"  {name:38s} {ms * 1e3:9.1f} us   {byt / ms / 1e6:8.1f} GB/s (algorithmic)   {fl / ms / 1e9:8.2f} TFLOP/s fp64")


if __name__ == "__main__":
    main()
