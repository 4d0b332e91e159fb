"""Offline (CPU, oracle): per pass of a prune, how long are the rows' ranges?

For every pass of the reference's schedule on a synthetic config this prints, per active row, the number of ACTIVE columns
between the row and its stop column (first cached column, or the end of its chunk) -- what a pair kernel may have to look at --
and, per 16-row tile, the largest such range.  It is the input to the question "can a whole row tile be walked by ONE
wavefront in ONE launch" (k_pass_rows).

    python tools/range_study.py [C3] [n_conformers]
"""

import sys
import os
import json

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from tscode_amd.synthetic import make_config  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    n_conf = int(sys.argv[2]) if len(sys.argv) > 2 else None
    ens = make_config(cfg, n_conf) if n_conf else make_config(cfg)
    poses = oracle.transform_batch(ens.frag_coords, ens.conf_idx, ens.rot, ens.pos)
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    heavy = np.ascontiguousarray(poses[cm][:, np.asarray(ens.atomnos) != 1])
    n = len(heavy)
    res = oracle.prune_heavy(heavy, 0.5, 0, row_parallel=True, trace=True)
    keys = res["keys"]
    out = []
    mask = np.ones(n, bool)
    key_pos = 0
    for p, st in enumerate(res["stats"]):
        k = int(st["k"])
        cs = n // k
        cache = keys[:key_pos]
        # deltas per chunk start
        by_start = {}
        for a, b in cache:
            by_start.setdefault(int(a), set()).add(int(b - a))
        act = np.flatnonzero(mask)
        rank = np.cumsum(mask) - 1
        ranges = np.zeros(len(act), np.int64)
        for c in range(k):
            f = c * cs
            l = n if c == k - 1 else f + cs
            rows = act[(act >= f) & (act < l)]
            if len(rows) == 0:
                continue
            d = by_start.get(f)
            end_rank = rank[rows[-1]] + 1
            if not d:
                ranges[rank[rows]] = end_rank - rank[rows] - 1
                continue
            dl = np.array(sorted(d))
            # stop column of row i: the smallest i + d (d in D_f) that is active and < l
            stop = np.full(len(rows), l, np.int64)
            for dd in dl:
                j = rows + dd
                ok = (j < l) & (stop > j)
                ok[ok] &= mask[j[ok]]
                stop[ok] = j[ok]
            stop_rank = np.where(stop < l, rank[np.minimum(stop, n - 1)], end_rank)
            ranges[rank[rows]] = stop_rank - rank[rows] - 1
        tiles = np.add.reduceat(np.zeros(1), [0])  # placeholder
        nt = (len(act) + 15) // 16
        pad = np.zeros(nt * 16, np.int64)
        pad[:len(act)] = ranges
        tmax = pad.reshape(nt, 16).max(axis=1)
        q = lambda a, x: int(np.quantile(a, x))
        rec = {"k": k, "active": int(len(act)), "evals": int(st["pairs_evaluated"]), "sum_ranges": int(ranges.sum()),
               "row_range_p50": q(ranges, .5), "p90": q(ranges, .9), "p99": q(ranges, .99), "max": int(ranges.max()),
               "tile_max_p50": q(tmax, .5), "tile_p90": q(tmax, .9), "tile_p99": q(tmax, .99), "tile_max": int(tmax.max()),
               "tiles": int(nt), "tiles_over_512": int((tmax > 512).sum()), "tiles_over_2048": int((tmax > 2048).sum()),
               "pairs_in_tiles_over_512": int(pad.reshape(nt, 16)[tmax > 512].sum())}
        out.append(rec)
        print(json.dumps(rec))
        mask = res["pass_masks"][p].copy()
        key_pos += int(st["new_keys"]) if "new_keys" in st else int(st.get("removed", 0))
    return out


if __name__ == "__main__":
    main()
