"""Predicted strong-scaling curve of the sharded single-ensemble protocol, from ONE GPU.

The builder has one MI355X at a time; the N = 2, 4, 8 line of bench.py can only run on the driver's node.  What one GPU can
measure is every rank's SHARE of the work: for N in {1, 2, 4, 8} and every rank r < N this script times, with HIP events,
  * the embed + clash + compaction of rank r's pose block (HipShardBackend.embed_clash_block),
  * for every pass that the protocol shards: tsc_prune_pass_local(r, N) -- rank r's row tiles -- one rank after the other on
    the same card (best[] accumulates by atomicMin, so after the last rank the pass is complete and the run continues with
    the true verdicts),
  * everything the ranks replicate (the small passes, the per-pass bookkeeping around a sharded pass, the export),
and combines them as the protocol would run:  max_r front(r) + counts all-reduce + coordinates all-gather +
sum over passes [replicated part + max_r local(r) + all-reduce(best[])] .  The collectives are MODELLED, not measured:
per-link xGMI bandwidth 153 GB/s x 0.7 efficiency, 20 us fixed cost per collective, and two readings of the mesh -- ring
collectives bound by ONE link (conservative) and every shard sent straight to its N - 1 peers over all links at once
(what the fully connected xGMI mesh allows) -- both stated in the output.  Beside it, for every N, the other form of the front
half (`front_replicate`: every rank embeds all poses itself, no coordinates travel; pipeline.py times both on the node and
keeps the faster).

usage (GPU box): python tools/predict_scaling.py [C3 C4] > profiles/r02_predicted_scaling.json
"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from tscode_amd.pipeline import SHARD_MIN_PAIRS, HipShardBackend
from tscode_amd.synthetic import make_config

LINK_GBS, LINK_EFF, COLL_FIXED_US = 153.0, 0.7, 20.0


class Timer:
    def __init__(self, stream):
        self.stream = stream

    def __call__(self, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            a.record(self.stream)
            out = fn()
            b.record(self.stream)
        b.synchronize()
        return a.elapsed_time(b), out


def allgather_ms(total_bytes, n, all_links=False):
    """ring: N - 1 steps of one shard over ONE link; all_links: every shard goes straight to its N - 1 peers, one link each."""
    if n == 1:
        return 0.0
    shard = total_bytes / n
    steps = 1 if all_links else n - 1
    return shard * steps / (LINK_GBS * LINK_EFF * 1e9) * 1e3 + COLL_FIXED_US / 1e3


def allreduce_ms(nbytes, n, all_links=False):
    """reduce-scatter + all-gather of the same pattern."""
    if n == 1:
        return 0.0
    steps = 1 if all_links else n - 1
    return 2.0 * (nbytes / n) * steps / (LINK_GBS * LINK_EFF * 1e9) * 1e3 + COLL_FIXED_US / 1e3


def measure(cfg, n_ranks, reps=3, front_all_ms=None):
    ens = make_config(cfg)
    # front half: every rank's block
    front, counts = [], []
    for r in range(n_ranks):
        be = HipShardBackend(ens, 0, r, n_ranks, 1.5, 0, 0.5, 0)
        be.eng.set_option("pass_timing", 0)
        tm = Timer(be.stream)
        runs_r = [tm(be.embed_clash_block) for _ in range(reps + 1)]
        front.append(min(t for t, _ in runs_r))
        counts.append(int(runs_r[-1][1]))
        del be
        torch.cuda.empty_cache()
    # the prune over the whole survivor list: one backend holding everything (world 1 = the whole pose axis)
    be = HipShardBackend(ens, 0, 0, 1, 1.5, 0, 0.5, 0)
    be.eng.set_option("pass_timing", 0)
    tm = Timer(be.stream)
    n_pass = int(be.embed_clash_block())
    with torch.cuda.stream(be.stream):
        be.heavy_all[:n_pass].copy_(be.heavy_local[:n_pass])
    h = be.h
    runs = []
    for rep in range(reps):
        passes, replicated = [], 0.0
        t_create, st = tm(lambda: be.make_stepper(n_pass))
        replicated += t_create
        while True:
            t_rep, k = tm(lambda: st.run_replicated(n_ranks, SHARD_MIN_PAIRS))
            replicated += t_rep
            if k == 0:
                break
            # rank 0's call opens the pass (scan, cache view, stop columns: replicated work) and searches rank 0's rows; the
            # other ranks' rows follow one by one (tsc_prune_pass_rows), rank 0's once more alone (idempotent) to separate its
            # pair search from the opening
            t_open0, _ = tm(lambda: st.pass_local(0, n_ranks))
            local = [tm(lambda r=r: st.pass_rows(r, n_ranks))[0] for r in range(n_ranks)]
            replicated += max(t_open0 - local[0], 0.0)
            n_best = st.n_active()
            t_fin, _ = tm(st.pass_finish)
            replicated += t_fin
            passes.append({"k": int(k), "best_entries": int(n_best), "local_ms_per_rank": local})
        t_tail, _ = tm(lambda: st.copy_mask(be.keep))
        replicated += t_tail
        stats = st.stats()
        st.close()
        runs.append({"replicated_ms": replicated, "passes": passes, "n_keep": int(stats[-1]["n_active_after"])})
    best = min(runs, key=lambda r: r["replicated_ms"] + sum(max(p["local_ms_per_rank"]) for p in r["passes"]))
    gather_bytes = n_ranks * max(counts) * h * 24             # shards padded to the largest count (pipeline.py)

    def comm_ms(all_links):
        return (allreduce_ms(8 * n_ranks, n_ranks, all_links) + allgather_ms(gather_bytes, n_ranks, all_links)
                + sum(allreduce_ms(4 * p["best_entries"], n_ranks, all_links) for p in best["passes"]))
    comm, comm_fast = comm_ms(False), comm_ms(True)
    compute = max(front) + best["replicated_ms"] + sum(max(p["local_ms_per_rank"]) for p in best["passes"])
    # the other form of the front half (pipeline.py, front="replicate"): every rank embeds and clash-filters ALL poses, no counts
    # all-reduce, no all-gather -- only the all-reduces of best[] remain
    front_all = max(front) if n_ranks == 1 or front_all_ms is None else front_all_ms
    rep_comm = [sum(allreduce_ms(4 * p["best_entries"], n_ranks, al) for p in best["passes"]) for al in (False, True)]
    rep_compute = front_all + best["replicated_ms"] + sum(max(p["local_ms_per_rank"]) for p in best["passes"])
    replicate = {"front_ms": front_all, "compute_ms": rep_compute, "modelled_comm_ms": rep_comm[0], "modelled_comm_ms_all_links": rep_comm[1],
                 "predicted_ms_per_step": rep_compute + rep_comm[0], "predicted_ms_per_step_all_links": rep_compute + rep_comm[1]}
    return {"n_ranks": n_ranks, "front_replicate": replicate, "front_ms_per_rank": front, "replicated_ms": best["replicated_ms"],
            "sharded_passes": [{"k": p["k"], "best_entries": p["best_entries"], "max_local_ms": max(p["local_ms_per_rank"]),
                                "sum_local_ms": sum(p["local_ms_per_rank"]), "imbalance": max(p["local_ms_per_rank"]) * n_ranks / max(sum(p["local_ms_per_rank"]), 1e-9)}
                               for p in best["passes"]],
            "compute_ms": compute, "modelled_comm_ms": comm, "modelled_comm_ms_all_links": comm_fast, "allgather_bytes": gather_bytes,
            "pass_counts_per_rank": counts, "predicted_ms_per_step": compute + comm, "predicted_ms_per_step_all_links": compute + comm_fast,
            "predicted_conformers_per_s": ens.n_poses / (compute + comm) * 1e3,
            "predicted_conformers_per_s_all_links": ens.n_poses / (compute + comm_fast) * 1e3, "n_pass_clash": n_pass, "n_survivors": best["n_keep"]}


def main():
    cfgs = [a for a in sys.argv[1:] if a.startswith("C")] or ["C3", "C4"]
    out = {"what": __doc__.split("\n\n")[1].replace("\n", " "),
           "model": {"xgmi_link_GBs": LINK_GBS, "link_efficiency": LINK_EFF, "collective_fixed_us": COLL_FIXED_US,
                     "ring (predicted_ms_per_step)": "per-link bound: all-gather = (N-1) steps of one shard over one link; all-reduce = reduce-scatter + all-gather of that pattern",
                     "all_links (predicted_ms_per_step_all_links)": "the fully connected xGMI mesh used at once: every shard straight to its N-1 peers, one link each",
                     "shard_min_pairs": SHARD_MIN_PAIRS},
           "measured_on": torch.cuda.get_device_name(0), "configs": {}}
    for cfg in cfgs:
        rows = []
        front_all = None
        for n in (1, 2, 4, 8):
            t0 = time.time()
            rows.append(measure(cfg, n, front_all_ms=front_all))
            if n == 1:
                front_all = max(rows[0]["front_ms_per_rank"])          # one rank's front half IS the whole pose list
            rp = rows[-1]["front_replicate"]
            print(f"{cfg} N={n}: front sharded {rows[-1]['predicted_ms_per_step']:.3f} ms/step (compute {rows[-1]['compute_ms']:.3f}, modelled comm "
                  f"{rows[-1]['modelled_comm_ms']:.3f}; all links {rows[-1]['predicted_ms_per_step_all_links']:.3f}) | front replicated "
                  f"{rp['predicted_ms_per_step']:.3f} (all links {rp['predicted_ms_per_step_all_links']:.3f}) [{time.time() - t0:.0f} s]", file=sys.stderr)
        base = rows[0]["predicted_ms_per_step"]
        for r in rows:
            r["speedup_vs_1_rank_protocol"] = base / r["predicted_ms_per_step"]
            r["speedup_vs_1_rank_protocol_all_links"] = base / r["predicted_ms_per_step_all_links"]
            r["front_replicate"]["speedup_vs_1_rank_protocol"] = base / r["front_replicate"]["predicted_ms_per_step"]
            r["front_replicate"]["speedup_vs_1_rank_protocol_all_links"] = base / r["front_replicate"]["predicted_ms_per_step_all_links"]
        out["configs"][cfg] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
