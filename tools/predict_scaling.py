"""Predicted strong-scaling curve of the sharded single-ensemble protocol, from ONE GPU.

The builder has one MI355X at a time; the N = 2, 4, 8 line of bench.py can only run on the driver's node.  What one GPU can
measure is every rank's SHARE of the work.  For N in {1, 2, 4, 8} this script keeps N prune runs ("ranks") over the same
survivor list in lockstep, exactly as tscode_amd/pipeline.py::sharded_step drives them, and times every library call of every
rank with HIP events:
  * the front half in its three forms -- rank r's pose block (embed + clash + compaction; `shard`), all poses on every rank
    (`replicate`), rank r's clash verdicts + the embed of ALL survivors on every rank (`hybrid`);
  * a PARTITIONED pass (k >= PARTITION_MIN_CHUNKS x N chunks): tsc_prune_pass_range of every rank (its own chunks: rows, stop
    columns, pair search, verdicts), the exchange buffers summed here with torch as the all-reduce would, tsc_prune_pass_merge
    of every rank;
  * a pass SHARDED BY ROW TILES: tsc_prune_pass_local(r, N) of every rank, best[] min-merged here, tsc_prune_pass_finish;
  * a small pass that every rank runs whole.
A pass costs  max_r(local part) + collective + max_r(closing part);  the collectives are MODELLED, not measured: per-link xGMI
bandwidth 153 GB/s x 0.7 efficiency, 20 us fixed cost per collective, and two readings of the mesh -- ring collectives bound by
ONE link (conservative) and every shard sent straight to its N - 1 peers over all links at once (what the fully connected xGMI
mesh allows) -- both stated in the output.

usage (GPU box): python tools/predict_scaling.py [C3 C4] [--chunks M] > profiles/r03_predicted_scaling.json
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from tscode_amd.engine import PruneStepper
from tscode_amd.pipeline import PARTITION_MIN_CHUNKS, SHARD_MIN_PAIRS, SHARDED_CULL_MIN_PAIRS, HipShardBackend
from tscode_amd.synthetic import make_config

LINK_GBS, LINK_EFF, COLL_FIXED_US = 153.0, 0.7, 20.0
# The fixed cost of a collective is the model's least certain number: 20 us is what a world of ONE shows for an RCCL all-reduce from
# torch.distributed; eight ranks, a Python frame per collective and torch's own work in between can make it 40 - 80 us.  Every predicted
# step is therefore ALSO given at these fixed costs (`by_collective_fixed_us`); a step makes `collectives_per_step` of them.  The library's own
# exchange (csrc/xchg.hip, profiles/r05_ipc_exchange.json: 6.5 - 10.5 us up to 230 KB, 25 us at 1.9 MB, two processes sharing one GPU) lies below
# the first of them.
SENSITIVITY_FIXED_US = (10.0, 20.0, 50.0, 100.0)
# The library's own exchange (tsc_xchg_*): a rank writes its whole contribution into every peer (one link each, all at once) and folds
# what it received: fixed cost from profiles/r05_ipc_exchange.json (two launches: 6.5 - 10.5 us measured up to 230 KB with two processes
# sharing one GPU; 8 us taken) + bytes over ONE link.  Used for the per-pass messages (`predicted_ms_per_step_ipc`); the front half's
# collectives stay with the collective library in the model as in the code.
IPC_FIXED_US = 8.0


def ipc_allreduce_ms(nbytes, n):
    if n == 1:
        return 0.0
    return IPC_FIXED_US / 1e3 + nbytes / (LINK_GBS * LINK_EFF * 1e9) * 1e3


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


REFERENCE_MASKS = {}      # (config, structures entering the prune) -> digest of the survivor mask of the first run measured (N = 1 comes first)


def measure_front(ens, n_ranks, reps):
    """(slowest rank of `shard`, pass counts per rank, slowest rank's clash verdicts alone)"""
    front, counts, clash_only = [], [], []
    for r in range(n_ranks):
        be = HipShardBackend(ens, 0, r, n_ranks, 1.5, 0, 0.5, 0)
        be.eng.set_option("pass_timing", 0)
        tm = Timer(be.stream)
        runs_r = [tm(be.embed_clash_block) for _ in range(reps + 1)]
        front.append(min(t for t, _ in runs_r))
        counts.append(int(runs_r[-1][1]))
        clash_only.append(min(tm(be.clash_block_into_all)[0] for _ in range(reps + 1)))    # hybrid: everything in front of the mask exchange
        del be
        torch.cuda.empty_cache()
    return front, counts, clash_only


def measure(cfg, n_ranks, chunks, reps=3, front_all_ms=None, embed_all_ms=None, setup_borrowed_ms=None):
    ens = make_config(cfg)
    front, counts, clash_only = measure_front(ens, n_ranks, reps)
    # the prune over the whole survivor list: one backend holding everything (world 1 = the whole pose axis)
    be = HipShardBackend(ens, 0, 0, 1, 1.5, 0, 0.5, 0)
    be.eng.set_option("pass_timing", 0)
    # every emulated rank must derive bit-identical descriptors, as the ranks of a real run do (HipShardBackend sets deterministic_basis
    # at world > 1): fixed-order sums, and no pose-sample basis left pending by the embed that only the FIRST stepper would consume
    # (layouts that differ between the emulated ranks leave pairs unvisited in a culled pass dealt by row tiles)
    be.eng.set_option("deterministic_basis", 1)
    be.world = 2      # (embed_clash_block forks a pose-sample basis for the NEXT prune run when its backend is alone in the world: the first stepper would
                      # take it, the others build their own from the survivors, and a culled pass dealt by row tiles then misses pairs -- which is what the
                      # mask comparison below caught at C4 x 2 on its first run; the pose block was cut for a world of one and stays whole)
    if n_ranks > 1:   # (what HipShardBackend sets on a rank of a real sharded run)
        be.eng.set_option("cull_min_pairs", SHARDED_CULL_MIN_PAIRS)
    for name, value in OPTIONS:
        be.eng.set_option(name, value)
    tm = Timer(be.stream)
    n_pass = int(be.embed_clash_block())
    with torch.cuda.stream(be.stream):
        be.heavy_all[:n_pass].copy_(be.heavy_local[:n_pass])
    if n_ranks == 1:                  # hybrid, behind the exchange: every survivor's heavy atoms + descriptors on every rank (the same at every N)
        ts = []
        for _ in range(reps + 1):
            be.clash_block_into_all()
            ts.append(tm(be.embed_masked_all)[0])
        embed_all_ms = min(ts)
        # ... and the prune run that follows it BORROWS the descriptors and the float32 copy that embed_masked_all wrote beside the heavy atoms
        # (tsc_prune_create on the same array): its set-up, not the one measured below on runs that build their own from heavy_all
        tb = []
        for _ in range(reps + 1):          # (the first creation also pays the run's allocations, which later ones take from the context's cache)
            be.clash_block_into_all()      # (forks the basis that embed_masked_all's descriptors are written in: the order of _front_hybrid)
            n_all = int(be.embed_masked_all())
            t_b, st_b = tm(lambda: be.eng.prune_stepper(be.heavy_all, n_all, be.h, 0.5, 0))
            st_b.close()
            tb.append(t_b)
        setup_borrowed_ms = min(tb)
        n_pass = int(be.embed_clash_block())
        with torch.cuda.stream(be.stream):
            be.heavy_all[:n_pass].copy_(be.heavy_local[:n_pass])
    h = be.h
    words = PruneStepper.exchange_words(be.eng.lib, n_pass, 0)
    per_pass_words = n_pass // 64 + 48
    exch = [torch.zeros(words, dtype=torch.int64, device=be.dev) for _ in range(n_ranks)]
    bests = [torch.empty(n_pass, dtype=torch.int32, device=be.dev) for _ in range(n_ranks)]
    runs = []
    for rep in range(reps):
        passes, setup = [], 0.0
        sts = []
        for r in range(n_ranks):
            t, st = tm(lambda r=r: be.eng.prune_stepper(be.heavy_all, n_pass, h, 0.5, 0))
            setup = max(setup, t)
            st.use_best_buffer(bests[r])
            if n_ranks > 1 and chunks > 0:
                with torch.cuda.stream(be.stream):
                    st.set_partition(r, n_ranks, chunks, exch[r])
            sts.append(st)
        while True:
            ks = {st.next_pass() for st in sts}
            assert len(ks) == 1
            k = ks.pop()
            if k == 0:
                break
            est = sts[0].pass_estimate()
            if n_ranks > 1 and chunks > 0 and sts[0].pass_partitioned():
                t_range = [tm(st.pass_range)[0] for st in sts]
                with torch.cuda.stream(be.stream):
                    total = torch.stack([e[:per_pass_words] for e in exch]).sum(0)
                    for e in exch:
                        e[:per_pass_words].copy_(total)
                t_merge = [tm(st.pass_merge)[0] for st in sts]
                passes.append({"k": int(k), "kind": "partitioned", "local_ms_per_rank": t_range, "close_ms": max(t_merge), "bytes": 8 * per_pass_words})
                continue
            extra = 0
            if n_ranks > 1 and chunks > 0:
                off, w = sts[0].views_range()
                for st in sts[1:]:
                    st.views_range()
                if w:
                    with torch.cuda.stream(be.stream):
                        total = torch.stack([e[off:off + w] for e in exch]).sum(0)
                        for e in exch:
                            e[off:off + w].copy_(total)
                    extra = 8 * w
                t_v = [tm(st.views_merged)[0] for st in sts]
                if w:
                    passes.append({"k": int(k), "kind": "views", "local_ms_per_rank": [0.0] * n_ranks, "close_ms": max(t_v), "bytes": extra})
            if n_ranks > 1 and est >= SHARD_MIN_PAIRS:
                t_loc = [tm(lambda r=r, st=st: st.pass_local(r, n_ranks))[0] for r, st in enumerate(sts)]
                n_best = sts[0].best_ptr()[1]
                with torch.cuda.stream(be.stream):
                    merged = torch.stack(bests).amin(0)
                    for b in bests:
                        b.copy_(merged)
                t_fin = [tm(st.pass_finish)[0] for st in sts]
                passes.append({"k": int(k), "kind": "row_tiles", "local_ms_per_rank": t_loc, "close_ms": max(t_fin), "bytes": 4 * n_best})
            else:
                t_all = [tm(lambda st=st: (st.pass_local(0, 1), st.pass_finish()))[0] for st in sts]
                passes.append({"k": int(k), "kind": "replicated", "local_ms_per_rank": t_all, "close_ms": 0.0, "bytes": 0})
        t_tail = max(tm(lambda st=st: st.copy_mask(be.keep))[0] for st in sts)
        stats = sts[0].stats()
        # every emulated rank ends with the same survivors and active counts -- and with the one-rank run's (REFERENCE_MASKS)
        digests = []
        for st in sts:
            st.copy_mask(be.keep)
            torch.cuda.synchronize()
            digests.append(hashlib.sha256(np.packbits(be.keep[:n_pass].cpu().numpy().astype(bool)).tobytes()).hexdigest()[:16])
            assert [x["n_active_after"] for x in st.stats()] == [x["n_active_after"] for x in stats], "the emulated ranks disagree on the active counts"
        assert len(set(digests)) == 1, f"the emulated ranks end with different survivor masks: {digests}"
        actives = [(x["k"], x["n_active_after"], x["pairs_evaluated"]) for x in stats]
        ref, ref_actives = REFERENCE_MASKS.setdefault((cfg, n_pass), (digests[0], actives))
        assert ref == digests[0], (f"{cfg} with {n_ranks} emulated ranks: survivors differ from the first run's ({digests[0]} != {ref}); first pass that differs "
                                   f"(k, active after, evaluations): {next(((a, b) for a, b in zip(actives, ref_actives) if a != b), None)}; kinds {[(p['k'], p['kind']) for p in passes]}")
        if os.environ.get("PREDICT_COUNTS"):          # work counters of the passes dealt by row tiles, summed over the ranks (they are per rank there)
            every = [st.stats() for st in sts]
            tiled = {p["k"] for p in passes if p["kind"] in ("row_tiles", "replicated")}
            for i, row in enumerate(stats):
                if row["k"] in tiled:
                    n = 1 if n_ranks == 1 else n_ranks
                    print(f"  counts N={n_ranks} k={row['k']}: " + " ".join(f"{f} {sum(e[i][f] for e in every[:n]) / (1 if n_ranks > 1 else 1):.4g}"
                                                                           for f in ("pairs_screened", "candidates", "pairs_computed")), file=sys.stderr)
        for st in sts:
            st.close()
        runs.append({"setup_ms": setup, "tail_ms": t_tail, "passes": passes, "n_keep": int(stats[-1]["n_active_after"])})

    def compute(run):
        return run["setup_ms"] + run["tail_ms"] + sum(max(p["local_ms_per_rank"]) + p["close_ms"] for p in run["passes"])
    best = min(runs, key=compute)
    gather_bytes = n_ranks * max(counts) * h * 24             # shards padded to the largest count (pipeline.py)

    def pass_comm(all_links):
        return sum(allreduce_ms(p["bytes"], n_ranks, all_links) for p in best["passes"] if p["bytes"])
    prune_ms = compute(best)
    fronts = {}
    # shard: pose blocks + counts all-reduce + all-gather of the survivors' heavy atoms
    fronts["shard"] = {"compute_ms": max(front), "comm_ms": [allreduce_ms(8 * n_ranks, n_ranks, al) + allgather_ms(gather_bytes, n_ranks, al) for al in (False, True)]}
    # replicate: every rank embeds and clash-filters all poses
    fa = max(front) if n_ranks == 1 or front_all_ms is None else front_all_ms
    fronts["replicate"] = {"compute_ms": fa, "comm_ms": [0.0, 0.0]}
    # hybrid: this rank's clash verdicts, the mask summed over the ranks (one byte per pose), every survivor embedded on every rank
    if embed_all_ms is not None:
        # (the set-up of the prune run behind it is the borrowing one: the descriptors and the float32 copy are in embed_all_ms already)
        fronts["hybrid"] = {"compute_ms": max(clash_only) + embed_all_ms,
                            "setup_ms": setup_borrowed_ms if (n_ranks > 1 and setup_borrowed_ms is not None) else best["setup_ms"],
                            "comm_ms": [allreduce_ms(ens.n_poses, n_ranks, al) for al in (False, True)] if n_ranks > 1 else [0.0, 0.0]}
    out_fronts = {}
    for name, f in fronts.items():
        # the prune's set-up depends on the front: `shard` and `replicate` leave heavy atoms only (the run builds descriptors and the float32 copy
        # from them: the set-up measured on the emulated ranks), `hybrid` leaves both beside the heavy atoms (the run borrows them)
        setup_f = f.get("setup_ms", best["setup_ms"])
        body = prune_ms - best["setup_ms"] + setup_f
        ring, fast = f["compute_ms"] + f["comm_ms"][0] + body + pass_comm(False), f["compute_ms"] + f["comm_ms"][1] + body + pass_comm(True)
        # collectives of a step: the front's (shard: counts + coordinates; hybrid: the clash mask) + one per pass that exchanges anything
        n_coll = 0 if n_ranks == 1 else {"shard": 2, "hybrid": 1}.get(name, 0) + sum(1 for p in best["passes"] if p["bytes"])
        pass_ipc = sum(ipc_allreduce_ms(p["bytes"], n_ranks) for p in best["passes"] if p["bytes"])
        out_fronts[name] = {"front_compute_ms": f["compute_ms"], "front_comm_ms_ring": f["comm_ms"][0], "front_comm_ms_all_links": f["comm_ms"][1],
                            "prune_setup_ms": setup_f, "collectives_per_step": n_coll,
                            "predicted_ms_per_step": ring, "predicted_ms_per_step_all_links": fast,
                            "predicted_ms_per_step_ipc": f["compute_ms"] + f["comm_ms"][1] + body + pass_ipc, "pass_comm_ms_ipc": pass_ipc,
                            "by_collective_fixed_us": {str(int(us)): {"ring": ring + n_coll * (us - COLL_FIXED_US) / 1e3,
                                                                      "all_links": fast + n_coll * (us - COLL_FIXED_US) / 1e3} for us in SENSITIVITY_FIXED_US}}
    by_kind = {}
    for p in best["passes"]:
        d = by_kind.setdefault(p["kind"], {"passes": 0, "local_ms": 0.0, "close_ms": 0.0, "comm_ms_ring": 0.0, "comm_ms_all_links": 0.0})
        d["passes"] += 1
        d["local_ms"] += max(p["local_ms_per_rank"])
        d["close_ms"] += p["close_ms"]
        d["comm_ms_ring"] += allreduce_ms(p["bytes"], n_ranks, False) if p["bytes"] else 0.0
        d["comm_ms_all_links"] += allreduce_ms(p["bytes"], n_ranks, True) if p["bytes"] else 0.0
    return {"n_ranks": n_ranks, "partition_min_chunks": chunks, "fronts": out_fronts, "front_ms_per_rank": front, "clash_only_ms_per_rank": clash_only,
            "embed_all_survivors_ms": embed_all_ms, "setup_ms": best["setup_ms"], "setup_borrowed_ms": setup_borrowed_ms, "tail_ms": best["tail_ms"], "prune_compute_ms": prune_ms,
            "pass_comm_ms_ring": pass_comm(False), "pass_comm_ms_all_links": pass_comm(True), "by_kind": by_kind,
            "passes": [{"k": p["k"], "kind": p["kind"], "max_local_ms": max(p["local_ms_per_rank"]), "sum_local_ms": sum(p["local_ms_per_rank"]),
                        "close_ms": p["close_ms"], "bytes": p["bytes"]} for p in best["passes"]],
            "allgather_bytes": gather_bytes, "pass_counts_per_rank": counts, "n_pass_clash": n_pass, "n_survivors": best["n_keep"]}, embed_all_ms, setup_borrowed_ms


OPTIONS = []


def main():
    args = sys.argv[1:]
    chunks = PARTITION_MIN_CHUNKS
    if "--chunks" in args:
        i = args.index("--chunks")
        chunks = int(args[i + 1])
        del args[i:i + 2]
    while "--opt" in args:                      # library options for the prune runs: --opt name=value
        i = args.index("--opt")
        name, value = args[i + 1].split("=")
        OPTIONS.append((name, float(value)))
        del args[i:i + 2]
    ranks = (1, 2, 4, 8)
    if "--ranks" in args:                       # e.g. --ranks 1,8 (the one-rank line is needed for the speed-up)
        i = args.index("--ranks")
        ranks = tuple(int(v) for v in args[i + 1].split(","))
        del args[i:i + 2]
    cfgs = [a for a in args if a.startswith("C")] or ["C3", "C4"]
    out = {"what": __doc__.split("\n\n")[1].replace("\n", " "),
           "model": {"xgmi_link_GBs": LINK_GBS, "link_efficiency": LINK_EFF, "collective_fixed_us": COLL_FIXED_US,
                     "sensitivity_fixed_us": list(SENSITIVITY_FIXED_US), "ipc_exchange_fixed_us": IPC_FIXED_US,
                     "ring (predicted_ms_per_step)": "per-link bound: all-gather = (N-1) steps of one shard over one link; all-reduce = reduce-scatter + all-gather of that pattern",
                     "all_links (predicted_ms_per_step_all_links)": "the fully connected xGMI mesh used at once: every shard straight to its N-1 peers, one link each",
                     "shard_min_pairs": SHARD_MIN_PAIRS, "partition_min_chunks": chunks,
                     "cull_min_pairs_of_a_rank_share": SHARDED_CULL_MIN_PAIRS},
           "measured_on": torch.cuda.get_device_name(0), "configs": {}}
    for cfg in cfgs:
        rows = []
        front_all = embed_all = setup_b = None
        for n in ranks:
            t0 = time.time()
            row, embed_all, setup_b = measure(cfg, n, chunks, front_all_ms=front_all, embed_all_ms=embed_all, setup_borrowed_ms=setup_b)
            rows.append(row)
            if n == ranks[0]:
                front_all = max(rows[0]["front_ms_per_rank"])          # one rank's front half IS the whole pose list
            print(f"{cfg} N={n}: " + " | ".join(f"{name} {f['predicted_ms_per_step']:.3f} (all links {f['predicted_ms_per_step_all_links']:.3f}, ipc {f['predicted_ms_per_step_ipc']:.3f})"
                                                for name, f in row["fronts"].items()) + f"  prune compute {row['prune_compute_ms']:.3f} [{time.time() - t0:.0f} s]",
                  file=sys.stderr)
        base = min(f["predicted_ms_per_step"] for f in rows[0]["fronts"].values())
        for r in rows:
            for f in r["fronts"].values():
                f["speedup_vs_1_rank_protocol"] = base / f["predicted_ms_per_step"]
                f["speedup_vs_1_rank_protocol_all_links"] = base / f["predicted_ms_per_step_all_links"]
                f["speedup_vs_1_rank_protocol_ipc"] = base / f["predicted_ms_per_step_ipc"]
                for v in f["by_collective_fixed_us"].values():
                    v["speedup_ring"], v["speedup_all_links"] = base / v["ring"], base / v["all_links"]
        out["configs"][cfg] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
