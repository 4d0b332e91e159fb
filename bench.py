#!/usr/bin/env python
"""bench.py -- conformers/sec of the geometry hot path on the BASELINE config 3 workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode 0|1] [--config C3] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic ensemble that is already resident in HBM:
fused embed+clash verdicts -> ordered compaction of the passing poses -> prune_conformers_rmsd
(reference-exact mode), verdict masks left on the device and copied to pinned host memory.

N > 1 (one process per GPU).  The 100k x 50 pipeline is a 0.9 ms job on one MI355X, a chain of ~50 dependent
launches, so the two ways the path shards behave very differently and the line reports both:
  * `value` (scaling "weak", --multi ensembles, the default): every GPU runs the whole pipeline on its own 100k x 50
    ensemble -- ensembles are independent, no data-path collective; this is how a batch of embeds uses a node;
  * `sharded_single_ensemble` (scaling "strong"; --multi sharded makes it the line's value): ONE ensemble sharded over
    the ranks -- pose blocks for embed/clash, one RCCL all-gather of the surviving heavy-atom coordinates, row tiles of
    every large prune pass dealt round-robin with an all-reduce(MIN) per pass (tscode_amd/pipeline.py::sharded_step).
    It is what an ensemble too large for one GPU's patience (C4: 1M x 50) needs.

The timed region (K steps between barrier + synchronize) carries the HIP start/stop events of every pair-kernel dispatch that
`roofline.avg_launch_us` needs; the same K steps with those events off follow (`events_off`), then three steps with every
library event on for `stage_ms_per_step` and `passes[].ms` -- both outside the timed region.

Rank 0 prints ONE JSON line (see the keys below).  The CPU baseline leg (rank 0, N = 1 only) times the
oracle -- this repo's C restatement of the reference algorithm, "port" -- on a bounded sample of the same
generator; it is a reported baseline, not the thing measured.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector FP64 (datasheet; = FP32 vector 157.3 / 2), no MFMA on this path
FP32_VALU_PEAK_TFLOPS = 157.3    # MI355X vector FP32 with packed v_pk_fma_f32
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
PMC_PROFILE = "r01_final_pmc_hbm_counters.json"   # committed rocprofv3 --pmc summary the roofline's `traffic` is read from


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", type=int, default=0, help="0 = reference-exact (parity), 1 = cache-free")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--n-poses", type=int, default=None, help="override the config's N (debugging)")
    ap.add_argument("--algo", type=int, default=0, help="pair kernel: 0/2 = descriptor sieve (default), 1 = register-tiled all-pairs")
    ap.add_argument("--opt", action="append", default=[], help="library tunable name=value (seg_cols, drain_min), repeatable")
    ap.add_argument("--multi", choices=["ensembles", "sharded"], default="ensembles",
                    help="N > 1: 'ensembles' = one whole ensemble per GPU, no data-path collective (weak scaling; the line's value), "
                         "'sharded' = ONE ensemble sharded over the ranks (strong scaling; all-gather + per-pass all-reduce). "
                         "With 'ensembles' the sharded protocol is timed too and reported beside it.")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' lets several ranks share one GPU to rehearse "
                                                      "the 'ensembles' mode (the sharded protocol needs nccl = RCCL)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-rank protocol (sharded_step + torch.distributed collectives) even with one rank")
    ap.add_argument("--pass-timing", type=int, default=1,
                    help="library HIP events in the timed region: 1 = start/stop events on every pair-kernel dispatch (feeds the roofline), 0 = none")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=40000, help="poses of the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(cfg_name, n_sample, mode):
    """Oracle (CPU restatement of the reference algorithm) on a bounded sample. Only called on rank 0 at N=1."""
    import numpy as np
    import oracle
    from tscode_amd.synthetic import make_config
    ens = make_config(cfg_name, n_sample)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("TSCODE_BENCH_CPU_THREADS", "16")))   # the GPU box's CPU share per GPU
    oracle.set_num_threads(cores)
    t0 = time.perf_counter()
    poses = oracle.transform_batch(ens.frag_coords, ens.conf_idx, ens.rot, ens.pos)
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    st = poses[cm]
    heavy = np.ascontiguousarray(st[:, ens.atomnos != 1])
    res = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=False)     # chunk-parallel like the reference's prange
    dt = time.perf_counter() - t0
    evals = sum(s["pairs_evaluated"] for s in res["stats"])
    # the prune alone once more with the rows of a chunk spread over the threads too (NOT the reference's
    # parallelisation: its k = 1, 2, 5 passes run on 1, 2, 5 threads) -- what the same cores could do
    t1 = time.perf_counter()
    res_rp = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True)
    dt_rp = time.perf_counter() - t1
    assert np.array_equal(res_rp["mask"], res["mask"])
    cpu_model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    return {"value": n_sample / dt, "unit": "conformers/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "sample": f"{cfg_name} generator at N={n_sample} (embed + clash + prune mode {mode}; {int(cm.sum())} pass the clash check, "
                      f"{int(res['mask'].sum())} survive; {evals} pair evaluations; chunk-parallel like the reference's prange)",
            "seconds": dt, "pair_evals_per_s": evals / dt,
            "row_parallel_variant": {"prune_seconds": dt_rp, "pair_evals_per_s": evals / dt_rp,
                                     "note": "prune only, rows of a chunk also spread over the threads: not the reference's parallelisation"}}


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: everything libraries print there (RCCL prints a version banner at
    # communicator creation) is sent to stderr; the line itself goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    pg = None
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                       # --force-sharded without a launcher: a one-rank group
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)      # rehearsal: ranks may share a GPU
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    red_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"      # where the small timing / parity reductions live

    from tscode_amd.pipeline import DevicePipeline
    from tscode_amd.synthetic import make_config

    # every rank draws the SAME ensemble, also in the one-ensemble-per-GPU mode: per-GPU work is then exactly fixed as N grows,
    # and every rank's survivor set is checked against the recorded oracle mask (parity_vs_recorded_oracle = all ranks agree)
    ens = make_config(args.config, args.n_poses)
    sharded_mode = (world > 1 and args.multi == "sharded") or args.force_sharded
    if sharded_mode:                                        # ONE ensemble, every rank keeps only its block of the pose axis
        pipe = DevicePipeline(ens, device_index=local_rank, rank=rank, world=world, mode=args.mode, process_group=pg,
                              force_sharded=args.force_sharded)
    else:                                                   # one whole ensemble per GPU: the single-GPU pipeline on every rank
        pipe = DevicePipeline(ens, device_index=local_rank, rank=0, world=1, mode=args.mode)
    units_per_step = ens.n_poses * (1 if sharded_mode else world)     # conformers all ranks process in one step
    pipe.set_option("prune_algo", args.algo)
    pipe.set_option("pass_timing", args.pass_timing)
    for opt in args.opt:
        name, val = opt.split("=")
        pipe.set_option(name, float(val))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_loop(steps, pipe=pipe):
        """K steps bracketed by barrier + synchronize on both sides; max over ranks. Returns (seconds, last result, sums)."""
        acc = {"tile_ms": 0.0, "evals": 0, "computed": 0, "screened": 0,
               "stage_ms": {"embed_clash": 0.0, "compact": 0.0, "prune": 0.0, "total": 0.0}}
        res = None
        results = []
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = pipe.step()
            results.append(res)        # statistics are read after the timed region (the result converts them lazily)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        for r in results:
            acc["tile_ms"] += sum(s["tile_ms"] for s in r["stats"] if s["algo"] in (1, 2))     # the pair kernel's own launches
            acc["evals"] += sum(s["pairs_evaluated"] for s in r["stats"])
            acc["computed"] += sum(s["pairs_computed"] for s in r["stats"])
            acc["screened"] += sum(s["pairs_screened"] for s in r["stats"])
            for k in acc["stage_ms"]:
                acc["stage_ms"][k] += r.get("ms", {}).get(k, 0.0)
        return dt, res, acc

    res = None
    for _ in range(args.warmup):
        res = pipe.step()
    dt, res, acc = timed_loop(args.steps)           # THE timed region (library events as --pass-timing says)
    tile_ms, evals, computed, screened, stage_ms = acc["tile_ms"], acc["evals"], acc["computed"], acc["screened"], acc["stage_ms"]
    # the same K steps once more with every library event off: what a production caller sees (an event record on the
    # stream costs about 4 us on MI355X; the kernel durations of the roofline need them, the product does not)
    events_off = None
    if args.pass_timing != 0:
        pipe.set_option("pass_timing", 0)
        dt0, _, _ = timed_loop(args.steps)
        pipe.set_option("pass_timing", args.pass_timing)
        events_off = {"ms_per_step": dt0 / args.steps * 1e3, "value": units_per_step * args.steps / dt0, "unit": "conformers/s"}
    # where a step goes: three more steps with every library event on (stages of the pipeline, whole passes), outside both
    # timed regions -- these events cost about 4 us each and would distort what they measure
    pass_ms = None
    if args.pass_timing != 0:
        pipe.set_option("pass_timing", 2)
        _, res_d, acc_d = timed_loop(3)
        pipe.set_option("pass_timing", args.pass_timing)
        stage_ms = {k: v * args.steps / 3 for k, v in acc_d["stage_ms"].items()}
        pass_ms = [s["gpu_ms"] for s in res_d["stats"]]
    # verdict fingerprint (after the timed region)
    n_pass, n_keep = res["n_pass"], res["n_keep"]
    keep = pipe.h_keep[:n_pass].numpy().copy()             # the host copy every step produces
    digest = hashlib.sha256(np.packbits(keep.astype(bool)).tobytes()).hexdigest()[:16]
    expected = None
    exp_path = os.path.join(ROOT, "tests", "golden", "expected_full.json")
    if os.path.exists(exp_path):
        expected = json.load(open(exp_path)).get(f"{args.config}:{ens.n_poses}:mode{args.mode}")
    parity = None
    if expected is not None:
        parity = bool(expected["n_pass"] == n_pass and expected["n_keep"] == n_keep and expected["keep_sha256_16"] == digest)
        if world > 1:                                       # every rank checks its own result
            flag = torch.tensor([1 if parity else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            parity = bool(flag.item())

    if rank == 0:
        h = ens.n_heavy
        n = ens.n_poses
        flops_per_eval = 46 * h + 500                       # F = 46h + 500 per pair evaluation
        tile_s = tile_ms / 1e3
        big = [s for s in res["stats"] if s["algo"] in (1, 2)]   # passes run by the pair kernel (the others: chunk-local kernel)
        n_launch = len(big)                                 # pair-kernel launches per step
        launches = n_launch * args.steps
        avg_launch_s = tile_s / launches if launches and tile_s > 0 else None
        # algorithmic bytes of the prune: sum over passes (A_p * h * 24 + 2 N)  (SURVEY.md 8d), per launch: the mean
        b_k3 = sum(s["n_active_before"] * h * 24 + 2 * n_pass for s in res["stats"])
        b_big = sum(s["n_active_before"] * h * 24 + 2 * n_pass for s in big)
        b_launch = b_big / n_launch if n_launch else None
        b_k12 = n * ens.frag_coords.__len__() * 96 + n_pass * ens.n_atoms * 24 + n
        ms_per_step = dt / args.steps * 1e3
        kernel = "k_rmsd_sieve" if res["stats"] and res["stats"][0]["algo"] == 2 else "k_rmsd_tile"
        # HBM bytes per launch from the committed PMC passes of this same command (tools/profile.sh: FETCH_SIZE and
        # WRITE_SIZE in separate runs, KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if os.path.exists(pmc_path) and args.config == "C3" and args.n_poses is None and world == 1:
            pmc = json.load(open(pmc_path))
            f = next((v for k, v in pmc.get("FETCH_SIZE", {}).items() if kernel in k), None)
            w = next((v for k, v in pmc.get("WRITE_SIZE", {}).items() if kernel in k), None)
            if f and w:
                traffic = (2.0 * f["per_call_KB"] + w["per_call_KB"]) * 1024.0
                traffic_src = f"profiles/{PMC_PROFILE}: (2 x FETCH_SIZE + WRITE_SIZE) per launch, rocprofv3 --pmc, separate passes"
        hbm_achieved = b_launch / avg_launch_s / 1e9 if avg_launch_s else None
        # SURVEY.md 8(d) flop accounting beside it: the reference's own pair evaluations x (46 h + 500) flop each over the
        # kernel time.  The sieve reaches the reference's verdicts without forming H for most pairs, so this "algorithmic"
        # figure exceeds the FP64 peak; `executed` is what the instructions really do: the fp32 screen (2 families x (8 fma +
        # add + fma) = 34 flop per screened pair, packed fp32) and fp64 H + quartic tests (18 h + 110 flop per
        # pair that reaches them; register-tiled kernel: every computed pair, h padded to a multiple of 4)
        evals_big = sum(s["pairs_evaluated"] for s in big)
        screened_big, computed_big = sum(s["pairs_screened"] for s in big), sum(s["pairs_computed"] for s in big)
        achieved_alg = evals_big * flops_per_eval / (tile_s / args.steps) / 1e12 if tile_s > 0 else None
        if screened_big:
            f32_flops = screened_big * 34        # dot-product form: 2 families x (8 fma + add + fma) per pair
            f64_flops = computed_big * (18 * h + 110)
        else:
            f32_flops = 0.0
            f64_flops = computed_big * (18 * ((h + 3) // 4 * 4) + 110)
        per_s = (lambda x: x / (tile_s / args.steps) / 1e12) if tile_s > 0 else (lambda x: None)
        out = {
            "metric": "conformers/sec, 100k x 50-atom prune_conformers pipeline (embed -> clash mask -> RMSD prune)",
            "value": units_per_step * args.steps / dt,
            "unit": "conformers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if sharded_mode else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {n} conformers x {ens.n_atoms} atoms ({h} heavy), 2 rigid fragments, "
                                   f"10 children per parent, seed {ens.seed}; clash_thresh 1.5, max_clashes 0, rmsd_thr 0.5, "
                                   f"mode {args.mode} ({'reference-exact' if args.mode == 0 else 'cache-free'})",
                       "n_pass_clash": n_pass, "n_survivors": n_keep, "keep_sha256_16": digest,
                       "parity_vs_recorded_oracle": parity,
                       "parallelism": (f"one ensemble sharded over {world} rank(s): pose blocks, all-gather, per-pass all-reduce" if sharded_mode else
                                       f"{world} x (one whole ensemble per GPU), no data-path collective"),
                       "conformers_per_step_all_ranks": units_per_step,
                       "library_events_in_timed_region": args.pass_timing},
            "roofline": {
                "kernel": kernel + " (all-pairs Kabsch RMSD of one pass; one launch per pass)",
                "bound": "hbm",
                "achieved": hbm_achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (hbm_achieved / HBM_PEAK_GBS) if hbm_achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": b_launch,
                "avg_launch_us": avg_launch_s * 1e6 if avg_launch_s else None,
                "launches_per_step": n_launch,
                "kernel_ms_per_step": tile_ms / args.steps,
                "timing": "HIP start/stop events attached to every dispatch of the kernel (hipExtLaunchKernel) inside the timed region",
                "passes_in_chunk_local_kernel": len(res["stats"]) - n_launch,
                "fp64_valu": {"algorithmic_pair_evals_per_step": evals_big, "flops_per_eval": flops_per_eval,
                              "achieved_algorithmic_TFLOPs": achieved_alg, "peak_TFLOPs": FP64_VALU_PEAK_TFLOPS,
                              "frac_algorithmic": (achieved_alg / FP64_VALU_PEAK_TFLOPS) if achieved_alg else None,
                              "pairs_screened_per_step": screened_big, "pairs_with_H_formed_per_step": computed_big,
                              "executed_fp32_TFLOPs": per_s(f32_flops), "executed_fp64_TFLOPs": per_s(f64_flops),
                              "peak_fp32_TFLOPs": FP32_VALU_PEAK_TFLOPS},
            },
            "pipeline_hbm": {"algorithmic_bytes": (b_k12 + b_k3) * (units_per_step // n),
                             "achieved_GBs": (b_k12 + b_k3) * (units_per_step // n) / (ms_per_step / 1e3) / 1e9,
                             "peak_GBs": HBM_PEAK_GBS * world,
                             "frac": (b_k12 + b_k3) * (units_per_step // n) / (ms_per_step / 1e3) / 1e9 / (HBM_PEAK_GBS * world)},
            "events_off": events_off,
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "passes": [{"k": s["k"], "active": s["n_active_before"], "evals": s["pairs_evaluated"], "screened": s["pairs_screened"], "H_formed": s["pairs_computed"],
                        "exact": s["candidates"], "ms": round(pass_ms[i] if pass_ms and i < len(pass_ms) else s["gpu_ms"], 4),
                        "tile_ms": round(s["tile_ms"], 4),
                        "kernel": {1: "k_rmsd_tile", 2: "k_rmsd_sieve", 3: "k_pass_chunks"}.get(s["algo"], "?")} for i, s in enumerate(res["stats"])],
            "detail_note": "stage_ms_per_step and passes[].ms come from 3 extra steps with every library event on (pass_timing 2), "
                           "outside the timed regions; passes[].tile_ms (the pair kernel's own events) from the timed region",
        }
    else:
        out = None
    # N > 1 in 'ensembles' mode: the sharded single-ensemble protocol (strong scaling) timed beside it, same K steps.
    # The line's value is already measured; a watchdog makes sure it is printed even if a collective of this extra
    # leg should hang on some node (this leg runs under RCCL with N > 1 only on the driver's hardware).
    sharded_beside = None
    if world > 1 and not sharded_mode and args.backend == "nccl":
        import threading

        def bail():
            if rank == 0:
                out["sharded_single_ensemble"] = {"error": "timed out after 120 s"}
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)
        watchdog = threading.Timer(120.0, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            spipe = DevicePipeline(ens, device_index=local_rank, rank=rank, world=world, mode=args.mode, process_group=pg)
            for _ in range(args.warmup):
                spipe.step()
            dts, sres, _ = timed_loop(args.steps, spipe)
            skeep = spipe.h_keep[:sres["n_pass"]].numpy().copy()
            sharded_beside = {"ms_per_step": dts / args.steps * 1e3, "value": ens.n_poses * args.steps / dts, "unit": "conformers/s",
                              "scaling": "strong", "n_survivors": int(sres["n_keep"]),
                              "keep_sha256_16": hashlib.sha256(np.packbits(skeep.astype(bool)).tobytes()).hexdigest()[:16],
                              "what": "ONE ensemble sharded over the ranks: pose blocks, one RCCL all-gather of the surviving heavy-atom "
                                      "shards, all-reduce(MIN) per large pass"}
        except Exception as exc:                             # the line's value above is already measured
            sharded_beside = {"error": f"{type(exc).__name__}: {exc}"}
        watchdog.cancel()
    if rank == 0:
        out["sharded_single_ensemble"] = sharded_beside
    if rank == 0:
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.config, min(args.cpu_sample, ens.n_poses), args.mode)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
