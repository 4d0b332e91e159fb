#!/usr/bin/env python
"""bench.py -- conformers/sec of the geometry hot path on the BASELINE config 3 workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode 0|1] [--config C3|C4|C5] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` from a bare shell starts the N ranks itself, as child processes under torch.distributed.run.)

A "step" is one pass of the hot path over one synthetic ensemble that is already resident in HBM:
fused embed+clash verdicts -> ordered compaction of the passing poses -> prune_conformers_rmsd
(reference-exact mode), verdict masks left on the device and copied to pinned host memory.

N = 1: the one-call pipeline (tsc_pipeline_dev).

N > 1 (one process per GPU, torch.distributed backend nccl = RCCL): for the 100k-conformer workloads (C2, C3: the metric's) the line's
`value` is `replicas` -- every GPU its own ensemble, `"scaling": "weak"`, no data-path collective: such an ensemble is a one-GPU job and does
not strong-scale (`multi.why`) -- with the sharded leg beside it (`sharded_single_ensemble`); for C4 / C5 / the C5 chain (and with
`--multi sharded`) the line's `value` is ONE ensemble sharded over the
ranks, `"scaling": "strong"` -- every prune pass with at least 2 chunks per rank PARTITIONED BY CHUNKS (a rank runs the whole pass
on the chunks that start inside its block of the structure axis; one bit per structure + statistics summed per pass,
`partitioned_passes`), the later passes with their row tiles dealt round-robin and an all-reduce(MIN) over best[]
(`sharded_passes`), and in front of it one of three forms of the embed / clash half: pose blocks + one RCCL all-gather of the
surviving heavy-atom shards (`"front": "shard"`), every rank embedding and clash-filtering all poses itself (`"replicate"`), or
clash verdicts per pose block, one byte per pose summed over the ranks, and every rank embedding the heavy atoms of all passing
poses itself (`"hybrid"`): the three are timed on the node before the warm-up and the fastest runs (`front_tuning`; `--front`
forces one) (tscode_amd/pipeline.py::sharded_step); `rccl_world`, `allgather_bytes_per_step` and `allreduce_bytes_per_step` say
what crossed xGMI.  `replicas` beside it (`"scaling": "weak"`): every GPU runs the whole pipeline on its own ensemble, no
data-path collective -- how a batch of independent embeds uses a node.  The 100k x 50 pipeline is a 0.9 ms chain of dependent
launches of which about 0.5 ms shards (the Amdahl ceiling of C3 is stated there); `--config C4` (1M x 50)
is the workload on which sharding one ensemble pays.  Should the sharded leg fail or hang (180 s watchdog), the line falls
back to the replicas figure, says so in `error`, and the process exits non-zero -- also when the sharded leg was only the side figure.
At N > 1 BOTH figures stand at top level whichever of them `value` is: `value_strong` / `ms_per_step_strong` (one ensemble sharded, with
`rccl_world`, `collectives_per_step`, `exchange`) and `value_weak` / `ms_per_step_weak` (replicas).  Before any leg the collectives meet the
step's real buffers once (`selftest_collectives`: all-reduce SUM of int64, all-reduce MIN of int32, all-gather, each checked, timed and under
a 30 s watchdog that names the collective that hung; then the library's own exchange over memory the ranks map into each other,
`--exchange`: used for the per-pass messages when it connects and agrees).

The timed region (K steps between barrier + synchronize) carries the HIP start/stop events of every pair-kernel dispatch that
`roofline.avg_launch_us` needs; the same K steps with those events off follow (`events_off`), then three steps with every
library event on for `stage_ms_per_step` and `passes[].ms` -- both outside the timed region.

Rank 0 prints ONE JSON line.  The CPU baseline leg (rank 0, N = 1 only) times the oracle -- this repo's C restatement of the
reference algorithm, "port" -- on the SAME workload (full C3: about a minute on 16 threads; other configs on a bounded
sample, extrapolation stated); it is a reported baseline, not the thing measured.
"""

from __future__ import annotations

import argparse
import gc
import glob
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector FP64 (datasheet; = FP32 vector 157.3 / 2), no MFMA on this path
FP32_VALU_PEAK_TFLOPS = 157.3    # MI355X vector FP32 with packed v_pk_fma_f32
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
CHAIN_CANDIDATES = 20000         # --config C5chain: angle sets the conformational search rotates per step
EVENT_EVERY = 4                  # the timed region carries the pair kernel's HIP events in every 4th step (timed_loop)
STEP_TIMES = bool(os.environ.get("BENCH_STEP_TIMES"))
if STEP_TIMES:      # debugging aid: every garbage collection with its generation, duration and yield
    _gc_t = [0.0]

    def _gc_note(phase, info):
        if phase == "start":
            _gc_t[0] = time.perf_counter()
        else:
            print(f"gc gen {info['generation']}: {(time.perf_counter() - _gc_t[0]) * 1e3:.2f} ms, collected {info['collected']}", file=sys.stderr)
    gc.callbacks.append(_gc_note)
SQ_PROFILES = {"C3": ("r05_sq_counters_c3.json",), "C4": ("r05_sq_counters_c4.json",)}   # tools/pmc_sq.sh summaries (profiles/): VALU issue of the pair kernels
GPU_CLOCK_GHZ = 2.4              # MI355X engine clock: CU-cycles of a launch = duration x clock x 256 CUs
SCREEN_FLOP_PER_PAIR = 36        # the screen as executed: 18 v_pk_fma_f32 per (row, 128 columns) = 2 families x 9 FMA per pair (8 of the dot product
                                 # + 1 that turns it into the squared distance); DESIGN.md section 4 counts the same
PMC_PROFILES = ("r05_pmc_hbm_counters.json", "r04_pmc_hbm_counters.json", "r03_pmc_hbm_counters.json", "r02_pmc_hbm_counters.json", "r01_final_pmc_hbm_counters.json")   # newest first (profiles/)
PMC_PROFILES_C4 = ("r05_pmc_hbm_counters_C4.json", "r04_pmc_hbm_counters_C4.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", type=int, default=0, help="0 = reference-exact (parity), 1 = cache-free")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--n-poses", type=int, default=None, help="override the config's N (debugging)")
    ap.add_argument("--algo", type=int, default=0, help="pair kernel: 0/2 = descriptor sieve (default), 1 = register-tiled all-pairs")
    ap.add_argument("--opt", action="append", default=[], help="library tunable name=value (seg_cols, drain_min), repeatable")
    ap.add_argument("--multi", choices=["auto", "sharded", "replicas"], default="auto",
                    help="N > 1: what the line's value is.  'auto' (default) = 'replicas' for the 100k-conformer workloads (C2, C3: a one-GPU job of "
                         "under a millisecond, of which only half divides by N -- a node runs N of them), 'sharded' for C4 / C5 / the C5 chain; "
                         "'sharded' = ONE ensemble sharded over the ranks under RCCL "
                         "(strong scaling; all-gather + per-pass all-reduce), 'replicas' = one whole ensemble per GPU, no data-path "
                         "collective (weak scaling).  The other one is timed too and reported beside it.")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' lets several ranks share one GPU to rehearse "
                                                      "(device tensors then travel over the host)")
    ap.add_argument("--front", choices=("auto", "shard", "replicate", "hybrid"), default="auto",
                    help="multi-rank front half: pose blocks + all-gather of the survivors' coordinates, every rank computing all "
                         "poses itself, clash verdicts per block + every rank embedding all survivors (hybrid), or whichever is fastest on "
                         "this node (timed before the warm-up)")
    ap.add_argument("--exchange", choices=("auto", "ipc", "callback"), default="auto",
                    help="N > 1: the per-pass exchanges of the sharded prune -- 'callback' = torch.distributed all-reduces (RCCL) the library calls back "
                         "for, 'ipc' = the library's own exchange over areas the ranks map into each other (hipIpc*; the process group only carries "
                         "the handles), 'auto' (default) = ipc if the collective self-test finds it connected and correct on this node, else callback")
    ap.add_argument("--no-selftest", action="store_true",
                    help="N > 1: skip the collective self-test (one all-reduce SUM of int64, one all-reduce MIN of int32 and one all-gather on the "
                         "step's real buffers, each under a 30 s watchdog, before any leg)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-rank protocol (sharded_step + torch.distributed collectives) even with one rank")
    ap.add_argument("--pass-timing", type=int, default=1,
                    help="library HIP events in the timed region: 1 = start/stop events on every pair-kernel dispatch (feeds the roofline), 0 = none")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-sample", type=int, default=None, help="poses of the CPU baseline run (default: the whole workload for C3, 40000 otherwise)")
    ap.add_argument("--no-side-leg", action="store_true", help="N > 1: time only the leg that is the line's value")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="N = 1 side figure `steps_in_flight`: this many independent steps at a time, each on its own library context, "
                         "HIP stream and buffers, driven by one host thread each (0 = skip).  Never the line's value.")
    return ap.parse_args()


def csrc_digest():
    """SHA-256 (16 hex digits) over the HIP sources: ties a profile under profiles/ to the kernels it was taken from."""
    from tscode_amd.build import csrc_digest as d
    return d()


def binary_digest():
    """The digest of the sources the LOADED libtscode_hip.so was built from (baked in by tscode_amd/build.py)."""
    from tscode_amd import _lib
    return _lib.load().tsc_build_digest().decode()


def cpu_baseline(cfg_name, n_sample, n_full, mode):
    """Oracle (CPU restatement of the reference algorithm). Only called on rank 0 at N=1."""
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
    whole = n_sample == n_full
    digest = hashlib.sha256(np.packbits(res["mask"].astype(bool)).tobytes()).hexdigest()[:16]
    out = {"value": n_sample / dt, "unit": "conformers/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
           "sample": (f"the whole workload: {cfg_name} at N={n_sample}" if whole else f"{cfg_name} generator at N={n_sample} of {n_full}")
                     + f" (embed + clash + prune mode {mode}; {int(cm.sum())} pass the clash check, {int(res['mask'].sum())} survive; "
                       f"{evals} pair evaluations; chunk-parallel like the reference's prange)",
           "same_workload_as_value": whole, "seconds": dt, "pair_evals_per_s": evals / dt, "n_pass_clash": int(cm.sum()),
           "n_survivors": int(res["mask"].sum()), "keep_sha256_16": digest,
           "row_parallel_variant": {"prune_seconds": dt_rp, "pair_evals_per_s": evals / dt_rp,
                                    "note": "prune only, rows of a chunk also spread over the threads: not the reference's parallelisation"}}
    if not whole:
        out["note"] = ("a sample, not the workload: the prune's work grows faster than N (pair evaluations ~ N^2 / k in the late passes), so "
                       "conformers/s on the sample overstates what the CPU would reach on the whole workload; compare pair_evals_per_s")
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell (no launcher's RANK / WORLD_SIZE in the environment): start the N ranks as CHILD
    processes under torch.distributed.run and leave with their status.  Decided before this process imports torch or touches the GPU
    (a process that has initialised the GPU must not be replaced by, or turn into, the launcher); the children inherit stdout, so
    rank 0's one JSON line is this command's one JSON line."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL and tensor sharing across processes need on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))
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
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): pass the same N to both (or run `python bench.py --gpus N` "
                         "from a bare shell: it starts its own ranks)")
    pg = None
    use_dist = world > 1 or args.force_sharded
    if use_dist:
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

    # every rank draws the SAME ensemble, also for the replicas leg: per-GPU work is then exactly fixed as N grows, and every
    # rank's survivor set is checked against the recorded oracle mask (parity_vs_recorded_oracle = all ranks agree)
    chain = args.config == "C5chain"          # config 5 as a chain: csearch rotations feed the pipeline's conformers (N > 1: the search
                                              # cut into blocks of the angle table, the kept candidates all-gathered in table order)
    ens = make_config("C5" if chain else args.config, args.n_poses)
    expected = None
    exp_path = os.path.join(ROOT, "tests", "golden", "expected_full.json")
    if os.path.exists(exp_path):
        expected = json.load(open(exp_path)).get(f"{args.config}:{ens.n_poses}:mode{args.mode}")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def collectives_selftest():
        """First contact of the node's collectives with the step's REAL buffers, before any leg: all-reduce(SUM) of int64 on the exchange
        buffer of the partitioned passes, all-reduce(MIN) of int32 on best[], all-gather on the heavy-atom send / receive buffers -- each
        checked against what every rank can compute for itself and timed (5 calls between stream events), under a 30 s watchdog that
        prints a line NAMING the collective that did not return and ends the process (exit code 4).  Then the same two reductions through
        the library's own exchange (tsc_xchg_*): connected?  equal to what the process group's gave?  Returns the record for the line."""
        from tscode_amd._lib import XCHG_MIN_I32, XCHG_SUM_I64
        from tscode_amd.pipeline import HipShardBackend
        rec = {"watchdog_s": 30, "backend": dist.get_backend()}
        now = {"what": "setup"}

        def hung():
            if rank == 0:
                line = {"metric": "conformers/sec, 100k x 50-atom prune_conformers pipeline (embed -> clash mask -> RMSD prune)", "value": None,
                        "unit": "conformers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                        "config": {"workload": f"{args.config}: {ens.n_poses} conformers x {ens.n_atoms} atoms"},
                        "error": f"collective self-test: `{now['what']}` did not return within 30 s on {world} ranks ({dist.get_backend()})",
                        "selftest_collectives": rec}
                os.write(real_stdout, (json.dumps(line) + "\n").encode())
            os._exit(4)

        def guarded(name, fn):
            now["what"] = name
            t = threading.Timer(30.0, hung)
            t.daemon = True
            t.start()
            try:
                return fn()
            finally:
                t.cancel()
        be = guarded("buffers of the sharded backend", lambda: HipShardBackend(ens, local_rank, rank, world, 1.5, 0, 0.5, args.mode))
        n = ens.n_poses
        words = n // 64 + 48
        staged = args.backend != "nccl"

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(be.stream):
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps

        def ar(t, op):
            with torch.cuda.stream(be.stream):
                if staged:
                    c = t.cpu()
                    dist.all_reduce(c, op=op)
                    t.copy_(c)
                else:
                    dist.all_reduce(t, op=op)
        idx64 = torch.arange(words, dtype=torch.int64, device=be.dev)
        idx32 = torch.arange(n, dtype=torch.int32, device=be.dev)

        def fill():                                   # rank-dependent patterns every rank can predict for all ranks
            with torch.cuda.stream(be.stream):
                be.exch[:words].copy_(idx64 * (rank + 1) + 7)
                be.best[:n].copy_((idx32 * 31 + rank * 1009) % 65521 - rank)
        want64 = idx64 * (world * (world + 1) // 2) + 7 * world
        want32 = torch.stack([(idx32 * 31 + r * 1009) % 65521 - r for r in range(world)]).amin(0)

        def check_sum():
            fill()
            ar(be.exch[:words], dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            return bool(torch.equal(be.exch[:words], want64))

        def check_min():
            fill()
            ar(be.best[:n], dist.ReduceOp.MIN)
            torch.cuda.synchronize()
            return bool(torch.equal(be.best[:n], want32))
        ok = guarded("all-reduce SUM of int64 (removed-row bits + statistics)", check_sum)
        rec["allreduce_sum_i64"] = {"count": words, "ok": ok,
                                    "us_per_call": guarded("all-reduce SUM of int64, timed", lambda: timed(lambda: ar(be.exch[:words], dist.ReduceOp.SUM)))}
        ok = guarded("all-reduce MIN of int32 (best[])", check_min)
        rec["allreduce_min_i32"] = {"count": n, "ok": ok,
                                    "us_per_call": guarded("all-reduce MIN of int32, timed", lambda: timed(lambda: ar(be.best[:n], dist.ReduceOp.MIN)))}
        rows = min(be.max_local, 4096)
        row_elems = be.heavy_pad[0].numel()

        def ag():
            with torch.cuda.stream(be.stream):
                out_t, in_t = be.gather.view(-1)[:world * rows * row_elems], be.heavy_pad.view(-1)[:rows * row_elems]
                if staged:
                    co, ci = out_t.cpu(), in_t.cpu()
                    dist.all_gather_into_tensor(co, ci)
                    out_t.copy_(co)
                else:
                    dist.all_gather_into_tensor(out_t, in_t)

        def check_gather():
            with torch.cuda.stream(be.stream):
                be.heavy_pad.view(-1)[:rows * row_elems].fill_(float(rank + 1))
            ag()
            torch.cuda.synchronize()
            got = be.gather.view(-1)[:world * rows * row_elems].view(world, -1)
            return bool(all(float(got[r].min()) == float(got[r].max()) == float(r + 1) for r in range(world)))
        ok = guarded("all-gather of float64 (heavy-atom shards)", check_gather)
        rec["allgather_f64"] = {"bytes_per_rank": rows * row_elems * 8, "ok": ok, "us_per_call": guarded("all-gather, timed", lambda: timed(ag))}
        # the library's own exchange: may fail to connect (no peer access, IPC refused) -- that is a result, not an error
        ipc = {"connected": False}
        try:
            guarded("tsc_xchg connect (hipIpcGetMemHandle / hipIpcOpenMemHandle + handles over the process group)", lambda: be.connect_exchange(dist, None))
            ipc["connected"] = be.xchg is not None
        except Exception as exc:   # noqa: BLE001
            ipc["error"] = f"{type(exc).__name__}: {exc}"
        flag = torch.tensor([1 if ipc["connected"] else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # one rank without it: nobody uses it
        if ipc["connected"] and not bool(flag.item()):
            ipc["connected"], ipc["error"] = False, "another rank could not connect"
            be.disconnect_exchange()
        if ipc["connected"]:
            be.xchg.set_timeout(20.0)

            def x_sum():
                with torch.cuda.stream(be.stream):
                    be.xchg.allreduce(XCHG_SUM_I64, be.exch[:words], words)

            def x_min():
                with torch.cuda.stream(be.stream):
                    be.xchg.allreduce(XCHG_MIN_I32, be.best[:n], n)

            def check_x():
                fill()
                x_sum()
                x_min()
                torch.cuda.synchronize()
                return bool(torch.equal(be.exch[:words], want64)), bool(torch.equal(be.best[:n], want32))
            ipc["sum_ok"], ipc["min_ok"] = guarded("tsc_xchg_allreduce (SUM int64 + MIN int32)", check_x)
            ipc["sum_us_per_call"] = guarded("tsc_xchg_allreduce SUM, timed", lambda: timed(x_sum))
            ipc["min_us_per_call"] = guarded("tsc_xchg_allreduce MIN, timed", lambda: timed(x_min))
            ipc["timeouts"] = be.xchg.status()[1]
            good = torch.tensor([1 if (ipc["sum_ok"] and ipc["min_ok"] and ipc["timeouts"] == 0) else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            ipc["usable"] = bool(good.item())
            be.disconnect_exchange()
        rec["ipc_exchange"] = ipc
        rec["all_ok"] = bool(rec["allreduce_sum_i64"]["ok"] and rec["allreduce_min_i32"]["ok"] and rec["allgather_f64"]["ok"])
        del be
        gc.collect()
        torch.cuda.synchronize()
        dist.barrier()
        return rec

    selftest = None
    if world > 1 and not args.no_selftest:
        selftest = collectives_selftest()
    if args.exchange == "auto":
        use_ipc = bool(selftest and selftest.get("ipc_exchange", {}).get("usable"))
    else:
        use_ipc = args.exchange == "ipc" and world > 1
    exchange_form = "ipc" if use_ipc else "callback"
    notes = []

    def configure(pipe, pass_timing):
        pipe.set_option("prune_algo", args.algo)
        pipe.set_option("pass_timing", pass_timing)
        for opt in args.opt:
            name, val = opt.split("=")
            pipe.set_option(name, float(val))

    def timed_loop(pipe, steps, timing_level=0):
        """K steps bracketed by barrier + synchronize on both sides; max over ranks. Returns (seconds, last result, sums)."""
        acc = {"tile_ms": 0.0, "evals": 0, "computed": 0, "screened": 0, "evented_steps": 0,
               "stage_ms": {"embed_clash": 0.0, "compact": 0.0, "prune": 0.0, "total": 0.0}}
        res = None
        results = []
        sync()
        # (a generation-2 collection of this process takes about 40 ms -- torch's and numpy's module graphs -- and was seen to land inside
        # 10-step loops of 1 to 6 ms steps, multiplying their figure by up to five.  Everything alive now goes to the permanent generation:
        # collections inside the loop then look at the loop's own objects only.  Switching the collector off instead cost the C3 step 30 us)
        # (the collection itself is made before the warm-up steps, in run_leg: 40 ms of idle GPU right in front of the loop cost its first steps
        # their clocks)
        if not os.environ.get("BENCH_KEEP_GC"):
            gc.freeze()
        t0 = time.perf_counter()
        marks = []
        for i in range(steps):
            # the pair kernel's start / stop events ride on its dispatches in every EVENT_EVERY-th step of the timed region only: a pair
            # of events costs a launch about 6 us, eight launches per step were 6 - 9 % of a C3 step -- measuring the kernel was slowing
            # down what `value` reports.  The average launch duration comes from the steps that carry them (3 - 5 of the default 20).
            if timing_level == 1 and EVENT_EVERY > 1:
                on = (steps - 1 - i) % EVENT_EVERY == 0       # (the last step among them: its per-pass figures are printed)
                pipe.set_option("pass_timing", 1 if on else 0)
                acc["evented_steps"] += 1 if on else 0
            elif timing_level == 1:
                acc["evented_steps"] += 1
            res = pipe.step()
            results.append(res)        # statistics are read after the timed region (the result converts them lazily)
            if STEP_TIMES:
                marks.append(time.perf_counter())
        sync()
        dt = time.perf_counter() - t0
        if STEP_TIMES and rank == 0:   # (debugging aid: when each step's call returned, ms since the loop began)
            print("step returns (ms):", " ".join(f"{(m - t0) * 1e3:.2f}" for m in marks), "| loop", f"{dt * 1e3:.2f}", file=sys.stderr)
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

    def verdict(pipe, res, expected=expected):
        """(digest of the packed survivor mask, parity with the recorded oracle run on every rank)."""
        keep = pipe.h_keep[:res["n_pass"]].numpy().copy()             # the host copy every step produces
        digest = hashlib.sha256(np.packbits(keep.astype(bool)).tobytes()).hexdigest()[:16]
        parity = None
        if expected is not None:
            parity = bool(expected["n_pass"] == res["n_pass"] and expected["n_keep"] == res["n_keep"] and expected["keep_sha256_16"] == digest
                          and expected.get("n_conformers", res.get("n_conformers")) == res.get("n_conformers"))
            if world > 1:                                       # every rank checks its own result
                flag = torch.tensor([1 if parity else 0], dtype=torch.int32, device=red_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                parity = bool(flag.item())
        return digest, parity

    def run_leg(sharded, ens=ens, expected=expected, detail=True):
        """One leg: its own pipeline, warm-up, THE timed region, then (detail) the same steps with the library's events off and
        three steps with all of them on.  Returns a dict of everything measured."""
        if chain:
            from tscode_amd.pipeline import CsearchChain, ShardedCsearchChain
            n0 = ens.frag_coords[0].shape[1]
            torsions, tmasks = CsearchChain.chain_torsions(n0, 8, seed=5)
            angle_table = np.random.default_rng(6).choice(np.array([0, 0, 60, 120, 180, 240, 300, 25]), size=(CHAIN_CANDIDATES, 8)).astype(np.int32)
            if sharded:
                pipe = ShardedCsearchChain(ens, torsions, tmasks, angle_table, rank, world, process_group=pg, thresh=1.4, device_index=local_rank,
                                           mode=args.mode, seed=7, front=args.front, exchange=exchange_form)
            else:
                pipe = CsearchChain(ens, torsions, tmasks, angle_table, thresh=1.4, device_index=local_rank, mode=args.mode, seed=7)
        elif sharded:
            pipe = DevicePipeline(ens, device_index=local_rank, rank=rank, world=world, mode=args.mode, process_group=pg,
                                  force_sharded=args.force_sharded or world == 1, front=args.front, exchange=exchange_form)
        else:
            pipe = DevicePipeline(ens, device_index=local_rank, rank=0, world=1, mode=args.mode)
        configure(pipe, args.pass_timing)
        xc = getattr(getattr(pipe, "backend", None), "xchg", None)
        if xc is not None:
            xc.set_timeout(2.0)      # (a peer that never delivers ends this leg within seconds: the repeat with the callback form follows)
        if sharded and world > 1 and pipe.front == "auto":
            pipe.tune_front()                                  # both forms of the front half timed on this node, before the warm-up
        if not os.environ.get("BENCH_KEEP_GC"):
            gc.collect()                                       # (see timed_loop)
            gc.freeze()
        for _ in range(args.warmup):
            pipe.step()
        dt, res, acc = timed_loop(pipe, args.steps, args.pass_timing)   # THE timed region (library events as --pass-timing says, sampled)
        pipe.set_option("pass_timing", args.pass_timing)
        leg = {"dt": dt, "res": res, "acc": acc, "pipe": pipe, "sharded": sharded, "events_off": None, "pass_ms": None, "stage_ms": acc["stage_ms"]}
        units = ens.n_poses * (1 if sharded else world)
        leg["units_per_step"] = units
        if args.pass_timing != 0 and detail:
            # the same K steps once more with every library event off: what a production caller sees (an event record on the
            # stream costs about 4 us on MI355X; the kernel durations of the roofline need them, the product does not)
            pipe.set_option("pass_timing", 0)
            dt0, _, _ = timed_loop(pipe, args.steps)
            leg["events_off"] = {"ms_per_step": dt0 / args.steps * 1e3, "value": units * args.steps / dt0, "unit": "conformers/s"}
            # where a step goes: three more steps with every library event on (stages of the pipeline, whole passes), outside
            # both timed regions -- these events cost about 4 us each and would distort what they measure
            pipe.set_option("pass_timing", 2)
            _, res_d, acc_d = timed_loop(pipe, 3)
            pipe.set_option("pass_timing", args.pass_timing)
            leg["stage_ms"] = {k: v * args.steps / 3 for k, v in acc_d["stage_ms"].items()}
            leg["pass_ms"] = [s["gpu_ms"] for s in res_d["stats"]]
        leg["digest"], leg["parity"] = verdict(pipe, res, expected)
        return leg

    def leg_summary(leg, what):
        res = leg["res"]
        out = {"ms_per_step": leg["dt"] / args.steps * 1e3, "value": leg["units_per_step"] * args.steps / leg["dt"], "unit": "conformers/s",
               "scaling": "strong" if leg["sharded"] else "weak", "conformers_per_step_all_ranks": leg["units_per_step"],
               "n_survivors": int(res["n_keep"]), "keep_sha256_16": leg["digest"], "parity_vs_recorded_oracle": leg["parity"],
               "events_off": leg["events_off"], "what": what}
        if leg["sharded"]:
            out["rccl_world"] = dist.get_world_size() if use_dist else 1
            out["collectives_per_step"] = collectives_of(res)
            out["exchange"] = res.get("exchange")
            out["front"] = res.get("front")
        return out

    def collectives_of(res):
        """Exchanges of one sharded step: counts all-reduce + coordinates all-gather (front = shard) or the clash mask (front = hybrid), one
        all-reduce per pass that is partitioned by chunks or dealt by row tiles, one for the cache views between the two kinds."""
        return ({"shard": 2, "hybrid": 1}.get(res.get("front"), 0) + len(res.get("exchanges", [])) + len(res.get("partitioned", []))
                + (1 if res.get("partitioned") and args.mode == 0 else 0))

    SHARDED_WHAT = ("ONE ensemble sharded over the ranks: passes with >= 2 chunks per rank partitioned by chunks (all-reduce SUM of one bit "
                    "per structure + statistics), the later ones by row tiles (all-reduce MIN over best[]); the front half as `front` says -- "
                    "'shard': pose blocks + one RCCL all-gather of the surviving heavy-atom shards, 'replicate': every rank embeds and "
                    "clash-filters all poses itself, 'hybrid': clash verdicts per block (one byte per pose summed over the ranks), all "
                    "survivors' heavy atoms embedded on every rank -- whichever `front_tuning` measured fastest on this node")
    REPLICAS_WHAT = "one whole ensemble per GPU (the same ensemble on every rank), no data-path collective"

    # what the line's value is at N > 1 (docstring): a 100k-conformer ensemble is a one-GPU job -- its 0.8 ms step is a chain of about
    # 30 dependent launches of which half divides by N, and every pass costs a collective (predicted 0.83 - 0.91 x
    # at 2 - 8 ranks) -- so a node runs N of them (`replicas`, weak scaling) and the sharded leg is reported beside it; the 1M-conformer
    # and 200-atom workloads are sharded (strong scaling), replicas beside them
    multi = args.multi if args.multi != "auto" else ("replicas" if args.config in ("C2", "C3") else "sharded")
    multi_reason = ("asked for (--multi)" if args.multi != "auto" else
                    ("automatic: a 100k-conformer ensemble is a one-GPU job (its step does not strong-scale, DESIGN.md 6); a node runs one per GPU"
                     if multi == "replicas" else "automatic: a workload this size is sharded over the ranks"))
    main_sharded = (world > 1 and multi == "sharded") or args.force_sharded
    side = None
    error = None
    exit_code = 0
    if world > 1:
        # the replicas leg first (it cannot hang on a collective), then the sharded leg under a watchdog: if RCCL should hang
        # on some node, a line is still printed -- with the replicas figure as its value, the failure named (exit code 3 when the
        # sharded leg was to be the line's value)
        replicas = run_leg(False) if (not main_sharded or not args.no_side_leg) else None
        sharded_leg = None
        if main_sharded or not args.no_side_leg:
            def bail():
                if rank == 0:
                    line = {"metric": "conformers/sec, 100k x 50-atom prune_conformers pipeline (embed -> clash mask -> RMSD prune)",
                            "value": None, "unit": "conformers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                            "config": {"workload": f"{args.config}: {ens.n_poses} conformers x {ens.n_atoms} atoms"},
                            "error": "the sharded single-ensemble leg did not finish within 180 s (collective hung?); value = replicas leg"}
                    if replicas is not None:
                        s = leg_summary(replicas, REPLICAS_WHAT)
                        line.update(value=s["value"], ms_per_step=s["ms_per_step"], replicas=s)
                    os.write(real_stdout, (json.dumps(line) + "\n").encode())
                os._exit(3)      # (also when the sharded leg was only the side figure: a hung collective is never a clean run)
            watchdog = threading.Timer(180.0, bail)
            watchdog.daemon = True
            watchdog.start()
            try:
                try:
                    sharded_leg = run_leg(True)
                except Exception as exc_ipc:
                    # the library's own exchange is the newest part of the path and has met more than one GPU nowhere yet: should it fail in a
                    # real step after passing the self-test, the leg runs once more with the process group's collectives before anything is
                    # called a failure (the first attempt's error stays in the line)
                    if exchange_form != "ipc":
                        raise
                    notes.append(f"sharded leg with exchange = ipc failed ({type(exc_ipc).__name__}: {exc_ipc}); repeated with exchange = callback")
                    exchange_form = "callback"
                    torch.cuda.synchronize()
                    sharded_leg = run_leg(True)
            except Exception as exc:
                error = f"sharded single-ensemble leg failed: {type(exc).__name__}: {exc}" + ("; value = replicas leg" if main_sharded else "")
                exit_code = 3        # (also when the sharded leg was only the side figure: the RCCL path gates every N > 1 run)
                if main_sharded and replicas is None:
                    raise
            watchdog.cancel()
        if main_sharded and sharded_leg is not None:
            leg = sharded_leg
            if replicas is not None:
                side = ("replicas", leg_summary(replicas, REPLICAS_WHAT))
        else:
            leg = replicas
            if sharded_leg is not None:
                side = ("sharded_single_ensemble", leg_summary(sharded_leg, SHARDED_WHAT))
    else:
        leg = run_leg(main_sharded)

    if selftest is not None and not selftest["all_ok"]:
        error = ((error + "; ") if error else "") + "collective self-test: a collective returned wrong data on the step's buffers (selftest_collectives)"
        exit_code = exit_code or 4
    res, acc, dt = leg["res"], leg["acc"], leg["dt"]
    tile_ms, evals, computed, screened = acc["tile_ms"], acc["evals"], acc["computed"], acc["screened"]
    stage_ms, pass_ms = leg["stage_ms"], leg["pass_ms"]
    n_pass, n_keep = res["n_pass"], res["n_keep"]
    sharded_mode = leg["sharded"]
    units_per_step = leg["units_per_step"]

    out = None
    if rank == 0:
        h = ens.n_heavy
        n = ens.n_poses
        flops_per_eval = 46 * h + 500                       # F = 46h + 500 per pair evaluation
        tile_s = tile_ms / 1e3
        ev_steps = acc.get("evented_steps") or args.steps    # steps of the timed region whose pair-kernel dispatches carried events
        big = [s for s in res["stats"] if s["algo"] in (1, 2)]   # passes run by the pair kernel (the others: chunk-local kernel)
        n_launch = len(big)                                 # pair-kernel launches per step
        launches = n_launch * (acc.get("evented_steps") or args.steps)   # (the dispatches that carried events: every EVENT_EVERY-th step)
        avg_launch_s = tile_s / launches if launches and tile_s > 0 else None
        # algorithmic bytes of the prune: sum over passes (A_p * h * 24 + 2 N)  (SURVEY.md 8d), per launch: the mean
        b_k3 = sum(s["n_active_before"] * h * 24 + 2 * n_pass for s in res["stats"])
        b_big = sum(s["n_active_before"] * h * 24 + 2 * n_pass for s in big)
        b_launch = b_big / n_launch if n_launch else None
        b_k12 = n * len(ens.frag_coords) * 96 + n_pass * ens.n_atoms * 24 + n
        ms_per_step = dt / args.steps * 1e3
        kernel = "k_rmsd_sieve" if res["stats"] and res["stats"][0]["algo"] == 2 else "k_rmsd_tile"
        # which form of the descriptor screen the sieve's pair kernel ran (csrc/prune.hip: decided per run from its size and the options)
        opts = dict(o.split("=") for o in args.opt)
        if kernel != "k_rmsd_sieve":
            screen = None
        elif float(opts.get("sieve_mm", 1)) == 2 or (float(opts.get("sieve_mm", 1)) == 1 and n_pass >= float(opts.get("mm_min_n", 100000))):
            screen = "mfma64"      # k_rmsd_sieve_mm (+ k_rmsd_sieve_sorted_mm): 64 rows per work item
        elif float(opts.get("sieve_mm16", 1)) == 1:
            screen = "mfma16"      # k_rmsd_sieve_mm16: k_rmsd_sieve's 16-row items
        else:
            screen = "pk_fp32"     # k_rmsd_sieve: the packed-fp32 screen
        kernel_label = {"mfma64": "k_rmsd_sieve_mm", "mfma16": "k_rmsd_sieve_mm16", "pk_fp32": "k_rmsd_sieve<16,2,...>", None: kernel}[screen]
        # HBM bytes per launch: NOT measurable inside this process (rocprofv3 --pmc wraps the whole command, one counter per
        # pass); taken from the committed PMC passes of this same command (tools/profile.sh: FETCH_SIZE and WRITE_SIZE in
        # separate runs, KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) -- and only if that profile was
        # taken from the kernels that run now (digest of tscode_amd/csrc recorded in it); otherwise null, labelled stale
        traffic, traffic_src = None, None
        now, built = csrc_digest(), binary_digest()
        if built != now:
            traffic_src = (f"the loaded libtscode_hip.so was built from csrc {built}, the sources beside it are {now}: traffic withheld; "
                           f"rebuild (python -m tscode_amd.build --force)")
        elif args.config == "C3" and args.n_poses is None and world == 1:
            for name in PMC_PROFILES:
                pmc_path = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pmc_path):
                    continue
                pmc = json.load(open(pmc_path))
                if pmc.get("csrc_sha256_16") != now:
                    traffic_src = (f"profiles/{name} is STALE (taken from csrc {pmc.get('csrc_sha256_16', 'unrecorded')}, the kernels now are "
                                   f"{now}): traffic withheld; re-run tools/profile.sh")
                    break
                f = next((v for k, v in pmc.get("FETCH_SIZE", {}).items() if kernel in k), None)
                w = next((v for k, v in pmc.get("WRITE_SIZE", {}).items() if kernel in k), None)
                if f and w:
                    traffic = (2.0 * f["per_call_KB"] + w["per_call_KB"]) * 1024.0
                    traffic_src = (f"profiles/{name} (csrc {now}, the kernels of this run): (2 x FETCH_SIZE + WRITE_SIZE) per launch, "
                                   f"rocprofv3 --pmc, separate passes")
                break
        hbm_achieved = b_launch / avg_launch_s / 1e9 if avg_launch_s else None
        # What the instructions execute (the PRIMARY compute figure): the fp32 screen (2 families x 9 FMA = 36 flop per screened pair:
        # 18 v_pk_fma_f32 per row and 128 columns) and fp64 H + quartic tests (18 h + 110 flop per pair that reaches them; register-tiled
        # kernel: every computed pair, h padded to a multiple of 4).  Beside it SURVEY.md 8(d)'s accounting: the reference's own
        # pair evaluations x (46 h + 500) flop each over the kernel time -- the sieve reaches the reference's verdicts without
        # forming H for most pairs, so that ratio exceeds 1 and is NOT a roofline fraction.
        evals_big = sum(s["pairs_evaluated"] for s in big)
        screened_big, computed_big = sum(s["pairs_screened"] for s in big), sum(s["pairs_computed"] for s in big)
        ref_equiv = evals_big * flops_per_eval / (tile_s / ev_steps) / 1e12 if tile_s > 0 else None
        f16_mfma_flops = 0.0
        if screened_big and screen in ("mfma16", "mfma64"):
            # the screen on the matrix cores: per pair and family one row of a 16 x 16 x 16 MFMA = 16 multiply-adds (8 components, 3 + 3 norm
            # pieces, 2 idle slots); no fp32 vector flops in it (its VALU work is integer: the results' sign bits)
            f32_flops = 0.0
            f16_mfma_flops = screened_big * 2 * 16 * 2
            f64_flops = computed_big * (18 * h + 110)
        elif screened_big:
            f32_flops = screened_big * SCREEN_FLOP_PER_PAIR
            f64_flops = computed_big * (18 * h + 110)
        else:
            f32_flops = 0.0
            f64_flops = computed_big * (18 * ((h + 3) // 4 * 4) + 110)
        per_s = (lambda x: x / (tile_s / ev_steps) / 1e12) if tile_s > 0 else (lambda x: None)
        ex32, ex64, ex16 = per_s(f32_flops), per_s(f64_flops), per_s(f16_mfma_flops)
        # the part of the path SURVEY.md 8(d) calls HBM-streaming: fused embed + clash verdicts, ordered compaction, the passing poses embedded
        # with their descriptors.  Algorithmic bytes B_K12 over the two stages' own HIP events (the 3 steps with every library event on,
        # outside the timed region: each stage carries some 4 us of event cost, so the figure is a lower bound)
        front_ms = (stage_ms.get("embed_clash", 0.0) + stage_ms.get("compact", 0.0)) / args.steps if stage_ms else 0.0
        front_roofline = None
        if front_ms > 0 and not sharded_mode:
            gbs = b_k12 / (front_ms / 1e3) / 1e9
            front_roofline = {"kernels": "k_clash_lanes (k_clash where it does not apply) + k_scan_block_sums + k_scan_write + k_transform_describe",
                              "bound": "hbm", "algorithmic_bytes": b_k12, "ms": front_ms, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": gbs / HBM_PEAK_GBS,
                              "what": "B_K12 = N n_mols 96 + N_pass n 24 + N (SURVEY.md 8d) over the embed_clash + compact stages' HIP events"}
        # VALU issue of the dominant kernel from the committed SQ-counter summary of this same command (tools/pmc_sq.sh: SQ_INSTS_VALU and
        # SQ_BUSY_CU_CYCLES over every launch of the kernel): a CU issues one wave64 VALU instruction per cycle (4 SIMDs x 4 cycles), so
        # `issue_frac` = instructions / (CU-cycles of the kernel's launches) is the fraction of the VALU issue ceiling the kernel reaches --
        # the resource that bounds it.  Only while the profile was taken from the kernels that run now (digest), like `traffic`.
        issue = None
        for name in SQ_PROFILES.get(args.config, ()):
            sq_path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(sq_path) or args.n_poses is not None or world != 1:
                continue
            sq = json.load(open(sq_path))
            run = sq.get("_run", {})
            if run.get("csrc_sha256_16") != now or built != now:
                issue = {"source": f"profiles/{name} is STALE (taken from csrc {run.get('csrc_sha256_16')}, the kernels now are {now}): withheld; re-run tools/pmc_sq.sh"}
                break
            kk = next((v for k_, v in sq.items() if k_ != "_run" and kernel in k_ and "sorted" not in k_), None)
            if kk and kk.get("SQ_BUSY_CU_CYCLES") and avg_launch_s:
                all_cu_cycles = kk["launches"] * avg_launch_s * GPU_CLOCK_GHZ * 1e9 * 256
                issue = {"valu_insts": kk["SQ_INSTS_VALU"], "busy_cu_cycles": kk["SQ_BUSY_CU_CYCLES"], "launches": kk["launches"],
                         "issue_frac_while_busy": kk["SQ_INSTS_VALU"] / kk["SQ_BUSY_CU_CYCLES"],
                         "issue_frac": kk["SQ_INSTS_VALU"] / all_cu_cycles,
                         "fma_f32_share_of_valu": kk.get("SQ_INSTS_VALU_FMA_F32", 0.0) / kk["SQ_INSTS_VALU"],
                         "wait_share_of_wave_cycles": kk.get("SQ_WAIT_ANY", 0.0) / kk["SQ_WAVE_CYCLES"] if kk.get("SQ_WAVE_CYCLES") else None,
                         "source": f"profiles/{name} (csrc {now}, the kernels of this run): rocprofv3 --pmc SQ_*, separate passes; CU-cycles of the "
                                   f"launches = launches x avg_launch_us (this run) x {GPU_CLOCK_GHZ} GHz x 256 CUs"}
            break
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
            "config": {"workload": f"{args.config}: {n} conformers x {ens.n_atoms} atoms ({h} heavy), {len(ens.frag_coords)} rigid fragments, "
                                   f"10 children per parent, seed {ens.seed}; clash_thresh 1.5, max_clashes 0, rmsd_thr 0.5, "
                                   f"mode {args.mode} ({'reference-exact' if args.mode == 0 else 'cache-free'})",
                       "n_pass_clash": n_pass, "n_survivors": n_keep, "keep_sha256_16": leg["digest"],
                       "parity_vs_recorded_oracle": leg["parity"],
                       "chain": (f"csearch_rotate of fragment 0 ({CHAIN_CANDIDATES} angle sets x 8 torsions, walk-back included) -> {res.get('n_conformers')} kept "
                                 f"candidates = its conformers -> embed -> clash -> prune; the candidate array stays on the device") if chain else None,
                       "parallelism": (f"one ensemble sharded over {world} rank(s): front half per `front`, passes partitioned by chunks or sharded by row tiles, one all-reduce per pass" if sharded_mode else
                                       f"{world} x (one whole ensemble per GPU), no data-path collective"),
                       "conformers_per_step_all_ranks": units_per_step,
                       "poses": ("synthetic rigid-body transforms of three fragments (tscode_amd/synthetic.py): the reference's trimolecular cyclical_embed is "
                                 "undefined -- vec_angle on 2-vectors, embeds.py:297-299 / algebra.py:87 -- so config 5 cannot be fed by it (INTEGRATION.md D)"
                                 if args.config in ("C5", "C5chain") else "synthetic rigid-body transforms (tscode_amd/synthetic.py, SURVEY.md 8d)"),
                       "library_events_in_timed_region": args.pass_timing},
            "roofline": {
                "kernel": kernel_label + " (all-pairs Kabsch RMSD of one pass; one launch per pass)" +
                          ("; descriptor screen on the matrix cores: v_mfma_f32_16x16x16_f16 on float16 records (csrc/mm.hpp)" if screen in ("mfma16", "mfma64") else ""),
                # what binds the kernel -- NOT HBM: its 33 MB per launch sit in the infinity cache.  `achieved` / `peak` / `frac`
                # below stay the bench contract's HBM figure (algorithmic bytes over the launch against 8 TB/s); `issue_frac` is the kernel against
                # the roof that does bind it
                "bound": "latency" if screen in ("mfma16", "mfma64") else "valu_issue",
                "bound_what": ("with the screen on the matrix cores the kernel is no longer bound by VALU issue (issue_frac below: a third of the ceiling; the MFMA "
                               "pipe a few per cent busy): a work item is a chain of dependent memory round trips -- stop columns and operands, the column "
                               "tiles, the candidates' indices, their coordinates, the atomics on the way out -- at four wavefronts per SIMD (MEASURED.md "
                               "section 8: per item 3.3 us prologue, 4.0 screen, 5.2 evaluation, 1.3 tail)" if screen in ("mfma16", "mfma64") else
                               "VALU issue of the packed-fp32 screen (issue_frac: wave64 VALU instructions per CU-cycle, ceiling 1)"),
                "frac_resource": "hbm",
                "issue_frac": issue.get("issue_frac") if issue else None,
                "issue": issue,
                "achieved": hbm_achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (hbm_achieved / HBM_PEAK_GBS) if hbm_achieved else None,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "csrc_sha256_16": now,
                "binary_csrc_sha256_16": built,
                "algorithmic_bytes_per_launch": b_launch,
                "avg_launch_us": avg_launch_s * 1e6 if avg_launch_s else None,
                "launches_per_step": n_launch,
                "kernel_ms_per_step": tile_ms / ev_steps,
                "timing": f"HIP start/stop events attached to the kernel's dispatches (hipExtLaunchKernel) inside the timed region, in every {EVENT_EVERY}th step "
                          f"of it ({acc.get('evented_steps')} of {args.steps} steps): a pair of events costs a launch about 6 us",
                "passes_in_chunk_local_kernel": len(res["stats"]) - n_launch,
                "executed": {"what": ("what the kernel's instructions do (the screen as float16 MFMAs with fp32 accumulation, 64 flop per pair -- a per cent of the "
                                      "matrix cores' 2.5 PFLOP/s: the screen is no longer what the kernel is made of --, H and the quartic tests in fp64): the "
                                      "kernel is bound by the VALU issue of the sign-bit collection behind the MFMAs and of the evaluation stages, and by the "
                                      "latency of the candidates' gathers" if screen in ("mfma16", "mfma64") else
                                      "what the kernel's instructions do (the screen in packed fp32, H and the quartic tests in fp64): the compute "
                                      "figure to read; the kernel is bound by VALU issue of the screen and by the latency of the candidates' gathers"),
                             "screen": screen,
                             "f16_mfma_TFLOPs": ex16 if f16_mfma_flops else None, "f16_mfma_peak_TFLOPs": 2500.0,
                             "f16_mfma_frac": (ex16 / 2500.0) if (f16_mfma_flops and ex16) else None,
                             "fp32_TFLOPs": ex32, "fp32_peak_TFLOPs": FP32_VALU_PEAK_TFLOPS, "fp32_frac": (ex32 / FP32_VALU_PEAK_TFLOPS) if ex32 else None,
                             "fp64_TFLOPs": ex64, "fp64_peak_TFLOPs": FP64_VALU_PEAK_TFLOPS, "fp64_frac": (ex64 / FP64_VALU_PEAK_TFLOPS) if ex64 else None,
                             "pairs_screened_per_step": screened_big, "pairs_with_H_formed_per_step": computed_big},
                "reference_flop_equivalent": {"what": "SURVEY.md 8(d)'s accounting, NOT a roofline fraction: the pair evaluations the REFERENCE would make "
                                                      "x (46 h + 500) flop over the kernel time; above the fp64 peak because the sieve decides most pairs "
                                                      "without the work this formula counts",
                                              "pair_evals_per_step": evals_big, "flops_per_eval": flops_per_eval, "TFLOPs_equivalent": ref_equiv,
                                              "ratio_to_fp64_peak": (ref_equiv / FP64_VALU_PEAK_TFLOPS) if ref_equiv else None},
            },
            "pipeline_hbm": {"algorithmic_bytes": (b_k12 + b_k3) * (units_per_step // n),
                             "achieved_GBs": (b_k12 + b_k3) * (units_per_step // n) / (ms_per_step / 1e3) / 1e9,
                             "peak_GBs": HBM_PEAK_GBS * world,
                             "frac": (b_k12 + b_k3) * (units_per_step // n) / (ms_per_step / 1e3) / 1e9 / (HBM_PEAK_GBS * world)},
            "events_off": leg["events_off"],
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "roofline_front": front_roofline,
            "passes": [{"k": s["k"], "active": s["n_active_before"], "evals": s["pairs_evaluated"], "screened": s["pairs_screened"], "H_formed": s["pairs_computed"],
                        "exact": s["candidates"], "ms": round(pass_ms[i] if pass_ms and i < len(pass_ms) else s["gpu_ms"], 4),
                        "tile_ms": round(s["tile_ms"], 4),
                        "kernel": {1: "k_rmsd_tile", 2: "k_rmsd_sieve", 3: "k_pass_chunks"}.get(s["algo"], "?")} for i, s in enumerate(res["stats"])],
            "detail_note": "stage_ms_per_step and passes[].ms come from 3 extra steps with every library event on (pass_timing 2), "
                           "outside the timed regions; passes[].tile_ms (the pair kernel's own events) from the timed region (0 for passes the chunk-local kernel runs: its events are taken at pass_timing 2 only)",
        }
        if sharded_mode:
            out["rccl_world"] = dist.get_world_size() if use_dist else 1
            out["collective_backend"] = dist.get_backend() if use_dist else None
            out["allgather_bytes_per_step"] = res.get("allgather_bytes")
            out["front"] = res.get("front")                    # "shard" or "replicate": what the line was measured with
            out["front_tuning"] = getattr(leg["pipe"], "front_tuning", None)
            out["allreduce_bytes_per_step"] = res.get("allreduce_bytes")
            out["sharded_passes"] = [{"k": k, "best_entries": nb} for k, nb in res.get("exchanges", [])]
            out["partitioned_passes"] = [{"k": k, "int64_words": w} for k, w in res.get("partitioned", [])]
            # counts all-reduce and coordinates all-gather (front = shard), the clash mask (front = hybrid), one all-reduce per pass that
            # is partitioned by chunks or sharded by row tiles, one for the cache views between the two kinds
            out["collectives_per_step"] = collectives_of(res)
            out["exchange"] = res.get("exchange")
        if world > 1:
            out["multi"] = {"value_is": "sharded_single_ensemble" if sharded_mode else "replicas", "why": multi_reason,
                            "beside_it": side[0] if side is not None else None}
            # both figures at top level, whichever of them `value` is: `value_strong` = ONE ensemble sharded over the ranks (RCCL / the library's
            # exchange in the data path; what north_star's ">= 6x further at 8 GPUs" speaks of), `value_weak` = one whole ensemble per GPU
            strong = ({"value": out["value"], "ms_per_step": ms_per_step, "res": res} if sharded_mode else
                      ({"value": side[1]["value"], "ms_per_step": side[1]["ms_per_step"], "res": sharded_leg["res"]} if side is not None and sharded_leg is not None else None))
            weak = ({"value": out["value"], "ms_per_step": ms_per_step} if not sharded_mode else
                    ({"value": side[1]["value"], "ms_per_step": side[1]["ms_per_step"]} if side is not None else None))
            out["value_strong"] = strong["value"] if strong else None
            out["ms_per_step_strong"] = strong["ms_per_step"] if strong else None
            out["value_weak"] = weak["value"] if weak else None
            out["ms_per_step_weak"] = weak["ms_per_step"] if weak else None
            if strong:
                out["rccl_world"] = dist.get_world_size()
                out["collectives_per_step"] = collectives_of(strong["res"])
                out["exchange"] = strong["res"].get("exchange")
            out["selftest_collectives"] = selftest
        if side is not None:
            out[side[0]] = side[1]
        if error is not None:
            out["error"] = error
        if notes:
            out["notes"] = notes
    # Beside the line's strictly sequential steps: D steps in flight at a time.  A step at 100k conformers is a chain of ~50
    # dependent launches most of which occupy a small part of the chip for a few microseconds; a caller with several
    # independent ensembles (TSCoDe embeds one per reactant pair / conformer set; multiembed runs them from several processes)
    # gives each its own context and stream, and the GPU overlaps one chain's latency with another's kernels.  Every step does
    # all of its work on its own buffers; the verdicts of every one are checked.  A side figure, never `value`.
    if world == 1 and args.in_flight > 1 and not args.no_side_leg and not chain and rank == 0:
        try:
            D = args.in_flight
            pipes = [DevicePipeline(ens, device_index=local_rank, rank=0, world=1, mode=args.mode) for _ in range(D)]
            for q in pipes:
                configure(q, 0)
                q.step()
                q.step()
            last = [None] * D

            def worker(i):
                for _ in range(args.steps):
                    last[i] = pipes[i].step()
            torch.cuda.synchronize()
            gc.collect()
            gc.freeze()
            runs = []
            for _ in range(2):      # (twice, the shorter one counts: on these boxes one run in seven has all three threads stand still for
                                    # 48 ms at the same moment -- not the collector's doing; both are kept below)
                threads = [threading.Thread(target=worker, args=(i,)) for i in range(D)]
                t0 = time.perf_counter()
                for t in threads:
                    t.start()
                for t in threads:
                    t.join()
                torch.cuda.synchronize()
                runs.append(time.perf_counter() - t0)
            dtc = min(runs)
            ok = all(verdict(pipes[i], last[i])[1] in (True, None) and last[i]["n_keep"] == n_keep for i in range(D))
            out["steps_in_flight"] = {"in_flight": D, "steps": args.steps * D, "ms_per_step": dtc / (args.steps * D) * 1e3,
                                      "value": ens.n_poses * args.steps * D / dtc, "unit": "conformers/s", "every_step_checked": bool(ok),
                                      "ms_per_step_of_both_runs": [r / (args.steps * D) * 1e3 for r in runs],
                                      "what": f"{D} independent steps at a time, each on its own library context, HIP stream and buffers (one host "
                                              f"thread each, library events off); the shorter of two runs; the line's `value` above is one step at a time"}
            del pipes
        except Exception as exc:
            out["steps_in_flight"] = {"error": f"{type(exc).__name__}: {exc}"}
    # beside the C3 line: config C4 (1M x 50), the workload large enough for sharding one ensemble to pay -- one GPU: the one-call
    # pipeline; N > 1: the same sharded protocol.  A side figure of the default run, so that every N of a scaling series has it.
    # Under a watchdog of its own: the line above is complete and is printed whatever happens here.
    if args.config == "C3" and args.n_poses is None and not args.no_side_leg and not (world > 1 and args.backend != "nccl") and exit_code == 0:
        def bail4():
            if rank == 0:
                out["c4"] = {"error": "the C4 side leg did not finish within 240 s"}
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(3)
        watchdog4 = threading.Timer(240.0, bail4)
        watchdog4.daemon = True
        watchdog4.start()
        try:
            leg["pipe"] = None                               # the C3 buffers are no longer needed
            ens4 = make_config("C4")
            exp4 = json.load(open(exp_path)).get(f"C4:{ens4.n_poses}:mode{args.mode}") if os.path.exists(exp_path) else None
            leg4 = run_leg(world > 1, ens4, exp4, detail=False)
            c4 = leg_summary(leg4, SHARDED_WHAT if world > 1 else "the one-call pipeline on one GPU")
            c4["workload"] = f"C4: {ens4.n_poses} conformers x {ens4.n_atoms} atoms ({ens4.n_heavy} heavy), seed {ens4.seed}"
            c4["n_pass_clash"] = int(leg4["res"]["n_pass"])
            if world == 1:
                c4["pair_kernels"] = ("k_rmsd_sieve_mm + k_rmsd_sieve_sorted_mm: the descriptor screen on the matrix cores (v_mfma_f32_16x16x16_f16 on float16 "
                                      "records, the fp32 screen behind it; csrc/mm.hpp, cull_mm.hpp) -- what runs of 150 000 structures and more take")
            # HBM traffic of the pair kernels at C4 against their algorithmic bytes, from the committed PMC passes of `bench.py --config C4`
            # (tools/profile.sh; only while that profile was taken from the kernels that run now)
            st4 = leg4["res"]["stats"]
            h4, np4 = ens4.n_heavy, int(leg4["res"]["n_pass"])
            alg4 = sum(s_["n_active_before"] * h4 * 24 + 2 * np4 for s_ in st4 if s_["algo"] in (1, 2))
            for name in PMC_PROFILES_C4:
                pmc_path = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(pmc_path) or world > 1:
                    continue
                pmc = json.load(open(pmc_path))
                if pmc.get("csrc_sha256_16") != csrc_digest() or binary_digest() != csrc_digest():
                    c4["traffic_source"] = f"profiles/{name} is STALE (csrc {pmc.get('csrc_sha256_16')}): traffic withheld"
                    break
                tot = 0.0
                for kname in ("k_rmsd_sieve_sorted", "k_rmsd_sieve<", "k_rmsd_sieve_mm<"):
                    f_ = [v for k_, v in pmc.get("FETCH_SIZE", {}).items() if kname in k_]
                    w_ = [v for k_, v in pmc.get("WRITE_SIZE", {}).items() if kname in k_]
                    tot += sum(2.0 * v["total_KB"] for v in f_) + sum(v["total_KB"] for v in w_)
                steps_prof = pmc.get("steps_profiled")
                if tot and steps_prof:
                    c4["pair_kernels_traffic_bytes_per_step"] = tot * 1024.0 / steps_prof
                    c4["pair_kernels_algorithmic_bytes_per_step"] = alg4
                    c4["traffic_over_algorithmic"] = tot * 1024.0 / steps_prof / alg4 if alg4 else None
                    c4["traffic_source"] = f"profiles/{name}: (2 x FETCH_SIZE + WRITE_SIZE) of k_rmsd_sieve + k_rmsd_sieve_sorted per step, rocprofv3 --pmc"
                break
            if world > 1:
                c4["front"] = leg4["res"].get("front")
                c4["front_tuning"] = getattr(leg4["pipe"], "front_tuning", None)
                c4["allgather_bytes_per_step"] = leg4["res"].get("allgather_bytes")
                c4["allreduce_bytes_per_step"] = leg4["res"].get("allreduce_bytes")
                c4["sharded_passes"] = [k for k, _ in leg4["res"].get("exchanges", [])]
                c4["partitioned_passes"] = [k for k, _ in leg4["res"].get("partitioned", [])]
            del leg4, ens4
        except Exception as exc:
            c4 = {"error": f"{type(exc).__name__}: {exc}"}
        watchdog4.cancel()
        if rank == 0:
            out["c4"] = c4
    # beside the favourable C3: the workloads on which the prune's descriptor sieve has the least to offer (tools/hard_workloads.py) --
    # children scattered around the threshold, and structures whose descriptors all coincide (reference-exact and cache-free mode),
    # each with the automatic kernel choice, the sieve and the all-pairs kernel, and the CPU port on a bounded sample.  Side
    # figures, never `value`.
    if world == 1 and rank == 0 and args.config == "C3" and args.n_poses is None and not args.no_side_leg and not chain and exit_code == 0:
        try:
            leg["pipe"] = None
            from tools.hard_workloads import measure as hard_measure
            hw = hard_measure(100_000, 12_000 if not args.no_cpu else 0, steps=5)
            out["hard_workloads"] = {name: {"workload": l["workload"],
                                            "ms_per_step": {a: round(v["ms_per_step"], 4) for a, v in l["gpu"].items()},
                                            "conformers_per_s": l["n"] / l["gpu"]["auto"]["ms_per_step"] * 1e3,
                                            "kernels_of_the_automatic_choice": l["gpu"]["auto"]["kernels"],
                                            "screen_pass_rate": l["gpu"]["sieve"]["screen_pass_rate"], "H_formed": l["gpu"]["sieve"]["H_formed"],
                                            "pairs_screened": l["gpu"]["sieve"]["pairs_screened"], "n_survivors": l["gpu"]["auto"]["n_survivors"],
                                            "kernels_agree": len({v["n_survivors"] for v in l["gpu"].values()}) == 1, "cpu": l.get("cpu")}
                                     for name, l in hw["legs"].items()}
        except Exception as exc:
            out["hard_workloads"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        if world == 1 and not args.no_cpu and not chain:
            n_cpu = args.cpu_sample if args.cpu_sample else (ens.n_poses if args.config == "C3" else 40000)
            out["cpu_baseline"] = cpu_baseline(args.config, min(n_cpu, ens.n_poses), ens.n_poses, args.mode)
            cb = out["cpu_baseline"]
            if cb["same_workload_as_value"]:
                cb["gpu_over_cpu"] = out["value"] / cb["value"]
                cb["survivors_identical_to_gpu"] = bool(cb["keep_sha256_16"] == leg["digest"])
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
