"""The library's own exchange (tsc_xchg_*, csrc/xchg.hip) on its own: correctness of both reductions at the message sizes of the sharded
prune, and microseconds per exchange from HIP events.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29701 tools/ipc_exchange_bench.py [out.json]

Ranks are processes; on a one-GPU box they share the card (gloo carries the 64-byte handles), which exercises the whole mechanism --
hipIpcGetMemHandle / hipIpcOpenMemHandle, writes into a peer's fine-grained area, system-scope flags -- but NOT xGMI, and two processes
time-slice one GPU: the microseconds are an upper bound of what two GPUs would show for the kernels and say nothing about link time.
With SHARD_BACKEND=nccl (one rank per GPU on a real node) the same script measures the real thing and RCCL's all-reduce beside it."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    backend = os.environ.get("SHARD_BACKEND", "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from tscode_amd._lib import XCHG_MIN_I32, XCHG_SUM_I64
    from tscode_amd.engine import Engine, IpcExchange
    eng = Engine(dev_index)
    stream = torch.cuda.Stream()
    eng.set_stream(stream.cuda_stream)
    dev = torch.device(f"cuda:{dev_index}")
    n_max = 1_000_000
    x = IpcExchange(eng, rank, world, IpcExchange.slot_bytes(eng.lib, n_max, 0))
    x.connect_over(dist, None)
    rng = np.random.default_rng(5)
    # message sizes of the protocol: bits + statistics of a partitioned pass at C3 / C4, the cache views at C3, best[] at C3 / C4 (odd count too)
    cases = [("sum_i64", 57_046 // 64 + 48), ("sum_i64", 482_855 // 64 + 48), ("sum_i64", 9 * (57_046 // 64 + 41)), ("sum_i64", 1),
             ("min_i32", 57_046), ("min_i32", 482_855), ("min_i32", 1), ("min_i32", 2)]
    out = {"world": world, "backend": backend, "ranks_share_one_gpu": backend != "nccl", "cases": []}
    ok_all = True
    reps = 50
    with torch.cuda.stream(stream):
        for kind_name, count in cases:
            kind = XCHG_SUM_I64 if kind_name == "sum_i64" else XCHG_MIN_I32
            # every rank draws ALL ranks' contributions from one seed: the expected result needs no second transport
            if kind == XCHG_SUM_I64:
                parts = rng.integers(0, 2 ** 40, size=(world, count), dtype=np.int64)
                want = parts.sum(axis=0)
                dt = torch.int64
            else:
                parts = rng.integers(-5, 2 ** 30, size=(world, count), dtype=np.int64).astype(np.int32)
                want = parts.min(axis=0)
                dt = torch.int32
            mine = torch.from_numpy(parts[rank].copy()).to(dev)
            buf = torch.empty(count + 8, dtype=dt, device=dev)[:count]
            good = True
            for _ in range(3):                               # several exchanges on the same slots: both parities, flags that move on
                buf.copy_(mine)
                x.allreduce(kind, buf, count)
                torch.cuda.synchronize()
                good = good and bool(np.array_equal(buf.cpu().numpy(), want))
            dist.barrier()
            # timing: `reps` exchanges back to back between two events on the stream (the buffer's content no longer matters)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                x.allreduce(kind, buf, count)
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            rccl_us = None
            if backend == "nccl":
                op = dist.ReduceOp.SUM if kind == XCHG_SUM_I64 else dist.ReduceOp.MIN
                dist.all_reduce(buf, op=op)
                torch.cuda.synchronize()
                e0.record(stream)
                for _ in range(reps):
                    dist.all_reduce(buf, op=op)
                e1.record(stream)
                torch.cuda.synchronize()
                rccl_us = e0.elapsed_time(e1) * 1e3 / reps
            t = torch.tensor([us, rccl_us or 0.0, 1.0 if good else 0.0], dtype=torch.float64)
            if backend == "nccl":
                td = t.to(dev)
                dist.all_reduce(td, op=dist.ReduceOp.MAX)
                tmax = td.cpu()
                td = t.to(dev)
                dist.all_reduce(td, op=dist.ReduceOp.MIN)
                tmin = td.cpu()
            else:
                tmax, tmin = t.clone(), t.clone()
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            ok_all = ok_all and bool(tmin[2] == 1.0)
            out["cases"].append({"kind": kind_name, "count": count, "bytes": count * (8 if kind == XCHG_SUM_I64 else 4), "correct_on_every_rank": bool(tmin[2] == 1.0),
                                 "ipc_us_per_exchange_slowest_rank": float(tmax[0]), "rccl_us_per_allreduce_slowest_rank": float(tmax[1]) if rccl_us else None})
    n_x, n_to = x.status()
    out["exchanges"], out["timeouts"], out["all_correct"] = n_x, n_to, bool(ok_all and n_to == 0)
    out["csrc_sha256_16"] = eng.lib.tsc_build_digest().decode()
    out["what"] = ("per exchange: k_xchg_push (this rank's contribution written into every peer's slot, flag raised) + k_xchg_reduce (wait for the peers' "
                   "flags, fold their slots into the buffer), HIP events around 50 back-to-back exchanges, slowest rank")
    if rank == 0:
        line = json.dumps(out)
        print(line, flush=True)
        if len(sys.argv) > 1:
            with open(sys.argv[1], "w") as f:
                f.write(json.dumps(out, indent=1) + "\n")
    dist.barrier()
    x.close()
    dist.destroy_process_group()
    if not out["all_correct"]:
        sys.exit(4)


if __name__ == "__main__":
    main()
