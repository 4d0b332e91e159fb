"""Worker of tests/test_gpu_parity.py::test_sharded_protocol_on_gpu_ranks: one rank of the sharded pipeline on the HIP
backend.  Launched by torch.distributed.run with the gloo backend, so that several ranks can share the box's one GPU
(device tensors are staged over the host for the collectives; under nccl = RCCL nothing is staged), or -- SHARD_BACKEND=nccl -- with RCCL
and a world of one: the protocol's collectives on device tensors, as a node runs them.
Rank 0 prints one JSON line."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    cfg, n_poses, min_pairs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    backend = os.environ.get("SHARD_BACKEND", "gloo")       # "nccl" (= RCCL): one rank per GPU, so on this box a world of one
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from tscode_amd.pipeline import DevicePipeline
    from tscode_amd.synthetic import make_config
    if cfg == "C5chain":          # config 5 as a chain, with the inputs of bench.py --config C5chain (tools/record_c5chain.py recorded the oracle's run)
        from tscode_amd.pipeline import CsearchChain, ShardedCsearchChain
        ens = make_config("C5", n_poses if n_poses > 0 else None)
        torsions, tmasks = CsearchChain.chain_torsions(ens.frag_coords[0].shape[1], 8, seed=5)
        n_cand = 20000 if n_poses <= 0 else max(200, n_poses // 25)
        angle_table = np.random.default_rng(6).choice(np.array([0, 0, 60, 120, 180, 240, 300, 25]), size=(n_cand, 8)).astype(np.int32)
        pipe = ShardedCsearchChain(ens, torsions, tmasks, angle_table, rank, world, thresh=1.4, device_index=0, mode=0, seed=7,
                                   shard_min_pairs=min_pairs if min_pairs > 0 else None)
    else:
        ens = make_config(cfg, n_poses if n_poses > 0 else None)
        pipe = DevicePipeline(ens, device_index=0, rank=rank, world=world, mode=0, shard_min_pairs=min_pairs if min_pairs > 0 else None,
                              force_sharded=world == 1)    # (a world of one still goes through the multi-rank protocol and its collectives)
    res, digest, unstable = None, None, 0
    for _ in range(int(os.environ.get("SHARD_STEPS", "2"))):   # more steps on the same state: buffers are reused; every step must agree
        res = pipe.step()
        torch.cuda.synchronize()
        keep = pipe.h_keep[:res["n_pass"]].numpy().copy()
        d = hashlib.sha256(np.packbits(keep.astype(bool)).tobytes()).hexdigest()[:16]
        unstable += digest is not None and d != digest
        digest = digest or d
    # the pipeline chose a form of the front half by timing both (tune_front, inside the first step); either form, forced, must
    # give the same survivors and the same evaluation counts
    forms_agree, evals = True, [s["pairs_evaluated"] for s in res["stats"]]
    for form in ("shard", "replicate", "hybrid"):
        pipe.front = form
        r2 = pipe.step()
        torch.cuda.synchronize()
        d2 = hashlib.sha256(np.packbits(pipe.h_keep[:r2["n_pass"]].numpy().astype(bool)).tobytes()).hexdigest()[:16]
        forms_agree = forms_agree and r2["front"] == form and d2 == digest and [s["pairs_evaluated"] for s in r2["stats"]] == evals
    # the pass loop behind the C ABI (tsc_prune_run_sharded: what every step above ran) against the same loop driven from the host, call by call
    be = getattr(pipe, "backend", None) or getattr(getattr(pipe, "pipe", None), "backend", None)
    loops_agree = None
    if be is not None:
        be.python_pass_loop = True
        r3 = pipe.step()
        torch.cuda.synchronize()
        d3 = hashlib.sha256(np.packbits(pipe.h_keep[:r3["n_pass"]].numpy().astype(bool)).tobytes()).hexdigest()[:16]
        loops_agree = bool(d3 == digest and [s["pairs_evaluated"] for s in r3["stats"]] == evals and r3["exchanges"] == res["exchanges"]
                           and r3["partitioned"] == res["partitioned"])
        be.python_pass_loop = False
    # the per-pass exchanges INSIDE the library (tsc_xchg_*: every rank's receive area mapped into the others with hipIpcGetMemHandle /
    # hipIpcOpenMemHandle -- here the ranks are processes sharing one card, which exercises the whole mechanism) against the callback form
    # (torch.distributed collectives) that every step above used: same survivors, evaluation counts and sequence of exchanges, with the
    # pass loop inside the library and driven from the host
    ipc_agrees, ipc_status = None, None
    if be is not None and world > 1:
        be.connect_exchange(dist, None)
        ok = True
        for host_loop in (False, True):
            be.python_pass_loop = host_loop
            r4 = pipe.step()
            torch.cuda.synchronize()
            d4 = hashlib.sha256(np.packbits(pipe.h_keep[:r4["n_pass"]].numpy().astype(bool)).tobytes()).hexdigest()[:16]
            ok = ok and bool(r4["exchange"] == "ipc" and d4 == digest and [s["pairs_evaluated"] for s in r4["stats"]] == evals
                             and r4["exchanges"] == res["exchanges"] and r4["partitioned"] == res["partitioned"])
        be.python_pass_loop = False
        ipc_status = list(be.xchg.status())
        ipc_agrees = bool(ok and ipc_status[1] == 0 and ipc_status[0] > 0 and res["exchange"] == "callback")
        be.disconnect_exchange()
    flags = torch.tensor([res["n_pass"], res["n_keep"], int(digest[:12], 16)], dtype=torch.int64, device="cuda:0" if backend == "nccl" else "cpu")
    gathered = [torch.zeros_like(flags) for _ in range(world)]
    dist.all_gather(gathered, flags)
    if backend == "nccl":      # the reductions the partitioned passes use (a world of one has no such pass): int64 SUM on a device tensor
        probe = torch.arange(8, dtype=torch.int64, device="cuda:0")
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)
        assert probe.tolist() == list(range(8))
    if rank == 0:
        sharded_passes = [s["k"] for s in res["stats"] if s["algo"] in (1, 2)]
        print(json.dumps({"world": world, "n_pass": res["n_pass"], "n_keep": res["n_keep"], "n_conformers": res.get("n_conformers"),
                          "keep_sha256_16": digest, "steps_that_differ": unstable,
                          "ranks_agree": all(torch.equal(g, flags) for g in gathered), "counts": res["counts"], "forms_agree": forms_agree,
                          "loops_agree": loops_agree, "ipc_agrees": ipc_agrees, "ipc_status": ipc_status, "exchanges": res["exchanges"], "partitioned": res["partitioned"],
                          "front_tuning": pipe.front_tuning,
                          "pairs_evaluated": [s["pairs_evaluated"] for s in res["stats"]], "global_path_passes": sharded_passes}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
