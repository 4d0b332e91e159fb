"""Device-resident pipeline: embed -> clash mask -> ordered compaction -> RMSD prune.

``DevicePipeline`` keeps one ensemble in HBM as torch tensors (PyTorch is only the allocator, the stream
and the torch.distributed/RCCL transport here) and runs the whole hot path on it through the C ABI.

One GPU:   tsc_pipeline_dev does everything in one call.
N GPUs:    one process per GPU, ``sharded_step``:
           1. the pose axis is cut into contiguous blocks; each rank embeds and clash-filters its block
              (no communication) and compacts the survivors in order;
           2. one exchange of coordinates: the surviving heavy-atom shards are all-gathered (RCCL over
              xGMI), so every rank holds the full heavy array in the original order;
           3. every pass of prune_conformers_rmsd deals its row tiles round-robin to the ranks (rows of a
              pass are independent, tscode/rmsd_pruning.py:92,101-113); a rank fills best[] for its rows,
              an all-reduce(MIN) over best[] merges them, and every rank applies the identical mask and
              cache update, so no rank ever needs another rank's mask.

``sharded_step`` only talks to a small backend interface, so the same protocol code runs under gloo on
CPU tensors in the tests (with a test-only backend) and under RCCL on the GPU (``HipShardBackend``).
"""

from __future__ import annotations

import numpy as np

from .engine import Engine, FragmentSet

__all__ = ["DevicePipeline", "HipShardBackend", "sharded_step", "block_bounds", "SHARD_MIN_PAIRS"]

# A pass smaller than this many pairs (estimate n * (n / k) / 2, identical on every rank) is not worth a collective:
# every rank runs it whole and reaches the same verdicts on its own.  On MI355X such a pass takes tens of
# microseconds, the same as one small all-reduce over xGMI.
SHARD_MIN_PAIRS = 50_000_000


def block_bounds(n: int, rank: int, world: int):
    """Contiguous block [lo, hi) of the pose axis owned by ``rank``."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _staged(dist, t, group):
    """gloo moves CUDA tensors only for a few collectives: with it (several ranks rehearsing on one GPU, tests) device tensors
    take the detour over the host.  Under nccl (= RCCL) nothing is staged."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce(dist, t, op, group):
    if _staged(dist, t, group):
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_gather(dist, out, inp, group):
    if _staged(dist, inp, group):
        co, ci = out.cpu(), inp.cpu()
        dist.all_gather_into_tensor(co, ci, group=group)
        out.copy_(co)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def sharded_step(backend, rank: int, world: int, dist, group=None, min_pairs=None):
    """One step of the hot path over an ensemble sharded across ``world`` ranks, run inside ``backend.stream_context()`` when the
    backend has one (the HIP backend makes its own torch stream current, so that its kernels, torch's copies and the
    collectives are ordered on one stream)."""
    enter = getattr(backend, "stream_context", None)
    if enter is None:
        return _sharded_step(backend, rank, world, dist, group, min_pairs)
    with enter():
        return _sharded_step(backend, rank, world, dist, group, min_pairs)


def _sharded_step(backend, rank: int, world: int, dist, group, min_pairs):
    """The protocol itself.

    backend interface (all tensors live where the backend computes):
        embed_clash_block() -> n_pass_local      fills backend.heavy_local[:n_pass_local]  (h, 3 per row)
        heavy_pad, gather, heavy_all, counts, keep, max_local, best      preallocated tensors / int
        make_stepper(n_pass) -> stepper with next_pass(), pass_estimate(), pass_local(rank, world), n_active(), pass_finish(),
                                stats(), copy_mask(dst), close(); it keeps best[] in backend.best
    """
    n_pass_local = int(backend.embed_clash_block())
    # how many poses of every block passed the clash check
    backend.counts.zero_()
    backend.counts[rank] = n_pass_local
    _all_reduce(dist, backend.counts, dist.ReduceOp.SUM, group)
    counts = [int(c) for c in backend.counts.cpu().tolist()]
    # the one exchange of coordinates: padded shards, one all-gather, then un-pad in block order
    if backend.heavy_pad.data_ptr() != backend.heavy_local.data_ptr():
        backend.heavy_pad[:n_pass_local].copy_(backend.heavy_local[:n_pass_local])
    _all_gather(dist, backend.gather.view(-1), backend.heavy_pad.view(-1), group)
    off = 0
    for r, c in enumerate(counts):
        backend.heavy_all[off:off + c].copy_(backend.gather[r * backend.max_local:r * backend.max_local + c])
        off += c
    n_pass = off
    stats = []
    exchanges = []                                      # (k, entries of best[] all-reduced) of every pass that was sharded
    if n_pass > 0:
        st = backend.make_stepper(n_pass)
        limit = SHARD_MIN_PAIRS if min_pairs is None else min_pairs
        try:
            while True:
                if hasattr(st, "run_replicated"):       # the small passes in one library call; back here for an exchange
                    k = st.run_replicated(world, limit)
                    if k == 0:
                        break
                    shard = True
                else:
                    k = st.next_pass()
                    if k == 0:
                        break
                    shard = world > 1 and st.pass_estimate() >= limit
                if shard:
                    st.pass_local(rank, world)          # this rank's row tiles only ...
                    n_best = st.n_active()
                    _all_reduce(dist, backend.best[:n_best], dist.ReduceOp.MIN, group)   # ... merged
                    exchanges.append((int(k), int(n_best)))
                else:
                    st.pass_local(0, 1)                 # small pass: replicated, no exchange
                st.pass_finish()
            st.copy_mask(backend.keep)
            if getattr(backend, "keep_host", None) is not None:      # verdicts back to the host before the run's one sync
                backend.keep_host[:n_pass].copy_(backend.keep[:n_pass], non_blocking=True)
            stats = st.stats()
        finally:
            st.close()
    n_keep = stats[-1]["n_active_after"] if stats else 0
    return {"n_pass": n_pass, "n_pass_local": n_pass_local, "n_keep": int(n_keep), "stats": stats, "counts": counts, "exchanges": exchanges,
            "allgather_bytes": int(backend.gather.numel() * backend.gather.element_size()), "allreduce_bytes": 4 * sum(n for _, n in exchanges) + 8 * world}


class _HipStepper:
    def __init__(self, stepper):
        self.s = stepper

    def next_pass(self):
        return self.s.next_pass()

    def pass_estimate(self):
        return self.s.pass_estimate()

    def run_replicated(self, world, min_pairs):
        return self.s.run_replicated(world, min_pairs)

    def pass_local(self, rank, world):
        self.s.pass_local(rank, world)

    def n_active(self):
        return self.s.best_ptr()[1]

    def pass_finish(self):
        self.s.pass_finish()

    def stats(self):
        return self.s.stats()

    def copy_mask(self, dst):
        self.s.copy_mask(dst)

    def close(self):
        self.s.close()


class HipShardBackend:
    """The product backend of sharded_step: this rank's block on its MI355X through libtscode_hip."""

    def __init__(self, ens, device_index, rank, world, clash_thresh, max_clashes, rmsd_thr, mode):
        import torch
        self.torch, self.ens = torch, ens
        self.clash_thresh, self.max_clashes, self.rmsd_thr, self.mode = clash_thresh, max_clashes, rmsd_thr, mode
        self.dev = torch.device(f"cuda:{device_index}")
        torch.cuda.set_device(self.dev)
        # An engine (library context) of its own: the stream a context launches on is part of its state, and a context shared
        # with other pipelines or with the drop-in functions (get_engine()) could be switched to another stream between two
        # steps -- kernels and collectives would then no longer be ordered.
        self.eng = Engine(device_index)
        # This library's kernels, torch's copies and the collectives must be ordered on ONE stream.  torch's default stream
        # has handle 0, which tsc_ctx_set_stream reads as "use the library's own stream": a dedicated torch stream is made
        # current for every step instead (torch.distributed orders its collectives against the current stream).
        self.stream = torch.cuda.Stream(device=self.dev)
        self.eng.set_stream(self.stream.cuda_stream)
        self.fs = FragmentSet(ens.frag_coords)
        n = ens.n_poses
        self.lo, self.hi = block_bounds(n, rank, world)
        self.n_local = self.hi - self.lo
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.d_frags = t(self.fs.flat)
        self.d_ci, self.d_rot, self.d_pos = t(ens.conf_idx[self.lo:self.hi]), t(ens.rot[self.lo:self.hi]), t(ens.pos[self.lo:self.hi])
        self.heavy_idx = np.flatnonzero(ens.atomnos != 1).astype(np.int32)
        self.h = h = len(self.heavy_idx)
        na, nl = ens.n_atoms, max(self.n_local, 1)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.clash = torch.empty(nl, dtype=torch.uint8, device=self.dev)
        self.structures = torch.empty((nl, na, 3), **f64)
        self.max_local = (n + world - 1) // world + 1
        self.heavy_pad = torch.zeros((self.max_local, h, 3), **f64)
        self.heavy_local = self.heavy_pad[:nl]                     # this rank's survivors land in the send buffer directly
        self.gather = torch.empty((world * self.max_local, h, 3), **f64)
        self.heavy_all = torch.empty((n, h, 3), **f64)
        self.best = torch.empty(n, dtype=torch.int32, device=self.dev)
        self.keep = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.counts = torch.zeros(world, dtype=torch.int64, device=self.dev)
        self.keep_host = torch.empty(n, dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize(self.dev)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def embed_clash_block(self):
        # fused verdicts, then only the passing poses are embedded: straight into `structures` and into the padded send
        # buffer of the all-gather (heavy_local is a view of it)
        return self.eng.embed_clash_compact_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, self.heavy_idx,
                                                self.clash_thresh, self.max_clashes, self.clash, self.structures, self.heavy_local)

    def make_stepper(self, n_pass):
        st = self.eng.prune_stepper(self.heavy_all, n_pass, self.h, self.rmsd_thr, self.mode)
        st.use_best_buffer(self.best)
        return _HipStepper(st)


class DevicePipeline:
    def __init__(self, ens, device_index=0, rank=0, world=1, clash_thresh=1.5, max_clashes=0, rmsd_thr=0.5, mode=0,
                 process_group=None, force_sharded=False, shard_min_pairs=None):
        import torch
        self.torch, self.ens = torch, ens
        self.rank, self.world, self.pg = int(rank), int(world), process_group
        self.params = (clash_thresh, max_clashes, rmsd_thr, mode)
        self._run = None
        self.sharded = self.world > 1 or force_sharded
        self.shard_min_pairs = shard_min_pairs                  # None: SHARD_MIN_PAIRS (tests lower it to shard small passes too)
        if self.sharded:
            self.backend = HipShardBackend(ens, device_index, self.rank, self.world, clash_thresh, max_clashes, rmsd_thr, mode)
            self.d_keep, self.d_clash, self.d_structures = self.backend.keep, self.backend.clash, self.backend.structures
            self.h_keep = self.backend.keep_host
            return
        self.dev = torch.device(f"cuda:{device_index}")
        torch.cuda.set_device(self.dev)
        self.eng = Engine(device_index)                     # its own context (and so its own stream): see HipShardBackend
        self.fs = FragmentSet(ens.frag_coords)
        n = ens.n_poses
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.d_frags, self.d_ci, self.d_rot, self.d_pos = t(self.fs.flat), t(ens.conf_idx), t(ens.rot), t(ens.pos)
        self.heavy_idx = np.flatnonzero(ens.atomnos != 1).astype(np.int32)
        self.d_clash = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.d_structures = torch.empty((n, ens.n_atoms, 3), dtype=torch.float64, device=self.dev)
        self.d_keep = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.h_keep = torch.empty(n, dtype=torch.uint8).pin_memory()      # verdicts, copied back at the end of every step
        torch.cuda.synchronize(self.dev)

    @property
    def engine(self):
        """The library context this pipeline launches on (its own, not get_engine()'s)."""
        return self.backend.eng if self.sharded else self.eng

    def set_option(self, name, value):
        self.engine.set_option(name, value)

    def step(self):
        """One pass of the hot path over the resident ensemble.  Returns counts/statistics; the verdicts
        stay on the device (d_clash, d_structures, d_keep)."""
        if not self.sharded:
            if self._run is None:      # arguments converted once: the buffers are resident and do not change between steps
                c, m, r, mode = self.params
                self._run = self.eng.pipeline_dev_prepare(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.ens.n_poses,
                                                          self.heavy_idx, c, m, r, mode, self.d_clash, self.d_structures, self.d_keep,
                                                          self.h_keep)
            return self._run()
        return sharded_step(self.backend, self.rank, self.world, self.torch.distributed, self.pg, self.shard_min_pairs)
