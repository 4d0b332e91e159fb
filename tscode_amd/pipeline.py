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

           Step 1-2 have a second form, ``front="replicate"``: every rank embeds and clash-filters ALL poses itself and
           nothing but best[] ever travels.  Moving the survivors' coordinates costs 24 B per heavy atom over xGMI;
           recomputing them costs a few flops and no bytes -- which is faster depends on the node (the all-gather of 350 MB at
           1M conformers is modelled at 0.4-2.9 ms, the embed of all 1M poses measures 0.87 ms), so ``DevicePipeline`` times
           both forms on the node it runs on and keeps the faster one (``tune_front``).

``sharded_step`` only talks to a small backend interface, so the same protocol code runs under gloo on
CPU tensors in the tests (with a test-only backend) and under RCCL on the GPU (``HipShardBackend``).
"""

from __future__ import annotations

import numpy as np

from ._lib import XCHG_MIN_I32, XCHG_SUM_I64
from .engine import Engine, FragmentSet

__all__ = ["DevicePipeline", "HipShardBackend", "CsearchChain", "ShardedCsearchChain", "sharded_step", "block_bounds", "partition_bounds", "SHARD_MIN_PAIRS",
           "PARTITION_MIN_CHUNKS", "SHARDED_CULL_MIN_PAIRS"]

# A pass smaller than this many pairs (estimate n * (n / k) / 2, identical on every rank) is not worth a collective:
# every rank runs it whole and reaches the same verdicts on its own.  On MI355X such a pass takes tens of
# microseconds, the same as one small all-reduce over xGMI.
SHARD_MIN_PAIRS = 50_000_000


# A pass with at least this many chunks per rank is PARTITIONED: every rank runs the whole pass on the chunks that start inside its
# block of the structure axis (chunks are independent, tscode/rmsd_pruning.py:139-157) and the ranks exchange one bit per structure.
# With fewer chunks per rank a rank's share differs from another's by up to a whole chunk in PARTITION_MIN_CHUNKS: those passes deal
# their ROW TILES round-robin instead and exchange best[].  2 since round 4 (4 before): a pass dealt by row tiles opens and applies ALL
# rows on every rank, a partitioned one only its own -- in the one-GPU model of an 8-rank C4 step the k = 20 pass (2.5 chunks per rank,
# 3 on the fullest) costs a rank 0.40 ms partitioned against 0.50 dealt by tiles, and the passes behind it get cheaper with it
# (tools/predict_scaling.py --chunks; profiles/r04_predicted_scaling.json).  With 1 the k = 10 pass on eight ranks leaves two of them idle: no gain.
PARTITION_MIN_CHUNKS = 2

# A sharded run culls a pass (cull.hpp: sorted layout + bounding boxes) from this many pairs of a rank's SHARE on, where one GPU on its own
# waits for 2e9 ("cull_min_pairs"): the walk's cost falls with the share, a layout's does not, but the shares of C4's five largest passes on
# eight ranks (0.3 - 1.5e9 pairs) still lie where the layout pays -- 0.5 ms of a 4.4 ms step in the same model.  The share of a pass
# PARTITIONED BY CHUNKS is compared with this number as it stands; for a pass dealt by ROW TILES the library doubles it
# (csrc/prune.hip, pass_launch: every rank lays the whole pass out for its fraction of the tiles), i.e. such a pass is culled from 1e9 pairs
# of a rank's share on.
SHARDED_CULL_MIN_PAIRS = 5.0e8


def partition_bounds(n: int, k: int, rank: int, world: int):
    """Chunks [c_lo, c_hi) of a pass of k chunks that start inside rank's block of the structure axis and the structures
    [s_lo, s_hi) they cover (mirror of the library's partition_bounds, csrc/prune_host.hpp)."""
    cs = n // k

    def first_chunk(r):
        return 0 if r <= 0 else (k if r >= world else min(-(-(n * r // world) // cs), k))
    c_lo, c_hi = first_chunk(rank), first_chunk(rank + 1)
    return c_lo, c_hi, (c_lo * cs if c_lo < k else n), (c_hi * cs if c_hi < k else n)


FRONTS = ("shard", "replicate", "hybrid")


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


def sharded_step(backend, rank: int, world: int, dist, group=None, min_pairs=None, front="shard", partition_chunks=PARTITION_MIN_CHUNKS):
    """One step of the hot path over an ensemble sharded across ``world`` ranks, run inside ``backend.stream_context()`` when the
    backend has one (the HIP backend makes its own torch stream current, so that its kernels, torch's copies and the
    collectives are ordered on one stream).  ``front``: "shard" = pose blocks + all-gather of the survivors' coordinates,
    "replicate" = every rank computes the whole front half, no coordinates travel (module docstring)."""
    if front not in FRONTS:
        raise ValueError(f"front must be one of {FRONTS}, got {front!r}")
    enter = getattr(backend, "stream_context", None)
    if enter is None:
        return _sharded_step(backend, rank, world, dist, group, min_pairs, front, partition_chunks)
    with enter():
        return _sharded_step(backend, rank, world, dist, group, min_pairs, front, partition_chunks)


def _front_sharded(backend, rank, world, dist, group):
    """Pose blocks -> counts -> all-gather of the heavy-atom shards; returns (n_pass, n_pass_local, counts, all-gathered bytes)."""
    n_pass_local = int(backend.embed_clash_block())
    # how many poses of every block passed the clash check
    backend.counts.zero_()
    backend.counts[rank] = n_pass_local
    _all_reduce(dist, backend.counts, dist.ReduceOp.SUM, group)
    counts = [int(c) for c in backend.counts.cpu().tolist()]
    # the one exchange of coordinates: shards padded to the LARGEST count (every rank knows all counts by now; blocks of a
    # shuffled pose list pass the clash check at nearly the same rate, so the padding is a per cent or two -- padding to the
    # block size would move the rejected poses' share as well: 720 MB instead of 350 at 1M conformers), one all-gather, then
    # un-pad in block order
    if backend.heavy_pad.data_ptr() != backend.heavy_local.data_ptr():
        backend.heavy_pad[:n_pass_local].copy_(backend.heavy_local[:n_pass_local])
    rows = max(max(counts), 1)
    row_elems = backend.heavy_pad[0].numel()
    _all_gather(dist, backend.gather.view(-1)[:world * rows * row_elems], backend.heavy_pad.view(-1)[:rows * row_elems], group)
    gathered = backend.gather.view(-1)[:world * rows * row_elems].view(world, rows, *backend.heavy_pad.shape[1:])
    off = 0
    for r, c in enumerate(counts):
        backend.heavy_all[off:off + c].copy_(gathered[r, :c])
        off += c
    return off, n_pass_local, counts, int(world * rows * row_elems * backend.gather.element_size())


def _front_hybrid(backend, rank, world, dist, group):
    """Clash verdicts of this rank's block -> the verdicts of all blocks (one byte per pose: all-reduce SUM of a mask that is zero
    outside the rank's block) -> every rank embeds the heavy atoms of ALL passing poses itself.  Returns (n_pass, mask bytes moved)."""
    backend.clash_block_into_all()            # (also forks the descriptor basis onto the side stream, embeds this block's survivors)
    xchg = getattr(backend, "xchg", None) if world > 1 else None
    words = getattr(backend, "clash_all_words", None)
    if xchg is not None and words is not None:
        # the library's own exchange: the byte mask is zero outside this rank's block, so the SUM of its 8-byte words over the ranks is
        # the union of the blocks' verdicts -- the one collective of this front then needs no collective library either
        xchg.allreduce(XCHG_SUM_I64, words, words.numel())
    else:
        _all_reduce(dist, backend.clash_all, dist.ReduceOp.SUM, group)
    return int(backend.embed_masked_all()), int(backend.clash_all.numel())


def _sharded_step(backend, rank: int, world: int, dist, group, min_pairs, front="shard", partition_chunks=PARTITION_MIN_CHUNKS):
    """The protocol itself.

    backend interface (all tensors live where the backend computes):
        embed_clash_block() -> n_pass_local      fills backend.heavy_local[:n_pass_local]  (h, 3 per row)
        embed_clash_all() -> n_pass              (front="replicate" only) fills backend.heavy_all[:n_pass] from ALL poses
        clash_block_into_all(), clash_all, embed_masked_all() -> n_pass   (front="hybrid" only): verdicts of this rank's block into
                                                 the all-poses mask (zero elsewhere); heavy_all[:n_pass] from the summed mask
        heavy_pad, gather, heavy_all, counts, keep, max_local, best      preallocated tensors / int
        make_stepper(n_pass) -> stepper with next_pass(), pass_estimate(), pass_local(rank, world), n_active(), pass_finish(),
                                stats(), copy_mask(dst), close(); it keeps best[] in backend.best.  Optional (partitioned passes,
                                ``partition_chunks`` > 0): set_partition(rank, world, min_chunks) -- the stepper then keeps its
                                exchange buffer in backend.exch (int64) --, pass_partitioned(), pass_range(), exchange_words(),
                                pass_merge(), views_range() -> (offset, words) into backend.exch, views_merged()
    """
    mask_bytes = 0
    if front == "replicate":
        n_pass = int(backend.embed_clash_all())
        n_pass_local, counts, gathered_bytes = n_pass, [n_pass], 0
    elif front == "hybrid":
        n_pass, mask_bytes = _front_hybrid(backend, rank, world, dist, group)
        n_pass_local, counts, gathered_bytes = n_pass, [n_pass], 0
    else:
        n_pass, n_pass_local, counts, gathered_bytes = _front_sharded(backend, rank, world, dist, group)
    stats = []
    exchanges = []                                      # (k, entries of best[] all-reduced) of every pass that was sharded by row tiles
    partitioned = []                                    # (k, int64 words summed) of every pass that was partitioned by chunks
    views_words = 0
    if n_pass > 0:
        st = backend.make_stepper(n_pass)
        limit = SHARD_MIN_PAIRS if min_pairs is None else min_pairs
        can_partition = world > 1 and partition_chunks > 0 and hasattr(st, "set_partition")
        in_library = hasattr(st, "run_sharded") and not getattr(backend, "python_pass_loop", False)
        if can_partition and not in_library:
            st.set_partition(rank, world, partition_chunks)
        # The per-pass exchanges: the library's own (backend.xchg: areas the ranks map into each other, one-shot all-reduce kernels on the
        # step's stream -- IpcExchange) where the backend has connected one, else torch.distributed collectives (RCCL) on the same buffers
        xchg = getattr(backend, "xchg", None) if world > 1 else None

        def reduce_sum(t):
            if xchg is not None:
                xchg.allreduce(XCHG_SUM_I64, t, t.numel())
            else:
                _all_reduce(dist, t, dist.ReduceOp.SUM, group)

        def reduce_min(t):
            if xchg is not None:
                xchg.allreduce(XCHG_MIN_I32, t, t.numel())
            else:
                _all_reduce(dist, t, dist.ReduceOp.MIN, group)
        try:
            if in_library:
                # the product backend: the pass loop lives behind the C ABI (tsc_prune_run_sharded); the host is called back for the
                # collectives only -- one Python frame per collective instead of three library calls + this loop per pass -- or, with
                # the library's own exchange, not at all
                def exchange(kind, addr, count):
                    if kind == XCHG_SUM_I64:
                        off = (addr - backend.exch.data_ptr()) // 8
                        _all_reduce(dist, backend.exch[off:off + count], dist.ReduceOp.SUM, group)
                    else:
                        assert addr == backend.best.data_ptr()
                        _all_reduce(dist, backend.best[:count], dist.ReduceOp.MIN, group)
                for k, kind, count in st.run_sharded(rank, world, partition_chunks if can_partition else 0, limit, xchg if xchg is not None else exchange):
                    if kind == XCHG_MIN_I32:
                        exchanges.append((int(k), int(count)))
                    elif k > 0:
                        partitioned.append((int(k), int(count)))
                    else:
                        views_words += int(count)
            while not in_library:                       # the same loop on the host: test backends, and the product's with python_pass_loop set
                if hasattr(st, "run_replicated"):       # the small passes in one library call; back here for an exchange
                    k = st.run_replicated(world, limit)
                    if k == 0:
                        break
                    shard = st.pass_estimate() >= limit
                else:
                    k = st.next_pass()
                    if k == 0:
                        break
                    shard = world > 1 and st.pass_estimate() >= limit
                if can_partition and st.pass_partitioned():
                    # the whole pass on this rank's chunks; what the ranks tell each other is which rows they removed
                    st.pass_range()
                    words = st.exchange_words()
                    reduce_sum(backend.exch[:words])
                    st.pass_merge()
                    partitioned.append((int(k), int(words)))
                    continue
                if can_partition:
                    # first pass after the partitioned ones: every rank needs every rank's cache keys from here on
                    off, words = st.views_range()
                    if words:
                        reduce_sum(backend.exch[off:off + words])
                        views_words += int(words)
                    st.views_merged()
                if shard:
                    st.pass_local(rank, world)          # this rank's row tiles only ...
                    n_best = st.n_active()
                    reduce_min(backend.best[:n_best])   # ... merged
                    exchanges.append((int(k), int(n_best)))
                else:
                    st.pass_local(0, 1)                 # small pass: replicated, no exchange
                st.pass_finish()
            st.copy_mask(backend.keep)
            if getattr(backend, "keep_host", None) is not None:      # verdicts back to the host before the run's one sync
                backend.keep_host[:n_pass].copy_(backend.keep[:n_pass], non_blocking=True)
            stats = st.stats()
            if xchg is not None:
                xchg.check_status()                     # (after the run's synchronisation: did every exchange hear from every peer?)
        finally:
            st.close()
    n_keep = stats[-1]["n_active_after"] if stats else 0
    return {"n_pass": n_pass, "n_pass_local": n_pass_local, "n_keep": int(n_keep), "stats": stats, "counts": counts, "exchanges": exchanges,
            "partitioned": partitioned, "front": front, "allgather_bytes": gathered_bytes,
            "allreduce_bytes": 4 * sum(n for _, n in exchanges) + 8 * sum(w for _, w in partitioned) + 8 * views_words
                               + (8 * world if front == "shard" else 0) + mask_bytes,
            "exchange": "ipc" if (world > 1 and getattr(backend, "xchg", None) is not None) else "callback"}


class _HipStepper:
    def __init__(self, stepper, exch=None):
        self.s, self.exch = stepper, exch

    def next_pass(self):
        return self.s.next_pass()

    def pass_estimate(self):
        return self.s.pass_estimate()

    def run_replicated(self, world, min_pairs):
        return self.s.run_replicated(world, min_pairs)

    def run_sharded(self, rank, world, min_chunks, min_pairs, exchange):
        return self.s.run_sharded(rank, world, min_chunks, min_pairs, self.exch, exchange)

    def pass_local(self, rank, world):
        self.s.pass_local(rank, world)

    def pass_rows(self, rank, world):
        self.s.pass_rows(rank, world)

    def set_partition(self, rank, world, min_chunks):
        self.s.set_partition(rank, world, min_chunks, self.exch)

    def pass_partitioned(self):
        return self.s.pass_partitioned()

    def pass_range(self):
        self.s.pass_range()

    def exchange_words(self):
        return self.s.n // 64 + 48

    def pass_merge(self):
        self.s.pass_merge()

    def views_range(self):
        return self.s.views_range()

    def views_merged(self):
        self.s.views_merged()

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
        self.xchg = None                       # the library's own per-pass exchange (connect_exchange); None: torch.distributed collectives
        self.clash_thresh, self.max_clashes, self.rmsd_thr, self.mode = clash_thresh, max_clashes, rmsd_thr, mode
        self.dev = torch.device(f"cuda:{device_index}")
        torch.cuda.set_device(self.dev)
        # An engine (library context) of its own: the stream a context launches on is part of its state, and a context shared
        # with other pipelines or with the drop-in functions (get_engine()) could be switched to another stream between two
        # steps -- kernels and collectives would then no longer be ordered.
        self.eng = Engine(device_index)
        # every rank must derive bit-identical descriptors (the culled passes deal the tiles of a layout sorted by them): fixed-order sums
        self.eng.set_option("deterministic_basis", 1 if world > 1 else 0)
        if world > 1:
            self.eng.set_option("cull_min_pairs", SHARDED_CULL_MIN_PAIRS)
        # This library's kernels, torch's copies and the collectives must be ordered on ONE stream.  torch's default stream
        # has handle 0, which tsc_ctx_set_stream reads as "use the library's own stream": a dedicated torch stream is made
        # current for every step instead (torch.distributed orders its collectives against the current stream).
        self.stream = torch.cuda.Stream(device=self.dev)
        self.eng.set_stream(self.stream.cuda_stream)
        self.fs = FragmentSet(ens.frag_coords)
        n = ens.n_poses
        self.world = int(world)
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
        # exchange buffer of the partitioned passes (removed-row bits + statistics, then the run's cache views), sized for all n poses
        from .engine import PruneStepper
        self.exch = torch.zeros(PruneStepper.exchange_words(self.eng.lib, max(n, 1), 0), dtype=torch.int64, device=self.dev)
        self.keep_host = torch.empty(n, dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize(self.dev)

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def connect_exchange(self, dist, group=None):
        """The per-pass exchanges of the prune inside the library from here on (engine.IpcExchange: every rank's receive area mapped into
        every other rank of the node, one-shot all-reduce kernels on this backend's stream).  Collective: every rank calls it.  The
        process group only carries the 64-byte handles, once.  Returns the exchange; on failure nothing changes and the error
        propagates (the caller may stay with the callback form)."""
        from .engine import IpcExchange
        if self.world <= 1 or self.xchg is not None:
            return self.xchg
        x = IpcExchange(self.eng, dist.get_rank(group), self.world, IpcExchange.slot_bytes(self.eng.lib, max(self.ens.n_poses, 1), self.mode))
        x.connect_over(dist, group)
        self.xchg = x
        return x

    def disconnect_exchange(self):
        if self.xchg is not None:
            self.torch.cuda.synchronize(self.dev)
            self.xchg.close()
            self.xchg = None

    def embed_clash_block(self):
        # fused verdicts, then only the passing poses are embedded: straight into `structures` and into the padded send
        # buffer of the all-gather (heavy_local is a view of it).
        # With several ranks the descriptor basis must NOT come from this rank's block: the culled passes of the prune deal the tiles
        # of a layout sorted by descriptor among the ranks, so every rank needs bit-identical descriptors -- the basis is then
        # estimated by tsc_prune_create from the all-gathered survivors, the same sample on every rank (csrc/cull.hpp)
        self.eng.set_option("early_basis", 1 if self.world == 1 else 0)
        try:
            return self.eng.embed_clash_compact_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, self.heavy_idx,
                                                    self.clash_thresh, self.max_clashes, self.clash, self.structures, self.heavy_local)
        finally:
            self.eng.set_option("early_basis", 1)

    def embed_clash_all(self):
        """front="replicate": the whole pose list on this rank; the survivors' heavy atoms land in heavy_all directly.  The
        full-size inputs and outputs (clash_all, structures_all) are set up on first use."""
        self._all_inputs()
        if getattr(self, "structures_all", None) is None:
            self.structures_all = self.torch.empty((self.ens.n_poses, self.ens.n_atoms, 3), dtype=self.torch.float64, device=self.dev)
        return self.eng.embed_clash_compact_dev(self.fs, self.d_frags, self.d_ci_all, self.d_rot_all, self.d_pos_all, self.ens.n_poses,
                                                self.heavy_idx, self.clash_thresh, self.max_clashes, self.clash_all, self.structures_all,
                                                self.heavy_all)

    def _all_inputs(self):
        if getattr(self, "d_ci_all", None) is None:
            torch, ens = self.torch, self.ens
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
            self.d_ci_all, self.d_rot_all, self.d_pos_all = t(ens.conf_idx), t(ens.rot), t(ens.pos)
            # (padded to whole 8-byte words: the library's exchange sums the mask as int64 -- clash_all_words is the same memory)
            self.clash_all_words = torch.zeros((ens.n_poses + 7) // 8, dtype=torch.int64, device=self.dev)
            self.clash_all = self.clash_all_words.view(torch.uint8)[:ens.n_poses]

    def clash_block(self):
        """The clash verdicts of this rank's block alone (no pose is written)."""
        self.eng.embed_clash_mask_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, self.clash_thresh, self.max_clashes,
                                      self.clash)

    def clash_block_into_all(self):
        """front="hybrid", before the exchange: the descriptor basis forked onto the side stream (from a sample of ALL poses), this
        block's verdicts written into the all-poses mask, which is zero elsewhere, and this block's passing poses embedded with all
        their atoms (``structures``: what the caller gets back; only the heavy atoms of the OTHER blocks' poses are ever needed here)."""
        self._all_inputs()
        self.eng.basis_from_poses_dev(self.fs, self.d_frags, self.d_ci_all, self.d_rot_all, self.d_pos_all, self.ens.n_poses, self.heavy_idx)
        self.clash_all.zero_()
        block = self.clash_all[self.lo:self.hi]
        if self.n_local:
            self.eng.embed_clash_mask_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, self.clash_thresh, self.max_clashes,
                                          block)
            self.eng.embed_masked_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, block, self.heavy_idx, self.structures,
                                      None, want_count=False)

    def embed_masked_all(self):
        """front="hybrid", after the exchange: the heavy atoms (and the prune's descriptors) of every passing pose, on this rank."""
        self._all_inputs()
        return self.eng.embed_masked_dev(self.fs, self.d_frags, self.d_ci_all, self.d_rot_all, self.d_pos_all, self.ens.n_poses, self.clash_all,
                                         self.heavy_idx, None, self.heavy_all)

    def make_stepper(self, n_pass):
        st = self.eng.prune_stepper(self.heavy_all, n_pass, self.h, self.rmsd_thr, self.mode)
        st.use_best_buffer(self.best)
        return _HipStepper(st, self.exch)


def time_fronts(torch, dev, pg, steps, step):
    """``step(form)`` timed for every form of the front half: one untimed call, then ``steps`` timed ones between a barrier and a
    device synchronisation, the slowest rank's time counting.  Returns {"ms_per_step": {form: ms}, "chosen": fastest}."""
    import time
    dist = torch.distributed
    times = {}
    for form in FRONTS:
        step(form)
        torch.cuda.synchronize(dev)
        dist.barrier(group=pg)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(form)
        torch.cuda.synchronize(dev)
        t = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64)
        if dist.get_backend(pg) == "gloo":
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
        else:                                   # nccl (= RCCL) reduces device tensors
            td = t.to(dev)
            dist.all_reduce(td, op=dist.ReduceOp.MAX, group=pg)
            t = td.cpu()
        times[form] = float(t[0])
    return {"ms_per_step": times, "chosen": min(times, key=times.get)}


class DevicePipeline:
    def __init__(self, ens, device_index=0, rank=0, world=1, clash_thresh=1.5, max_clashes=0, rmsd_thr=0.5, mode=0,
                 process_group=None, force_sharded=False, shard_min_pairs=None, front="auto", partition_chunks=PARTITION_MIN_CHUNKS,
                 exchange="callback"):
        """``exchange`` (multi-rank runs): "callback" = the per-pass collectives are torch.distributed all-reduces (RCCL) the library calls back
        for; "ipc" = the library's own exchange over areas the ranks map into each other (``HipShardBackend.connect_exchange``; the process
        group only carries the handles).  Same results either way.
        ``front`` (multi-rank runs): "shard" = pose blocks + all-gather of the survivors' coordinates, "replicate" = every rank
        computes the whole front half and only best[] travels, "auto" = ``tune_front()`` decides on the node at hand (called by
        the first ``step()`` unless the caller did; one rank: "shard", there is nothing to choose)."""
        import torch
        self.torch, self.ens = torch, ens
        self.rank, self.world, self.pg = int(rank), int(world), process_group
        self.params = (clash_thresh, max_clashes, rmsd_thr, mode)
        self._run = None
        self.sharded = self.world > 1 or force_sharded
        self.shard_min_pairs = shard_min_pairs                  # None: SHARD_MIN_PAIRS (tests lower it to shard small passes too)
        self.partition_chunks = int(partition_chunks)           # chunks per rank from which a pass is partitioned by chunks (0: never)
        if front != "auto" and front not in FRONTS:
            raise ValueError(f"front must be 'auto' or one of {FRONTS}, got {front!r}")
        self.front = front if self.world > 1 or front != "auto" else "shard"
        self.front_tuning = None
        if self.sharded:
            self.backend = HipShardBackend(ens, device_index, self.rank, self.world, clash_thresh, max_clashes, rmsd_thr, mode)
            if exchange not in ("callback", "ipc"):
                raise ValueError(f"exchange must be 'callback' or 'ipc', got {exchange!r}")
            if exchange == "ipc" and self.world > 1:
                self.backend.connect_exchange(torch.distributed, process_group)
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
        if self.front == "auto":
            self.tune_front()
        return sharded_step(self.backend, self.rank, self.world, self.torch.distributed, self.pg, self.shard_min_pairs, self.front,
                            self.partition_chunks)

    def tune_front(self, steps=2):
        """Times the forms of the front half on this node -- one untimed step each (buffers, communicators), then ``steps`` timed
        ones each, the slowest rank's time counting (all-reduce MAX) -- and keeps the fastest.  Every rank reaches the same
        decision.  All forms give the same result, so the tuning steps are ordinary steps."""
        if not self.sharded:
            return None
        self.front_tuning = time_fronts(self.torch, self.backend.dev, self.pg, steps,
                                        lambda form: sharded_step(self.backend, self.rank, self.world, self.torch.distributed, self.pg,
                                                                  self.shard_min_pairs, form, self.partition_chunks))
        self.front = self.front_tuning["chosen"]
        return self.front_tuning


class CsearchChain:
    """BASELINE config 5 as ONE chain on the device: conformational-search rotations of a fragment
    (tscode/torsion_module.py:463-509: every angle set of the table rotates the fragment's torsions, walking back in 5-degree
    steps where a rotation clashes; a candidate is kept iff at least one bond really rotated, up to ``n_out``) -> the kept
    candidates ARE the conformer stack of that fragment -> embed -> clash mask -> compaction -> prune_conformers_rmsd.

    The candidate array (``n_candidates x n_atoms x 24`` B: 480 MB for 100 000 candidates of a 200-atom complex) never leaves
    the GPU: ``tsc_csearch_rotate_dev`` writes it, ``tsc_compact_rows_dev`` packs the kept rows in order into the fragment
    buffer the pipeline reads its conformers from.  The one number that returns to the host between the two halves is how
    many conformers were kept (the pose list indexes them)."""

    def __init__(self, ens, torsions, masks, angles, n_out=None, fragment=0, thresh=1.5, device_index=0, clash_thresh=1.5, max_clashes=0,
                 rmsd_thr=0.5, mode=0, seed=0):
        import torch
        self.torch, self.ens, self.fragment = torch, ens, fragment
        self.params = (clash_thresh, max_clashes, rmsd_thr, mode)
        self.thresh = thresh
        self.dev = torch.device(f"cuda:{device_index}")
        torch.cuda.set_device(self.dev)
        self.eng = Engine(device_index)
        self.stream = torch.cuda.Stream(device=self.dev)          # torch ops (a modulo, a compare) run between the two library calls
        self.eng.set_stream(self.stream.cuda_stream)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        base = np.ascontiguousarray(ens.frag_coords[fragment][0], dtype=np.float64)
        self.n0 = base.shape[0]
        self.torsions_h = np.ascontiguousarray(torsions, dtype=np.int32).reshape(-1, 4)
        self.masks_h = np.ascontiguousarray(masks, dtype=np.uint8).reshape(len(self.torsions_h), self.n0)
        self.angles_h = np.ascontiguousarray(angles, dtype=np.int32).reshape(-1, len(self.torsions_h))
        self.n_cand = len(self.angles_h)
        self.n_out = self.n_cand if n_out is None else int(n_out)
        self.d_base, self.d_tors, self.d_masks, self.d_angles = t(base), t(self.torsions_h), t(self.masks_h), t(self.angles_h)
        self.d_cand = torch.empty((self.n_cand, self.n0, 3), dtype=torch.float64, device=self.dev)
        self.d_rb = torch.empty(self.n_cand, dtype=torch.int32, device=self.dev)
        # fragment buffer: the searched fragment first (its conformer stack is rewritten every step), the others behind it
        self.others = [np.ascontiguousarray(f, dtype=np.float64) for m, f in enumerate(ens.frag_coords) if m != fragment]
        self.order = [fragment] + [m for m in range(len(ens.frag_coords)) if m != fragment]
        other_flat = np.concatenate([f.ravel() for f in self.others]) if self.others else np.zeros(0)
        self.cap0 = self.n_cand * self.n0 * 3
        self.d_frags = torch.empty(self.cap0 + len(other_flat), dtype=torch.float64, device=self.dev)
        self.d_frags[self.cap0:].copy_(t(other_flat))
        n = ens.n_poses
        rng = np.random.default_rng(seed)
        self.d_draw = t(rng.integers(0, 2 ** 30, size=n).astype(np.int64))       # which conformer a pose uses: draw % n_kept
        # poses list the fragments in the buffer's order
        self.d_ci = t(np.ascontiguousarray(ens.conf_idx[:, self.order]))
        self.d_rot, self.d_pos = t(np.ascontiguousarray(ens.rot[:, self.order])), t(np.ascontiguousarray(ens.pos[:, self.order]))
        atomnos = np.concatenate([np.asarray(ens.atomnos)[sl] for sl in self._atom_slices()])
        self.atomnos = atomnos
        self.heavy_idx = np.flatnonzero(atomnos != 1).astype(np.int32)
        self.d_clash = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.d_structures = torch.empty((n, ens.n_atoms, 3), dtype=torch.float64, device=self.dev)
        self.d_keep = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.h_keep = torch.empty(n, dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize(self.dev)

    @property
    def engine(self):
        return self.eng

    def set_option(self, name, value):
        self.eng.set_option(name, value)

    @staticmethod
    def chain_torsions(n_atoms, n_torsions, seed=0):
        """Torsions along a chain-like fragment (the synthetic fragments are self-avoiding walks: atom a is bonded to a + 1):
        ``(c - 1, c, c + 1, c + 2)`` about random bonds ``c - c + 1``, everything behind the bond rotates (what
        tscode/torsion_module.py:301-325 ``_get_rotation_mask`` gives for a chain).  Returns (torsions i32[T, 4], masks u8[T, n])."""
        rng = np.random.default_rng(seed)
        centres = np.sort(rng.choice(np.arange(2, n_atoms - 3), size=n_torsions, replace=False))
        torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres], dtype=np.int32)
        masks = np.zeros((n_torsions, n_atoms), dtype=np.uint8)
        for t, c in enumerate(centres):
            masks[t, c + 1:] = 1
        return torsions, masks

    def _atom_slices(self):
        off = np.concatenate([[0], np.cumsum([f.shape[1] for f in self.ens.frag_coords])])
        return [slice(int(off[m]), int(off[m + 1])) for m in self.order]

    def step(self):
        torch = self.torch
        c, m, r, mode = self.params
        with torch.cuda.stream(self.stream):
            self.eng.csearch_rotate_dev(self.d_base, self.n0, self.d_tors, self.d_masks, len(self.torsions_h), self.d_angles, self.n_cand,
                                        self.thresh, 0, self.d_cand, self.d_rb)
            kept_mask = (self.d_rb != 0).to(torch.uint8)                                      # torsion_module.py:505
            n_kept = self.eng.compact_rows_dev(self.d_cand, kept_mask, self.n_cand, self.n0 * 24, self.d_frags)
            n_kept = min(n_kept, self.n_out)                                                  # :510: stop at n_out structures
            if n_kept == 0:
                return {"n_conformers": 0, "n_pass": 0, "n_keep": 0, "stats": []}
            self.d_ci[:, 0] = (self.d_draw % n_kept).to(torch.int32)
            sizes = [(n_kept, self.n0)] + [(f.shape[0], f.shape[1]) for f in self.others]
            fs = FragmentSet.__new__(FragmentSet)
            fs.n_mols = len(sizes)
            fs.n_atoms = np.array([s[1] for s in sizes], dtype=np.int32)
            fs.n_conf = np.array([s[0] for s in sizes], dtype=np.int32)
            offs, o = [0], self.cap0
            for f in self.others:
                offs.append(o)
                o += f.size
            fs.frag_off = np.array(offs[:fs.n_mols], dtype=np.int64)
            fs.n_total = int(fs.n_atoms.sum())
            res = self.eng.pipeline_dev(fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.ens.n_poses, self.heavy_idx, c, m, r, mode,
                                        self.d_clash, self.d_structures, self.d_keep, self.h_keep)
        return {"n_conformers": n_kept, "n_pass": res["n_pass"], "n_keep": res["n_keep"], "stats": res["stats"]}


class ShardedCsearchChain:
    """BASELINE config 5 as a chain over SEVERAL ranks (one process per GPU): the conformational search is cut into blocks of the
    angle table, the kept candidates of all blocks are put together IN TABLE ORDER -- the reference keeps candidates in the order it
    generates them and stops at ``n_out`` (tscode/torsion_module.py:505-510), and a pose names its conformer by that position -- and
    the poses then run through the sharded pipeline (``sharded_step``).

        1. rank r rotates the angle sets [n_cand r / W, n_cand (r + 1) / W) (tsc_csearch_rotate_dev, walk-back loop included) and
           packs the kept ones in order (tsc_compact_rows_dev);
        2. counts all-reduce (8 B per rank) + ONE all-gather of the packed blocks, padded to the largest block (n_atoms x 24 B per
           candidate: 34 MB for 20 000 candidates of a 70-atom fragment); every rank un-pads in rank order into the conformer stack
           of the searched fragment, cut at n_out;
        3. conformer index of every pose = draw % n_kept (as in CsearchChain);
        4. sharded_step on the poses: front half as ``front`` says, prune passes partitioned by chunks / sharded by row tiles.

    Same result as CsearchChain on one GPU by construction; tests/test_gpu_parity.py runs two ranks against the recorded oracle run."""

    def __init__(self, ens, torsions, masks, angles, rank, world, process_group=None, n_out=None, fragment=0, thresh=1.5, device_index=0,
                 clash_thresh=1.5, max_clashes=0, rmsd_thr=0.5, mode=0, seed=0, front="auto", shard_min_pairs=None,
                 partition_chunks=PARTITION_MIN_CHUNKS, exchange="callback"):
        import copy

        import torch
        self.torch, self.rank, self.world, self.pg = torch, int(rank), int(world), process_group
        self.thresh, self.shard_min_pairs, self.partition_chunks = thresh, shard_min_pairs, int(partition_chunks)
        if front != "auto" and front not in FRONTS:
            raise ValueError(f"front must be 'auto' or one of {FRONTS}, got {front!r}")
        self.front, self.front_tuning = front, None
        # the searched fragment first (its conformer stack is rewritten every step), the others behind it: an ensemble in that order
        order = [fragment] + [m for m in range(len(ens.frag_coords)) if m != fragment]
        off = np.concatenate([[0], np.cumsum([f.shape[1] for f in ens.frag_coords])])
        atom_order = np.concatenate([np.arange(off[m], off[m + 1]) for m in order])
        e2 = copy.copy(ens)
        e2.frag_coords = [ens.frag_coords[m] for m in order]
        e2.conf_idx, e2.rot, e2.pos = (np.ascontiguousarray(a[:, order]) for a in (ens.conf_idx, ens.rot, ens.pos))
        e2.ids, e2.atomnos = np.asarray(ens.ids)[order], np.asarray(ens.atomnos)[atom_order]
        self.ens = e2
        self.backend = be = HipShardBackend(e2, device_index, self.rank, self.world, clash_thresh, max_clashes, rmsd_thr, mode)
        if exchange not in ("callback", "ipc"):
            raise ValueError(f"exchange must be 'callback' or 'ipc', got {exchange!r}")
        if exchange == "ipc" and self.world > 1:
            be.connect_exchange(torch.distributed, process_group)
        self.eng, self.dev, self.stream = be.eng, be.dev, be.stream
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        base = np.ascontiguousarray(e2.frag_coords[0][0], dtype=np.float64)
        self.n0 = base.shape[0]
        tors = np.ascontiguousarray(torsions, dtype=np.int32).reshape(-1, 4)
        self.n_tors = len(tors)
        angles = np.ascontiguousarray(angles, dtype=np.int32).reshape(-1, self.n_tors)
        self.n_cand = len(angles)
        self.n_out = self.n_cand if n_out is None else int(n_out)
        self.c_lo, self.c_hi = block_bounds(self.n_cand, self.rank, self.world)
        self.max_block = (self.n_cand + self.world - 1) // self.world + 1
        self.d_base, self.d_tors, self.d_masks = t(base), t(tors), t(np.ascontiguousarray(masks, dtype=np.uint8).reshape(self.n_tors, self.n0))
        self.d_angles = t(angles[self.c_lo:self.c_hi])
        nb = max(self.c_hi - self.c_lo, 1)
        self.d_cand = torch.empty((nb, self.n0, 3), dtype=torch.float64, device=self.dev)
        self.d_rb = torch.empty(nb, dtype=torch.int32, device=self.dev)
        self.d_send = torch.zeros((self.max_block, self.n0, 3), dtype=torch.float64, device=self.dev)
        self.d_gather = torch.empty((self.world * self.max_block, self.n0, 3), dtype=torch.float64, device=self.dev)
        self.d_counts = torch.zeros(self.world, dtype=torch.int64, device=self.dev)
        # fragment buffer of the pipeline: room for every candidate as a conformer of the searched fragment, the others behind it
        others = [np.ascontiguousarray(f, dtype=np.float64) for f in e2.frag_coords[1:]]
        self.cap0 = self.n_cand * self.n0 * 3
        other_flat = np.concatenate([f.ravel() for f in others]) if others else np.zeros(0)
        be.d_frags = torch.empty(self.cap0 + len(other_flat), dtype=torch.float64, device=self.dev)
        be.d_frags[self.cap0:].copy_(t(other_flat))
        self._other_sizes = [(f.shape[0], f.shape[1], f.size) for f in others]
        rng = np.random.default_rng(seed)
        draw = rng.integers(0, 2 ** 30, size=e2.n_poses).astype(np.int64)       # which conformer a pose uses: draw % n_kept
        self.d_draw_all = t(draw)
        be._all_inputs()                                                          # (the all-poses inputs of the replicate / hybrid fronts)
        self.d_keep, self.d_clash, self.d_structures, self.h_keep = be.keep, be.clash, be.structures, be.keep_host
        torch.cuda.synchronize(self.dev)

    @property
    def engine(self):
        return self.eng

    def set_option(self, name, value):
        self.eng.set_option(name, value)

    def _search(self):
        """Steps 1-3; returns the number of conformers (0: nothing was kept anywhere)."""
        torch, be, dist = self.torch, self.backend, self.torch.distributed
        nb = self.c_hi - self.c_lo
        n_local = 0
        if nb > 0:
            self.eng.csearch_rotate_dev(self.d_base, self.n0, self.d_tors, self.d_masks, self.n_tors, self.d_angles, nb, self.thresh, 0, self.d_cand,
                                        self.d_rb)
            kept_mask = (self.d_rb[:nb] != 0).to(torch.uint8)                                  # torsion_module.py:505
            n_local = self.eng.compact_rows_dev(self.d_cand, kept_mask, nb, self.n0 * 24, self.d_send)
        self.d_counts.zero_()
        self.d_counts[self.rank] = n_local
        _all_reduce(dist, self.d_counts, dist.ReduceOp.SUM, self.pg)
        counts = [int(c) for c in self.d_counts.cpu().tolist()]
        rows = max(max(counts), 1)
        row_elems = self.n0 * 3
        if self.world > 1:
            _all_gather(dist, self.d_gather.view(-1)[:self.world * rows * row_elems], self.d_send.view(-1)[:rows * row_elems], self.pg)
            gathered = self.d_gather.view(-1)[:self.world * rows * row_elems].view(self.world, rows, self.n0, 3)
        else:
            gathered = self.d_send[:rows].view(1, rows, self.n0, 3)
        n_kept = min(sum(counts), self.n_out)                                                  # :510: stop at n_out structures
        stack = be.d_frags[:self.cap0].view(self.n_cand, self.n0, 3)
        o = 0
        for r, c in enumerate(counts):                                                         # rank order = table order
            c = min(c, n_kept - o)
            if c > 0:
                stack[o:o + c].copy_(gathered[r, :c])
            o += c
        self.allgather_bytes = int(self.world * rows * row_elems * 8) if self.world > 1 else 0
        if n_kept == 0:
            return 0
        which = (self.d_draw_all % n_kept).to(torch.int32)
        be.d_ci_all[:, 0] = which
        be.d_ci[:, 0] = which[be.lo:be.hi]
        fs = FragmentSet.__new__(FragmentSet)
        sizes = [(n_kept, self.n0)] + [(s[0], s[1]) for s in self._other_sizes]
        fs.n_mols = len(sizes)
        fs.n_atoms = np.array([s[1] for s in sizes], dtype=np.int32)
        fs.n_conf = np.array([s[0] for s in sizes], dtype=np.int32)
        offs, o2 = [0], self.cap0
        for s in self._other_sizes:
            offs.append(o2)
            o2 += s[2]
        fs.frag_off = np.array(offs[:fs.n_mols], dtype=np.int64)
        fs.n_total = int(fs.n_atoms.sum())
        be.fs = fs
        return n_kept

    def tune_front(self, steps=2):
        with self.torch.cuda.stream(self.stream):
            if self._search() == 0:
                self.front = "hybrid"
                return None
        self.front_tuning = time_fronts(self.torch, self.dev, self.pg, steps,
                                        lambda form: sharded_step(self.backend, self.rank, self.world, self.torch.distributed, self.pg,
                                                                  self.shard_min_pairs, form, self.partition_chunks))
        self.front = self.front_tuning["chosen"]
        return self.front_tuning

    def step(self):
        if self.front == "auto":
            self.tune_front()
        with self.torch.cuda.stream(self.stream):
            n_kept = self._search()
        if n_kept == 0:
            return {"n_conformers": 0, "n_pass": 0, "n_keep": 0, "stats": []}
        res = sharded_step(self.backend, self.rank, self.world, self.torch.distributed, self.pg, self.shard_min_pairs, self.front, self.partition_chunks)
        res["n_conformers"] = n_kept
        res["allgather_bytes"] = res.get("allgather_bytes", 0) + self.allgather_bytes
        return res
