"""Device-resident pipeline: embed -> clash mask -> ordered compaction -> RMSD prune.

``DevicePipeline`` keeps one synthetic (or user) ensemble in HBM as torch tensors (PyTorch is only the
allocator, the stream and the torch.distributed/RCCL transport here) and runs the whole hot path on it
through the C ABI.

One GPU:   tsc_pipeline_dev does everything in one call.
N GPUs:    one process per GPU.  The pose axis is sharded in contiguous blocks; each rank embeds and
           clash-filters its block (no communication), the surviving heavy-atom coordinates are
           all-gathered once (RCCL over xGMI), and every pass of the prune splits its row tiles
           round-robin over the ranks: a rank computes best[] for its rows, an all-reduce(MIN) over best[]
           merges them, and every rank applies the identical mask/cache update.
"""

from __future__ import annotations

import numpy as np

from .engine import FragmentSet, get_engine

__all__ = ["DevicePipeline"]


class DevicePipeline:
    def __init__(self, ens, device_index=0, rank=0, world=1, clash_thresh=1.5, max_clashes=0, rmsd_thr=0.5, mode=0,
                 process_group=None):
        import torch
        self.torch = torch
        self.ens = ens
        self.rank, self.world = int(rank), int(world)
        self.clash_thresh, self.max_clashes, self.rmsd_thr, self.mode = clash_thresh, max_clashes, rmsd_thr, mode
        self.pg = process_group
        self.dev = torch.device(f"cuda:{device_index}")
        torch.cuda.set_device(self.dev)
        self.eng = get_engine(device_index)
        # everything (this library's kernels, torch copies, RCCL collectives) is ordered on torch's current stream
        self.eng.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.fs = FragmentSet(ens.frag_coords)
        n = ens.n_poses
        self.n_total = n
        # contiguous block of the pose axis for this rank
        self.lo = (n * self.rank) // self.world
        self.hi = (n * (self.rank + 1)) // self.world
        self.n_local = self.hi - self.lo
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.d_frags = t(self.fs.flat)
        self.d_ci = t(ens.conf_idx[self.lo:self.hi])
        self.d_rot = t(ens.rot[self.lo:self.hi])
        self.d_pos = t(ens.pos[self.lo:self.hi])
        self.heavy_idx = np.flatnonzero(ens.atomnos != 1).astype(np.int32)
        self.h = len(self.heavy_idx)
        na = ens.n_atoms
        self.d_clash = torch.empty(max(self.n_local, 1), dtype=torch.uint8, device=self.dev)
        self.d_structures = torch.empty((max(self.n_local, 1), na, 3), dtype=torch.float64, device=self.dev)
        self.d_keep = torch.empty(max(n if self.world > 1 else self.n_local, 1), dtype=torch.uint8, device=self.dev)
        if self.world > 1:
            self.d_heavy_local = torch.empty((max(self.n_local, 1), self.h, 3), dtype=torch.float64, device=self.dev)
            self.max_local = (n + self.world - 1) // self.world + 1
            self.d_heavy_pad = torch.zeros((self.max_local, self.h, 3), dtype=torch.float64, device=self.dev)
            self.d_gather = torch.empty((self.world * self.max_local, self.h, 3), dtype=torch.float64, device=self.dev)
            self.d_heavy_all = torch.empty((n, self.h, 3), dtype=torch.float64, device=self.dev)
            self.d_best = torch.empty(n, dtype=torch.int32, device=self.dev)
            self.d_counts = torch.zeros(self.world, dtype=torch.int64, device=self.dev)
        torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------------------------------
    def step(self):
        """One pass of the hot path over the resident ensemble. Returns a dict of counts/statistics;
        the verdicts stay on the device (d_clash, d_structures, d_keep)."""
        if self.world == 1:
            return self.eng.pipeline_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, self.heavy_idx,
                                         self.clash_thresh, self.max_clashes, self.rmsd_thr, self.mode, self.d_clash,
                                         self.d_structures, self.d_keep)
        return self._step_sharded()

    def _step_sharded(self):
        torch, dist, eng = self.torch, self.torch.distributed, self.eng
        # 1. this rank's block: fused embed+clash verdicts, ordered compaction (all atoms + heavy atoms)
        eng.embed_clash_mask_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, self.clash_thresh,
                                 self.max_clashes, self.d_clash)
        # materialise the passing poses of this block (structures) and their heavy atoms
        n_pass_local = self._embed_passing()
        # 2. exchange: counts, then one all-gather of the (padded) heavy-atom shards
        self.d_counts.zero_()
        self.d_counts[self.rank] = n_pass_local
        dist.all_reduce(self.d_counts, op=dist.ReduceOp.SUM, group=self.pg)
        counts = self.d_counts.cpu().tolist()
        self.d_heavy_pad[:n_pass_local].copy_(self.d_heavy_local[:n_pass_local])
        dist.all_gather_into_tensor(self.d_gather.view(-1), self.d_heavy_pad.view(-1), group=self.pg)
        off = 0
        for r, c in enumerate(counts):
            self.d_heavy_all[off:off + c].copy_(self.d_gather[r * self.max_local:r * self.max_local + c])
            off += c
        n_pass = off
        # 3. prune: row tiles of every pass dealt round-robin over ranks, merged by all-reduce(MIN)
        stats = []
        if n_pass > 0:
            st = eng.prune_stepper(self.d_heavy_all, n_pass, self.h, self.rmsd_thr, self.mode)
            st.use_best_buffer(self.d_best)
            try:
                while True:
                    k = st.next_pass()
                    if k == 0:
                        break
                    st.pass_local(self.rank, self.world)
                    _, n_act = st.best_ptr()
                    dist.all_reduce(self.d_best[:n_act], op=dist.ReduceOp.MIN, group=self.pg)
                    st.pass_finish()
                stats = st.stats()
                st.copy_mask(self.d_keep)          # verdicts: stepper state -> d_keep[:n_pass], on the shared stream
            finally:
                st.close()
        n_keep = stats[-1]["n_active_after"] if stats else 0
        return {"n_pass": n_pass, "n_pass_local": n_pass_local, "n_keep": n_keep, "stats": stats, "counts": counts}

    def _embed_passing(self):
        """Embed the poses of this block that passed the clash check, in order: d_structures, d_heavy_local."""
        eng = self.eng
        # all atoms: transform the whole block then compact (two streaming kernels); heavy atoms from the compacted poses
        # (the single-GPU path fuses this into one kernel inside tsc_pipeline_dev)
        torch = self.torch
        if not hasattr(self, "d_all"):
            self.d_all = torch.empty((max(self.n_local, 1), self.ens.n_atoms, 3), dtype=torch.float64, device=self.dev)
        eng.transform_batch_dev(self.fs, self.d_frags, self.d_ci, self.d_rot, self.d_pos, self.n_local, self.d_all)
        n_pass = eng.compact_rows_dev(self.d_all, self.d_clash, self.n_local, self.ens.n_atoms * 24, self.d_structures)
        if n_pass:
            eng.gather_heavy_dev(self.d_structures, None, n_pass, self.ens.n_atoms, self.heavy_idx, self.d_heavy_local)
        return int(n_pass)
