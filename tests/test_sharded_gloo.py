"""The multi-rank protocol of tscode_amd.pipeline.sharded_step under gloo on CPU, world_size 2 and 3.

The protocol code (block sharding, count exchange, padded all-gather of heavy-atom shards, per-pass
all-reduce(MIN) of best[], identical mask/cache update on every rank) is the product's; the arithmetic
behind it comes from a TEST-ONLY backend built on the CPU oracle, because the product has no CPU path.
Every rank must end with the mask the single-process oracle gives.
"""

import os
import socket

import numpy as np
import pytest

KS = (5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1)
INT_MAX = np.iinfo(np.int32).max


class OracleStepper:
    """Stepping form of prune_conformers_rmsd on CPU tensors (mirror of tsc_prune_* for the tests)."""

    def __init__(self, oracle, heavy, best_tensor, thr, mode, tile_rows):
        self.o, self.heavy, self.best_t = oracle, heavy, best_tensor
        self.thr, self.mode, self.tile_rows = thr, mode, tile_rows
        self.n = len(heavy)
        self.mask = np.ones(self.n, dtype=np.uint8)
        self.keys = np.zeros((0, 2), dtype=np.int64)
        self.ks = list(KS)
        self._stats = []
        self.k = 0

    def next_pass(self):
        while self.ks:
            k = int(self.ks.pop(0))
            if k == 1 or 20 * k < int(self.mask.sum()):
                self.k = k
                self.act = np.flatnonzero(self.mask)
                return k
        return 0

    def n_active(self):
        return len(self.act)

    def pass_estimate(self):
        # alternate between "small" (replicated, no collective) and "large" (sharded + all-reduce) so that both
        # branches of the protocol run in every test
        self._flip = not getattr(self, "_flip", False)
        return 10 ** 12 if self._flip else 0

    def pass_local(self, rank, world):
        best = self.best_t.numpy()                      # shares memory with the torch tensor
        best[:len(self.act)] = INT_MAX
        self.o.prune_pass_rows(self.heavy, self.mask, self.keys, self.k, rank, world, self.tile_rows, best, self.thr, self.mode)

    def pass_finish(self):
        best = self.best_t.numpy()[:len(self.act)]
        cs = self.n // self.k
        rows = np.flatnonzero(best != INT_MAX)
        i, j = self.act[rows], self.act[best[rows]]
        first = np.minimum(i // cs, self.k - 1) * cs
        self.keys = np.concatenate([self.keys, np.stack([first, first + (j - i)], axis=1)])
        before = int(self.mask.sum())
        self.mask[i] = 0
        self._stats.append({"k": self.k, "n_active_before": before, "n_active_after": int(self.mask.sum())})
        self.k = 0

    # ---- partitioned passes: the whole pass on the chunks that start inside this rank's block of the structure axis.  The oracle
    # computes best[] with the keys THIS rank holds (its own removals only); rows outside the rank's chunks are discarded, so a
    # wrong ownership rule (a key needed by another rank's chunk) would show as a wrong mask.
    def set_partition(self, rank, world, min_chunks):
        self.rank, self.world, self.min_chunks = rank, world, min_chunks
        self.bit_words = self.n // 64 + 40
        self.exch = self.backend_exch.numpy()
        self.exch[:] = 0
        self.views_split = False

    def pass_partitioned(self):
        return getattr(self, "world", 1) > 1 and self.k >= self.min_chunks * self.world

    def exchange_words(self):
        return self.bit_words + 8

    def pass_range(self):
        from tscode_amd.pipeline import partition_bounds
        best = np.full(len(self.act), INT_MAX, dtype=np.int32)
        self.o.prune_pass_rows(self.heavy, self.mask, self.keys, self.k, 0, 1, self.tile_rows, best, self.thr, self.mode)
        _, _, s_lo, s_hi = partition_bounds(self.n, self.k, self.rank, self.world)
        rows = np.flatnonzero((best != INT_MAX) & (self.act >= s_lo) & (self.act < s_hi))
        i, j = self.act[rows], self.act[best[rows]]
        cs = self.n // self.k
        first = np.minimum(i // cs, self.k - 1) * cs
        self.keys = np.concatenate([self.keys, np.stack([first, first + (j - i)], axis=1)])      # this rank's keys only
        bits = np.zeros(self.bit_words * 64, dtype=np.uint8)
        bits[i] = 1
        self.exch[:self.bit_words] = np.packbits(bits, bitorder="little").view(np.int64)
        self.exch[self.bit_words] = len(rows)                                                    # a statistic that must come out as the sum
        self.views_split = True

    def pass_merge(self):
        removed = np.flatnonzero(np.unpackbits(self.exch[:self.bit_words].view(np.uint8), bitorder="little"))
        assert int(self.exch[self.bit_words]) == len(removed)                                    # the ranks' removals are disjoint
        before = int(self.mask.sum())
        self.mask[removed] = 0
        self.exch[:self.bit_words + 8] = 0
        self._stats.append({"k": self.k, "n_active_before": before, "n_active_after": int(self.mask.sum())})
        self.k = 0

    def _remaining(self):
        return [self.k] + [int(k) for k in self.ks if int(k) == 1 or 20 * int(k) < self.n]

    def views_range(self):
        if not getattr(self, "views_split", False) or self.mode != 0:
            return 0, 0
        # the keys of this rank as one row of n flags per remaining pass: flag b of pass k2 = a key (a, b) with a the start of the
        # chunk of b in that pass (the only keys that pass can hit)
        off = self.bit_words + 8
        rem = self._remaining()
        dense = np.zeros((len(rem), self.n), dtype=np.int64)
        for q, k2 in enumerate(rem):
            cs = self.n // k2
            for a, b in self.keys:
                if 0 <= b < self.n and np.minimum(b // cs, k2 - 1) * cs == a:
                    dense[q, b] = 1
        self.exch[off:off + dense.size] = dense.ravel()
        return off, dense.size

    def views_merged(self):
        if getattr(self, "views_split", False) and self.mode == 0:
            off = self.bit_words + 8
            rem = self._remaining()
            dense = self.exch[off:off + len(rem) * self.n].reshape(len(rem), self.n)
            assert dense.max(initial=0) <= 1                                                     # disjoint between ranks
            keys = set()
            for q, k2 in enumerate(rem):
                cs = self.n // k2
                for b in np.flatnonzero(dense[q]):
                    keys.add((int(np.minimum(b // cs, k2 - 1) * cs), int(b)))
            self.keys = np.array(sorted(keys), dtype=np.int64).reshape(-1, 2)
        self.views_split = False

    def stats(self):
        return self._stats

    def copy_mask(self, dst):
        import torch
        dst[:self.n].copy_(torch.from_numpy(self.mask))

    def close(self):
        pass


class OracleShardBackend:
    def __init__(self, oracle, ens, rank, world, thr=0.5, mode=0, tile_rows=16):
        import torch

        from tscode_amd.pipeline import block_bounds
        self.o, self.ens, self.thr, self.mode, self.tile_rows = oracle, ens, thr, mode, tile_rows
        n, h = ens.n_poses, ens.n_heavy
        self.lo, self.hi = block_bounds(n, rank, world)
        self.h = h
        self.max_local = (n + world - 1) // world + 1
        self.heavy_local = torch.zeros((max(self.hi - self.lo, 1), h, 3), dtype=torch.float64)
        self.heavy_pad = torch.zeros((self.max_local, h, 3), dtype=torch.float64)
        self.gather = torch.zeros((world * self.max_local, h, 3), dtype=torch.float64)
        self.heavy_all = torch.zeros((n, h, 3), dtype=torch.float64)
        self.best = torch.zeros(n, dtype=torch.int32)
        self.keep = torch.zeros(n, dtype=torch.uint8)
        self.counts = torch.zeros(world, dtype=torch.int64)
        self.exch = torch.zeros(n // 64 + 48 + 19 * n, dtype=torch.int64)

    def embed_clash_block(self):
        import torch
        e = self.ens
        sl = slice(self.lo, self.hi)
        poses = self.o.transform_batch(e.frag_coords, e.conf_idx[sl], e.rot[sl], e.pos[sl])
        cm = self.o.compenetration_mask(poses, e.ids, 1.5, 0)
        heavy = np.ascontiguousarray(poses[cm][:, e.atomnos != 1])
        self.heavy_local[:len(heavy)].copy_(torch.from_numpy(heavy))
        return len(heavy)

    def embed_clash_all(self):
        """front="replicate": every rank computes the whole pose list itself."""
        import torch
        e = self.ens
        poses = self.o.transform_batch(e.frag_coords, e.conf_idx, e.rot, e.pos)
        cm = self.o.compenetration_mask(poses, e.ids, 1.5, 0)
        heavy = np.ascontiguousarray(poses[cm][:, e.atomnos != 1])
        self.heavy_all[:len(heavy)].copy_(torch.from_numpy(heavy))
        return len(heavy)

    def clash_block_into_all(self):
        """front="hybrid": this block's verdicts into the all-poses mask (zero elsewhere)."""
        import torch
        e = self.ens
        sl = slice(self.lo, self.hi)
        poses = self.o.transform_batch(e.frag_coords, e.conf_idx[sl], e.rot[sl], e.pos[sl])
        self.clash_all = torch.zeros(e.n_poses, dtype=torch.uint8)
        self.clash_all[sl] = torch.from_numpy(self.o.compenetration_mask(poses, e.ids, 1.5, 0).astype(np.uint8))

    def embed_masked_all(self):
        import torch
        e = self.ens
        cm = self.clash_all.numpy().astype(bool)
        assert self.clash_all.max() <= 1                      # the blocks' verdicts do not overlap
        poses = self.o.transform_batch(e.frag_coords, e.conf_idx[cm], e.rot[cm], e.pos[cm])
        heavy = np.ascontiguousarray(poses[:, e.atomnos != 1])
        self.heavy_all[:len(heavy)].copy_(torch.from_numpy(heavy))
        return len(heavy)

    def make_stepper(self, n_pass):
        st = OracleStepper(self.o, self.heavy_all[:n_pass].numpy(), self.best, self.thr, self.mode, self.tile_rows)
        st.backend_exch = self.exch
        return st


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_poses, mode, out_dir, front="shard", partition_chunks=4):
    import torch
    import torch.distributed as dist

    import oracle
    from tscode_amd.pipeline import sharded_step
    from tscode_amd.synthetic import make_config
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    oracle.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ens = make_config("C2", n_poses)
        backend = OracleShardBackend(oracle, ens, rank, world, mode=mode)
        res = sharded_step(backend, rank, world, dist, front=front, partition_chunks=partition_chunks)
        assert res["front"] == front and (res["allgather_bytes"] > 0) == (front == "shard")
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keep=backend.keep[:res["n_pass"]].numpy(), n_pass=res["n_pass"],
                 n_keep=res["n_keep"], counts=np.array(res["counts"]), ks=np.array([s["k"] for s in res["stats"]]),
                 partitioned=np.array([k for k, _ in res["partitioned"]], dtype=np.int64))
    finally:
        dist.destroy_process_group()


# partition_chunks: 2 = the default since round 4, 4 = rounds 2 - 3 (passes with at least that many chunks per rank are partitioned by chunks, the
# rest sharded by row tiles or replicated), 1 = every pass down to k = world partitioned (chunks as long as a rank's block: every block boundary is a seam
# that a chunk straddles), 0 = no partitioned pass (the protocol of round 2)
@pytest.mark.parametrize("world,n_poses,mode,front,partition_chunks", [
    (2, 3000, 0, "shard", 2), (3, 2001, 0, "shard", 4), (2, 1500, 1, "shard", 2), (2, 2500, 0, "replicate", 4), (3, 1201, 1, "replicate", 2),
    (2, 2999, 0, "replicate", 1), (3, 2503, 0, "shard", 1), (3, 1801, 0, "replicate", 0), (2, 2200, 0, "hybrid", 4), (3, 1901, 1, "hybrid", 2)])
def test_sharded_step_gloo(oracle, tmp_path, world, n_poses, mode, front, partition_chunks):
    import torch.multiprocessing as mp

    from tscode_amd.synthetic import make_config
    mp.spawn(_worker, args=(world, _free_port(), n_poses, mode, str(tmp_path), front, partition_chunks), nprocs=world, join=True)
    ens = make_config("C2", n_poses)
    poses = oracle.transform_batch(ens.frag_coords, ens.conf_idx, ens.rot, ens.pos)
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    ref = oracle.prune_heavy(np.ascontiguousarray(poses[cm][:, ens.atomnos != 1]), 0.5, mode=mode)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert int(got["n_pass"]) == int(cm.sum())
        assert got["counts"].sum() == cm.sum() and len(got["counts"]) == (world if front == "shard" else 1)
        assert np.array_equal(got["keep"].astype(bool), ref["mask"]), f"rank {rank}"
        assert int(got["n_keep"]) == int(ref["mask"].sum())
        assert got["ks"].tolist() == [s["k"] for s in ref["stats"]]
        want = [s["k"] for s in ref["stats"] if partition_chunks and s["k"] >= partition_chunks * world]
        assert got["partitioned"].tolist() == want, (got["partitioned"].tolist(), want)


def test_partition_bounds_are_whole_chunks_that_cover_the_pass():
    from tscode_amd.pipeline import partition_bounds
    for n in (57, 1000, 1003, 57046, 483472):
        for k in (1, 2, 5, 10, 50, 200, 2000):
            if k > n:
                continue
            cs = n // k
            for world in (1, 2, 3, 8):
                b = [partition_bounds(n, k, r, world) for r in range(world)]
                assert b[0][0] == 0 and b[0][2] == 0 and b[-1][1] == k and b[-1][3] == n
                for r in range(world):
                    c_lo, c_hi, s_lo, s_hi = b[r]
                    assert c_lo <= c_hi and s_lo == (c_lo * cs if c_lo < k else n) and s_hi == (c_hi * cs if c_hi < k else n)
                    if r + 1 < world:
                        assert b[r + 1][0] == c_hi and b[r + 1][2] == s_hi
                    # a chunk belongs to the rank whose block [n r / W, n (r + 1) / W) holds its first structure
                    for c in range(c_lo, c_hi):
                        assert n * r // world <= c * cs and (r + 1 == world or c * cs < n * (r + 1) // world)


def test_block_bounds_cover_the_pose_axis():
    from tscode_amd.pipeline import block_bounds
    for n in (0, 1, 7, 100, 100001):
        for world in (1, 2, 3, 8):
            b = [block_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
