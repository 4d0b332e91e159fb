"""Engine: one libtscode_hip context (one GPU, one stream) and typed Python entry points.

Host methods take and return NumPy arrays (the drop-in functions are built on them);
``*_dev`` methods take anything with ``data_ptr()`` (torch tensors on the engine's GPU) and
leave results on the device.  Every call goes through the C ABI of include/tscode_hip.h.
"""

from __future__ import annotations

import ctypes as C
import threading
import weakref

import numpy as np

from . import _lib
from ._lib import PassStats, TSC_MAX_PASSES, check, ptr

__all__ = ["Engine", "FragmentSet", "get_engine", "device_count"]


def device_count() -> int:
    n = _lib.load().tsc_device_count()
    if n < 0:
        check(n)
    return n


class FragmentSet:
    """Rigid fragments in the layout tsc_transform_batch expects:
    fragment m is a conformer stack f64[n_conf_m, n_m, 3]; all stacks sit back to back in one buffer."""

    def __init__(self, frag_coords):
        stacks = []
        for f in frag_coords:
            f = np.ascontiguousarray(f, dtype=np.float64)
            if f.ndim == 2:
                f = f[None]
            if f.ndim != 3 or f.shape[2] != 3:
                raise ValueError("each fragment must be (n_conf, n_atoms, 3) or (n_atoms, 3)")
            stacks.append(f)
        self.n_mols = len(stacks)
        self.n_atoms = np.array([s.shape[1] for s in stacks], dtype=np.int32)
        self.n_conf = np.array([s.shape[0] for s in stacks], dtype=np.int32)
        sizes = np.array([s.size for s in stacks], dtype=np.int64)
        self.frag_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.flat = np.ascontiguousarray(np.concatenate([s.ravel() for s in stacks]))
        self.n_total = int(self.n_atoms.sum())

    def table_args(self):
        return (self.frag_off.ctypes.data_as(_lib.c_i64p), self.n_atoms.ctypes.data_as(_lib.c_i32p),
                self.n_conf.ctypes.data_as(_lib.c_i32p), C.c_int(self.n_mols))


def _ids_arg(ids):
    if ids is None:
        return None, 0, np.zeros(0, dtype=np.int32)
    a = np.ascontiguousarray(ids, dtype=np.int32).ravel()
    return a.ctypes.data_as(_lib.c_i32p), len(a), a


def _stats_list(stats, n):
    return [stats[i].as_dict() for i in range(n)]


class PipelineResult:
    """What one tsc_pipeline_dev call returned.  Reads like the dict it used to be (res["n_pass"], res["stats"], ...);
    the per-pass statistics are turned into Python objects only when somebody asks for them -- a dozen structs of a dozen
    fields cost more host time than several of the step's kernels take."""
    __slots__ = ("n_pass", "n_keep", "_stats", "_n_passes", "_tm", "_cache")

    def __init__(self, n_pass, n_keep, stats, n_passes, tm):
        self.n_pass, self.n_keep, self._stats, self._n_passes, self._tm, self._cache = n_pass, n_keep, stats, n_passes, tm, None

    @property
    def stats(self):
        if self._cache is None:
            self._cache = _stats_list(self._stats, self._n_passes)
        return self._cache

    @property
    def ms(self):
        t = self._tm
        return {"embed_clash": t[0], "compact": t[1], "prune": t[2], "total": t[3]}

    def __getitem__(self, key):
        if key in ("n_pass", "n_keep", "stats", "ms"):
            return getattr(self, key)
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def keys(self):
        return ("n_pass", "n_keep", "stats", "ms")


class Engine:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.tsc_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._runs = weakref.WeakSet()      # live PruneSteppers / IpcExchanges: they hold blocks and events of this context

    def close(self):
        if getattr(self, "_h", None):
            # (a run and its engine can become garbage together -- e.g. both held by the traceback of an exception -- and the
            # collector finalises them in no particular order: the runs go first, whichever finaliser is called first.  When they die
            # in ONE collection CPython has cleared the weak references before any finaliser runs and this set is already empty:
            # tsc_ctx_destroy then destroys the runs the context still lists itself, and their close() finds the engine closed)
            for run in list(getattr(self, "_runs", ())):
                run.close()
            self.lib.tsc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stream / timing ------------------------------------------------------------------
    def set_stream(self, hip_stream):
        check(self.lib.tsc_ctx_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        check(self.lib.tsc_ctx_synchronize(self._h))

    def set_option(self, name: str, value: float):
        """Tunables of the library: "prune_algo" (0 auto, 1 register-tiled, 2 sieve), "seg_cols"."""
        check(self.lib.tsc_ctx_set_option(self._h, name.encode(), C.c_double(value)))

    def timer_begin(self):
        check(self.lib.tsc_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        check(self.lib.tsc_timer_end(self._h, C.byref(ms)))
        return ms.value

    # ---- K1 ---------------------------------------------------------------------------------
    def transform_batch(self, frags: FragmentSet, conf_idx, rot, pos) -> np.ndarray:
        conf_idx = np.ascontiguousarray(conf_idx, dtype=np.int32).reshape(-1, frags.n_mols)
        rot = np.ascontiguousarray(rot, dtype=np.float64).reshape(-1, frags.n_mols, 3, 3)
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, frags.n_mols, 3)
        n = len(rot)
        if not (len(conf_idx) == n == len(pos)):
            raise ValueError("conf_idx, rot and pos disagree on the number of poses")
        out = np.empty((n, frags.n_total, 3), dtype=np.float64)
        check(self.lib.tsc_transform_batch(self._h, ptr(frags.flat), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos),
                                           C.c_int64(n), ptr(out)))
        return out

    def transform_batch_dev(self, frags: FragmentSet, frags_dev, conf_idx, rot, pos, n_poses, out):
        check(self.lib.tsc_transform_batch_dev(self._h, ptr(frags_dev), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos),
                                               C.c_int64(n_poses), ptr(out)))

    # ---- K2 ---------------------------------------------------------------------------------
    def clash_mask(self, coords, ids=None, thresh=1.5, max_clashes=0, return_counts=False):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        if coords.ndim != 3 or coords.shape[2] != 3:
            raise ValueError("coords must be (n_poses, n_atoms, 3)")
        n, na = coords.shape[0], coords.shape[1]
        ids_p, n_ids, _keep = _ids_arg(ids)
        mask = np.zeros(n, dtype=np.uint8)
        counts = np.zeros(n, dtype=np.int32) if return_counts else None
        check(self.lib.tsc_clash_mask(self._h, ptr(coords), C.c_int64(n), C.c_int(na), ids_p, C.c_int(n_ids), C.c_double(thresh),
                                      C.c_int64(int(max_clashes)), ptr(mask), ptr(counts)))
        return (mask.astype(bool), counts) if return_counts else mask.astype(bool)

    def clash_mask_dev(self, coords, n_poses, n_atoms, ids, thresh, max_clashes, mask, counts=None):
        ids_p, n_ids, _keep = _ids_arg(ids)
        check(self.lib.tsc_clash_mask_dev(self._h, ptr(coords), C.c_int64(n_poses), C.c_int(n_atoms), ids_p, C.c_int(n_ids),
                                          C.c_double(thresh), C.c_int64(int(max_clashes)), ptr(mask), ptr(counts)))

    def embed_clash_mask_dev(self, frags: FragmentSet, frags_dev, conf_idx, rot, pos, n_poses, thresh, max_clashes, mask, counts=None):
        check(self.lib.tsc_embed_clash_mask_dev(self._h, ptr(frags_dev), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos),
                                                C.c_int64(n_poses), C.c_double(thresh), C.c_int64(int(max_clashes)), ptr(mask),
                                                ptr(counts)))

    def embed_clash_compact_dev(self, frags: FragmentSet, frags_dev, conf_idx, rot, pos, n_poses, heavy_idx, thresh, max_clashes, mask,
                                structures, heavy) -> int:
        """Fused embed + clash verdicts, then the passing poses embedded straight into ``structures`` (may be None) and ``heavy``.
        Returns how many passed."""
        heavy_idx = np.ascontiguousarray(heavy_idx, dtype=np.int32)
        n_pass = C.c_int64()
        check(self.lib.tsc_embed_clash_compact_dev(self._h, ptr(frags_dev), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos),
                                                   C.c_int64(n_poses), heavy_idx.ctypes.data_as(_lib.c_i32p), C.c_int(len(heavy_idx)),
                                                   C.c_double(thresh), C.c_int64(int(max_clashes)), ptr(mask), ptr(structures), ptr(heavy),
                                                   C.byref(n_pass)))
        return n_pass.value

    def basis_from_poses_dev(self, frags: FragmentSet, frags_dev, conf_idx, rot, pos, n_poses, heavy_idx):
        """Fork the estimate of the prune's descriptor basis from a sample of these poses onto the side stream."""
        heavy_idx = np.ascontiguousarray(heavy_idx, dtype=np.int32)
        check(self.lib.tsc_basis_from_poses_dev(self._h, ptr(frags_dev), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos), C.c_int64(n_poses),
                                                heavy_idx.ctypes.data_as(_lib.c_i32p), C.c_int(len(heavy_idx))))

    def embed_masked_dev(self, frags: FragmentSet, frags_dev, conf_idx, rot, pos, n_poses, mask, heavy_idx, structures, heavy, want_count=True):
        """The poses selected by a device mask, embedded in order into ``structures`` and / or ``heavy`` (either may be None).
        Returns how many were selected (None with want_count=False: the call then does not synchronise)."""
        heavy_idx = np.ascontiguousarray(heavy_idx, dtype=np.int32)
        n_sel = C.c_int64()
        check(self.lib.tsc_embed_masked_dev(self._h, ptr(frags_dev), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos), C.c_int64(n_poses),
                                            ptr(mask), heavy_idx.ctypes.data_as(_lib.c_i32p), C.c_int(len(heavy_idx)), ptr(structures), ptr(heavy),
                                            C.byref(n_sel) if want_count else None))
        return n_sel.value if want_count else None

    def all_dists(self, a, b) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != 3 or b.shape[1] != 3:
            raise AssertionError("all_dists needs (n, 3) arrays")          # reference: assert A.shape[1]==B.shape[1]
        out = np.empty((len(a), len(b)), dtype=np.float64)
        check(self.lib.tsc_all_dists(self._h, ptr(a), C.c_int(len(a)), ptr(b), C.c_int(len(b)), ptr(out)))
        return out

    # ---- compaction ---------------------------------------------------------------------------
    def compact_rows_dev(self, src, mask, n_rows, row_bytes, dst) -> int:
        kept = C.c_int64()
        check(self.lib.tsc_compact_rows_dev(self._h, ptr(src), ptr(mask), C.c_int64(n_rows), C.c_int64(row_bytes), ptr(dst),
                                            C.byref(kept)))
        return kept.value

    def gather_heavy_dev(self, coords, mask, n_poses, n_atoms, heavy_idx, heavy_out) -> int:
        heavy_idx = np.ascontiguousarray(heavy_idx, dtype=np.int32)
        kept = C.c_int64()
        check(self.lib.tsc_gather_heavy_dev(self._h, ptr(coords), ptr(mask), C.c_int64(n_poses), C.c_int(n_atoms),
                                            heavy_idx.ctypes.data_as(_lib.c_i32p), C.c_int(len(heavy_idx)), ptr(heavy_out),
                                            C.byref(kept)))
        return kept.value

    # ---- K3 ---------------------------------------------------------------------------------
    def rmsd_pairs(self, heavy, pairs):
        heavy = np.ascontiguousarray(heavy, dtype=np.float64)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        r = np.empty(len(pairs), dtype=np.float64)
        m = np.empty(len(pairs), dtype=np.float64)
        check(self.lib.tsc_rmsd_pairs(self._h, ptr(heavy), C.c_int64(heavy.shape[0]), C.c_int(heavy.shape[1]), ptr(pairs),
                                      C.c_int64(len(pairs)), ptr(r), ptr(m)))
        return r, m

    def greedy_group_filter(self, poses, group_off, rmsd_thr=1.0):
        """accepted bool[n_poses]: per group, keep a pose iff it is not similar to a pose kept before it."""
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        group_off = np.ascontiguousarray(group_off, dtype=np.int32)
        if poses.ndim != 3 or poses.shape[2] != 3 or group_off.ndim != 1 or len(group_off) < 1 or group_off[-1] != len(poses):
            raise ValueError("poses must be (n_poses, n_atoms, 3) and group_off[-1] == n_poses")
        acc = np.zeros(len(poses), dtype=np.uint8)
        check(self.lib.tsc_greedy_group_filter(self._h, ptr(poses), ptr(group_off), C.c_int(len(group_off) - 1), C.c_int(poses.shape[1]),
                                               C.c_double(rmsd_thr), ptr(acc)))
        return acc.astype(bool)

    # ---- N2: torsion fingerprints / TFD pair search -----------------------------------------------
    def torsion_fingerprints(self, structures, quadruplets):
        """_get_tf_mat (tscode/numba_functions.py:233-240): f32[N, T] dihedral angles in degrees."""
        structures = np.ascontiguousarray(structures, dtype=np.float64)
        if structures.ndim != 3 or structures.shape[2] != 3:
            raise ValueError("structures must be (N, n_atoms, 3)")
        quads = np.ascontiguousarray(quadruplets, dtype=np.int32).reshape(-1, 4)
        out = np.zeros((len(structures), len(quads)), dtype=np.float32)
        check(self.lib.tsc_torsion_fingerprints(self._h, ptr(structures), C.c_int64(len(structures)), C.c_int(structures.shape[1]), ptr(quads),
                                                C.c_int(len(quads)), ptr(out)))
        return out

    def tfd_first_similar(self, tf_mat, d, k, num_active, thresh=10.0):
        """The pair search of one pass of prune_conformers_tfd (tscode/numba_functions.py:171-199): i32[N], -1 = none."""
        tf = np.ascontiguousarray(tf_mat, dtype=np.float32)
        first = np.full(len(tf), -1, dtype=np.int32)
        check(self.lib.tsc_tfd_first_similar(self._h, ptr(tf), C.c_int64(len(tf)), C.c_int(tf.shape[1]), C.c_int64(int(d)), C.c_int64(int(k)),
                                             C.c_int64(int(num_active)), C.c_double(float(thresh)), ptr(first)))
        return first

    # ---- N4: moments of inertia, embed scores -------------------------------------------------------
    def inertia_moments(self, structures, masses):
        structures = np.ascontiguousarray(structures, dtype=np.float64)
        masses = np.ascontiguousarray(masses, dtype=np.float64)
        if structures.ndim != 3 or structures.shape[2] != 3 or masses.shape != (structures.shape[1],):
            raise ValueError("structures must be (N, n_atoms, 3) and masses (n_atoms,)")
        out = np.empty((len(structures), 3))
        check(self.lib.tsc_inertia_moments(self._h, ptr(structures), C.c_int64(len(structures)), C.c_int(structures.shape[1]), ptr(masses), ptr(out)))
        return out

    def moi_first_similar(self, moments, max_deviation=1e-2):
        moments = np.ascontiguousarray(moments, dtype=np.float64).reshape(-1, 3)
        first = np.full(len(moments), -1, dtype=np.int32)
        check(self.lib.tsc_moi_first_similar(self._h, ptr(moments), C.c_int64(len(moments)), C.c_double(float(max_deviation)), ptr(first)))
        return first

    def embed_scores(self, structures, indices, distances):
        structures = np.ascontiguousarray(structures, dtype=np.float64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        distances = np.ascontiguousarray(distances, dtype=np.float64)
        n = len(structures)
        if indices.ndim != 3 or indices.shape[0] != n or indices.shape[2] != 2 or distances.shape != indices.shape[:2]:
            raise ValueError("indices must be (N, n_c, 2) and distances (N, n_c)")
        sc, err = np.zeros(n, dtype=np.float32), np.zeros(n)
        check(self.lib.tsc_embed_scores(self._h, ptr(structures), C.c_int64(n), C.c_int(structures.shape[1]), ptr(indices), ptr(distances),
                                        C.c_int(indices.shape[1]), ptr(sc), ptr(err)))
        return sc, err

    # ---- N1: string-embed pose parameters -------------------------------------------------------
    def string_embed_params(self, p1, p2, ref_vec, mol_vec, conf_pair, angles):
        """tscode/embeds.py:98-116 for every (site, angle): rot f64[S*A, 2, 3, 3], pos f64[S*A, 2, 3], conf_idx i32[S*A, 2]."""
        p1, p2, ref_vec, mol_vec = (np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64) for x in (p1, p2, ref_vec, mol_vec))
        conf_pair = np.ascontiguousarray(np.atleast_2d(conf_pair), dtype=np.int32)
        angles = np.ascontiguousarray(angles, dtype=np.float64)
        S, A = len(p1), len(angles)
        if not (p1.shape == p2.shape == ref_vec.shape == mol_vec.shape == (S, 3)) or conf_pair.shape != (S, 2):
            raise ValueError("p1, p2, ref_vec, mol_vec must be (n_sites, 3) and conf_pair (n_sites, 2)")
        rot, pos, ci = np.empty((S * A, 2, 3, 3)), np.empty((S * A, 2, 3)), np.empty((S * A, 2), dtype=np.int32)
        check(self.lib.tsc_string_embed_params(self._h, ptr(p1), ptr(p2), ptr(ref_vec), ptr(mol_vec), ptr(conf_pair), C.c_int64(S), ptr(angles),
                                               C.c_int(A), ptr(rot), ptr(pos), ptr(ci)))
        return rot, pos, ci

    def cyclical_embed_params(self, start, end, direction, pivot, meanpoint, r0, r1, n_reactive, angle):
        """tscode/embeds.py:676-713 per (pose, molecule) row: rot f64[n, 3, 3], pos f64[n, 3]."""
        arrs = [np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64) for x in (start, end, direction, pivot, meanpoint, r0, r1)]
        n_reactive = np.ascontiguousarray(n_reactive, dtype=np.int32)
        angle = np.ascontiguousarray(angle, dtype=np.float64)
        n = len(angle)
        if any(a.shape != (n, 3) for a in arrs) or n_reactive.shape != (n,):
            raise ValueError("all vector inputs must be (n, 3), n_reactive and angle (n,)")
        rot, pos = np.empty((n, 3, 3)), np.empty((n, 3))
        check(self.lib.tsc_cyclical_embed_params(self._h, *[ptr(a) for a in arrs], ptr(n_reactive), ptr(angle), C.c_int64(n), ptr(rot), ptr(pos)))
        return rot, pos

    # ---- N1: the embed loops as one call each ----------------------------------------------------
    def tfd_greedy_filter(self, tf_mat, thresh=10.0):
        """is_new_structure (tscode/embeds.py:47-69) over an ordered list of fingerprints f32[N, T]: bool[N]."""
        tf = np.ascontiguousarray(tf_mat, dtype=np.float32)
        if tf.ndim != 2:
            raise ValueError("tf_mat must be (N, n_quadruplets)")
        acc = np.zeros(len(tf), dtype=np.uint8)
        nk = C.c_int64()
        check(self.lib.tsc_tfd_greedy_filter(self._h, ptr(tf), C.c_int64(len(tf)), C.c_int(tf.shape[1]), C.c_double(float(thresh)), ptr(acc), C.byref(nk)))
        return acc.astype(bool)

    def string_embed(self, frags: FragmentSet, p1, p2, ref_vec, mol_vec, conf_pair, angles, clash_thresh, max_clashes, quadruplets, tfd_thresh=10.0,
                     want_poses=True):
        """tscode/embeds.py:91-120 for every (site, angle) candidate: (clash_ok bool[N], kept bool[N], poses f64[n_kept, n, 3] or None)."""
        if frags.n_mols != 2:
            raise ValueError("the string embed takes two fragments")
        p1, p2, ref_vec, mol_vec = (np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64) for x in (p1, p2, ref_vec, mol_vec))
        conf_pair = np.ascontiguousarray(np.atleast_2d(conf_pair), dtype=np.int32)
        angles = np.ascontiguousarray(angles, dtype=np.float64)
        quads = np.ascontiguousarray(quadruplets, dtype=np.int32).reshape(-1, 4)
        S, A = len(p1), len(angles)
        if not (p1.shape == p2.shape == ref_vec.shape == mol_vec.shape == (S, 3)) or conf_pair.shape != (S, 2):
            raise ValueError("p1, p2, ref_vec, mol_vec must be (n_sites, 3) and conf_pair (n_sites, 2)")
        N = S * A
        ok, kept = np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.uint8)
        poses = np.empty((N, frags.n_total, 3)) if want_poses else None        # rows beyond n_kept are never touched
        n_pass, n_kept = C.c_int64(), C.c_int64()
        check(self.lib.tsc_string_embed(self._h, ptr(frags.flat), *frags.table_args()[:3], ptr(p1), ptr(p2), ptr(ref_vec), ptr(mol_vec), ptr(conf_pair),
                                        C.c_int64(S), ptr(angles), C.c_int(A), C.c_double(float(clash_thresh)), C.c_int64(int(max_clashes)),
                                        ptr(quads), C.c_int(len(quads)), C.c_double(float(tfd_thresh)), ptr(ok), ptr(kept), ptr(poses), C.c_int64(N),
                                        C.byref(n_pass), C.byref(n_kept)))
        return ok.astype(bool), kept.astype(bool), (poses[:n_kept.value].copy() if want_poses else None)

    def cyclical_embed(self, frags: FragmentSet, start, end, direction, pivot, meanpoint, r0, r1, n_reactive, angle, conf_idx, group_off, clash_thresh,
                       max_clashes, rmsd_thr=1.0, want_poses=True):
        """tscode/embeds.py:657-717 / :785-847 over rows = pose * n_mols + molecule: (clash_ok bool[N], kept bool[N], poses or None)."""
        arrs = [np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64) for x in (start, end, direction, pivot, meanpoint, r0, r1)]
        n_reactive = np.ascontiguousarray(n_reactive, dtype=np.int32).ravel()
        angle = np.ascontiguousarray(angle, dtype=np.float64).ravel()
        conf_idx = np.ascontiguousarray(conf_idx, dtype=np.int32).ravel()
        group_off = np.ascontiguousarray(group_off, dtype=np.int32)
        rows = len(angle)
        if rows % frags.n_mols or any(a.shape != (rows, 3) for a in arrs) or n_reactive.shape != (rows,) or conf_idx.shape != (rows,):
            raise ValueError("every per-row input must have n_poses * n_mols rows")
        N = rows // frags.n_mols
        ok, kept = np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.uint8)
        poses = np.empty((N, frags.n_total, 3)) if want_poses else None
        n_pass, n_kept = C.c_int64(), C.c_int64()
        check(self.lib.tsc_cyclical_embed(self._h, ptr(frags.flat), *frags.table_args(), *[ptr(a) for a in arrs], ptr(n_reactive), ptr(angle), ptr(conf_idx),
                                          C.c_int64(N), ptr(group_off), C.c_int(len(group_off) - 1), C.c_double(float(clash_thresh)),
                                          C.c_int64(int(max_clashes)), C.c_double(float(rmsd_thr)), ptr(ok), ptr(kept), ptr(poses), C.c_int64(N),
                                          C.byref(n_pass), C.byref(n_kept)))
        return ok.astype(bool), kept.astype(bool), (poses[:n_kept.value].copy() if want_poses else None)

    # ---- N3: conformational-search rotations ---------------------------------------------------
    def csearch_rotate(self, coords, torsions, masks, angles, thresh=1.5, max_clashes=0):
        """Every candidate of tscode/torsion_module.py:463-500: (new_coords f64[M, n, 3], rotated_bonds i32[M])."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        torsions = np.ascontiguousarray(torsions, dtype=np.int32).reshape(-1, 4)
        n, nt = len(coords), len(torsions)
        masks = np.ascontiguousarray(masks, dtype=np.uint8).reshape(nt, n)
        angles = np.ascontiguousarray(angles, dtype=np.int32).reshape(-1, nt)
        if coords.ndim != 2 or coords.shape[1] != 3:
            raise ValueError("coords must be (n_atoms, 3)")
        out = np.empty((len(angles), n, 3))
        rb = np.zeros(len(angles), dtype=np.int32)
        check(self.lib.tsc_csearch_rotate(self._h, ptr(coords), C.c_int(n), ptr(torsions), ptr(masks), C.c_int(nt), ptr(angles),
                                          C.c_int64(len(angles)), C.c_double(thresh), C.c_int64(int(max_clashes)), ptr(out), ptr(rb)))
        return out, rb

    def csearch_rotate_dev(self, coords, n_atoms, torsions, masks, n_tors, angles, n_cand, thresh, max_clashes, out, rotated_bonds):
        """The same on device buffers (torch tensors / raw pointers), asynchronous on the engine's stream."""
        check(self.lib.tsc_csearch_rotate_dev(self._h, ptr(coords), C.c_int(n_atoms), ptr(torsions), ptr(masks), C.c_int(n_tors), ptr(angles),
                                              C.c_int64(n_cand), C.c_double(thresh), C.c_int64(int(max_clashes)), ptr(out), ptr(rotated_bonds)))

    def rotate_dihedral_batch(self, coords, torsion, mask, angles):
        """rotate_dihedral (tscode/utils.py:389-414) on structures f64[M, n, 3] sharing torsion and mask: structure s by angles[s]
        degrees (floats).  Returns the rotated copies."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        if coords.ndim != 3 or coords.shape[2] != 3:
            raise ValueError("coords must be (n_structs, n_atoms, 3)")
        torsion = np.ascontiguousarray(torsion, dtype=np.int32).reshape(4)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(coords.shape[1])
        angles = np.ascontiguousarray(np.broadcast_to(np.asarray(angles, dtype=np.float64), (len(coords),)))
        out = np.empty_like(coords)
        check(self.lib.tsc_rotate_dihedral(self._h, ptr(coords), C.c_int64(len(coords)), C.c_int(coords.shape[1]), ptr(torsion), ptr(mask),
                                           ptr(angles), ptr(out)))
        return out

    def torsion_comp_check(self, coords, torsion, mask, thresh=1.5, max_clashes=0):
        """ok i32[M] for structures f64[M, n, 3] sharing one torsion and mask (tscode/numba_functions.py:26-47)."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        if coords.ndim != 3 or coords.shape[2] != 3:
            raise ValueError("coords must be (n_structs, n_atoms, 3)")
        torsion = np.ascontiguousarray(torsion, dtype=np.int32).reshape(4)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(coords.shape[1])
        ok = np.zeros(len(coords), dtype=np.int32)
        check(self.lib.tsc_torsion_comp_check(self._h, ptr(coords), C.c_int64(len(coords)), C.c_int(coords.shape[1]), ptr(torsion), ptr(mask),
                                              C.c_double(thresh), C.c_int64(int(max_clashes)), ptr(ok)))
        return ok

    def prune_heavy(self, heavy, rmsd_thr=0.5, mode=0):
        """prune_conformers_rmsd on the heavy-atom array f64[N, h, 3]. Returns (mask bool[N], per-pass stats)."""
        heavy = np.ascontiguousarray(heavy, dtype=np.float64)
        if heavy.ndim != 3 or heavy.shape[2] != 3:
            raise ValueError("heavy must be (N, h, 3)")
        n, h = heavy.shape[0], heavy.shape[1]
        mask = np.zeros(n, dtype=np.uint8)
        stats = (PassStats * TSC_MAX_PASSES)()
        np_ = C.c_int()
        check(self.lib.tsc_prune_rmsd(self._h, ptr(heavy), C.c_int64(n), C.c_int(h), C.c_double(rmsd_thr), C.c_int(mode), ptr(mask),
                                      stats, C.byref(np_)))
        return mask.astype(bool), _stats_list(stats, np_.value)

    def prune_structures(self, structures, heavy_idx, rmsd_thr=0.5, mode=0):
        """The same from all-atom structures f64[N, n_atoms, 3] (C-contiguous) and the indices of the heavy atoms: the gather of
        tscode/rmsd_pruning.py:178-179 runs on the device (on the host it costs ten times the prune at 57k structures)."""
        n, na = structures.shape[0], structures.shape[1]
        heavy_idx = np.ascontiguousarray(heavy_idx, dtype=np.int32)
        mask = np.zeros(n, dtype=np.uint8)
        stats = (PassStats * TSC_MAX_PASSES)()
        np_ = C.c_int()
        check(self.lib.tsc_prune_structures(self._h, ptr(structures), C.c_int64(n), C.c_int(na), heavy_idx.ctypes.data_as(_lib.c_i32p),
                                            C.c_int(len(heavy_idx)), C.c_double(rmsd_thr), C.c_int(mode), ptr(mask), stats, C.byref(np_)))
        return mask.astype(bool), _stats_list(stats, np_.value)

    def prune_heavy_dev(self, heavy, n, h, rmsd_thr, mode, mask):
        stats = (PassStats * TSC_MAX_PASSES)()
        np_ = C.c_int()
        check(self.lib.tsc_prune_rmsd_dev(self._h, ptr(heavy), C.c_int64(n), C.c_int(h), C.c_double(rmsd_thr), C.c_int(mode),
                                          ptr(mask), stats, C.byref(np_)))
        return _stats_list(stats, np_.value)

    def prune_stepper(self, heavy_dev, n, h, rmsd_thr, mode):
        return PruneStepper(self, heavy_dev, n, h, rmsd_thr, mode)

    # ---- pipeline -----------------------------------------------------------------------------
    def pipeline_dev(self, frags: FragmentSet, frags_dev, conf_idx, rot, pos, n_poses, heavy_idx, clash_thresh, max_clashes,
                     rmsd_thr, mode, clash_mask, structures, keep_mask, keep_mask_host=None):
        heavy_idx = np.ascontiguousarray(heavy_idx, dtype=np.int32)
        n_pass, n_keep = C.c_int64(), C.c_int64()
        stats = (PassStats * TSC_MAX_PASSES)()
        np_ = C.c_int()
        tm = (C.c_float * 4)()
        check(self.lib.tsc_pipeline_dev(self._h, ptr(frags_dev), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos),
                                        C.c_int64(n_poses), heavy_idx.ctypes.data_as(_lib.c_i32p), C.c_int(len(heavy_idx)),
                                        C.c_double(clash_thresh), C.c_int64(int(max_clashes)), C.c_double(rmsd_thr), C.c_int(mode),
                                        ptr(clash_mask), ptr(structures), ptr(keep_mask), ptr(keep_mask_host), C.byref(n_pass),
                                        C.byref(n_keep), stats,
                                        C.byref(np_), tm))
        return PipelineResult(n_pass.value, n_keep.value, stats, np_.value, tm)

    def pipeline_dev_prepare(self, frags: FragmentSet, frags_dev, conf_idx, rot, pos, n_poses, heavy_idx, clash_thresh, max_clashes,
                             rmsd_thr, mode, clash_mask, structures, keep_mask, keep_mask_host=None):
        """The same call with every argument converted once: returns run() -> PipelineResult for callers that repeat the
        step on resident buffers (the ctypes conversions of 25 arguments are host time in front of the first launch)."""
        heavy_idx = np.ascontiguousarray(heavy_idx, dtype=np.int32)
        fixed = (self._h, ptr(frags_dev), *frags.table_args(), ptr(conf_idx), ptr(rot), ptr(pos), C.c_int64(n_poses),
                 heavy_idx.ctypes.data_as(_lib.c_i32p), C.c_int(len(heavy_idx)), C.c_double(clash_thresh), C.c_int64(int(max_clashes)),
                 C.c_double(rmsd_thr), C.c_int(mode), ptr(clash_mask), ptr(structures), ptr(keep_mask), ptr(keep_mask_host))
        keep_alive = (frags, frags_dev, conf_idx, rot, pos, heavy_idx, clash_mask, structures, keep_mask, keep_mask_host)
        fn, byref = self.lib.tsc_pipeline_dev, C.byref
        stats_t, tm_t = PassStats * TSC_MAX_PASSES, C.c_float * 4

        def run(_keep=keep_alive):
            n_pass, n_keep, np_ = C.c_int64(), C.c_int64(), C.c_int()
            stats, tm = stats_t(), tm_t()
            check(fn(*fixed, byref(n_pass), byref(n_keep), stats, byref(np_), tm))
            return PipelineResult(n_pass.value, n_keep.value, stats, np_.value, tm)

        return run


class IpcExchange:
    """The exchange inside the library (include/tscode_hip.h, tsc_xchg_*): a receive area of fine-grained device memory per rank, mapped
    into every other rank of the node; an all-reduce is one kernel that writes into all peers and one that waits for their flags and
    folds what they delivered.  ``handle`` (64 bytes) has to reach every other rank -- the caller's business (``connect_over`` does it
    with torch.distributed) -- then ``connect(handles in rank order)``."""

    def __init__(self, engine: Engine, rank, world, slot_bytes):
        self.e, self.rank, self.world = engine, int(rank), int(world)
        x = C.c_void_p()
        buf = C.create_string_buffer(_lib.XCHG_HANDLE_BYTES)
        check(engine.lib.tsc_xchg_create(engine._h, C.c_int(rank), C.c_int(world), C.c_int64(int(slot_bytes)), C.byref(x), buf))
        self._x, self.handle = x, bytes(buf.raw)
        engine._runs.add(self)               # closed with the engine, before its context goes

    @staticmethod
    def slot_bytes(lib, n, mode=0) -> int:
        b = C.c_int64()
        check(lib.tsc_xchg_slot_bytes(C.c_int64(int(n)), C.c_int(mode), C.byref(b)))
        return b.value

    def connect(self, handles):
        handles = [bytes(h) for h in handles]
        if len(handles) != self.world or any(len(h) != _lib.XCHG_HANDLE_BYTES for h in handles):
            raise ValueError("connect() takes the 64-byte handle of every rank, in rank order")
        check(self.e.lib.tsc_xchg_connect(self._x, C.c_char_p(b"".join(handles))))

    def connect_over(self, dist, group=None):
        """All-gather the handles over a torch.distributed process group and map the peers."""
        handles = [None] * self.world
        dist.all_gather_object(handles, self.handle, group=group)
        self.connect(handles)
        dist.barrier(group=group)            # every rank has mapped every area before the first exchange writes into one

    def set_timeout(self, seconds):
        check(self.e.lib.tsc_xchg_set_timeout(self._x, C.c_double(seconds)))

    def allreduce(self, kind, buf_dev, count):
        """In place over the ranks, enqueued on the engine's stream: kind = XCHG_SUM_I64 / XCHG_MIN_I32 (tscode_amd._lib)."""
        check(self.e.lib.tsc_xchg_allreduce(self._x, C.c_int(kind), ptr(buf_dev), C.c_int64(int(count))))

    def status(self):
        """(exchanges made, exchanges that gave up waiting for a peer).  Read it after a synchronisation."""
        n, t = C.c_int64(), C.c_int()
        check(self.e.lib.tsc_xchg_status(self._x, C.byref(n), C.byref(t)))
        return n.value, t.value

    def check_status(self):
        n, t = self.status()
        if t:
            raise _lib.TscodeHipError(-5, f"{t} of {n} in-library exchanges gave up waiting for a peer rank (tsc_xchg_set_timeout): the results of "
                                     "the runs they belonged to are not valid")

    def close(self):
        if getattr(self, "_x", None):
            if getattr(self.e, "_h", None):
                self.e.lib.tsc_xchg_destroy(self._x)
            self._x = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PruneStepper:
    """Stepping form of one prune run (tsc_prune_*), used to shard a pass over ranks."""

    def __init__(self, engine: Engine, heavy_dev, n, h, rmsd_thr, mode):
        self.e = engine
        self.n = int(n)
        h_ = C.c_void_p()
        check(engine.lib.tsc_prune_create(engine._h, ptr(heavy_dev), C.c_int64(n), C.c_int(h), C.c_double(rmsd_thr), C.c_int(mode),
                                          C.byref(h_)))
        self._p = h_
        self._keep = heavy_dev
        engine._runs.add(self)

    def next_pass(self) -> int:
        k = C.c_int64()
        check(self.e.lib.tsc_prune_next_pass(self._p, C.byref(k)))
        return k.value

    def pass_estimate(self) -> int:
        n = C.c_int64()
        check(self.e.lib.tsc_prune_pass_estimate(self._p, C.byref(n)))
        return n.value

    def run_replicated(self, world, min_pairs) -> int:
        """Every pass that needs no exchange, in one call; returns the k of the first pass that does (left open) or 0."""
        k = C.c_int64()
        check(self.e.lib.tsc_prune_run_replicated(self._p, C.c_int(world), C.c_int64(int(min_pairs)), C.byref(k)))
        return k.value

    def pass_local(self, rank=0, world=1):
        check(self.e.lib.tsc_prune_pass_local(self._p, C.c_int(rank), C.c_int(world)))

    def pass_rows(self, rank, world):
        """The pair search of another rank's row tiles of the open pass (after pass_local), into the same best[]."""
        check(self.e.lib.tsc_prune_pass_rows(self._p, C.c_int(rank), C.c_int(world)))

    # ---- rank-partitioned passes (include/tscode_hip.h) ----
    @staticmethod
    def exchange_words(lib, n, mode) -> int:
        w = C.c_int64()
        check(lib.tsc_prune_exchange_words(C.c_int64(int(n)), C.c_int(mode), C.byref(w)))
        return w.value

    def set_partition(self, rank, world, min_chunks_per_rank, exch_dev):
        """exch_dev: device tensor of at least exchange_words(n) int64 that the caller can all-reduce."""
        check(self.e.lib.tsc_prune_set_partition(self._p, C.c_int(rank), C.c_int(world), C.c_int(min_chunks_per_rank), ptr(exch_dev),
                                                 C.c_int64(exch_dev.numel())))
        self._keep_exch = exch_dev

    def pass_partitioned(self) -> bool:
        f = C.c_int()
        check(self.e.lib.tsc_prune_pass_partitioned(self._p, C.byref(f)))
        return bool(f.value)

    def pass_range(self):
        check(self.e.lib.tsc_prune_pass_range(self._p))

    def pass_merge(self):
        check(self.e.lib.tsc_prune_pass_merge(self._p))

    def views_range(self):
        """(offset, words) of the block of the exchange buffer that holds the cache views of the passes still to run, to be summed
        over the ranks before the open (not partitioned) pass; words = 0: nothing to exchange."""
        p, off, n = C.c_void_p(), C.c_int64(), C.c_int64()
        check(self.e.lib.tsc_prune_views_ptr(self._p, C.byref(p), C.byref(off), C.byref(n)))
        return off.value, n.value

    def views_merged(self):
        check(self.e.lib.tsc_prune_views_merged(self._p))

    def best_ptr(self):
        p, n = C.c_void_p(), C.c_int64()
        check(self.e.lib.tsc_prune_best_ptr(self._p, C.byref(p), C.byref(n)))
        return p.value, n.value

    def use_best_buffer(self, best_dev):
        check(self.e.lib.tsc_prune_use_best_buffer(self._p, ptr(best_dev)))
        self._keep_best = best_dev

    def pass_finish(self):
        check(self.e.lib.tsc_prune_pass_finish(self._p))

    def run_sharded(self, rank, world, min_chunks_per_rank, min_pairs, exch_dev, exchange):
        """The whole pass loop of a sharded run in ONE library call (tsc_prune_run_sharded): the library walks the schedule and calls
        ``exchange(kind, ptr, count)`` for every collective -- kind XCHG_SUM_I64 / XCHG_MIN_I32 (tscode_amd._lib), ``ptr`` the device
        address of ``count`` elements to reduce over the ranks in place, in stream order with the context's stream.  Returns the list
        of exchanges made as (k, kind, count); k < 0 = the cache views in front of pass -k.  An exception raised by ``exchange``
        aborts the run and is re-raised here."""
        from ._lib import EXCHANGE_FN, ExchangeRecord
        failure = []
        user = None
        if isinstance(exchange, IpcExchange):
            # the library's own exchange (tsc_xchg_allreduce has the signature of tsc_exchange_fn): the pass loop calls it directly,
            # no Python frame per collective
            cb, user = C.cast(self.e.lib.tsc_xchg_allreduce, EXCHANGE_FN), exchange._x
        elif exchange is not None:
            def _cb(_user, kind, buf, count):
                try:
                    exchange(int(kind), int(buf), int(count))
                    return 0
                except BaseException as exc:  # noqa: BLE001  (must not propagate through the C frames)
                    failure.append(exc)
                    return 1
            cb = EXCHANGE_FN(_cb)
        else:
            cb = C.cast(None, EXCHANGE_FN)     # (no callback: a world of one needs none)
        log = (ExchangeRecord * (2 * TSC_MAX_PASSES + 2))()
        n_log = C.c_int()
        rc = self.e.lib.tsc_prune_run_sharded(self._p, C.c_int(rank), C.c_int(world), C.c_int(min_chunks_per_rank), C.c_int64(int(min_pairs)),
                                              ptr(exch_dev), C.c_int64(exch_dev.numel() if exch_dev is not None else 0), cb, user, log,
                                              C.c_int(len(log)), C.byref(n_log))
        if failure:
            raise failure[0]
        check(rc)
        if exch_dev is not None:
            self._keep_exch = exch_dev
        return [(log[i].k, log[i].kind, log[i].count) for i in range(n_log.value)]

    def mask_ptr(self) -> int:
        p = C.c_void_p()
        check(self.e.lib.tsc_prune_mask_dev(self._p, C.byref(p)))
        return p.value

    def copy_mask(self, dst_dev):
        check(self.e.lib.tsc_prune_copy_mask_dev(self._p, ptr(dst_dev)))

    def stats(self):
        stats = (PassStats * TSC_MAX_PASSES)()
        np_ = C.c_int()
        check(self.e.lib.tsc_prune_stats(self._p, stats, C.byref(np_)))
        return _stats_list(stats, np_.value)

    def close(self):
        if getattr(self, "_p", None):
            if getattr(self.e, "_h", None):      # (a closed engine has closed its runs already)
                self.e.lib.tsc_prune_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_engines = {}
_engines_lock = threading.Lock()


def get_engine(device: int | None = None) -> Engine:
    """Process-wide engine for a device (default: LOCAL_RANK or 0).  multiembed-style use from several
    processes is safe: each process owns its own HIP context (Threading)."""
    import os
    if device is None:
        device = int(os.environ.get("TSCODE_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _engines_lock:
        eng = _engines.get(device)
        if eng is None:
            eng = _engines[device] = Engine(device)
    return eng
