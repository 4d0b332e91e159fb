"""Drop-in mirror of tscode/rmsd_pruning.py on the MI355X engine.

Same names, arguments and return values as the reference; the arithmetic runs in
libtscode_hip (hand-written gfx950 kernels) through the C ABI.  No CPU fallback.
"""

from __future__ import annotations

import numpy as np

from .engine import get_engine

__all__ = ["prune_conformers_rmsd", "rmsd_and_max_numba", "_rmsd_similarity", "last_prune_stats"]

_last_stats = []


def last_prune_stats():
    """Per-pass statistics (k, active before/after, pair evaluations ...) of the latest prune call."""
    return list(_last_stats)


def prune_conformers_rmsd(structures, atomnos, rmsd_thr=0.5, mode=0, **_ignored):
    """tscode/rmsd_pruning.py:164-206.  Returns (structures[mask], mask).

    ``mode=0`` reproduces the reference bit for bit, including its pair-cache behaviour
    (SURVEY.md F5); ``mode=1`` is the cache-free variant.  Extra keyword arguments are accepted
    and ignored: tscode/atropisomer_module.py:499 passes ``verbose=False``.
    """
    global _last_stats
    structures = np.asarray(structures)
    atomnos = np.asarray(atomnos)
    if structures.ndim != 3 or structures.shape[1] != atomnos.shape[0]:
        raise ValueError("structures must be (N, n_atoms, 3) with len(atomnos) == n_atoms")
    n = structures.shape[0]
    if n == 0:
        return structures[:0], np.zeros(0, dtype=bool)
    heavy_idx = np.flatnonzero(atomnos != 1)                                        # :178
    if len(heavy_idx) == 0:
        raise ZeroDivisionError("no non-hydrogen atoms: the reference divides by zero (rmsd_pruning.py:35)")
    if structures.dtype == np.float64 and structures.flags.c_contiguous and n >= 2048:
        # the gather structures[:, heavy] (:179) on the device: as a strided host copy it is most of a large call's time
        mask, _last_stats = get_engine().prune_structures(structures, heavy_idx, float(rmsd_thr), int(mode))
    else:
        heavy = np.ascontiguousarray(structures[:, heavy_idx], dtype=np.float64)    # :179
        mask, _last_stats = get_engine().prune_heavy(heavy, float(rmsd_thr), int(mode))
    if n > 1 and _last_stats and _last_stats[0].get("nonfinite_input"):
        # the reference's np.linalg.svd (rmsd_pruning.py:19) raises on such a structure as soon as a pair with it is evaluated; the
        # library itself defines the verdict (similar to nothing, kept) and only reports that it met one
        raise np.linalg.LinAlgError("Array must not contain infs or NaNs")
    return structures[mask], mask                                                   # :206


def rmsd_and_max_numba(p, q):
    """tscode/rmsd_pruning.py:6-41: (rmsd, max deviation) after the optimal rotation of p onto q, no centring."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    if p.shape != q.shape or p.ndim != 2 or p.shape[1] != 3:
        raise ValueError("p and q must both be (n, 3)")
    r, m = get_engine().rmsd_pairs(np.stack([p, q]), np.array([[0, 1]], dtype=np.int32))
    return float(r[0]), float(m[0])


def _rmsd_similarity(ref, structures, rmsd_thr=0.5):
    """tscode/rmsd_pruning.py:208-224: True if ref is similar to any of structures (all atoms, no cache)."""
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    if len(structures) == 0:
        return False
    stack = np.concatenate([ref[None], np.ascontiguousarray(structures, dtype=np.float64).reshape(-1, *ref.shape)])
    pairs = np.stack([np.zeros(len(stack) - 1, dtype=np.int32), np.arange(1, len(stack), dtype=np.int32)], axis=1)
    r, m = get_engine().rmsd_pairs(stack, pairs)
    return bool(np.any((r < rmsd_thr) & (m < 2 * rmsd_thr)))
