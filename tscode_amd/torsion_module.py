"""Drop-in names of the conformational-search rotations and their batched form.

Reference: ``tscode/utils.py:389-414`` (rotate_dihedral), ``tscode/numba_functions.py:26-47`` (torsion_comp_check),
``tscode/torsion_module.py:463-509`` (the loop over angle sets of random_csearch / csearch).  The rotation masks
(``_get_rotation_mask``, a graph walk) and the shuffled angle table stay with the caller; everything per candidate
runs on the GPU, one wavefront per candidate.
"""

from __future__ import annotations

import numpy as np

from .algebra import rot_mat_from_pointer
from .engine import get_engine

__all__ = ["rotate_dihedral", "rotate_dihedral_batch", "torsion_comp_check", "csearch_rotate", "csearch_candidates"]


def csearch_rotate(coords, torsions, masks, angles, thresh=1.5, max_clashes=0):
    """All candidates at once: ``(new_coords [M, n, 3], rotated_bonds [M])`` for ``angles [M, n_torsions]`` (degrees, ints)."""
    return get_engine().csearch_rotate(coords, torsions, masks, angles, thresh, max_clashes)


def csearch_candidates(coords, torsions, masks, angles, n_out=100, max_tries=10000, thresh=1.5, block=8192):
    """The ``new_structures`` array of tscode/torsion_module.py:463-509: candidates in the order of ``angles`` (shuffle it
    first, :459), kept iff at least one bond really rotated (:505), until ``n_out`` are kept or row ``max_tries`` is reached.
    The angle table is walked in blocks of ``block`` rows and the walk stops where the reference's loop stops: a
    cartesian-product table (3^12 rows x 100 atoms ...) is never rotated, stored or downloaded as a whole."""
    angles = np.asarray(angles)
    coords = np.asarray(coords, dtype=np.float64)
    out, n_kept = [], 0
    for lo in range(0, len(angles), block):
        new_coords, rotated = csearch_rotate(coords, torsions, masks, angles[lo:lo + block], thresh)
        kept, done = [], False
        for a in np.flatnonzero(rotated != 0):
            kept.append(a)
            n_kept += 1
            if n_kept == n_out or lo + a == max_tries:          # :510 (tested only when a structure has just been kept)
                done = True
                break
        out.append(new_coords[kept])
        if done:
            break
    return np.concatenate(out) if out else np.zeros((0,) + coords.shape)


def rotate_dihedral_batch(coords, dihedral, angles, mask):
    """Structures f64[M, n, 3] sharing torsion and mask, structure s turned by ``angles[s]`` degrees (floats): what a search over
    correction angles (tscode/torsion_module.py:984-1005: rotate, look, rotate back, per angle) becomes as ONE call -- rotate M copies."""
    return get_engine().rotate_dihedral_batch(coords, [int(i) for i in dihedral], mask, angles)


def rotate_dihedral(coords, dihedral, angle, mask=None, indices_to_be_moved=None):
    """tscode/utils.py:389-414.  Like the reference it changes ``coords`` in place and returns it."""
    coords_arr = np.asarray(coords)
    n = len(coords_arr)
    if indices_to_be_moved is not None:
        mask = np.array([i in indices_to_be_moved for i in range(n)])
    if mask is None:
        m = np.zeros(n, dtype=bool)
        m[dihedral[0]] = True
        mask = m
    mask = np.asarray(mask, dtype=bool)
    if float(angle) == 0.0:
        return coords                                    # (the identity; the candidate loop never rotates by zero, :482)
    # ONE structure, one rotation: on the host, like the other single 3x3 helpers of the path (algebra.py) -- the
    # reference's search loops call this once per torsion and candidate and once per 5-degree walk-back step
    # (tscode/torsion_module.py:482-489, :756-763), and a GPU call per molecule-sized rotation (three uploads, a launch, a download, a
    # synchronisation: 50 us and more) would slow them down.  Whole tables of rotations belong on rotate_dihedral_batch / csearch_rotate.
    _, i2, i3, _ = (int(i) for i in dihedral)
    mat = rot_mat_from_pointer(coords_arr[i2] - coords_arr[i3], float(angle))
    center = coords_arr[i3].copy()
    coords_arr[mask] = (mat @ (coords_arr[mask] - center).T).T + center
    return coords


def torsion_comp_check(coords, torsion, mask, thresh=1.5, max_clashes=0) -> int:
    """tscode/numba_functions.py:26-47: 1 if at most ``max_clashes`` distances between the moved side and the rest (the bond
    atoms aside) are below ``thresh``, else 0."""
    coords = np.asarray(coords, dtype=np.float64)
    return int(get_engine().torsion_comp_check(coords[None], torsion, mask, thresh, max_clashes)[0])
