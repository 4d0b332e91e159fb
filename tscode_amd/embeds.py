"""Mirror of the pose-generation boundary of tscode/embeds.py on the MI355X engine.

get_embed keeps the reference's duck-typed signature (objects with .rotation, .position,
.atomcoords); embed_batch is the batched form the embed loops should call instead of one
get_embed per pose: all (R, t) of a run are built on the host, then K1 (+K2) run once.
"""

from __future__ import annotations

import numpy as np

from .algebra import rot_mat_from_pointer, rotation_matrix_from_vectors  # noqa: F401  (host-side singles, re-exported)
from .engine import FragmentSet, get_engine

__all__ = ["get_embed", "embed_batch", "string_embed_poses", "filter_angular_groups"]


def get_embed(mols, conf_ids):
    """tscode/embeds.py:961-969: concatenated coordinates of every molecule in its current pose."""
    mols = list(mols)
    frags = FragmentSet([np.asarray(m.atomcoords[c], dtype=np.float64) for m, c in zip(mols, conf_ids)])
    rot = np.stack([np.asarray(m.rotation, dtype=np.float64) for m in mols])[None]
    pos = np.stack([np.asarray(m.position, dtype=np.float64) for m in mols])[None]
    return get_engine().transform_batch(frags, np.zeros((1, len(mols)), np.int32), rot, pos)[0]


def embed_batch(frag_coords, conf_idx, rot, pos):
    """Batched get_embed: frag_coords[m] f64[n_conf_m, n_m, 3]; conf_idx i32[N, n_mols];
    rot f64[N, n_mols, 3, 3]; pos f64[N, n_mols, 3]  ->  f64[N, sum n_m, 3]."""
    return get_engine().transform_batch(FragmentSet(frag_coords), conf_idx, rot, pos)


def string_embed_poses(coords1, coords2, p1, p2, ref_vec, mol_vec, angles):
    """Pose parameters of the string-embed inner loop (tscode/embeds.py:98-116) for one
    (conformer pair, reactive-centre pair), then the batched embed:

        R0 = rotation_matrix_from_vectors(mol_vec, -ref_vec)                    (:108)
        R  = rot_mat_from_pointer(ref_vec, angle) @ R0   for angle != 0         (:110-112)
        t  = p1 - R @ p2                                                        (:114)

    molecule 1 stays at identity.  Returns (poses f64[len(angles), n1+n2, 3], rot, pos).
    The reactive centres coincide by construction: R @ p2 + t == p1.
    """
    coords1 = np.asarray(coords1, dtype=np.float64)
    coords2 = np.asarray(coords2, dtype=np.float64)
    rot, pos, conf_idx = string_embed_params([p1], [p2], [ref_vec], [mol_vec], [[0, 0]], angles)
    poses = embed_batch([coords1[None], coords2[None]], conf_idx, rot, pos)
    return poses, rot, pos


def string_embed_params(p1, p2, ref_vec, mol_vec, conf_pair, angles):
    """Pose parameters of the whole string embed (tscode/embeds.py:91-116) in one launch: for every site (a conformer pair
    ``conf_pair[s] = (c1, c2)`` with one reactive-centre pair ``p1[s], p2[s], ref_vec[s], mol_vec[s]``) and every angle,
    ``rot [S*A, 2, 3, 3]``, ``pos [S*A, 2, 3]``, ``conf_idx [S*A, 2]`` in the reference's loop order (site-major) -- the
    inputs of ``embed_batch`` / ``compenetration_mask`` / ``pipeline.DevicePipeline``."""
    return get_engine().string_embed_params(p1, p2, ref_vec, mol_vec, conf_pair, angles)


def cyclical_embed_params(start, end, direction, pivot, meanpoint, r0, r1, n_reactive, angle):
    """Rotation and position of every (pose, molecule) of the cyclical embed (tscode/embeds.py:676-713) in one launch.  One
    row per (pose, molecule): ``start, end`` = the polygon side (``vec_pair``), ``direction`` = ``directions[i]``, ``pivot`` /
    ``meanpoint`` = the active pivot's vector and mean point, ``r0, r1`` = the reactive atoms (``n_reactive`` 1 or 2),
    ``angle`` in degrees.  Returns ``(rot [n, 3, 3], pos [n, 3])`` -- ``mol.rotation`` / ``mol.position`` of :711-714."""
    return get_engine().cyclical_embed_params(start, end, direction, pivot, meanpoint, r0, r1, n_reactive, angle)


def filter_angular_groups(poses, group_sizes, rmsd_thr=1.0):
    """The per-group greedy filter of the cyclical embed loops (tscode/embeds.py:713-717, :841-845), batched:

        for pose in group:  keep pose iff  not _rmsd_similarity(pose, kept_so_far, rmsd_thr=1)

    poses f64[n_poses, n, 3] with the groups laid out one after the other; group_sizes int[n_groups].
    Returns bool[n_poses].  One GPU wavefront per group."""
    group_sizes = np.asarray(group_sizes, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(group_sizes)]).astype(np.int32)
    return get_engine().greedy_group_filter(poses, off, float(rmsd_thr))
