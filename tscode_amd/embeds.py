"""Mirror of the pose-generation boundary of tscode/embeds.py on the MI355X engine.

get_embed keeps the reference's duck-typed signature (objects with .rotation, .position,
.atomcoords); embed_batch is the batched form the embed loops should call instead of one
get_embed per pose: all (R, t) of a run are built on the host, then K1 (+K2) run once.
"""

from __future__ import annotations

import numpy as np

from .algebra import align_vec_pair, norm, rot_mat_from_pointer, rotation_matrix_from_vectors, vec_angle  # noqa: F401  (host-side singles, re-exported)
from .engine import FragmentSet, get_engine
from .utils import cartesian_product, polygonize

__all__ = ["get_embed", "embed_batch", "string_embed_poses", "filter_angular_groups", "string_embed_batch", "cyclical_embed_batch",
           "EmbedTrace", "string_embed", "cyclical_embed"]


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


class EmbedTrace:
    """What an embed driver decided about every candidate, in the reference's loop order: ``clash_ok`` (compenetration_check),
    ``kept`` (the pose was appended), ``group_of`` (cyclical embeds: index of the candidate's angular group), plus the driver's
    own bookkeeping (``sites`` / ``groups``: what each candidate row was built from)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def string_embed_batch(coords1, coords2, centers1, orb_vecs1, centers2, orb_vecs2, angles, clash_thresh=1.5, max_clashes=0,
                       quadruplets=(), tfd_thresh=10, return_trace=False):
    """The whole loop of ``string_embed`` (tscode/embeds.py:91-120) in one GPU call.

    The reference walks conformer pairs x reactive-centre pairs x angles in Python, one pose at a time: rotation and position
    of molecule 2 (:100-114), ``get_embed`` (:116), ``compenetration_check`` (:118), ``is_new_structure`` (:119, torsion
    fingerprints against every pose kept so far -- its "LRU" never evicts) and appends what survives.  Here the candidate list
    is built in that order on the host (a few integers per candidate) and everything per pose runs on the device
    (``tsc_string_embed``); the order-dependent filter included.

    coords1 f64[n_conf1, n1, 3], coords2 f64[n_conf2, n2, 3]   ``mol.atomcoords``
    centers*, orb_vecs* f64[n_conf, n_centers, 3]               ``mol.get_r_atoms(c)[0].center`` / ``.orb_vecs`` per conformer
    angles                                                      ``embedder.systematic_angles`` (degrees)
    quadruplets i32[T, 4]                                       ``_get_quadruplets(get_sum_graph(...))`` (:87), embedded-structure indices

    Returns the kept poses f64[n_kept, n1 + n2, 3] (the reference's ``np.array(poses)``; empty where the reference raises
    ZeroCandidatesError); with ``return_trace`` also an EmbedTrace.
    """
    coords1, coords2 = (np.ascontiguousarray(x, dtype=np.float64) for x in (coords1, coords2))
    centers1, orb_vecs1, centers2, orb_vecs2 = (np.asarray(x, dtype=np.float64) for x in (centers1, orb_vecs1, centers2, orb_vecs2))
    if coords1.ndim != 3 or coords2.ndim != 3 or centers1.ndim != 3 or centers2.ndim != 3:
        raise ValueError("coords* must be (n_conf, n_atoms, 3) and centers* / orb_vecs* (n_conf, n_centers, 3)")
    conf_indices = cartesian_product(np.arange(len(coords1)), np.arange(len(coords2)))               # :73-74
    centre_indices = cartesian_product(np.arange(centers1.shape[1]), np.arange(centers2.shape[1]))   # :77 (counted on conformer 0)
    # site = (conformer pair, centre pair), conformer pairs outermost (:91, :97)
    c1 = np.repeat(conf_indices[:, 0], len(centre_indices))
    c2 = np.repeat(conf_indices[:, 1], len(centre_indices))
    a1 = np.tile(centre_indices[:, 0], len(conf_indices))
    a2 = np.tile(centre_indices[:, 1], len(conf_indices))
    p1, p2 = centers1[c1, a1], centers2[c2, a2]                                                      # :103-104
    ref_vec, mol_vec = orb_vecs1[c1, a1], orb_vecs2[c2, a2]                                          # :105-106
    conf_pair = np.stack([c1, c2], axis=1)
    ok, kept, poses = get_engine().string_embed(FragmentSet([coords1, coords2]), p1, p2, ref_vec, mol_vec, conf_pair, angles, clash_thresh,
                                                max_clashes, quadruplets, tfd_thresh)
    if return_trace:
        return poses, EmbedTrace(clash_ok=ok, kept=kept, sites=np.stack([c1, c2, a1, a2], axis=1), n_angles=len(np.atleast_1d(angles)))
    return poses


MAX_GROUP_POSES = 8192      # poses per (conformers, pivots, orientation) group that tsc_cyclical_embed takes (GF_MAX_GROUP, group_filter.hpp)


def _in_constraints(pair, internal_constraints):
    """``pair in embedder.internal_constraints`` as the reference evaluates it (tscode/embeds.py:642, :777): the constraints are an
    ndarray of index pairs (tscode/embedder.py:499) or an empty list, and ``in`` on an ndarray is ``(array == pair).any()`` -- true as
    soon as ONE index matches in its column, not only for a whole pair."""
    if internal_constraints is None or len(internal_constraints) == 0:
        return False
    return bool((np.asarray(internal_constraints) == np.asarray(pair)).any())


# ---- three molecules (tscode/embeds.py:244-465).  UNPINNED WHERE THE REFERENCE IS UNDEFINED: _get_directions calls vec_angle on 2-vectors
# (:297-299) and algebra.norm reads vec[2] of them (tscode/algebra.py:87) -- an IndexError as plain NumPy, an out-of-bounds read under Numba.
# What follows takes the mathematically intended reading (a plane vector has z = 0); with that one change the reference's own code runs, and
# fixture G18 records it ("reference patched").  Everything else below is the reference's behaviour, quirks included.
_ORIENT3 = np.array([(0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1)], dtype=bool)      # :886-893
_SWAPS3 = ((1, 2), (2, 1), (3, 1), (3, 2), (4, 0), (5, 0), (5, 1), (6, 0), (6, 2), (7, 0), (7, 1), (7, 2))                       # tscode/utils.py:255
_ADJUST_ANGLES = cartesian_product(np.arange(7), np.arange(7), np.arange(7)) * 10.0 - 30.0                                      # :415-420


def _plane_angle(v1, v2):
    """vec_angle (tscode/algebra.py:58-62) of two vectors with up to three components, the missing ones zero.  Degrees."""
    a, b = np.zeros(3), np.zeros(3)
    a[:len(v1)], b[:len(v2)] = v1, v2
    return vec_angle(a, b)


def _triangle3(norms):
    """Vertices of the triangle with sides norms (first vertex at the origin, first side along x): tscode/utils.py:240-250, embeds.py:256-267."""
    a, b, c = norms[0] ** 2, norms[1] ** 2, norms[2] ** 2
    x = (a - b + c) / (2 * a ** 0.5)
    y = (c - x ** 2) ** 0.5
    return np.array([[0.0, 0.0, 0.0], [norms[0], 0.0, 0.0], [x, y, 0.0]])


def _polygonize3(norms):
    """tscode/utils.py:234-261: the eight orientations of the three (start, end) pairs, f64[8, 3, 2, 3]."""
    v = _triangle3(norms)
    arr = np.zeros((3, 2, 3))
    arr[0, 1] = arr[1, 0] = v[1]
    arr[1, 1] = arr[2, 0] = v[2]
    out = np.repeat(arr[None], 8, axis=0)
    for t, m in _SWAPS3:
        out[t, m] = out[t, m][::-1].copy()
    return out


def _directions3(norms):
    """_get_directions for three molecules (embeds.py:244-310): from the middle of every side towards the circumcentre, away from it where
    the opposite angle is obtuse.  `norms` is nudged in place for a right triangle, as the reference does (:287-293)."""
    v = _triangle3(norms)[:, :2]
    a, b, c = v[1, 0], v[2, 0], v[2, 1]
    cc = np.array([a / 2, (b ** 2 + c ** 2 - a * b) / (2 * c)])
    d = [cc - (v[0] + v[1]) / 2, cc - (v[1] + v[2]) / 2, cc - (v[2] + v[0]) / 2]
    if any(np.all(x == 0) for x in d):
        norms[0] += 1e-5
        d = [t[:-1] for t in _directions3(norms)]
    obtuse = (_plane_angle(v[1] - v[0], v[2] - v[0]) > 90, _plane_angle(v[0] - v[1], v[2] - v[1]) > 90, _plane_angle(v[0] - v[2], v[1] - v[2]) > 90)
    flip = (obtuse[2], obtuse[0], obtuse[1])                                                          # :299-301
    out = np.zeros((3, 3))
    for i in range(3):
        out[i, :2] = -d[i] if flip[i] else d[i]
        out[i] = norm(out[i])
    return out


def _adjust_directions3(coords, reactive, cumnums, norms, directions, ids, vecs, pivot, meanpoint, conf_ids):
    """_adjust_directions (embeds.py:312-465): the molecules pre-aligned with `directions`; every combination of -30 .. 30 degree turns of the
    three about their pivots (7^3) scored by the angles between the facing orbitals; the displacement vectors of the best one (the first
    of the smallest cost).  All 343 candidates at once."""
    p = vecs[:, 1] - vecs[:, 0]
    p_mean = vecs.mean(axis=1)
    tri = _triangle3(norms)
    rot, pos = [], []
    for i in range(3):
        mol_direction = meanpoint[i] - coords[i][conf_ids[i]][reactive[i]].mean(axis=0)
        if np.all(mol_direction == 0.0):
            mol_direction = meanpoint[i]
        r_i = align_vec_pair(np.array([p[i], directions[i]]), np.array([pivot[i], mol_direction]))     # :368-369
        rot.append(r_i), pos.append(p_mean[i] - r_i @ meanpoint[i])
    facing = np.zeros((3, 3), dtype=np.int64)            # facing[m, partner] = m's reactive atom that faces `partner` (:376-397)
    for c in ids:
        found = [None, None]
        for m in range(3):
            for index, cum in cumnums[m]:
                if cum == c[0]:
                    found[0] = (m, int(index))
                if cum == c[1]:
                    found[1] = (m, int(index))
        (m0, i0), (m1, i1) = found
        facing[m0, m1], facing[m1, m0] = i0, i1
    at = lambda m, partner: rot[m] @ coords[m][0][facing[m, partner]] + pos[m]                        # (conformer 0: as the reference, :403-411)
    # R(axis, angle) for the 7 angles of every molecule, then the 343 combinations by indexing
    steps = np.arange(7) * 10.0 - 30.0
    turn = [np.array([rot_mat_from_pointer(p[i], a) for a in steps]) for i in range(3)]               # [7, 3, 3] each
    idx = ((_ADJUST_ANGLES + 30.0) / 10.0).round().astype(np.int64)                                   # [343, 3]
    new = {}
    for m, partner in ((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)):
        new[m, partner] = turn[m][idx[:, m]] @ at(m, partner)                                         # [343, 3]

    def ang(u, w):                                                                                    # vec_angle row by row
        un = u / np.sqrt((u * u).sum(axis=1, keepdims=True))
        wn = w / np.sqrt((w * w).sum(axis=1, keepdims=True))
        return np.degrees(np.arccos(np.clip((un * wn).sum(axis=1), -1.0, 1.0)))
    cost = ang(tri[0] - new[0, 2], new[2, 0] - tri[0]) + ang(tri[1] - new[0, 1], new[1, 0] - tri[1]) + ang(tri[2] - new[2, 1], new[1, 2] - tri[2])
    best = int(np.argmin(cost))                                                                       # (the first of the smallest: sorted() is stable)
    return np.array([p_mean[0] - (new[0, 1][best] + new[0, 2][best]) / 2, p_mean[1] - (new[1, 0][best] + new[1, 2][best]) / 2,
                     p_mean[2] - (new[2, 0][best] + new[2, 1][best]) / 2])


def _trimolecular_groups(coords, reactive, pivots, cumnums, pairings, internal_constraints):
    """The (conformers, pivots, orientation) groups of a three-molecule cyclical embed under RIGID (embeds.py:470-655): one 23-field record
    per molecule and group, the facing atoms of every group, and (conformers, pivots, orientations) per pivot triple for the trace."""
    blocks, group_ids, group_meta = [], [], []
    for conf_ids in cartesian_product(*[np.arange(len(c)) for c in coords]):
        pv = [pivots[m][conf_ids[m]] for m in range(3)]
        vec = [np.asarray(q[0], dtype=np.float64).reshape(-1, 3) for q in pv]
        mean = [np.asarray(q[1], dtype=np.float64).reshape(-1, 3) for q in pv]
        cum = [np.asarray(q[2]).reshape(-1, 2) for q in pv]
        for pi in cartesian_product(*[np.arange(len(v)) for v in vec]):
            pivot = [vec[m][pi[m]] for m in range(3)]
            meanpoint = [mean[m][pi[m]] for m in range(3)]
            norms = np.array([np.linalg.norm(q) for q in pivot])                                     # :487
            if not all(norms[i] < norms[i - 1] + norms[i - 2] for i in (0, 1, 2)):                    # an impossible triangle: skipped under RIGID (:527-573)
                continue
            poly = _polygonize3(norms)
            directions = _directions3(norms)
            for v in range(8):
                c3 = [cum[m][pi[m]][::-1] if _ORIENT3[v][m] else cum[m][pi[m]] for m in range(3)]
                ids = [sorted([int(c3[0][1]), int(c3[1][0])]), sorted([int(c3[1][1]), int(c3[2][0])]), sorted([int(c3[2][1]), int(c3[0][0])])]   # :895-897
                if pairings and not all((list(pair) in ids) or _in_constraints(pair, internal_constraints) for pair in pairings):                 # :642
                    continue
                # (carried over from orientation to orientation, as the reference's loop does: `directions` is rebound inside it, :650-655)
                directions = _adjust_directions3(coords, reactive, cumnums, norms, directions, ids, poly[v], pivot, meanpoint, conf_ids)
                rec = np.zeros((3, 23))
                for m in range(3):
                    r = coords[m][conf_ids[m]][reactive[m]]
                    rec[m, 0:3], rec[m, 3:6], rec[m, 6:9] = poly[v, m, 0], poly[v, m, 1], directions[m]
                    rec[m, 9:12], rec[m, 12:15] = pivot[m], meanpoint[m]
                    rec[m, 15:18], rec[m, 18:21] = r[0], (r[1] if len(r) == 2 else r[0])
                    rec[m, 21], rec[m, 22] = len(r), conf_ids[m]
                blocks.append(rec), group_ids.append(ids)
                group_meta.append((tuple(int(c) for c in conf_ids), tuple(int(i) for i in pi), v))
    return (np.array(blocks).reshape(-1, 3, 23), np.array(group_ids, dtype=np.int64).reshape(-1, 3, 2), group_meta)


def cyclical_embed_batch(mols, systematic_angles, clash_thresh=1.5, max_clashes=0, rigid_shortcut=True, max_norm_delta=5, pairings=None,
                         internal_constraints=(), rmsd_thr=1, return_trace=False):
    """The loops of a BImolecular ``cyclical_embed`` (tscode/embeds.py:470-732, rigid shortcut :734-860) in one GPU call.

    Per (conformer pair, pivot pair) the reference takes the pivots' lengths, ``polygonize``s them into two orientations of two
    superposed segments, and per orientation walks ``systematic_angles``: alignment + step rotation of both molecules (:659-709),
    ``get_embed``, ``compenetration_check``, and keeps a pose unless it is ``_rmsd_similarity``-similar (threshold 1) to a pose
    already kept in the same (conformers, pivots, orientation) group.  Here the host lays out one row per (pose, molecule) in that
    order and ``tsc_cyclical_embed`` does everything per pose, the greedy per-group filter included.

    mols: two objects (or dicts) with
        coords            f64[n_conf, n, 3]            ``mol.atomcoords``
        reactive_indices  1 or 2 atom indices          ``mol.reactive_indices``
        pivots            per conformer: (pivot f64[P, 3], meanpoint f64[P, 3], cumnums int[P, 2])
                          ``[(p.pivot, p.meanpoint, (p.start_atom.cumnum, p.end_atom.cumnum)) for p in mol.pivots[c]]``
    systematic_angles     f64[A, 2]                    ``embedder.systematic_angles`` (tscode/embedder.py:714-715)
    rigid_shortcut        True: ``_fast_bimol_rigid_cyclical_embed`` (pivot pairs are skipped when their lengths differ by MORE than
                          max_norm_delta, :762); False: the general loop (skipped unless they differ by LESS, :492).  The general
                          loop's bending of such pairs (ase_bend, an external optimiser) is outside this path: they are skipped, as
                          the reference does under RIGID.
    pairings              ``embedder.pairings_table.values()``: orientations that do not contain every pairing are skipped (:642)

    Returns ``(poses f64[n_kept, n_total, 3], constrained_indices int[n_kept, 2, 2])`` -- the reference's return value and
    ``embedder.constrained_indices`` -- plus an EmbedTrace with ``return_trace``.

    THREE molecules (RIGID: triangles that cannot be closed are skipped, the reference bends them otherwise): each molecule also brings
    ``reactive_cumnums`` int[R, 2] = (atom index, cumnum) of its reactive atoms (``[(i, a.cumnum) for i, a in
    mol.reactive_atoms_classes_dict[0].items()]``).  UNPINNED WHERE THE REFERENCE IS UNDEFINED: its ``_get_directions`` calls ``vec_angle``
    on 2-vectors (:297-299) and ``algebra.norm`` reads ``vec[2]`` of them (tscode/algebra.py:87) -- an unchecked out-of-bounds read under
    Numba, an IndexError as plain NumPy.  This driver takes the intended reading (z = 0); fixture G18 is the reference's own code run with
    that one change.  ``rigid_shortcut`` / ``max_norm_delta`` do not apply to three molecules.
    """
    get = lambda m, k: m[k] if isinstance(m, dict) else getattr(m, k)
    if len(mols) not in (2, 3):
        raise ValueError("cyclical_embed_batch embeds two or three molecules")
    n_mols = len(mols)
    coords = [np.ascontiguousarray(get(m, "coords"), dtype=np.float64) for m in mols]
    reactive = [np.atleast_1d(np.asarray(get(m, "reactive_indices"), dtype=np.int64)) for m in mols]
    pivots = [get(m, "pivots") for m in mols]
    angles = np.asarray(systematic_angles, dtype=np.float64).reshape(-1, n_mols)
    A = len(angles)
    if A > MAX_GROUP_POSES:
        raise ValueError(f"{A} angle sets per group: tsc_cyclical_embed takes groups of up to {MAX_GROUP_POSES} poses (rotation steps up to 89)")
    if n_mols == 3:
        cumnums = [np.asarray(get(m, "reactive_cumnums"), dtype=np.int64).reshape(-1, 2) for m in mols]
        blocks3, group_ids3, meta3 = _trimolecular_groups(coords, reactive, pivots, cumnums, pairings, internal_constraints)
        return _run_cyclical_groups(coords, blocks3, group_ids3, [(c, p, np.array([v])) for c, p, v in meta3], angles, clash_thresh, max_clashes, rmsd_thr,
                                    return_trace, flat_meta=True)
    directions = np.array([[0.0, 1.0, 0.0], [0.0, -1.0, 0.0]])                                      # _get_directions for two, :252-253 / :768
    conf_indices = cartesian_product(*[np.arange(len(c)) for c in coords])                           # :470-471
    blocks, group_ids, group_meta = [], [], []
    # Every conformer of a molecule has the same number of pivots (the usual case: pivots are a property of the reactive atoms) and no
    # pairing filter: all conformer pairs at once.  The arrays below are the loop's, with one more leading axis; flattening them under
    # the mask in row-major order (conformer pair, pivot pair, orientation) is the order the loop appends in.
    uniform = not pairings and len(conf_indices) > 0 and all(len({len(np.asarray(p[0]).reshape(-1, 3)) for p in pivots[m]}) == 1 for m in range(2))
    if uniform:
        vecs = [np.stack([np.asarray(p[0], dtype=np.float64).reshape(-1, 3) for p in pivots[m]]) for m in range(2)]      # [conformer, pivot, 3]
        means = [np.stack([np.asarray(p[1], dtype=np.float64).reshape(-1, 3) for p in pivots[m]]) for m in range(2)]
        cums = [np.stack([np.asarray(p[2]).reshape(-1, 2) for p in pivots[m]]) for m in range(2)]
        pi = cartesian_product(*[np.arange(v.shape[1]) for v in vecs])
        if len(pi):
            ci = np.asarray(conf_indices, dtype=np.int64)
            nc, npi = len(ci), len(pi)
            pick = lambda arr, m: arr[m][ci[:, m]][:, pi[:, m]]                                      # [conformer pair, pivot pair, ...]
            pv = [pick(vecs, m) for m in range(2)]
            norms = np.stack([np.linalg.norm(pv[m], axis=2) for m in range(2)], axis=2)              # :487
            delta = np.abs(norms[:, :, 0] - norms[:, :, 1])
            ok = ~(delta > max_norm_delta) if rigid_shortcut else (delta < max_norm_delta)           # :762 / :492-496, :630-631
            half = norms / 2.0
            rec = np.zeros((nc, npi, 2, 2, 23))                                                      # [conformer pair, pivot pair, orientation, molecule, field]
            for m in range(2):
                rec[:, :, 0, m, 0], rec[:, :, 0, m, 3] = -half[:, :, m], +half[:, :, m]
                rec[:, :, 1, m, 0], rec[:, :, 1, m, 3] = (-half[:, :, m], +half[:, :, m]) if m == 0 else (+half[:, :, m], -half[:, :, m])
                r = coords[m][ci[:, m]][:, reactive[m]]                                              # reactive_coords per conformer pair, :667
                rec[:, :, :, m, 6:9] = directions[m]
                rec[:, :, :, m, 9:12] = pv[m][:, :, None, :]
                rec[:, :, :, m, 12:15] = pick(means, m)[:, :, None, :]
                rec[:, :, :, m, 15:18] = r[:, 0][:, None, None, :]
                rec[:, :, :, m, 18:21] = (r[:, 1] if r.shape[1] == 2 else r[:, 0])[:, None, None, :]
                rec[:, :, :, m, 21], rec[:, :, :, m, 22] = r.shape[1], ci[:, m][:, None, None]
            c0, c1 = pick(cums, 0), pick(cums, 1)
            ids = np.empty((nc, npi, 2, 2, 2), dtype=np.int64)
            ids[:, :, :, :, 0] = c0[:, :, None, :]
            ids[:, :, 0, :, 1], ids[:, :, 1, :, 1] = c1, c1[:, :, ::-1]
            take = np.repeat(ok[:, :, None], 2, axis=2)
            blocks.append(rec[take])
            group_ids.append(ids[take])
            if return_trace:
                for n in range(nc):
                    qs, vs = np.nonzero(take[n])
                    if len(qs):
                        group_meta.append((ci[n], pi[qs], vs))
        conf_indices = ()
    for conf_ids in conf_indices:
        pv = [pivots[m][conf_ids[m]] for m in range(2)]
        vec = [np.asarray(p[0], dtype=np.float64).reshape(-1, 3) for p in pv]
        mean = [np.asarray(p[1], dtype=np.float64).reshape(-1, 3) for p in pv]
        cum = [np.asarray(p[2]).reshape(-1, 2) for p in pv]
        rc = [coords[m][conf_ids[m]][reactive[m]] for m in range(2)]                                 # reactive_coords, :667
        # every pivot pair of this conformer pair at once (the order of pivots_indices, :477), both polygon orientations
        pi = cartesian_product(*[np.arange(len(v)) for v in vec])
        if not len(pi):
            continue
        norms = np.stack([np.linalg.norm(vec[m][pi[:, m]], axis=1) for m in range(2)], axis=1)       # :487
        delta = np.abs(norms[:, 0] - norms[:, 1])
        ok = ~(delta > max_norm_delta) if rigid_shortcut else (delta < max_norm_delta)               # :762 / :492-496, :630-631
        pi, norms = pi[ok], norms[ok]
        n_pi = len(pi)
        if not n_pi:
            continue
        # polygonize(norms) for two lengths (tscode/utils.py:224-232): both segments centred on the origin along x,
        # orientation 1 = the second one reversed (vertices_out[1, 1] *= -1)
        half = norms / 2.0
        rec = np.zeros((n_pi, 2, 2, 23))                                                             # [pivot pair, orientation, molecule, field]
        for m in range(2):
            rec[:, 0, m, 0], rec[:, 0, m, 3] = -half[:, m], +half[:, m]                                # start.x, end.x
            rec[:, 1, m, 0], rec[:, 1, m, 3] = (-half[:, m], +half[:, m]) if m == 0 else (+half[:, m], -half[:, m])
            r = rc[m]
            rec[:, :, m, 6:9] = directions[m]
            rec[:, :, m, 9:12] = vec[m][pi[:, m]][:, None, :]
            rec[:, :, m, 12:15] = mean[m][pi[:, m]][:, None, :]
            rec[:, :, m, 15:18], rec[:, :, m, 18:21] = r[0], r[1] if len(r) == 2 else r[0]
            rec[:, :, m, 21], rec[:, :, m, 22] = len(r), conf_ids[m]
        # which atoms face each other in each of the two orientations (_get_cyclical_reactive_indices for two molecules, :862-883):
        # orientation 1 walks the second molecule's pivot backwards -- for all pivot pairs at once, [pivot pair, orientation, 2, 2]
        c0, c1 = cum[0][pi[:, 0]], cum[1][pi[:, 1]]
        ids = np.empty((n_pi, 2, 2, 2), dtype=np.int64)
        ids[:, :, :, 0] = c0[:, None, :]
        ids[:, 0, :, 1], ids[:, 1, :, 1] = c1, c1[:, ::-1]
        take = np.ones((n_pi, 2), dtype=bool)
        if pairings:                                                                                 # :642 / :777 (rare: a user's pairing constraints)
            for q in range(n_pi):
                for v in range(2):
                    faces = ids[q, v].tolist()
                    take[q, v] = all((list(pair) in faces) or _in_constraints(pair, internal_constraints) for pair in pairings)
        blocks.append(rec[take])                                                                     # (row-major: pivot pair, then orientation -- the loop's order)
        group_ids.append(ids[take])
        qs, vs = np.nonzero(take)
        group_meta.append((np.asarray(conf_ids, dtype=np.int64), pi[qs], vs))
    n_groups = sum(len(b) for b in blocks)
    blocks = np.concatenate(blocks) if n_groups else np.zeros((0, 2, 23))                            # [group, molecule, field]
    group_ids = np.concatenate(group_ids) if n_groups else np.zeros((0, 2, 2), dtype=np.int64)
    return _run_cyclical_groups(coords, blocks, group_ids, group_meta, angles, clash_thresh, max_clashes, rmsd_thr, return_trace)


def _run_cyclical_groups(coords, blocks, group_ids, group_meta, angles, clash_thresh, max_clashes, rmsd_thr, return_trace, flat_meta=False):
    """The groups' records -> one row per (pose, molecule) -> ONE library call (tsc_cyclical_embed): pose parameters, embedding, clash
    verdicts, the greedy per-group similarity filter; the kept poses and the facing atoms of their groups."""
    n_mols = len(coords)
    A = len(angles)
    n_total = sum(c.shape[1] for c in coords)
    n_groups = len(blocks)
    if not n_groups:
        out = (np.zeros((0, n_total, 3)), np.zeros((0, n_mols, 2), dtype=np.int64))
        return (*out, EmbedTrace(clash_ok=np.zeros(0, bool), kept=np.zeros(0, bool), group_of=np.zeros(0, np.int64), groups=[])) if return_trace else out
    # one row per (pose, molecule): a group's A poses share everything but the angles.  Every field is expanded straight into the
    # contiguous array the library takes (expanding the whole 23-field record and slicing it afterwards copied everything twice)
    expand = lambda lo, hi, dtype=np.float64: np.repeat(np.ascontiguousarray(blocks[:, :, lo:hi], dtype=dtype), A, axis=0).reshape(-1, hi - lo)
    angle_rows = np.tile(angles, (n_groups, 1)).reshape(-1)                                          # angles[i] for molecule i, :665
    group_off = np.arange(n_groups + 1, dtype=np.int32) * A
    ok, kept, poses = get_engine().cyclical_embed(FragmentSet(coords), expand(0, 3), expand(3, 6), expand(6, 9), expand(9, 12), expand(12, 15),
                                                  expand(15, 18), expand(18, 21), expand(21, 22, np.int32).reshape(-1), angle_rows,
                                                  expand(22, 23, np.int32).reshape(-1), group_off, clash_thresh, max_clashes, rmsd_thr)
    group_of = np.repeat(np.arange(n_groups), A)
    constrained = group_ids[group_of[kept]].reshape(-1, n_mols, 2)                                   # :718
    if return_trace:
        groups, g = [], 0
        for conf, pis, vs in group_meta:                                                             # (conformers, pivots, orientation, facing atoms) per group
            for q in range(len(vs)):
                groups.append((tuple(int(c) for c in conf), tuple(int(i) for i in (pis if flat_meta else pis[q])), int(vs[q]), group_ids[g].tolist()))
                g += 1
        return poses, constrained, EmbedTrace(clash_ok=ok, kept=kept, group_of=group_of, groups=groups)
    return poses, constrained


# ----------------------------------------------------------------------------------------------------------------------
# Drop-ins with the reference's own signatures: what tscode_amd.install() puts in place of tscode.embeds.string_embed /
# cyclical_embed (and of the names tscode/embedder.py:39-40 bound at import time).  They read the plain arrays out of the
# Embedder's molecules, make ONE call of the batch drivers above, and leave the same return value and side effects.  The
# graph helpers and the exception type are the reference's own (taken from the imported tscode.embeds), and whatever these
# drivers do not cover is handed to the reference's original function.

_originals = {}          # name -> the reference's function, recorded by install()


def _reference_embeds():
    import sys
    ref = sys.modules.get("tscode.embeds")
    if ref is None:
        raise RuntimeError("tscode.embeds is not imported: these drop-ins run inside a live TSCoDe (tscode_amd.install()); "
                           "from plain arrays call string_embed_batch / cyclical_embed_batch")
    return ref


def string_embed(embedder):
    """Drop-in for tscode/embeds.py:36-133 ``string_embed(embedder)``: same poses, same ``embedder.constrained_indices``, same
    ZeroCandidatesError -- the three nested Python loops replaced by one GPU call."""
    ref = _reference_embeds()
    assert len(embedder.objects) == 2
    mol1, mol2 = embedder.objects
    embedder.log(f"\n--> Performing string embed ({ref.pretty_num(embedder.candidates)} candidates, MI355X engine)")
    constrained_indices = [[int(mol1.reactive_indices[0]), int(mol2.reactive_indices[0] + embedder.ids[0])]]            # :84-85
    quadruplets = ref._get_quadruplets(ref.get_sum_graph((mol1.graph, mol2.graph), constrained_indices))               # :87
    arrays = []
    for mol in (mol1, mol2):
        n_centers = len(mol.get_centers(0)[0])                                                                         # :77
        r_atoms = [mol.get_r_atoms(c)[0] for c in range(len(mol.atomcoords))]
        arrays.append((np.array([np.asarray(r.center)[:n_centers] for r in r_atoms]), np.array([np.asarray(r.orb_vecs)[:n_centers] for r in r_atoms])))
    poses = string_embed_batch(np.asarray(mol1.atomcoords), np.asarray(mol2.atomcoords), arrays[0][0], arrays[0][1], arrays[1][0], arrays[1][1],
                               embedder.systematic_angles, clash_thresh=embedder.options.clash_thresh, quadruplets=quadruplets)
    if not len(poses):
        msg = ("\n--> Cyclical embed did not find any suitable disposition of molecules.\n"
               "    This is probably because the two molecules cannot find a correct interlocking pose.\n"
               "    Try expanding the conformational space with the csearch> operator or see the SHRINK keyword.")
        embedder.log(msg, p=False)
        raise ref.ZeroCandidatesError(msg)
    embedder.constrained_indices = ref._get_string_constrained_indices(embedder, len(poses))                           # :131
    return poses


# Three molecules through this driver?  False (default): handed to the reference's own function, which is undefined there (vec_angle on 2-vectors,
# embeds.py:297-299 / algebra.py:87: an IndexError as plain NumPy, an out-of-bounds read under Numba).  True: the batch driver with the
# intended reading (z = 0), for RIGID runs -- what fixture G18 records; the result is then this package's, labelled so in the log.
TRIMOLECULAR = False


def cyclical_embed(embedder, max_norm_delta=5):
    """Drop-in for tscode/embeds.py:234-860 ``cyclical_embed(embedder)`` for two molecules (the rigid shortcut, and the general
    loop where it bends nothing); pivot pairs the reference would bend go to the reference's own function, and so do trimolecular
    embeds unless ``tscode_amd.embeds.TRIMOLECULAR`` is set (see there)."""
    ref = _reference_embeds()
    original = _originals.get("cyclical_embed")
    mols = embedder.objects
    rigid = bool(embedder.options.rigid)
    if len(mols) == 3 and TRIMOLECULAR and rigid and len(embedder.systematic_angles) <= MAX_GROUP_POSES:
        packed = []
        for mol in mols:
            pivots = [(np.array([p.pivot for p in pv]).reshape(-1, 3), np.array([p.meanpoint for p in pv]).reshape(-1, 3),
                       np.array([[p.start_atom.cumnum, p.end_atom.cumnum] for p in pv]).reshape(-1, 2)) for pv in mol.pivots]
            packed.append(dict(coords=np.asarray(mol.atomcoords), reactive_indices=np.asarray(mol.reactive_indices), pivots=pivots,
                               reactive_cumnums=np.array([[int(i), int(a.cumnum)] for i, a in mol.reactive_atoms_classes_dict[0].items()]).reshape(-1, 2)))
        embedder.log(f"\n--> Performing {embedder.embed} embed of three molecules ({ref.pretty_num(embedder.candidates)} candidates, MI355X engine; "
                     "vec_angle of plane vectors read with z = 0: the reference is undefined there)")
        pairings = [list(p) for p in embedder.pairings_table.values()] if embedder.pairings_table else None
        poses, constrained = cyclical_embed_batch(packed, embedder.systematic_angles, clash_thresh=embedder.options.clash_thresh, pairings=pairings,
                                                  internal_constraints=getattr(embedder, "internal_constraints", ()))
        embedder.constrained_indices = constrained
        if not len(poses):
            msg = ("\n--> Cyclical embed did not find any suitable disposition of molecules.\n"
                   "    This is probably because one molecule has two reactive centers at a great distance,\n"
                   "    preventing the other two molecules from forming a closed, cyclical structure.")
            embedder.log(msg, p=False)
            raise ref.ZeroCandidatesError(msg)
        return poses

    def theirs():
        if original is None:
            raise RuntimeError("this embed needs the reference's own cyclical_embed (three molecules, or a pivot pair it would bend): "
                               "tscode_amd.install() records it")
        return original(embedder, max_norm_delta) if max_norm_delta != 5 else original(embedder)
    if len(mols) != 2 or len(embedder.systematic_angles) > MAX_GROUP_POSES:       # (groups beyond the library's size: STEPS > 89)
        return theirs()
    packed = []
    for mol in mols:
        pivots = [(np.array([p.pivot for p in pv]).reshape(-1, 3), np.array([p.meanpoint for p in pv]).reshape(-1, 3),
                   np.array([[p.start_atom.cumnum, p.end_atom.cumnum] for p in pv]).reshape(-1, 2)) for pv in mol.pivots]
        packed.append(dict(coords=np.asarray(mol.atomcoords), reactive_indices=np.asarray(mol.reactive_indices), pivots=pivots))
    if not rigid:
        # the general loop bends a molecule when two pivots differ by max_norm_delta or more (:574-628): not this driver's
        for pv0 in packed[0]["pivots"]:
            for pv1 in packed[1]["pivots"]:
                n0, n1 = np.linalg.norm(pv0[0], axis=1), np.linalg.norm(pv1[0], axis=1)
                if len(n0) and len(n1) and not (np.abs(n0[:, None] - n1[None, :]) < max_norm_delta).all():
                    return theirs()
    embedder.log(f"\n--> Performing {embedder.embed} embed ({ref.pretty_num(embedder.candidates)} candidates, MI355X engine)")
    pairings = [list(p) for p in embedder.pairings_table.values()] if embedder.pairings_table else None
    poses, constrained = cyclical_embed_batch(packed, embedder.systematic_angles, clash_thresh=embedder.options.clash_thresh,
                                              rigid_shortcut=rigid, max_norm_delta=max_norm_delta, pairings=pairings,
                                              internal_constraints=getattr(embedder, "internal_constraints", ()))
    embedder.constrained_indices = constrained                                                                         # :723 / :851
    if not len(poses):
        msg = ("\n--> Cyclical embed did not find any suitable disposition of molecules.\n"
               "    This is probably because one molecule has two reactive centers at a great distance,\n"
               "    preventing the other two molecules from forming a closed, cyclical structure.")
        embedder.log(msg, p=False)
        raise ref.ZeroCandidatesError(msg)
    return poses
