"""Drop-in names of the moment-of-inertia pruning and the embed-fitness checks.

Reference: ``tscode/algebra.py:165-213`` (get_inertia_moments, get_moi_similarity_matches, diagonalize),
``tscode/optimization_methods.py:327-358`` (prune_by_moment_of_inertia), ``:544-557`` (fitness_check),
``tscode/numba_functions.py:273-288`` (_score_embed_poses).
"""

from __future__ import annotations

import numpy as np

from .engine import get_engine

__all__ = ["get_inertia_moments", "get_moi_similarity_matches", "prune_by_moment_of_inertia", "fitness_check", "fitness_mask",
           "_score_embed_poses"]

# Standard atomic weights as the `periodictable` package tabulates them (tscode/pt.py builds `pt` from it; the package is
# not installed in the build image, so the table could not be checked against it) for the elements TSCoDe systems are
# usually made of; pass `masses=` for anything else or to be independent of the table
_MASS = {1: 1.00794, 3: 6.941, 5: 10.811, 6: 12.0107, 7: 14.0067, 8: 15.9994, 9: 18.9984032, 11: 22.98976928, 12: 24.305,
         13: 26.9815386, 14: 28.0855, 15: 30.973762, 16: 32.065, 17: 35.453, 19: 39.0983, 20: 40.078, 29: 63.546, 30: 65.409,
         34: 78.96, 35: 79.904, 46: 106.42, 53: 126.90447}


_WARNED_MASSES = False


def get_inertia_moments(coords, masses):
    """tscode/algebra.py:165-186 for one structure: (3,) moments, ordered by absolute value (the input is not shifted)."""
    return get_engine().inertia_moments(np.asarray(coords, dtype=np.float64)[None], masses)[0]


def _moi_first_similar(structures, masses, max_deviation):
    eng = get_engine()
    return eng.moi_first_similar(eng.inertia_moments(structures, masses), max_deviation)


def get_moi_similarity_matches(structures, masses, max_deviation=1e-2):
    """tscode/algebra.py:188-205: [(i, j)] with j the first structure after i whose moments all lie within max_deviation."""
    first = _moi_first_similar(structures, masses, max_deviation)
    return [(int(i), int(first[i])) for i in np.flatnonzero(first >= 0)]


def prune_by_moment_of_inertia(structures, atomnos, max_deviation=1e-2, masses=None):
    """tscode/optimization_methods.py:327-358.  ``masses`` (per atom, hydrogens included) overrides the built-in table of
    periodictable values.  The member of a cluster that survives is ``tuple(graph.nodes)[0]`` of a networkx subgraph view,
    as in the reference (and, like there, decided by CPython's set order): the graph step builds the same objects."""
    import networkx as nx
    structures = np.asarray(structures)
    atomnos = np.asarray(atomnos)
    heavy = atomnos != 1
    if masses is None:
        global _WARNED_MASSES
        if not _WARNED_MASSES:                                           # said once, loudly: the table is this package's, not periodictable's
            import warnings
            warnings.warn("prune_by_moment_of_inertia: masses= not given; using tscode_amd's built-in table of standard atomic weights, "
                          "which has not been checked against the `periodictable` package the reference reads its masses from "
                          "(tscode/pt.py).  Pass masses=[pt[z].mass for z in atomnos] to use the reference's own values.", stacklevel=2)
            _WARNED_MASSES = True
        try:
            heavy_masses = np.array([_MASS[int(a)] for a in atomnos[heavy]])
        except KeyError as exc:
            raise ValueError(f"no built-in mass for element Z = {exc.args[0]}; pass masses=") from None
    else:
        heavy_masses = np.asarray(masses, dtype=np.float64)[heavy]
    heavy_structures = np.ascontiguousarray(structures[:, heavy], dtype=np.float64)
    matches = get_moi_similarity_matches(heavy_structures, heavy_masses, max_deviation=max_deviation)
    G = nx.Graph(matches)                                                # :341
    from .numba_functions import _cluster_heads
    mask = np.ones(structures.shape[0], dtype=bool)
    for members, head in _cluster_heads(G):                              # :342-346: tuple(subgraph.nodes)[0] survives
        for i in members:
            if i != head:
                mask[i] = False
    return structures[mask], mask


def _score_embed_poses(structures, constrained_indices, constrained_distances):
    """tscode/numba_functions.py:273-288: float32 [N] sums of |distance - target| over each pose's constraints."""
    return get_engine().embed_scores(structures, constrained_indices, constrained_distances)[0]


def fitness_mask(structures, constrained_indices, constrained_distances, threshold):
    """The loop of tscode/embedder.py:1283-1290 in one launch: fitness_check for every structure (targets of None -> NaN)."""
    dist = np.array([[np.nan if t is None else t for t in row] for row in constrained_distances], dtype=np.float64)
    return get_engine().embed_scores(structures, constrained_indices, dist)[1] < threshold


def fitness_check(coords, constraints, targets, threshold) -> bool:
    """tscode/optimization_methods.py:544-557: the SIGNED deviations from the target distances sum to less than threshold."""
    return bool(fitness_mask(np.asarray(coords, dtype=np.float64)[None], np.asarray(constraints)[None], [list(targets)], threshold)[0])
