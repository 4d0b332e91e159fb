"""Drop-in mirror of the clash functions of tscode/numba_functions.py on the MI355X engine."""

from __future__ import annotations

import numpy as np

from .engine import get_engine

__all__ = ["compenetration_check", "count_clashes", "compenetration_mask"]


def compenetration_mask(coords, ids=None, thresh=1.5, max_clashes=0, return_counts=False):
    """Batched compenetration_check over poses: coords f64[N, n, 3] -> bool[N]
    (the loop of tscode/embedder.py:1243-1248 in one launch)."""
    return get_engine().clash_mask(coords, ids, float(thresh), int(max_clashes), return_counts)


def compenetration_check(coords, ids=None, thresh=1.5, max_clashes=0) -> int:
    """tscode/numba_functions.py:59-105: 1 if the pose has at most max_clashes inter-fragment
    distances below thresh (ids=None: at most max_clashes count_clashes), else 0."""
    coords = np.asarray(coords, dtype=np.float64)
    return int(get_engine().clash_mask(coords[None], ids, float(thresh), int(max_clashes))[0])


def count_clashes(coords) -> int:
    """tscode/numba_functions.py:49-56: ordered atom pairs with 0 < d < 0.5 (each clash counts twice)."""
    coords = np.asarray(coords, dtype=np.float64)
    _, counts = get_engine().clash_mask(coords[None], None, 0.5, 0, return_counts=True)
    return int(counts[0])
