"""Mirror of the three helpers of tscode/utils.py the embed loops call.

Host-side NumPy: they run once per (conformer pair, pivot pair), not once per pose.  Pinned by the reference's own
outputs in tests/golden/G10_embed_helpers.npz.
"""

from __future__ import annotations

import numpy as np

from .algebra import rotation_matrix_from_vectors  # noqa: F401  (tscode/utils.py:183-208 lives with the 3x3 helpers)

__all__ = ["TriangleError", "cartesian_product", "polygonize", "rotation_matrix_from_vectors"]


class TriangleError(Exception):
    """tscode/errors.py: raised by polygonize when three lengths do not close a triangle."""


def cartesian_product(*arrays):
    """tscode/utils.py:180-181: all combinations, as rows, in the order NumPy's meshgrid ('xy' indexing) gives them -- NOT
    lexicographic: for two inputs the FIRST index varies fastest ([0 0] [1 0] [0 1] [1 1] ...), for three the third varies
    fastest, then the first, then the second.  The expression is kept as it is because the order of every embed loop
    (conformer pairs, centre pairs, pivots, angle sets) is whatever it produces; pinned by fixture G10."""
    return np.stack(np.meshgrid(*arrays), -1).reshape(-1, len(arrays))


# Which sides of the triangle run backwards (end -> start) in each of the reference's eight orientation patterns, in its order
# (tscode/utils.py:253: the enumeration every trimolecular loop walks; data, pinned by fixture G10)
_TRIANGLE_REVERSED = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 1, 0], [1, 0, 1], [1, 1, 1]], dtype=bool)
_SEGMENT_REVERSED = np.array([[0, 0], [0, 1]], dtype=bool)          # two pivots: the second orientation turns the second segment round


def _oriented(sides, reversed_):
    """sides f64[S, 2, 3] (start, end) under every orientation pattern reversed_[T, S] -> f64[T, S, 2, 3]."""
    return np.where(reversed_[:, :, None, None], sides[None, :, ::-1, :], sides[None])


def polygonize(lengths):
    """tscode/utils.py:210-261: where a cyclical embed puts its pivots -- per orientation pattern, per pivot, a (start, end) pair.

    Two lengths: two segments centred on the origin along x, superposed; the second pattern reverses the second -> f64[2, 2, 2, 3].
    Three lengths: the sides of the triangle (0,0,0) - (L0,0,0) - (x,y,0) walked head to tail, under the reference's eight
    reversal patterns -> f64[8, 3, 2, 3]; TriangleError when the lengths violate the triangle inequality."""
    L = [float(v) for v in lengths]
    if len(L) == 2:
        sides = np.zeros((2, 2, 3))
        sides[:, 0, 0], sides[:, 1, 0] = [-L[0] / 2, -L[1] / 2], [L[0] / 2, L[1] / 2]
        return _oriented(sides, _SEGMENT_REVERSED)
    if len(L) != 3:
        raise AssertionError("polygonize takes two or three lengths")
    if not (L[0] < L[1] + L[2] and L[1] < L[0] + L[2] and L[2] < L[0] + L[1]):
        raise TriangleError(f"Impossible to build a triangle with sides {lengths}")
    # third vertex from the law of cosines, sides 1 and 2 meeting there (the operations and their order are the reference's:
    # the fixture compares the vertices to the last bit)
    a, b, c = np.power(L[0], 2), np.power(L[1], 2), np.power(L[2], 2)
    x = (a - b + c) / (2 * a ** 0.5)
    y = (c - x ** 2) ** 0.5
    corners = np.array([[0.0, 0.0, 0.0], [L[0], 0.0, 0.0], [x, y, 0.0]])
    sides = np.stack([corners, np.roll(corners, -1, axis=0)], axis=1)                        # side s: corner s -> corner s + 1
    return _oriented(sides, _TRIANGLE_REVERSED)
