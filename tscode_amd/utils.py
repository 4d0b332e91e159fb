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


def polygonize(lengths):
    """tscode/utils.py:210-261: vertices of the polygon a cyclical embed arranges its pivots on.

    Two lengths: the two orientations of two centred, superposed segments -> f64[2, 2, 2, 3].
    Three lengths: the eight orientation patterns of the triangle's sides -> f64[8, 3, 2, 3]; TriangleError when the lengths
    violate the triangle inequality."""
    assert len(lengths) in (2, 3)
    arr = np.zeros((len(lengths), 2, 3))
    if len(lengths) == 2:
        arr[0, 0] = np.array([-lengths[0] / 2, 0, 0])
        arr[0, 1] = np.array([+lengths[0] / 2, 0, 0])
        arr[1, 0] = np.array([-lengths[1] / 2, 0, 0])
        arr[1, 1] = np.array([+lengths[1] / 2, 0, 0])
        out = np.vstack(([arr], [arr]))
        out[1, 1] *= -1
        return out
    if not all(lengths[i] < lengths[i - 1] + lengths[i - 2] for i in (0, 1, 2)):
        raise TriangleError(f"Impossible to build a triangle with sides {lengths}")
    arr[0, 1] = np.array([lengths[0], 0, 0])
    arr[1, 0] = np.array([lengths[0], 0, 0])
    a, b, c = np.power(lengths[0], 2), np.power(lengths[1], 2), np.power(lengths[2], 2)
    x = (a - b + c) / (2 * a ** 0.5)
    y = (c - x ** 2) ** 0.5
    arr[1, 1] = np.array([x, y, 0])
    arr[2, 0] = np.array([x, y, 0])
    out = np.vstack([[arr]] * 8)
    for t, v in ((1, 2), (2, 1), (3, 1), (3, 2), (4, 0), (5, 0), (5, 1), (6, 0), (6, 2), (7, 0), (7, 1), (7, 2)):
        out[t, v][[0, 1]] = out[t, v][[1, 0]]                       # triangle t: start and end of side v swapped
    return out
