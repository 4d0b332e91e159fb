"""Mirror of the hot-path part of tscode/algebra.py (+ the 2 helpers of tscode/utils.py the embed
loops call).  all_dists / transform_coords run on the GPU; the 3x3 pose-parameter helpers are
host-side NumPy (they produce one (R, t) per pose and molecule)."""

from __future__ import annotations

import math

import numpy as np

from .engine import FragmentSet, get_engine

__all__ = ["all_dists", "transform_coords", "norm", "norm_of", "vec_angle", "rot_mat_from_pointer",
           "quaternion_to_rotation_matrix", "align_vec_pair", "rotation_matrix_from_vectors"]


def all_dists(A, B):
    """tscode/algebra.py:98-157: matrix of distances between two point sets."""
    return get_engine().all_dists(A, B)


def transform_coords(coords, rot, pos):
    """tscode/algebra.py:390-400: (rot @ coords.T).T + pos."""
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    out = get_engine().transform_batch(FragmentSet([coords]), np.zeros((1, 1), np.int32),
                                       np.asarray(rot, dtype=np.float64)[None, None], np.asarray(pos, dtype=np.float64)[None, None])
    return out[0]


def norm_of(vec):
    """tscode/algebra.py:89-96."""
    return math.sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2])


def norm(vec):
    """tscode/algebra.py:80-87."""
    return np.asarray(vec, dtype=np.float64) / norm_of(vec)


def vec_angle(v1, v2):
    """tscode/algebra.py:58-62, degrees."""
    d = float(np.dot(norm(v1), norm(v2)))
    return math.acos(min(1.0, max(-1.0, d))) * 180 / math.pi


def quaternion_to_rotation_matrix(Q):
    """tscode/algebra.py:284-323; Q = (x, y, z, w), scalar last."""
    q1, q2, q3, q0 = (float(v) for v in Q)
    return np.array([[2 * (q0 * q0 + q1 * q1) - 1, 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2)],
                     [2 * (q1 * q2 + q0 * q3), 2 * (q0 * q0 + q2 * q2) - 1, 2 * (q2 * q3 - q0 * q1)],
                     [2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 2 * (q0 * q0 + q3 * q3) - 1]])


def rot_mat_from_pointer(pointer, angle):
    """tscode/algebra.py:325-344: rotation of `angle` degrees about `pointer`."""
    pointer = np.asarray(pointer, dtype=np.float64)
    assert pointer.shape[0] == 3
    u = norm(pointer)
    half = angle * (math.pi / 180) / 2
    s = math.sin(half)
    return quaternion_to_rotation_matrix((s * u[0], s * u[1], s * u[2], math.cos(half)))


def align_vec_pair(ref, tgt):
    """tscode/algebra.py:258-282: rotation that best aligns the two tgt vectors to the two ref vectors."""
    ref = np.asarray(ref, dtype=np.float64)
    tgt = np.asarray(tgt, dtype=np.float64)
    B = ref[0][:, None] * tgt[0][None, :] + ref[1][:, None] * tgt[1][None, :]
    u, s, vh = np.linalg.svd(B)
    if np.linalg.det(u @ vh) < 0:
        u[:, -1] = -u[:, -1]
    return np.ascontiguousarray(u @ vh)


def rotation_matrix_from_vectors(vec1, vec2):
    """tscode/utils.py:183-208: rotation taking vec1 onto vec2 (exact-zero tests kept)."""
    vec1 = np.asarray(vec1, dtype=np.float64)
    vec2 = np.asarray(vec2, dtype=np.float64)
    assert vec1.shape == (3,) and vec2.shape == (3,)
    a, b = vec1 / norm_of(vec1), vec2 / norm_of(vec2)
    v = np.cross(a, b)
    s = norm_of(v)
    if s != 0:
        c = float(np.dot(a, b))
        k = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        return np.eye(3) + k + k.dot(k) * ((1 - c) / (s ** 2))
    if norm_of(a + b) == 0:
        return rot_mat_from_pointer(np.array([0.0, 0.0, 1.0]), 180)
    return np.eye(3)
