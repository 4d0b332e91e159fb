"""Deterministic synthetic conformer ensembles.

Every benchmark and every large parity test draws its input from here, so
that the GPU path, the CPU oracle and the committed golden vectors all see
the same numbers for a given (config, seed).

Layout of one ensemble
----------------------
* ``n_mols`` rigid fragments (2, or 3 for the trimolecular config); atom
  positions are a self-avoiding random walk (1.5 A steps, >= 1.2 A between
  non-bonded atoms), centred at the origin;
* ``atomnos``: 3 heavy atoms (Z=6) then 2 hydrogens, repeating -> 60 % heavy;
* fragment 0 is fixed (R = I, t = 0); the other fragments take ``N / c``
  "parent" transforms (R uniform on SO(3), t in a 4-9 A shell) and ``c``
  slightly perturbed children per parent, shuffled over the whole array.

Nothing in here touches the GPU or the reference.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

__all__ = ["Ensemble", "make_fragment", "make_ensemble", "make_unscreenable", "CONFIGS", "make_config", "quat_to_mat"]


def make_fragment(rng: np.random.Generator, n_atoms: int, step: float = 1.5, min_dist: float = 1.2) -> np.ndarray:
    """Self-avoiding random walk, centred at the origin. Returns f64[n_atoms, 3]."""
    pts = np.zeros((n_atoms, 3))
    i = 1
    tries = 0
    while i < n_atoms:
        v = rng.normal(size=3)
        v *= step / np.sqrt(v @ v)
        # grow from a random earlier atom now and then, so fragments are branched, not chains
        base = i - 1 if rng.random() < 0.7 else int(rng.integers(0, i))
        cand = pts[base] + v
        d = np.sqrt(((pts[:i] - cand) ** 2).sum(axis=1))
        d[base] = np.inf
        tries += 1
        if d.min() >= min_dist:
            pts[i] = cand
            i += 1
        if tries > 100000:
            raise RuntimeError("random walk stuck")
    return pts - pts.mean(axis=0)


def quat_to_mat(q: np.ndarray) -> np.ndarray:
    """Unit quaternions (w, x, y, z) f64[..., 4] -> rotation matrices f64[..., 3, 3]."""
    q = q / np.sqrt((q * q).sum(axis=-1, keepdims=True))
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = np.empty(q.shape[:-1] + (3, 3))
    m[..., 0, 0] = 1 - 2 * (y * y + z * z)
    m[..., 0, 1] = 2 * (x * y - w * z)
    m[..., 0, 2] = 2 * (x * z + w * y)
    m[..., 1, 0] = 2 * (x * y + w * z)
    m[..., 1, 1] = 1 - 2 * (x * x + z * z)
    m[..., 1, 2] = 2 * (y * z - w * x)
    m[..., 2, 0] = 2 * (x * z - w * y)
    m[..., 2, 1] = 2 * (y * z + w * x)
    m[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return m


def _axis_angle_to_mat(axis: np.ndarray, angle: np.ndarray) -> np.ndarray:
    axis = axis / np.sqrt((axis * axis).sum(axis=-1, keepdims=True))
    half = 0.5 * angle
    q = np.concatenate([np.cos(half)[..., None], np.sin(half)[..., None] * axis], axis=-1)
    return quat_to_mat(q)


@dataclass
class Ensemble:
    """One synthetic ensemble in the batched form of SURVEY.md section 8 row a16.

    frag_coords[m]: f64[1, n_m, 3] conformer stack of fragment m (one conformer each)
    conf_idx:       i32[N, n_mols]  conformer picked for each pose and fragment (all 0 here)
    rot:            f64[N, n_mols, 3, 3]
    pos:            f64[N, n_mols, 3]
    ids:            i64[n_mols]      atoms per fragment
    atomnos:        i64[n]           atomic numbers of the concatenated pose
    """

    frag_coords: list
    conf_idx: np.ndarray
    rot: np.ndarray
    pos: np.ndarray
    ids: np.ndarray
    atomnos: np.ndarray
    seed: int
    meta: dict = field(default_factory=dict)

    @property
    def n_poses(self) -> int:
        return int(self.rot.shape[0])

    @property
    def n_atoms(self) -> int:
        return int(self.ids.sum())

    @property
    def n_heavy(self) -> int:
        return int((self.atomnos != 1).sum())

    def poses(self, lo: int = 0, hi: int | None = None) -> np.ndarray:
        """NumPy materialisation of poses [lo, hi): the get_embed formula
        ``(R @ X.T).T + t`` per fragment, concatenated (reference embeds.py:961-969)."""
        hi = self.n_poses if hi is None else hi
        parts = []
        for m, frag in enumerate(self.frag_coords):
            x = frag[self.conf_idx[lo:hi, m]]                    # (P, n_m, 3)
            r = self.rot[lo:hi, m]                               # (P, 3, 3)
            parts.append(np.einsum("pij,paj->pai", r, x) + self.pos[lo:hi, m][:, None, :])
        return np.ascontiguousarray(np.concatenate(parts, axis=1))


def make_ensemble(n_poses: int, atoms_per_frag, seed: int, children: int = 10,
                  sigma_rot_deg: float = 1.0, sigma_t: float = 0.03,
                  shell=(4.0, 9.0), local_spread: float | None = None) -> Ensemble:
    """``local_spread`` = None: the children of all parents are shuffled over the whole array (every BASELINE config).
    A number S > 1: a LOCAL shuffle -- every family gets a home position, uniform over the array, and each child lands at
    home +- a log-uniform offset in [1, S]: a few siblings share a chunk of 20, more share a chunk of 200, the rest only meet in
    the coarse passes.  The prune's fine passes (k = 5000 ... 200, rmsd_pruning.py:186-192) then each find duplicates to remove
    while more than 20 k structures stay active, which a global shuffle never gives them (golden fixtures G16 / G17)."""
    rng = np.random.default_rng(seed)
    atoms_per_frag = [int(a) for a in atoms_per_frag]
    n_mols = len(atoms_per_frag)
    frags = [make_fragment(rng, a)[None, :, :] for a in atoms_per_frag]
    atomnos = np.concatenate([np.where(np.arange(a) % 5 < 3, 6, 1) for a in atoms_per_frag]).astype(np.int64)

    n_par = max(1, (n_poses + children - 1) // children)
    rot = np.zeros((n_poses, n_mols, 3, 3))
    pos = np.zeros((n_poses, n_mols, 3))
    rot[:, 0] = np.eye(3)
    parent_of = np.repeat(np.arange(n_par), children)[:n_poses]
    for m in range(1, n_mols):
        pr = quat_to_mat(rng.normal(size=(n_par, 4)))
        direction = rng.normal(size=(n_par, 3))
        direction /= np.sqrt((direction ** 2).sum(axis=1, keepdims=True))
        radius = rng.uniform(shell[0], shell[1], size=(n_par, 1))
        pt = direction * radius
        # children = parent o (small body-frame rotation) + small translation
        d_rot = _axis_angle_to_mat(rng.normal(size=(n_poses, 3)),
                                   np.deg2rad(sigma_rot_deg) * rng.normal(size=n_poses))
        d_t = sigma_t * rng.normal(size=(n_poses, 3))
        rot[:, m] = np.einsum("pij,pjk->pik", pr[parent_of], d_rot)
        pos[:, m] = pt[parent_of] + d_t
    if local_spread is None:
        perm = rng.permutation(n_poses)
    else:
        home = rng.uniform(0.0, n_poses, size=n_par)
        off = np.exp(rng.uniform(0.0, np.log(float(local_spread)), size=n_poses)) * rng.choice([-1.0, 1.0], size=n_poses)
        perm = np.argsort(home[parent_of] + off, kind="stable")
    rot, pos, parent_of = rot[perm], pos[perm], parent_of[perm]
    return Ensemble(
        frag_coords=frags,
        conf_idx=np.zeros((n_poses, n_mols), dtype=np.int32),
        rot=np.ascontiguousarray(rot), pos=np.ascontiguousarray(pos),
        ids=np.asarray(atoms_per_frag, dtype=np.int64), atomnos=atomnos, seed=seed,
        meta={"children": children, "sigma_rot_deg": sigma_rot_deg, "sigma_t": sigma_t,
              "shell": tuple(shell), "parent_of": parent_of, "local_spread": local_spread},
    )


def make_unscreenable(n: int, seed: int = 99, h: int = 30, children: int = 10, jitter: float = 0.004) -> np.ndarray:
    """Heavy-atom array f64[n, h, 3] on which the descriptor sieve of the prune can drop NOTHING: two rigid bodies about the
    origin, the first fixed, the second turned by one of n / children random rotations (+ a tiny jitter) ABOUT THE ORIGIN.  Every
    atom keeps its distance from the origin and every atom pair (a, a + h / 2) lies inside one body and keeps its length, so the
    rotation-invariant descriptors of all n structures coincide, while structures of different parents are far apart in RMSD.
    Every pair a pass looks at therefore reaches H = p^T q (tools/worstcase.py, bench.py's `unscreenable` leg)."""
    rng = np.random.default_rng(seed)
    half = h // 2
    na = half // 2
    ia = list(range(0, na)) + list(range(half, half + na))                   # body A: atoms a and a + h / 2 for a < na
    ib = [a for a in range(h) if a not in set(ia)]                           # body B: the other pairs (and an unpaired last atom)
    A, B = rng.normal(size=(len(ia), 3)) * 3, rng.normal(size=(len(ib), 3)) * 3
    n_par = max(1, n // children)
    rots = quat_to_mat(rng.normal(size=(n_par, 4)))
    which = rng.integers(0, n_par, size=n)
    jit = quat_to_mat(np.concatenate([np.ones((n, 1)), rng.normal(size=(n, 3)) * jitter], axis=1))
    heavy = np.empty((n, h, 3))
    heavy[:, ia] = A
    heavy[:, ib] = np.einsum("nij,njk,ak->nai", rots[which], jit, B)
    return np.ascontiguousarray(heavy)


# BASELINE.json configs -> (N, atoms per fragment, seed); thresholds are fixed for all of them.
CONFIGS = {
    "C2": dict(n_poses=10_000, atoms_per_frag=(15, 15), seed=1002),
    "C3": dict(n_poses=100_000, atoms_per_frag=(25, 25), seed=1003),
    "C4": dict(n_poses=1_000_000, atoms_per_frag=(25, 25), seed=1004),
    # 70-atom fragments are ~6 A in radius: with the 4-9 A shell of the bimolecular configs only 1.5 % of the
    # poses pass the clash check, so the trimolecular config places its fragments in an 8-15 A shell
    "C5": dict(n_poses=500_000, atoms_per_frag=(70, 70, 60), seed=1005, shell=(8.0, 15.0)),
}
RMSD_THR = 0.5
CLASH_THRESH = 1.5
MAX_CLASHES = 0


def make_config(name: str, n_poses: int | None = None, attempt: int = 0) -> Ensemble:
    """Ensemble for a named BASELINE config. ``attempt`` re-draws with seed + 1000*attempt
    (the guard-band rule of SURVEY.md 8d); ``n_poses`` overrides N for bounded CPU samples."""
    cfg = dict(CONFIGS[name])
    if n_poses is not None:
        cfg["n_poses"] = int(n_poses)
    cfg["seed"] = cfg["seed"] + 1000 * attempt
    ens = make_ensemble(**cfg)
    ens.meta["config"] = name
    return ens
