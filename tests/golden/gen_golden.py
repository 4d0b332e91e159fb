"""Golden-vector generator: runs ONLY in the build container, never on the GPU box (.gpurunignore lists it).

It imports the reference's own Python from /root/reference (tscode.algebra, tscode.rmsd_pruning,
tscode.numba_functions for G1-G6 and G8; tscode.utils, tscode.torsion_module and tscode.optimization_methods for
G7 and G9) and records their outputs on seeded inputs as small .npz fixtures next to this file.  The reference is
executed as plain NumPy: Numba is not loadable in this image, so name-only stand-ins for it and for
the off-path packages the reference imports at module level are registered in ``sys.modules`` first
(tests/golden/_reference.py; nothing is written next to the reference, no bytecode either).  What the Numba stand-in
does not reproduce (fastmath re-association, Numba's own LAPACK build) only moves last bits; the fixtures are compared
at 1e-9 or looser.  G10-G12 (embed helpers and the embed loops) are made by gen_golden_embeds.py.

Usage:  python -B tests/golden/gen_golden.py [G1 G2 ...]
"""

from __future__ import annotations

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _reference as R                      # noqa: E402

R.install_standins(full=True)
import networkx as nx                       # noqa: E402

if not hasattr(nx, "from_numpy_matrix"):    # networkx 3.x renamed it; the reference was written for 2.x
    nx.from_numpy_matrix = nx.from_numpy_array

import tscode.algebra as ref_alg            # noqa: E402
import tscode.numba_functions as ref_nf     # noqa: E402
import tscode.rmsd_pruning as ref_rp        # noqa: E402
import tscode.optimization_methods as ref_om   # noqa: E402
import tscode.torsion_module as ref_tm      # noqa: E402
import tscode.utils as ref_utils            # noqa: E402
from numba.typed import List                # noqa: E402

from tscode_amd.synthetic import make_ensemble, make_fragment, quat_to_mat   # noqa: E402


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------- G1
def _g1_pairs(rng, h):
    """Pairs (p, q) of (h,3) heavy-atom sets, including the special cases SURVEY 7.3/8c lists."""
    out = []
    base = make_fragment(rng, h) + rng.normal(size=3) * 3.0       # not centred, like a docked pose

    def rot(x, deg=None):
        r = quat_to_mat(rng.normal(size=4))
        if deg is not None:
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            a = np.deg2rad(deg) / 2
            r = quat_to_mat(np.array([np.cos(a), *(np.sin(a) * ax)]))
        return x @ r.T

    out.append(("identical", base, base.copy()))
    out.append(("translated", base, base + np.array([0.3, -0.2, 0.1])))
    out.append(("mirrored", base, base * np.array([1.0, 1.0, -1.0])))
    out.append(("rot_small", base, rot(base, 2.0)))
    out.append(("rot_big", base, rot(base)))
    planar = base.copy()
    planar[:, 2] = 0.0
    out.append(("planar_through_origin", planar, rot(planar, 5.0) + 0.0))
    planar_off = planar + np.array([0.0, 0.0, 2.5])
    out.append(("planar_offset", planar_off, planar_off + rng.normal(size=planar.shape) * 0.05))
    coll = np.outer(np.linspace(1.0, 1.0 + 1.4 * (h - 1), h), np.array([0.6, 0.0, 0.8]))
    out.append(("collinear", coll, coll + rng.normal(size=coll.shape) * 0.02))
    for s in (0.01, 0.05, 0.1, 0.2, 0.3, 0.5, 1.0):
        out.append((f"noise_{s}", base, rot(base, 3.0 * s) + rng.normal(size=base.shape) * s))
    while len(out) < 40:
        other = make_fragment(rng, h) + rng.normal(size=3) * 3.0
        out.append(("random", base if len(out) % 2 else rot(base), other))
    return out


def gen_g1():
    print("G1 rmsd_and_max_numba")
    rng = np.random.default_rng(9101)
    data = {}
    for h in (3, 5, 9, 18, 30, 60, 120):
        pairs = _g1_pairs(rng, h)
        p = np.array([a for _, a, _ in pairs])
        q = np.array([b for _, _, b in pairs])
        res = np.array([ref_rp.rmsd_and_max_numba(a.copy(), b.copy()) for a, b in zip(p, q)])
        data[f"p_{h}"], data[f"q_{h}"], data[f"out_{h}"] = p, q, res
        data[f"tags_{h}"] = np.array([t for t, _, _ in pairs])
    _save("G1_rmsd_and_max", seed=9101, **data)


# --------------------------------------------------------------------------- G2
def gen_g2():
    print("G2 all_dists / compenetration_check / count_clashes")
    rng = np.random.default_rng(9102)
    data = {}
    # all_dists on block-remainder sizes
    k = 0
    for na, nb_ in ((31, 32), (32, 33), (33, 65), (65, 31), (25, 25), (7, 3), (1, 40)):
        a = rng.normal(size=(na, 3)) * 3
        b = rng.normal(size=(nb_, 3)) * 3
        data[f"ad_a{k}"], data[f"ad_b{k}"], data[f"ad_out{k}"] = a, b, ref_alg.all_dists(a, b)
        k += 1
    data["ad_n"] = k
    # compenetration_check: bi- and tri-molecular poses, ids variants
    cases = []
    for ids in ((15, 15), (25, 25), (31, 33), (5, 8), (10, 12, 9), (33, 20, 32), (70, 70, 60)):
        ens = make_ensemble(60, ids, seed=int(rng.integers(1 << 30)), shell=(2.5, 7.0))
        cases.append((np.asarray(ids), ens.poses()))
    combos = [(t, mc) for t in (1.4, 1.5) for mc in (0, 1, 3)]
    data["cc_combos"] = np.array(combos)
    k = 0
    for ids, poses in cases:
        off = np.concatenate([[0], np.cumsum(ids)])
        frs = [poses[:, off[i]:off[i + 1]] for i in range(len(ids))]
        order = [(1, 0)] if len(ids) == 2 else [(1, 0), (2, 1), (0, 2)]
        data[f"cc_ids{k}"], data[f"cc_coords{k}"] = ids, poses
        # verdict per (thresh, max_clashes) combo, and the raw counts per fragment pair in the
        # reference's order (m2,m1),(m3,m2),(m1,m3) for the two thresholds
        data[f"cc_out{k}"] = np.array([[ref_nf.compenetration_check(c, ids=ids, thresh=t, max_clashes=int(mc))
                                        for c in poses] for t, mc in combos])
        data[f"cc_counts{k}"] = np.array([[[np.count_nonzero(ref_alg.all_dists(frs[a][s], frs[b][s]) < t)
                                            for a, b in order] for s in range(len(poses))] for t in (1.4, 1.5)])
        k += 1
    data["cc_n"] = k
    # ids=None -> count_clashes (threshold 0.5, ordered pairs, d>0)
    k = 0
    for n in (8, 30, 50):
        for _ in range(4):
            c = rng.normal(size=(n, 3)) * (0.6 if _ % 2 else 2.0)
            c[n // 2] = c[0]                       # an exact duplicate: d == 0 is not a clash
            data[f"cl_coords{k}"] = c
            data[f"cl_count{k}"] = ref_nf.count_clashes(c)
            data[f"cl_check{k}"] = np.array([ref_nf.compenetration_check(c, max_clashes=mc) for mc in (0, 2, 10)])
            k += 1
    data["cl_n"] = k
    _save("G2_clash", seed=9102, **data)


# --------------------------------------------------------------------------- G3
def _prune_traced(structures, atomnos, thr):
    """prune_conformers_rmsd (rmsd_pruning.py:164-206) re-run pass by pass through the reference's own
    _similarity_mask_rmsd_group so that the per-pass masks and cache keys can be recorded."""
    heavy = np.array([s[atomnos != 1] for s in structures])
    mask = np.ones(len(structures), dtype=np.bool_)
    cache = List([(-1, -1)])
    ks, masks, nkeys = [], [], []
    for k in (5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1):
        if k == 1 or 20 * k < np.count_nonzero(mask):
            mask, pairs = ref_rp._similarity_mask_rmsd_group(heavy, mask, cache=cache, k=k, rmsd_thr=thr)
            cache.extend(pairs)
            ks.append(int(k))
            masks.append(mask.copy())
            nkeys.append(len(cache) - 1)
    keys = np.array(sorted(set(cache[1:])), dtype=np.int64).reshape(-1, 2)
    return mask, np.array(ks), np.array(masks), np.array(nkeys), keys


def gen_g3():
    print("G3 prune_conformers_rmsd")
    data = {}
    cases = [
        # (N, atoms per fragment, children, sigma_t, thr, seed)
        (40, (5, 5), 4, 0.03, 0.5, 9301),
        (600, (8, 7), 20, 0.03, 0.5, 9302),
        (600, (8, 7), 3, 0.10, 0.25, 9303),
        (720, (15, 15), 12, 0.05, 0.5, 9304),
        (1003, (15, 15), 10, 0.03, 0.5, 9305),      # N not divisible by any k: remainder chunk
        (2500, (15, 15), 10, 0.03, 0.5, 9306),
    ]
    for c, (n, apf, ch, st, thr, seed) in enumerate(cases):
        t0 = time.time()
        ens = make_ensemble(n, apf, seed=seed, children=ch, sigma_t=st)
        structures = ens.poses()
        pruned, mask = ref_rp.prune_conformers_rmsd(structures, ens.atomnos, rmsd_thr=thr)
        mask2, ks, masks, nkeys, keys = _prune_traced(structures, ens.atomnos, thr)
        assert np.array_equal(mask, mask2)
        assert np.array_equal(pruned, structures[mask])
        data[f"structures{c}"], data[f"atomnos{c}"], data[f"thr{c}"] = structures, ens.atomnos, thr
        data[f"mask{c}"], data[f"ks{c}"], data[f"pass_masks{c}"] = mask, ks, masks
        data[f"pass_nkeys{c}"], data[f"keys{c}"] = nkeys, keys
        print(f"  case {c}: N={n} h={ens.n_heavy} thr={thr} survivors={mask.sum()} passes={ks.tolist()} "
              f"keys={len(keys)} ({time.time() - t0:.1f}s)")
    data["n_cases"] = len(cases)
    _save("G3_prune", **data)


# --------------------------------------------------------------------------- G4
def gen_g4():
    print("G4 _rmsd_similarity greedy group filter")
    data = {}
    for c, (seed, n, apf) in enumerate(((9401, 216, (10, 12, 9)), (9402, 36, (15, 15)), (9403, 216, (6, 5, 5)))):
        ens = make_ensemble(n, apf, seed=seed, children=6, sigma_t=0.3, sigma_rot_deg=8.0)
        poses = ens.poses()
        kept, flags = [], []
        for pose in poses:                              # embeds.py:715 greedy use, thr=1
            new = not ref_rp._rmsd_similarity(pose, kept, rmsd_thr=1)
            flags.append(new)
            if new:
                kept.append(pose)
        data[f"poses{c}"], data[f"accepted{c}"] = poses, np.array(flags)
        print(f"  case {c}: {n} poses -> {int(np.sum(flags))} accepted")
    data["n_cases"] = 3
    _save("G4_rmsd_similarity", **data)


# --------------------------------------------------------------------------- G5
def gen_g5():
    print("G5 rotation helpers")
    rng = np.random.default_rng(9105)
    ptr = rng.normal(size=(40, 3))
    ang = rng.uniform(-360, 360, size=40)
    ang[:4] = (0.0, 180.0, 90.0, 360.0)
    rmp = np.array([ref_alg.rot_mat_from_pointer(p.copy(), a) for p, a in zip(ptr, ang)])
    quat = rng.normal(size=(20, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    q2m = np.array([ref_alg.quaternion_to_rotation_matrix(q) for q in quat])
    ref_v = rng.normal(size=(30, 2, 3))
    tgt_v = rng.normal(size=(30, 2, 3))
    tgt_v[0] = ref_v[0]                                 # identity
    tgt_v[1] = -ref_v[1]                                # inversion of both vectors
    avp = np.array([ref_alg.align_vec_pair(r, t) for r, t in zip(ref_v, tgt_v)])
    v1 = rng.normal(size=(30, 3))
    v2 = rng.normal(size=(30, 3))
    v2[0], v2[1] = v1[0] * 2.0, -v1[1]
    va = np.array([ref_alg.vec_angle(a, b) for a, b in zip(v1, v2)])
    coords = rng.normal(size=(17, 3))
    tc = ref_alg.transform_coords(coords, rmp[5], np.array([1.0, -2.0, 0.5]))
    _save("G5_rotations", seed=9105, ptr=ptr, ang=ang, rot_mat_from_pointer=rmp, quat=quat, quat_to_mat=q2m,
          avp_ref=ref_v, avp_tgt=tgt_v, align_vec_pair=avp, va_v1=v1, va_v2=v2, vec_angle=va,
          tc_coords=coords, tc_rot=rmp[5], tc_pos=np.array([1.0, -2.0, 0.5]), transform_coords=tc)


# --------------------------------------------------------------------------- G6
def gen_g6():
    print("G6 torsion fingerprints (next-row N2)")
    rng = np.random.default_rng(9106)
    coords = rng.normal(size=(50, 12, 3)) * 2
    quads = np.array([[0, 1, 2, 3], [1, 2, 3, 4], [4, 5, 6, 7], [8, 9, 10, 11], [0, 5, 9, 11]])
    fp = np.array([ref_nf.get_torsion_fingerprint(c, quads) for c in coords])
    sim = np.array([[ref_nf.tfd_similarity(fp[i], fp[j], thresh=t) for j in range(10)] for i in range(10) for t in (10, 90, 300)])
    _save("G6_tfd", seed=9106, coords=coords, quadruplets=quads, fingerprints=fp, similarity=sim)


# --------------------------------------------------------------------------- G7
def _chain_molecule(rng, n_backbone, branches):
    """A branched chain: backbone atoms 0..n_backbone-1 bonded in sequence (1.5 A steps, self-avoiding), plus `branches`
    extra atoms each bonded to a random backbone atom.  Returns coords and the bond list."""
    coords = [np.zeros(3)]
    bonds = []
    while len(coords) < n_backbone:
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        cand = coords[-1] + 1.5 * d
        if all(np.linalg.norm(cand - c) > 1.3 for c in coords[:-1]):
            bonds.append((len(coords) - 1, len(coords)))
            coords.append(cand)
    for _ in range(branches):
        while True:
            root = int(rng.integers(1, n_backbone - 1))
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            cand = coords[root] + 1.1 * d
            if all(np.linalg.norm(cand - c) > 1.0 for i, c in enumerate(coords) if i != root):
                bonds.append((root, len(coords)))
                coords.append(cand)
                break
    return np.array(coords), bonds


def gen_g7():
    """csearch rotations (next-row N3): every number comes from the reference's own random_csearch
    (tscode/torsion_module.py:399-521), which calls its _get_rotation_mask, rotate_dihedral and torsion_comp_check.
    Part A runs it once per candidate (duck-typed torsions whose get_angles() offer that candidate's angle only), with
    note-taking wrappers around rotate_dihedral / torsion_comp_check, so that every candidate's final coordinates, every
    verdict and the number of bonds that really rotated are recorded -- also for the candidates the function drops.  Part B
    runs it as its callers do: the n-fold angle tables, np.random.shuffle (seeded) and the n_out / max_tries selection."""
    print("G7 csearch dihedral rotations (next-row N3): the reference's random_csearch")
    rng = np.random.default_rng(9107)

    class OneAngle(ref_tm.Torsion):
        def __init__(self, torsion, angle):
            super().__init__(*[int(i) for i in torsion])
            self.n_fold, self._angle = 1, angle

        def get_angles(self):
            return (self._angle,)

    events = []
    real_rd, real_cc = ref_tm.rotate_dihedral, ref_tm.torsion_comp_check

    def rd(coords, dihedral, angle, mask=None, indices_to_be_moved=None):
        out = real_rd(coords, dihedral, angle, mask=mask, indices_to_be_moved=indices_to_be_moved)
        events.append(("rot", angle, out))
        return out

    def cc(coords, torsion, mask, thresh=1.5, max_clashes=0):
        ok = real_cc(coords, torsion=torsion, mask=mask, thresh=thresh, max_clashes=max_clashes)
        events.append(("chk", bool(ok), None))
        return ok

    quiet = dict(logfunction=lambda *a, **k: None, interactive_print=False)
    cases = []
    for case, (nb, br, n_tors, n_cand) in enumerate(((10, 4, 3, 60), (24, 12, 5, 80), (40, 30, 6, 60))):
        coords, bonds = _chain_molecule(rng, nb, br)
        n = len(coords)
        graph = nx.Graph()
        graph.add_nodes_from(range(n))                               # node order 0..n-1: _get_rotation_mask walks graph.nodes
        graph.add_edges_from(bonds)
        atomnos = np.full(n, 6)
        centres = rng.choice(np.arange(1, nb - 2), size=n_tors, replace=False)
        torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres], dtype=np.int32)
        masks = np.array([ref_tm._get_rotation_mask(graph, tuple(int(i) for i in t)) for t in torsions])
        angles = rng.choice(np.array([0, 60, 120, 180, 240, 300, -60, -120, 30, 7]), size=(n_cand, n_tors)).astype(np.int32)
        out = np.empty((n_cand, n, 3))
        rb = np.zeros(n_cand, dtype=np.int32)
        kept_by_ref = np.zeros(n_cand, dtype=bool)
        first_checks = []
        ref_tm.rotate_dihedral, ref_tm.torsion_comp_check = rd, cc
        try:
            for a, angle_set in enumerate(angles):
                events.clear()
                tors = [OneAngle(t, int(x)) for t, x in zip(torsions, angle_set)]
                res = ref_tm.random_csearch(coords.copy(), atomnos, tors, graph, n_out=1, **quiet)
                rots = [e for e in events if e[0] == "rot"]
                out[a] = rots[-1][2] if rots else coords
                rb[a] = sum(1 for e in events if e[0] == "chk" and e[1])
                kept_by_ref[a] = len(res) == 1
                if len(res):
                    assert np.array_equal(res[0], out[a])
                for i, e in enumerate(events):                      # the check right after a forward rotation
                    if e[0] == "rot" and e[1] != -5:
                        first_checks.append(int(events[i + 1][1]))
        finally:
            ref_tm.rotate_dihedral, ref_tm.torsion_comp_check = real_rd, real_cc
        assert np.array_equal(kept_by_ref, rb != 0)
        cases.append(dict(coords=coords, torsions=torsions, masks=masks, angles=angles, out=out, rotated_bonds=rb,
                          first_checks=np.array(first_checks, dtype=np.int8), bonds=np.array(bonds)))
        print(f"  case {case}: n = {n}, {n_tors} torsions, {n_cand} candidates, rotated_bonds histogram {np.bincount(rb)}, "
              f"{np.mean(first_checks):.2f} of the first checks pass")
    flat = {"n_cases": len(cases), "seed": 9107}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat[f"{k}{i}"] = v

    # Part B: the whole function, as torsion_module.csearch calls it
    coords, bonds = _chain_molecule(rng, 16, 8)
    n = len(coords)
    graph = nx.Graph()
    graph.add_nodes_from(range(n))
    graph.add_edges_from(bonds)
    tors = []
    for c, nf in zip((2, 6, 9, 12), (3, 2, 6, 3)):
        t = ref_tm.Torsion(c - 1, c, c + 1, c + 2)
        t.n_fold = nf
        tors.append(t)
    table = ref_utils.cartesian_product(*[t.get_angles() for t in tors])
    for b, (seed, n_out, max_tries, rotations) in enumerate(((41, 25, 10000, None), (42, 1000, 40, None), (43, 30, 10000, 2))):
        tab = table.copy()
        if rotations is not None:
            tab = tab[np.count_nonzero(tab, axis=1) == rotations]
        np.random.seed(seed)
        np.random.shuffle(tab)                                      # the table random_csearch will walk (same seed, same call)
        np.random.seed(seed)
        res = ref_tm.random_csearch(coords.copy(), np.full(n, 6), tors, graph, n_out=n_out, max_tries=max_tries, rotations=rotations, **quiet)
        flat[f"b_angles{b}"], flat[f"b_out{b}"], flat[f"b_n_out{b}"], flat[f"b_max_tries{b}"] = tab, res, n_out, max_tries
        print(f"  part B run {b}: {len(tab)} angle sets, n_out {n_out}, max_tries {max_tries}: {len(res)} structures")
    flat["b_coords"], flat["b_torsions"] = coords, np.array([t.torsion for t in tors], dtype=np.int32)
    flat["b_masks"] = np.array([ref_tm._get_rotation_mask(graph, t.torsion) for t in tors])
    flat["b_n"] = 3
    _save("G7_csearch", **flat)


# --------------------------------------------------------------------------- G8
def gen_g8():
    """prune_conformers_tfd (next-row N2) on clustered ensembles: the reference's own function, networkx 3.x and all.
    Which member of a cluster survives is `tuple(graph.nodes)[0]` of a networkx subgraph view (numba_functions.py:221-226),
    i.e. it depends on CPython's set iteration order; the fixture records what THIS interpreter + networkx give, and the
    product reproduces it by building the same Python set / graph objects on the host from the GPU's matches."""
    import networkx
    print(f"G8 prune_conformers_tfd (next-row N2), networkx {networkx.__version__}, python {sys.version.split()[0]}")
    rng = np.random.default_rng(9108)
    quads = np.array([[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 5], [4, 5, 6, 7], [6, 7, 8, 9], [0, 4, 8, 11]])
    flat = {"n_cases": 0, "seed": 9108, "networkx": networkx.__version__, "python": sys.version.split()[0]}
    for case, (n_par, n_child, noise, thresh) in enumerate(((12, 5, 0.01, 10), (140, 5, 0.02, 10), (400, 6, 0.03, 20), (30, 4, 0.02, 10))):
        parents = rng.normal(size=(n_par, 12, 3)) * 2
        structures = (parents[:, None] + rng.normal(size=(n_par, n_child, 12, 3)) * noise).reshape(-1, 12, 3)
        structures = np.ascontiguousarray(structures[rng.permutation(len(structures))])
        if case == 3:                                         # exact duplicates and a wrap-around pair
            structures[7] = structures[3]
            structures[11] = structures[3]
        t0 = time.time()
        tf_mat = ref_nf._get_tf_mat(structures, quads)
        pruned, mask = ref_nf.prune_conformers_tfd(structures, quads, thresh=thresh)
        assert np.array_equal(pruned, structures[mask])
        flat[f"structures{case}"], flat[f"tf_mat{case}"], flat[f"mask{case}"], flat[f"thresh{case}"] = structures, tf_mat, mask, thresh
        flat["n_cases"] = case + 1
        print(f"  case {case}: N = {len(structures)}, thresh {thresh}: {mask.sum()} survive ({time.time() - t0:.1f} s)")
    flat["quadruplets"] = quads
    _save("G8_tfd_prune", **flat)


# --------------------------------------------------------------------------- G9
def gen_g9():
    """Moment-of-inertia pruning and embed scores (next-row N4): the reference's get_inertia_moments /
    get_moi_similarity_matches (tscode.algebra), prune_by_moment_of_inertia and fitness_check
    (tscode.optimization_methods:327-358, :544-557) and _score_embed_poses (tscode.numba_functions).  The masses
    prune_by_moment_of_inertia reads from tscode.pt are the element data of tests/golden/_reference.py."""
    import networkx
    print(f"G9 moments of inertia / embed scores (next-row N4), networkx {networkx.__version__}")
    rng = np.random.default_rng(9109)
    flat = {"n_cases": 0, "seed": 9109, "networkx": networkx.__version__, "python": sys.version.split()[0]}
    for case, (n_par, n_child, n_atoms) in enumerate(((10, 4, 9), (60, 5, 14))):
        atomnos = rng.choice(np.array([6, 7, 8, 16, 17]), size=n_atoms)
        masses = np.array([R.ELEMENTS[int(z)][2] for z in atomnos])
        parents = rng.normal(size=(n_par, n_atoms, 3)) * 2
        structures = (parents[:, None] + rng.normal(size=(n_par, n_child, n_atoms, 3)) * 0.004).reshape(-1, n_atoms, 3)
        # rigid motions and mirror images keep the moments: rotamer / enantiomer duplicates
        for s in range(0, len(structures), 3):
            q = rng.normal(size=4)
            Rm = ref_alg.quaternion_to_rotation_matrix(q / np.linalg.norm(q))
            structures[s] = structures[s] @ Rm.T + rng.normal(size=3)
        structures[1] = structures[0] * np.array([1.0, 1.0, -1.0])
        structures = np.ascontiguousarray(structures[rng.permutation(len(structures))])
        moments = np.array([ref_alg.get_inertia_moments(s.copy(), masses) for s in structures])
        matches = ref_alg.get_moi_similarity_matches(structures.copy(), masses, max_deviation=1e-2)
        pruned, mask = ref_om.prune_by_moment_of_inertia(structures.copy(), atomnos, max_deviation=1e-2)
        assert np.array_equal(pruned, structures[mask])
        flat[f"structures{case}"], flat[f"masses{case}"], flat[f"moments{case}"] = structures, masses, moments
        flat[f"atomnos{case}"] = atomnos
        flat[f"matches{case}"], flat[f"mask{case}"] = np.array(matches, dtype=np.int64).reshape(-1, 2), mask
        flat["n_cases"] = case + 1
        print(f"  case {case}: N = {len(structures)}, {len(matches)} matches, {mask.sum()} survive")
    # a case with hydrogens: prune_by_moment_of_inertia drops them before taking the moments (:335-336)
    atomnos = np.array([6, 1, 1, 8, 6, 1, 7, 1, 17, 6, 1, 1])
    parents = rng.normal(size=(25, 12, 3)) * 2
    structures = (parents[:, None] + rng.normal(size=(25, 4, 12, 3)) * 0.003).reshape(-1, 12, 3)
    structures = np.ascontiguousarray(structures[rng.permutation(len(structures))])
    pruned, mask = ref_om.prune_by_moment_of_inertia(structures.copy(), atomnos, max_deviation=1e-2)
    flat["h_structures"], flat["h_atomnos"], flat["h_mask"] = structures, atomnos, mask
    flat["h_masses"] = np.array([R.ELEMENTS[int(z)][2] for z in atomnos])
    print(f"  with hydrogens: N = {len(structures)}, {mask.sum()} survive")
    # embed scores and fitness
    structures = rng.normal(size=(40, 12, 3)) * 2
    ci = rng.integers(0, 12, size=(40, 3, 2))
    ci[:, :, 1] = (ci[:, :, 0] + 1 + rng.integers(0, 10, size=(40, 3))) % 12
    cd = rng.uniform(1.0, 4.0, size=(40, 3))
    scores = ref_nf._score_embed_poses(structures, ci, cd)
    thr = 0.5
    fit = np.array([ref_om.fitness_check(structures[s], ci[s], cd[s], thr) for s in range(40)])
    targets_none = [[None if (s + c) % 4 == 0 else cd[s, c] for c in range(3)] for s in range(40)]
    fit_none = np.array([ref_om.fitness_check(structures[s], ci[s], targets_none[s], thr) for s in range(40)])
    flat.update(sc_structures=structures, sc_indices=ci, sc_distances=cd, scores=scores, fitness_threshold=thr, fitness_ok=fit,
                fitness_none=np.array([[np.nan if t is None else t for t in row] for row in targets_none]), fitness_ok_none=fit_none)
    _save("G9_moi_scores", **flat)


if __name__ == "__main__":
    which = sys.argv[1:] or ["G1", "G2", "G3", "G4", "G5", "G6", "G7", "G8", "G9"]
    for g in which:
        globals()["gen_" + g.lower()]()
