"""Golden vectors G10-G12: the embed loops and their helpers, recorded from the REFERENCE'S OWN functions.

BUILD CONTAINER ONLY (listed in .gpurunignore).  Imports tscode.utils / embeds / embedder / hypermolecule_class from
/root/reference through the name-only stand-ins of tests/golden/_reference.py and records:

  G10  rotation_matrix_from_vectors (generic, parallel, antiparallel), polygonize (2 and 3 lengths, TriangleError),
       cartesian_product (the loop order of every embed), get_embed on duck-typed molecules, rotate_dihedral
  G11  string_embed (tscode/embeds.py:36-133) run by the reference on its own test molecules CH3Cl.xyz + HCOOH.xyz
       (config C1 of BASELINE.json, DIST A=2.5 as tests/string.txt asks) and on a multi-conformer variant: every
       candidate pose in loop order, every compenetration_check verdict, every torsion fingerprint and the poses the
       reference keeps (is_new_structure's never-evicting list)
  G12  cyclical_embed (tscode/embeds.py:234-860), bimolecular: the rigid shortcut on tests/cyclical.txt's 2 x C2H4, the
       general loop on the same pair, two different molecules, conformer ensembles: candidate poses, group boundaries,
       clash verdicts, greedy _rmsd_similarity verdicts, kept poses and their constrained indices.
       (TRImolecular embeds cannot be recorded: the reference's _get_directions calls vec_angle on 2-vectors
       (embeds.py:297-299) and algebra.norm reads vec[2] (algebra.py:87) -- IndexError as plain NumPy, an unchecked
       out-of-bounds read under Numba, i.e. undefined in the reference itself.)

The molecules (bond graph, reactive-atom orbitals, pivots) are built by the reference's own Hypermolecule /
reactive_atoms_classes / Embedder._set_pivots; the .xyz files are read as data.  networkx 3.x dropped
``from_numpy_matrix`` (the reference was written for 2.x): the generator aliases it to ``from_numpy_array``, its renamed
self.  The tracing wrappers call the reference's function and note what went in and what came out; they change nothing.

Usage:  python -B tests/golden/gen_golden_embeds.py [G10 G11 G12]
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference as R        # noqa: E402

R.install_standins(full=True)
import networkx as nx         # noqa: E402

if not hasattr(nx, "from_numpy_matrix"):
    nx.from_numpy_matrix = nx.from_numpy_array

import tscode.algebra as ref_alg             # noqa: E402
import tscode.embedder as ref_embedder       # noqa: E402
import tscode.embeds as ref_embeds           # noqa: E402
import tscode.hypermolecule_class as ref_hc  # noqa: E402
import tscode.utils as ref_utils             # noqa: E402
from tscode.errors import TriangleError      # noqa: E402

ref_hc.read_xyz = R.ccread_like              # cclib's reader -> the same two attributes read as data
ref_embeds.loadbar = lambda *a, **k: None    # the progress bar of the embed loops: printing only
TESTS = os.path.join(R.REFERENCE, "tscode", "tests")


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------- G10
def gen_g10():
    print("G10 rotation_matrix_from_vectors / polygonize / cartesian_product / get_embed / rotate_dihedral")
    rng = np.random.default_rng(9110)
    v1 = rng.normal(size=(40, 3))
    v2 = rng.normal(size=(40, 3))
    v2[0] = v1[0] * 2.5                                  # parallel -> identity
    v2[1] = -v1[1]                                       # antiparallel -> 180 degrees about z
    v1[2], v2[2] = np.array([0.0, 0.0, 2.0]), np.array([0.0, 0.0, -0.5])
    v1[3], v2[3] = np.array([1.0, 0.0, 0.0]), np.array([-3.0, 0.0, 0.0])
    v1[4], v2[4] = np.array([0.0, 1.0, 0.0]), np.array([0.0, 4.0, 0.0])
    v2[5] = v1[5] + 1e-9 * rng.normal(size=3)            # nearly parallel: the generic branch with a tiny sine
    v2[6] = -v1[6] + 1e-7 * rng.normal(size=3)           # nearly antiparallel
    v1[7], v2[7] = np.array([1.0, 2.0, 3.0]), np.array([-1.0, -2.0, -3.0])
    rmfv = np.array([ref_utils.rotation_matrix_from_vectors(a.copy(), b.copy()) for a, b in zip(v1, v2)])
    data = dict(seed=9110, rmfv_v1=v1, rmfv_v2=v2, rmfv_out=rmfv)

    # polygonize
    l2 = np.array([[1.5, 1.5], [2.0, 3.1], [0.7, 4.4]])
    data["poly2_lengths"], data["poly2_out"] = l2, np.array([ref_utils.polygonize(x) for x in l2])
    l3 = np.array([[3.0, 4.0, 5.0], [2.0, 2.0, 2.0], [2.5, 3.1, 1.2], [1.0, 1.7, 2.5], [4.0, 2.2, 2.1]])
    data["poly3_lengths"], data["poly3_out"] = l3, np.array([ref_utils.polygonize(x) for x in l3])
    bad = np.array([[1.0, 1.0, 2.0], [5.0, 1.0, 1.0], [1.0, 3.0, 1.5]])
    flags = []
    for x in bad:
        try:
            ref_utils.polygonize(x)
            flags.append(False)
        except TriangleError:
            flags.append(True)
    data["poly3_bad"], data["poly3_bad_raises"] = bad, np.array(flags)

    # cartesian_product: the order of every embed loop (np.meshgrid order, not lexicographic)
    for i, sizes in enumerate(((2, 3), (3, 2), (1, 2), (2, 2, 3), (3, 1, 2))):
        data[f"cp_sizes{i}"] = np.array(sizes)
        data[f"cp_out{i}"] = ref_utils.cartesian_product(*[np.array(range(s)) for s in sizes])
    data["cp_n"] = 5

    # get_embed on duck-typed molecules (2 and 3 molecules, conformer stacks)
    k = 0
    for sizes, confs in (((5, 8), (2, 3)), ((10, 12, 9), (1, 2, 2)), ((33,), (2,))):
        mols, conf_ids = [], []
        for n, nc in zip(sizes, confs):
            q = rng.normal(size=4)
            mols.append(types.SimpleNamespace(atomcoords=rng.normal(size=(nc, n, 3)) * 2,
                                              rotation=ref_alg.quaternion_to_rotation_matrix(q / np.linalg.norm(q)),
                                              position=rng.normal(size=3) * 3))
            conf_ids.append(int(rng.integers(nc)))
        out = ref_embeds.get_embed(mols, conf_ids)
        for m, mol in enumerate(mols):
            data[f"ge{k}_coords{m}"], data[f"ge{k}_rot{m}"], data[f"ge{k}_pos{m}"] = mol.atomcoords, mol.rotation, mol.position
        data[f"ge{k}_conf_ids"], data[f"ge{k}_out"], data[f"ge{k}_n_mols"] = np.array(conf_ids), out, len(mols)
        k += 1
    data["ge_n"] = k

    # rotate_dihedral (utils.py:389-414): mask, indices_to_be_moved, and neither (only i1 moves)
    coords = rng.normal(size=(14, 3)) * 2
    mask = rng.random(14) < 0.4
    mask[[4, 5]] = False
    data["rd_coords"], data["rd_dihedral"], data["rd_mask"] = coords, np.array([2, 4, 5, 9]), mask
    data["rd_angles"] = np.array([30.0, -75.0, 180.0, 5.0])
    data["rd_out_mask"] = np.array([ref_utils.rotate_dihedral(coords.copy(), [2, 4, 5, 9], a, mask=mask) for a in data["rd_angles"]])
    data["rd_moved"] = np.array([0, 1, 2, 7])
    data["rd_out_indices"] = np.array([ref_utils.rotate_dihedral(coords.copy(), [2, 4, 5, 9], a, indices_to_be_moved=[0, 1, 2, 7])
                                       for a in data["rd_angles"]])
    data["rd_out_first"] = np.array([ref_utils.rotate_dihedral(coords.copy(), [2, 4, 5, 9], a) for a in data["rd_angles"]])
    _save("G10_embed_helpers", **data)


# --------------------------------------------------------------------------- molecules through the reference
def _write_xyz(path, atomnos, frames):
    with open(path, "w") as f:
        for fr in frames:
            f.write(f"{len(atomnos)}\nframe\n")
            for z, c in zip(atomnos, fr):
                f.write("%s %.8f %.8f %.8f\n" % (R.ELEMENTS[int(z)][0], *c))


def _molecule(path, reactive_indices, dist=None):
    """A reference Hypermolecule with its orbitals computed; ``dist`` applies a DIST(x=dist) keyword to every reactive atom
    (orb_dim = dist / 2, tscode/embedder.py:527-535)."""
    mol = ref_hc.Hypermolecule(path, list(reactive_indices))
    mol.compute_orbitals()
    if dist is not None:
        for c, _ in enumerate(mol.atomcoords):
            for index, r_atom in mol.reactive_atoms_classes_dict[c].items():
                r_atom.init(mol, index, update=True, orb_dim=dist / 2, conf=c)
    return mol


def _embedder(mols, embed, **options):
    opts = dict(clash_thresh=1.5, max_clashes=0, rigid=True, suprafacial=False, simpleorbitals=False, debug=False, threads=1)
    opts.update(options)
    e = types.SimpleNamespace(objects=list(mols), ids=np.array([len(m.atomnos) for m in mols]), embed=embed,
                              options=types.SimpleNamespace(**opts), candidates=0, log=lambda *a, **k: None,
                              pairings_table={}, internal_constraints=[])
    e._get_pivots = lambda mol: ref_embedder.Embedder._get_pivots(e, mol)      # the reference's own methods, unbound
    ref_embedder.Embedder._set_reactive_atoms_cumnums(e)        # r_atom.cumnum, tscode/embedder.py:355-367
    return e


# --------------------------------------------------------------------------- G11
def _trace_string_embed(embedder):
    """Run the reference's string_embed with note-taking wrappers around the functions its loop calls."""
    cand, clash, fps = [], [], []
    real_cc, real_fp = ref_embeds.compenetration_check, ref_embeds.get_torsion_fingerprint

    def cc(coords, **kw):
        ok = real_cc(coords, **kw)
        cand.append(coords)
        clash.append(bool(ok))
        return ok

    def fp(coords, quadruplets):
        out = real_fp(coords, quadruplets)
        fps.append((len(cand) - 1, out, quadruplets))
        return out

    ref_embeds.compenetration_check, ref_embeds.get_torsion_fingerprint = cc, fp
    try:
        poses = ref_embeds.string_embed(embedder)
    finally:
        ref_embeds.compenetration_check, ref_embeds.get_torsion_fingerprint = real_cc, real_fp
    cand = np.array(cand)
    kept = np.zeros(len(cand), dtype=bool)
    ids_of = {id(c): i for i, c in enumerate(cand)}
    # the kept poses are a subsequence of the candidates: match them in order
    j = 0
    for p in poses:
        while not np.array_equal(cand[j], p) or not clash[j]:
            j += 1
        kept[j] = True
        j += 1
    del ids_of
    quadruplets = fps[0][2] if fps else np.zeros((0, 4), dtype=int)
    fp_index = np.array([i for i, _, _ in fps], dtype=np.int64)
    fp_vals = np.array([v for _, v, _ in fps], dtype=np.float32).reshape(len(fps), -1)
    return dict(candidates=cand, clash_ok=np.array(clash), kept=kept, poses=np.asarray(poses), quadruplets=np.asarray(quadruplets),
                fp_index=fp_index, fingerprints=fp_vals, constrained_indices=np.asarray(embedder.constrained_indices))


def _string_inputs(embedder):
    """The plain arrays the loop of embeds.py:91-120 reads from its molecules."""
    out = {}
    for m, mol in enumerate(embedder.objects):
        out[f"coords{m}"] = np.asarray(mol.atomcoords)
        out[f"atomnos{m}"] = np.asarray(mol.atomnos)
        out[f"centers{m}"] = np.array([mol.get_r_atoms(c)[0].center for c in range(len(mol.atomcoords))])
        out[f"orb_vecs{m}"] = np.array([mol.get_r_atoms(c)[0].orb_vecs for c in range(len(mol.atomcoords))])
        out[f"reactive_index{m}"] = int(mol.reactive_indices[0])
    out["ids"] = np.asarray(embedder.ids)
    out["angles"] = np.asarray(embedder.systematic_angles, dtype=np.float64)
    out["clash_thresh"] = embedder.options.clash_thresh
    return out


def _conformers(atomnos, coords, rng, n_conf, axis_atoms, moved, jitter=0.02, jitter_first=False):
    """Conformers as data: atoms ``moved`` rotated about the axis_atoms bond by random angles, plus a small jitter.
    ``jitter_first`` also perturbs the first frame: the reference's C2H4 is exactly planar and symmetric, and once it is no
    longer centred on a round number its antarafacial pivots get a ``mol_direction`` (embeds.py:673) that is pure rounding
    noise (1e-17, not the exact zero the fallback of :674 tests for); align_vec_pair's SVD then returns whatever LAPACK's
    rounding makes of a rank-one matrix, which no other implementation can be asked to reproduce."""
    frames = [coords + rng.normal(size=coords.shape) * jitter if jitter_first else coords]
    for _ in range(n_conf - 1):
        c = coords.copy()
        a, b = axis_atoms
        rot = ref_alg.rot_mat_from_pointer(c[a] - c[b], float(rng.uniform(40, 320)))
        c[moved] = (rot @ (c[moved] - c[b]).T).T + c[b]
        frames.append(c + rng.normal(size=c.shape) * jitter)
    return np.array(frames)


def gen_g11():
    print("G11 string_embed on the reference's CH3Cl + HCOOH (config C1)")
    flat = {"seed": 9111}
    rng = np.random.default_rng(9111)
    scratch = "/tmp/tscode_amd_golden"
    os.makedirs(scratch, exist_ok=True)
    cases = []
    # case 0: tests/string.txt -- CH3Cl 0A, HCOOH 3A, DIST(A=2.5), 36 angles
    cases.append((os.path.join(TESTS, "CH3Cl.xyz"), [0], os.path.join(TESTS, "HCOOH.xyz"), [3], 2.5, 36, 1.5))
    # case 1: the same molecules, default orbital lengths, a larger clash threshold so that both verdicts occur
    cases.append((os.path.join(TESTS, "CH3Cl.xyz"), [0], os.path.join(TESTS, "HCOOH.xyz"), [3], None, 36, 1.5))
    # case 2: conformer ensembles (2 x 3 conformers, the loop order of cartesian_product), 24 angles
    z1, c1 = R.read_xyz_data(os.path.join(TESTS, "CH3Cl.xyz"))
    z2, c2 = R.read_xyz_data(os.path.join(TESTS, "HCOOH.xyz"))
    p1, p2 = os.path.join(scratch, "CH3Cl_confs.xyz"), os.path.join(scratch, "HCOOH_confs.xyz")
    _write_xyz(p1, z1, _conformers(z1, c1[0], rng, 2, (0, 4), [1, 2, 3]))
    _write_xyz(p2, z2, _conformers(z2, c2[0], rng, 3, (0, 3), [4]))
    cases.append((p1, [0], p2, [3], 2.2, 24, 2.05))
    # case 3: HCOOH attacked at its carbonyl oxygen (index 1, two lobes) by HCOOH's hydroxyl oxygen: 2 x 2 centre pairs
    cases.append((os.path.join(TESTS, "HCOOH.xyz"), [1], os.path.join(TESTS, "HCOOH.xyz"), [3], 2.0, 36, 1.8))
    for k, (f1, r1, f2, r2, dist, steps, thresh) in enumerate(cases):
        mols = [_molecule(f1, r1, dist), _molecule(f2, r2, dist)]
        e = _embedder(mols, "string", clash_thresh=thresh)
        e.systematic_angles = [n * 360 / steps for n in range(steps)]                 # tscode/embedder.py:735
        inp = _string_inputs(e)
        tr = _trace_string_embed(e)
        for key, v in {**inp, **tr}.items():
            flat[f"{key}_{k}"] = v
        print(f"  case {k}: {len(tr['candidates'])} candidates ({inp['centers0'].shape[1]} x {inp['centers1'].shape[1]} centre pairs, "
              f"{len(inp['coords0'])} x {len(inp['coords1'])} conformers), {int(tr['clash_ok'].sum())} pass the clash check, "
              f"{len(tr['poses'])} kept, {len(tr['quadruplets'])} quadruplets")
    flat["n_cases"] = len(cases)
    _save("G11_string_embed", **flat)


# --------------------------------------------------------------------------- G12
def _trace_cyclical_embed(embedder):
    ev_group, cand, group_of, clash, sim = [], [], [], [], []
    real_cc, real_sim, real_ids = ref_embeds.compenetration_check, ref_embeds._rmsd_similarity, ref_embeds._get_cyclical_reactive_indices

    def ids(emb, pivots, n):
        out = real_ids(emb, pivots, n)
        ev_group.append(np.array(out))
        return out

    def cc(coords, **kw):
        ok = real_cc(coords, **kw)
        cand.append(coords)
        group_of.append(len(ev_group) - 1)
        clash.append(bool(ok))
        return ok

    def sm(ref, structures, rmsd_thr=0.5):
        out = real_sim(ref, structures, rmsd_thr=rmsd_thr)
        sim.append((len(cand) - 1, bool(out)))
        return out

    ref_embeds.compenetration_check, ref_embeds._rmsd_similarity, ref_embeds._get_cyclical_reactive_indices = cc, sm, ids
    try:
        poses = ref_embeds.cyclical_embed(embedder)
    finally:
        ref_embeds.compenetration_check, ref_embeds._rmsd_similarity, ref_embeds._get_cyclical_reactive_indices = real_cc, real_sim, real_ids
    kept = np.zeros(len(cand), dtype=bool)
    for i, similar in sim:
        kept[i] = not similar
    assert int(kept.sum()) == len(poses) and np.array_equal(np.array(cand)[kept], poses)
    return dict(candidates=np.array(cand), group_of=np.array(group_of), group_ids=np.array(ev_group), clash_ok=np.array(clash), kept=kept,
                poses=np.asarray(poses), constrained_indices=np.asarray(embedder.constrained_indices))


def _cyclical_inputs(embedder):
    out = {}
    for m, mol in enumerate(embedder.objects):
        out[f"coords{m}"] = np.asarray(mol.atomcoords)
        out[f"atomnos{m}"] = np.asarray(mol.atomnos)
        out[f"reactive_indices{m}"] = np.asarray(mol.reactive_indices)
        for c in range(len(mol.atomcoords)):
            pv = mol.pivots[c]
            out[f"pivot_vec{m}_{c}"] = np.array([p.pivot for p in pv]).reshape(-1, 3)
            out[f"pivot_mean{m}_{c}"] = np.array([p.meanpoint for p in pv]).reshape(-1, 3)
            out[f"pivot_cumnums{m}_{c}"] = np.array([[p.start_atom.cumnum, p.end_atom.cumnum] for p in pv]).reshape(-1, 2)
    out["n_mols"] = len(embedder.objects)
    out["ids"] = np.asarray(embedder.ids)
    out["angles"] = np.asarray(embedder.systematic_angles, dtype=np.float64)
    out["clash_thresh"] = embedder.options.clash_thresh
    out["rigid"] = bool(embedder.options.rigid)
    return out


def gen_g12():
    print("G12 cyclical_embed (bimolecular: rigid shortcut, general loop, conformer ensembles)")
    flat = {"seed": 9112}
    C2H4, HCOOH = (os.path.join(TESTS, f) for f in ("C2H4.xyz", "HCOOH.xyz"))
    rng = np.random.default_rng(9112)
    scratch = "/tmp/tscode_amd_golden"
    os.makedirs(scratch, exist_ok=True)
    pc, ph = os.path.join(scratch, "C2H4_confs.xyz"), os.path.join(scratch, "HCOOH_confs2.xyz")
    zc, cc_ = R.read_xyz_data(C2H4)
    zh, ch_ = R.read_xyz_data(HCOOH)
    _write_xyz(pc, zc, _conformers(zc, cc_[0], rng, 2, (0, 3), [1, 2], jitter=0.03, jitter_first=True))
    _write_xyz(ph, zh, _conformers(zh, ch_[0], rng, 3, (0, 3), [4]))
    cases = [
        # tests/cyclical.txt: C2H4 0A 3B + C2H4 0B 3A, DIST(A=2.2, B=2.3 -> 2.2 here for both), default 5 rotation steps
        dict(files=[(C2H4, [0, 3]), (C2H4, [0, 3])], dist=2.2, steps=5, rot_range=45, rigid=True, general=False, thresh=1.5),
        # the general loop (embeds.py:453-732) on the same pair: RIGID only suppresses the bending branches there
        dict(files=[(C2H4, [0, 3]), (C2H4, [0, 3])], dist=2.2, steps=2, rot_range=45, rigid=True, general=True, thresh=1.3),
        # two different molecules, two reactive atoms each
        dict(files=[(C2H4, [0, 3]), (HCOOH, [1, 3])], dist=2.0, steps=3, rot_range=30, rigid=True, general=False, thresh=1.4),
        # conformer ensembles on both sides: the order of conf_indices / pivots_indices (cartesian_product)
        dict(files=[(pc, [0, 3]), (ph, [1, 3])], dist=2.1, steps=2, rot_range=40, rigid=True, general=False, thresh=1.45),
    ]
    for k, cs in enumerate(cases):
        mols = [_molecule(f, r, cs["dist"]) for f, r in cs["files"]]
        e = _embedder(mols, "cyclical", clash_thresh=cs["thresh"], rigid=cs["rigid"])
        for m in mols:
            ref_embedder.Embedder._set_pivots(e, m)                                  # tscode/embedder.py:542-570
        steps, rr = cs["steps"], cs["rot_range"]
        e.systematic_angles = ref_utils.cartesian_product(*[range(steps + 1) for _ in mols]) * 2 * rr / steps - rr   # embedder.py:714-715
        inp = _cyclical_inputs(e)
        if cs["general"] and len(mols) == 2:
            # the general loop is entered for bimolecular embeds only when RIGID is off; no branch of it bends a molecule
            # when the two pivots differ by less than max_norm_delta, which holds for these molecules
            e.options.rigid = False
            inp["rigid"] = False
        tr = _trace_cyclical_embed(e)
        for key, v in {**inp, **tr}.items():
            flat[f"{key}_{k}"] = v
        print(f"  case {k}: {len(mols)} molecules, pivots {[len(m.pivots[0]) for m in mols]}, {len(tr['candidates'])} candidates in "
              f"{len(tr['group_ids'])} groups, {int(tr['clash_ok'].sum())} pass the clash check, {len(tr['poses'])} kept")
    flat["n_cases"] = len(cases)
    _save("G12_cyclical_embed", **flat)


# --------------------------------------------------------------------------- G18
def gen_g18():
    """cyclical_embed for THREE molecules (tscode/embeds.py:244-465, 470-732) -- with ONE change to the reference: `vec_angle` as
    `_get_directions` calls it on 2-vectors (:297-299) pads them with z = 0.  As shipped that call is undefined (algebra.norm reads vec[2]
    of a 2-vector, tscode/algebra.py:87: an IndexError as plain NumPy, an unchecked out-of-bounds read under Numba); the padded form is the
    mathematically intended one (an angle between two plane vectors).  Everything else -- polygonize, _get_directions, _adjust_directions
    (its carry-over of `directions` from one orientation to the next, its use of conformer 0 for the reactive atoms), the pose loop -- is the
    reference's own code, run here.  Fixtures made this way are labelled "reference patched: unpinned where the reference is undefined"."""
    print("G18 cyclical_embed, three molecules (vec_angle of 2-vectors padded with z = 0)")
    flat = {"seed": 9118, "patch": "tscode.embeds.vec_angle pads 2-vectors with z = 0 (embeds.py:297-299 / algebra.py:87)"}
    real_va = ref_embeds.vec_angle

    def va(v1, v2):
        v1, v2 = np.asarray(v1, dtype=np.float64), np.asarray(v2, dtype=np.float64)
        if v1.shape[0] == 2:
            v1 = np.concatenate((v1, [0.0]))
        if v2.shape[0] == 2:
            v2 = np.concatenate((v2, [0.0]))
        return real_va(v1, v2)
    CH3Cl, HCOOH = (os.path.join(TESTS, f) for f in ("CH3Cl.xyz", "HCOOH.xyz"))
    rng = np.random.default_rng(9118)
    scratch = "/tmp/tscode_amd_golden"
    os.makedirs(scratch, exist_ok=True)
    ph = os.path.join(scratch, "HCOOH_confs3.xyz")
    zh, ch_ = R.read_xyz_data(HCOOH)
    _write_xyz(ph, zh, _conformers(zh, ch_[0], rng, 2, (0, 3), [4]))
    cases = [
        # tests/trimolecular.txt: CH3Cl 0A 4y + HCOOH 1A 4x + HCOOH 1x 4y (one DIST for all pairings here), STEPS=1 ROTRANGE=10
        dict(files=[(CH3Cl, [0, 4]), (HCOOH, [1, 4]), (HCOOH, [1, 4])], dist=2.3, steps=1, rot_range=10, thresh=1.3),
        # more angle sets per group, a tighter clash threshold
        dict(files=[(CH3Cl, [0, 4]), (HCOOH, [1, 4]), (HCOOH, [1, 4])], dist=2.6, steps=2, rot_range=20, thresh=1.2),
        # a conformer ensemble on one side: the order of conf_indices / pivots_indices
        dict(files=[(CH3Cl, [0, 4]), (ph, [1, 4]), (HCOOH, [1, 4])], dist=2.4, steps=1, rot_range=15, thresh=1.25),
    ]
    ref_embeds.vec_angle = va
    try:
        for k, cs in enumerate(cases):
            mols = [_molecule(f, r, cs["dist"]) for f, r in cs["files"]]
            e = _embedder(mols, "cyclical", clash_thresh=cs["thresh"], rigid=True)
            for m in mols:
                ref_embedder.Embedder._set_pivots(e, m)
            steps, rr = cs["steps"], cs["rot_range"]
            e.systematic_angles = ref_utils.cartesian_product(*[range(steps + 1) for _ in mols]) * 2 * rr / steps - rr
            inp = _cyclical_inputs(e)
            for m, mol in enumerate(mols):     # which reactive atom carries which cumulative number (_adjust_directions pairs them up, :378-385)
                inp[f"reactive_cumnums{m}"] = np.array([[int(i), int(a.cumnum)] for i, a in mol.reactive_atoms_classes_dict[0].items()]).reshape(-1, 2)
            tr = _trace_cyclical_embed(e)
            for key, v in {**inp, **tr}.items():
                flat[f"{key}_{k}"] = v
            print(f"  case {k}: pivots {[len(m.pivots[0]) for m in mols]}, {len(tr['candidates'])} candidates in {len(tr['group_ids'])} groups, "
                  f"{int(tr['clash_ok'].sum())} pass the clash check, {len(tr['poses'])} kept")
    finally:
        ref_embeds.vec_angle = real_va
    flat["n_cases"] = len(cases)
    _save("G18_cyclical_embed_trimolecular", **flat)


if __name__ == "__main__":
    which = sys.argv[1:] or ["G10", "G11", "G12"]
    for g in which:
        globals()["gen_" + g.lower()]()
