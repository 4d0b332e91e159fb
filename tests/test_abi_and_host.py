"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the
host-side mirrors of the reference's small helpers match the golden vectors, install() patches
every binding site.  No compute call is made (there is no GPU here)."""

import os
import re
import sys
import types

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "tscode_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    from tscode_amd import _lib
    from tscode_amd.build import build
    build()
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/tscode_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == syms, "ctypes prototype table and header disagree"
    assert lib.tsc_version() == 100


def test_no_cpp_exception_can_cross_the_c_abi():
    """"no C++ exception crosses the boundary".  The library's context holds std::map / std::vector and calls `new`:
    every `extern "C"` entry point that returns a status must run its body inside the try / catch of TSC_API_GUARD_BEGIN / _END
    (csrc/common.hpp: std::bad_alloc -> TSC_ERR_NOMEM, anything else -> TSC_ERR_INVALID).  Checked on the source text: a thrown
    exception cannot be provoked from here without a GPU."""
    csrc = os.path.join(ROOT, "tscode_amd", "csrc")
    guarded, bare = [], []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".hpp")):
            continue
        text = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r'^extern "C"[^\n]*?\bint (tsc_[a-z0-9_]+)\(', text, flags=re.M):
            body_start = text.index("{", m.end())
            first = text[body_start + 1:body_start + 200].lstrip()
            if first.startswith("TSC_API_GUARD_BEGIN"):
                guarded.append(m.group(1))
            elif first.startswith("return TSC_VERSION"):      # a constant: nothing in it can throw
                continue
            else:
                bare.append(m.group(1))
    assert not bare, f"entry points without the exception barrier: {bare}"
    assert len(guarded) >= 60
    common = open(os.path.join(csrc, "common.hpp")).read()
    assert "catch (const std::bad_alloc &)" in common and "catch (...)" in common


def test_build_covers_every_translation_unit_and_the_binary_is_current():
    """tscode_amd/build.py compiles the units it lists: a .hip file under csrc/ that is not in SOURCES would never reach the library (its entry
    points would be missing symbols at load, its kernels silently absent), a header that is not hashed could change under a binary that still
    calls itself current.  And the library beside the sources carries the digest of exactly these sources (what bench.py and
    tools/collect_profiles.py tie a profile to)."""
    from tscode_amd import build as b
    from tscode_amd import _lib
    on_disk = sorted(f for f in os.listdir(b.CSRC) if f.endswith(".hip"))
    assert sorted(b.SOURCES) == on_disk
    assert b.DIGEST_UNIT in b.SOURCES
    hashed = {os.path.basename(h) for h in b.HEADERS}
    assert {f for f in os.listdir(b.CSRC) if f.endswith(".hpp")} <= hashed and "tscode_hip.h" in hashed
    b.build()
    assert _lib.load().tsc_build_digest().decode() == b.csrc_digest()
    # every unit's dependency list (hipcc -MD) names the unit itself and only files the digest covers
    if os.path.isdir(b.OBJ):
        for src in b.SOURCES:
            deps = b._unit_deps(src)
            if deps is None:
                continue                                   # (a checkout whose objects were never built here)
            names = {os.path.basename(d) for d in deps}
            assert src in names and names <= hashed | set(b.SOURCES), (src, names)


def test_product_does_not_import_oracle():
    import tscode_amd  # noqa: F401
    pkg = os.path.join(ROOT, "tscode_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in src and "tsc_oracle" not in src, f"{f} references the oracle library"


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tscode_amd import _lib
    from tscode_amd.engine import Engine
    with pytest.raises(_lib.TscodeHipError):
        Engine(0)
    import tscode_amd
    with pytest.raises(_lib.TscodeHipError):
        tscode_amd.prune_conformers_rmsd(np.zeros((4, 3, 3)), np.array([6, 6, 1]))


def test_host_rotation_helpers_match_golden():
    from tscode_amd import algebra as alg
    g = load_golden("G5_rotations")
    for p, a, ref in zip(g["ptr"], g["ang"], g["rot_mat_from_pointer"]):
        assert np.allclose(alg.rot_mat_from_pointer(p, a), ref, atol=1e-14, rtol=0)
    for q, ref in zip(g["quat"], g["quat_to_mat"]):
        assert np.allclose(alg.quaternion_to_rotation_matrix(q), ref, atol=1e-15, rtol=0)
    for r, t, ref in zip(g["avp_ref"], g["avp_tgt"], g["align_vec_pair"]):
        assert np.allclose(alg.align_vec_pair(r, t), ref, atol=1e-12, rtol=0)
    for a, b, ref in zip(g["va_v1"], g["va_v2"], g["vec_angle"]):
        assert abs(alg.vec_angle(a, b) - ref) < 1e-9


def test_rotation_matrix_from_vectors_matches_oracle(oracle):
    from tscode_amd import algebra as alg
    rng = np.random.default_rng(5)
    cases = [(rng.normal(size=3), rng.normal(size=3)) for _ in range(20)]
    v = rng.normal(size=3)
    cases += [(v, 2.5 * v), (v, -v), (np.array([0, 0, 1.0]), np.array([0, 0, -3.0]))]   # parallel / antiparallel
    for a, b in cases:
        got, ref = alg.rotation_matrix_from_vectors(a, b), oracle.rotation_matrix_from_vectors(a, b)
        assert np.allclose(got, ref, atol=1e-13, rtol=0)
        if np.linalg.norm(np.cross(a, b)) > 1e-9:
            assert np.allclose(got @ (a / np.linalg.norm(a)), b / np.linalg.norm(b), atol=1e-12)


def test_fragment_set_layout():
    from tscode_amd.engine import FragmentSet
    fs = FragmentSet([np.zeros((2, 5, 3)), np.ones((7, 3)), np.zeros((3, 4, 3))])
    assert fs.n_mols == 3 and fs.n_total == 16
    assert fs.n_atoms.tolist() == [5, 7, 4] and fs.n_conf.tolist() == [2, 1, 3]
    assert fs.frag_off.tolist() == [0, 30, 51] and fs.flat.size == 30 + 21 + 36
    assert np.all(fs.flat[30:51] == 1)


def test_install_table_equals_the_reference_binding_sites():
    """G14 (tests/golden/gen_install_sites.py, build container only): every module of the REAL reference imported, and for every
    hot-path name the modules whose namespace binds it to the defining module's object.  install.py's hand-written table must be
    exactly that -- a module missing from it keeps calling the reference's function after install()."""
    import json
    from tscode_amd.install import _PATCHES
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "G14_install_sites.json")))
    assert not g["modules_not_importable_here"], g["modules_not_importable_here"]
    assert len(g["modules_imported"]) >= 25
    for attr, (fn, names) in _PATCHES.items():
        site = g["sites"][attr]
        assert site["defined_in"] is not None, f"{attr}: not defined anywhere in the reference"
        assert sorted(names) == site["bound_in"], f"{attr}: install.py patches {sorted(names)}, the reference binds it in {site['bound_in']}"
        assert callable(fn)
    assert set(g["sites"]) == set(_PATCHES)


def test_install_patches_every_binding_site():
    import tscode_amd
    names = ["tscode.rmsd_pruning", "tscode.embedder", "tscode.operators", "tscode.optimization_methods",
             "tscode.atropisomer_module", "tscode.numba_functions", "tscode.embeds", "tscode.algebra", "tscode.automep"]
    fake = {}
    for n in names:
        m = types.ModuleType(n)
        for attr in ("prune_conformers_rmsd", "compenetration_check", "_rmsd_similarity", "get_embed", "rmsd_and_max_numba",
                     "all_dists", "count_clashes"):
            setattr(m, attr, "reference")
        fake[n] = m
    done = tscode_amd.install(modules=fake)               # default: only what takes a whole ensemble per call
    assert ("tscode.embedder", "prune_conformers_rmsd") in done and ("tscode.embeds", "compenetration_check") not in done
    assert fake["tscode.embeds"].compenetration_check == "reference" and fake["tscode.embedder"].prune_conformers_rmsd is tscode_amd.prune_conformers_rmsd
    tscode_amd.uninstall(modules=fake)
    done = tscode_amd.install(modules=fake, per_item=True)
    assert ("tscode.embedder", "prune_conformers_rmsd") in done and ("tscode.embeds", "compenetration_check") in done
    for n in ("tscode.rmsd_pruning", "tscode.embedder", "tscode.operators", "tscode.optimization_methods", "tscode.atropisomer_module"):
        assert fake[n].prune_conformers_rmsd is tscode_amd.prune_conformers_rmsd
    for n in ("tscode.numba_functions", "tscode.embedder", "tscode.embeds"):
        assert fake[n].compenetration_check is tscode_amd.compenetration_check
    assert fake["tscode.embeds"].get_embed is tscode_amd.get_embed
    assert fake["tscode.operators"].compenetration_check == "reference"     # operators.py does not bind it
    tscode_amd.uninstall(modules=fake)
    assert fake["tscode.embedder"].prune_conformers_rmsd == "reference"
    assert "tscode" not in sys.modules                                       # the package never imports tscode


def test_synthetic_configs_are_deterministic():
    from tscode_amd.synthetic import make_config
    a, b = make_config("C2", 500), make_config("C2", 500)
    assert np.array_equal(a.rot, b.rot) and np.array_equal(a.pos, b.pos)
    assert a.n_atoms == 30 and a.n_heavy == 18
    c3 = make_config("C3", 100)
    assert c3.n_atoms == 50 and c3.n_heavy == 30
    p = a.poses(0, 10)
    assert p.shape == (10, 30, 3)
    assert np.allclose(p[:, :15], a.frag_coords[0][0])                       # fragment 0 is fixed


def test_pairing_filter_reads_internal_constraints_like_the_reference():
    """tscode/embeds.py:642 evaluates `pair in embedder.internal_constraints` on an ndarray of index pairs: NumPy's `in` is
    (array == pair).any() -- one matching index in its column is enough."""
    import numpy as np

    from tscode_amd.embeds import _in_constraints
    ic = np.array([[3, 7], [10, 12]])
    for pair in ([3, 7], [3, 99], [99, 12], [7, 3], [1, 2]):
        assert _in_constraints(pair, ic) == (pair in ic), pair               # the reference's own expression
    assert _in_constraints([3, 99], ic) and not _in_constraints([7, 3], ic)
    assert not _in_constraints([3, 7], []) and not _in_constraints([3, 7], None)


def test_dropins_read_from_duck_types_what_they_read_from_the_reference_objects():
    """tests/golden/G13_dropin_reads.json: the attributes tscode_amd.embeds.string_embed / cyclical_embed read from the reference's
    own Embedder-side objects (real Hypermolecule, Pivot and orbital objects; recorded in the build container by
    tests/golden/gen_dropin_reads.py) and what each held.  The duck-typed objects the GPU parity test hands the drop-ins must be
    read the same way: the same attributes, holding the same kinds of value (array rank and dtype kind included)."""
    import json
    import os
    import sys
    import dropin_reads
    import tscode_amd.embeds as E
    from conftest import load_golden
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "G13_dropin_reads.json")))
    g11, g12 = load_golden("G11_string_embed"), load_golden("G12_cyclical_embed")

    class Zero(Exception):
        pass
    saved = sys.modules.get("tscode.embeds")
    sys.modules["tscode.embeds"] = dropin_reads.reference_module_standin(Zero, g11, 0)
    try:
        got = {"string_embed": dropin_reads.record(E.string_embed, dropin_reads.duck_string_embedder(g11, 0, [])),
               "cyclical_embed": dropin_reads.record(E.cyclical_embed, dropin_reads.duck_cyclical_embedder(g12, 0, []))}
    finally:
        if saved is None:
            sys.modules.pop("tscode.embeds", None)
        else:
            sys.modules["tscode.embeds"] = saved

    def same(a, b):
        if a is None or b is None:                      # an empty sequence on either side says nothing about its elements
            return True
        if a["type"] != b["type"]:
            return {a["type"], b["type"]} == {"none", "object"} or {a["type"], b["type"]} <= {"int", "float"}
        if a["type"] == "ndarray":
            return a["ndim"] == b["ndim"] and a["kind"] == b["kind"]
        return same(a.get("of"), b.get("of")) if a["type"] == "sequence" else True
    for dropin in want:
        assert set(got[dropin]) == set(want[dropin]), dropin
        for kind in want[dropin]:
            assert set(got[dropin][kind]) == set(want[dropin][kind]), (dropin, kind, sorted(got[dropin][kind]), sorted(want[dropin][kind]))
            for name, held in want[dropin][kind].items():
                assert same(got[dropin][kind][name], held), (dropin, kind, name, got[dropin][kind][name], held)


def test_rotate_dihedral_single_structure_is_host_side():
    """The one-structure drop-in of rotate_dihedral (tscode/utils.py:389-414) computes on the host like the other single 3x3 helpers
    (no library call: the reference's search loops call it per torsion and per 5-degree walk-back step, torsion_module.py:482-489) and
    matches G15 -- masks, `dihedral[0]` alone, and the rotate / undo trail of torsion_module.py:984-1005."""
    from tscode_amd.torsion_module import rotate_dihedral
    g = load_golden("G15_rotate_dihedral_fractional")
    coords, dih, mask = g["coords"], g["dihedral"], g["mask"].astype(bool)
    for a, rm, rf in zip(g["angles"], g["out_mask"], g["out_first"]):
        c = coords.copy()
        assert rotate_dihedral(c, dih, float(a), mask=mask) is c and np.abs(c - rm).max() < 1e-12
        assert np.abs(rotate_dihedral(coords.copy(), dih, float(a)) - rf).max() < 1e-12
        moved = list(np.flatnonzero(mask))
        assert np.abs(rotate_dihedral(coords.copy(), dih, float(a), indices_to_be_moved=moved) - rm).max() < 1e-12
    seq, q = coords.copy(), 0
    for a in g["angles"][:4]:
        for sign in (1.0, -1.0):
            seq = rotate_dihedral(seq, dih, sign * float(a), mask=mask)
            assert np.abs(seq - g["trail"][q]).max() < 1e-12
            q += 1
    same = coords.copy()
    assert rotate_dihedral(same, dih, 0.0, mask=mask) is same and np.array_equal(same, coords)


def test_trimolecular_groups_host_logic_vs_g18(oracle):
    """The host side of the three-molecule cyclical embed (tscode_amd.embeds._trimolecular_groups: polygonize, _get_directions with the
    plane-vector reading of vec_angle, _adjust_directions with its carry-over from orientation to orientation) against G18, the
    reference's own code run with that one change: the groups and their facing atoms, and -- the groups' records pushed through the
    oracle's per-row pose parameters and embedding on the CPU -- every candidate pose."""
    from tscode_amd.embeds import _trimolecular_groups
    g = load_golden("G18_cyclical_embed_trimolecular")
    for k in range(int(g["n_cases"])):
        coords = [np.ascontiguousarray(g[f"coords{m}_{k}"], dtype=np.float64) for m in range(3)]
        reactive = [np.atleast_1d(g[f"reactive_indices{m}_{k}"]).astype(np.int64) for m in range(3)]
        pivots = [[(g[f"pivot_vec{m}_{c}_{k}"], g[f"pivot_mean{m}_{c}_{k}"], g[f"pivot_cumnums{m}_{c}_{k}"]) for c in range(len(coords[m]))] for m in range(3)]
        cumnums = [g[f"reactive_cumnums{m}_{k}"] for m in range(3)]
        blocks, gids, meta = _trimolecular_groups(coords, reactive, pivots, cumnums, None, ())
        assert np.array_equal(gids, g[f"group_ids_{k}"]) and len(meta) == len(gids)
        angles = g[f"angles_{k}"]
        A = len(angles)
        assert len(blocks) * A == len(g[f"candidates_{k}"])
        worst = 0.0
        for gi in range(0, len(blocks), max(1, len(blocks) // 12)):          # a dozen groups spread over the list, all their angle sets
            rec = blocks[gi]
            for ai, ang in enumerate(angles):
                rot, pos = np.empty((1, 3, 3, 3)), np.empty((1, 3, 3))
                for m in range(3):
                    rm, pm = oracle.cyclical_embed_params([rec[m, 0:3]], [rec[m, 3:6]], [rec[m, 6:9]], [rec[m, 9:12]], [rec[m, 12:15]], [rec[m, 15:18]],
                                                          [rec[m, 18:21]], [int(rec[m, 21])], [ang[m]])
                    rot[0, m], pos[0, m] = rm[0], pm[0]
                pose = oracle.transform_batch(coords, [[int(rec[m, 22]) for m in range(3)]], rot, pos)[0]
                worst = max(worst, float(np.abs(pose - g[f"candidates_{k}"][gi * A + ai]).max()))
        assert worst < 1e-9, (k, worst)
