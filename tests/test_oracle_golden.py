"""The CPU oracle against the golden vectors recorded from the reference's own Python
(tests/golden/gen_golden.py).  These pin the oracle; the GPU parity tests then compare
the HIP path with the oracle."""

import numpy as np
import pytest

from conftest import LARGE_PRUNE_CASES, load_golden, load_large_prune

TOL = 1e-9


def test_g1_rmsd_and_max(oracle):
    g = load_golden("G1_rmsd_and_max")
    worst = 0.0
    for h in (3, 5, 9, 18, 30, 60, 120):
        p, q, ref = g[f"p_{h}"], g[f"q_{h}"], g[f"out_{h}"]
        tags = g[f"tags_{h}"]
        for a, b, r, tag in zip(p, q, ref, tags):
            rm, mx = oracle.rmsd_and_max_numba(a, b)
            assert abs(rm - r[0]) < TOL, (h, tag, rm, r[0])
            assert abs(mx - r[1]) < TOL, (h, tag, mx, r[1])
            worst = max(worst, abs(rm - r[0]), abs(mx - r[1]))
        pairs = np.stack([np.arange(len(p)), np.arange(len(p)) + len(p)], axis=1)
        rr, mm = oracle.rmsd_pairs(np.concatenate([p, q]), pairs)
        assert np.allclose(rr, ref[:, 0], atol=TOL, rtol=0) and np.allclose(mm, ref[:, 1], atol=TOL, rtol=0)
    print("G1 worst abs deviation", worst)


def test_g2_all_dists(oracle):
    g = load_golden("G2_clash")
    for k in range(int(g["ad_n"])):
        out = oracle.all_dists(g[f"ad_a{k}"], g[f"ad_b{k}"])
        assert out.shape == g[f"ad_out{k}"].shape
        assert np.allclose(out, g[f"ad_out{k}"], atol=1e-12, rtol=0)


def test_g2_compenetration_check(oracle):
    g = load_golden("G2_clash")
    combos = g["cc_combos"]
    seen = set()
    for k in range(int(g["cc_n"])):
        ids, coords = g[f"cc_ids{k}"], g[f"cc_coords{k}"]
        for ci, (thresh, mc) in enumerate(combos):
            ref = g[f"cc_out{k}"][ci]
            counts_ref = g[f"cc_counts{k}"][0 if thresh == 1.4 else 1]
            got = np.array([oracle.compenetration_check(c, ids, thresh, int(mc)) for c in coords])
            assert np.array_equal(got, ref), (k, thresh, mc)
            assert np.array_equal(oracle.compenetration_mask(coords, ids, thresh, int(mc)), ref.astype(bool))
            # per-fragment-pair counts, in the reference's order, wherever the oracle evaluated them
            for s, c in enumerate(coords):
                _, cnt = oracle.compenetration_check(c, ids, thresh, int(mc), return_counts=True)
                for j in range(counts_ref.shape[1]):
                    if cnt[j] >= 0:
                        assert cnt[j] == counts_ref[s, j]
            seen.update(ref.tolist())
    assert seen == {0, 1}          # both verdicts occur in the fixtures


def test_g2_count_clashes(oracle):
    g = load_golden("G2_clash")
    total = 0
    for k in range(int(g["cl_n"])):
        c = g[f"cl_coords{k}"]
        assert oracle.count_clashes(c) == int(g[f"cl_count{k}"])
        got = [oracle.compenetration_check(c, None, 1.5, mc) for mc in (0, 2, 10)]
        assert got == g[f"cl_check{k}"].tolist()
        total += int(g[f"cl_count{k}"])
    assert total > 0


@pytest.mark.parametrize("row_parallel", [False, True])
def test_g3_prune(oracle, row_parallel):
    g = load_golden("G3_prune")
    for c in range(int(g["n_cases"])):
        structures, atomnos, thr = g[f"structures{c}"], g[f"atomnos{c}"], float(g[f"thr{c}"])
        pruned, mask = oracle.prune_conformers_rmsd(structures, atomnos, thr, row_parallel=row_parallel)
        assert np.array_equal(mask, g[f"mask{c}"]), f"case {c}: {mask.sum()} vs {g[f'mask{c}'].sum()}"
        assert np.array_equal(pruned, structures[g[f"mask{c}"]])
        heavy = structures[:, atomnos != 1]
        res = oracle.prune_heavy(heavy, thr, trace=True, row_parallel=row_parallel)
        assert [s["k"] for s in res["stats"]] == g[f"ks{c}"].tolist()
        assert np.array_equal(res["pass_masks"], g[f"pass_masks{c}"])
        assert np.cumsum([s["new_keys"] for s in res["stats"]]).tolist() == g[f"pass_nkeys{c}"].tolist()
        keys = np.array(sorted(set(map(tuple, res["keys"].tolist()))), dtype=np.int64).reshape(-1, 2)
        assert np.array_equal(keys, g[f"keys{c}"])


@pytest.mark.parametrize("row_parallel", [False, True])
@pytest.mark.parametrize("name", LARGE_PRUNE_CASES)
def test_g16_g17_prune_large(oracle, name, row_parallel):
    """The FINE passes of the schedule (k = 5000 ... 200) and their cache-key collisions with the coarse ones, pinned by the
    reference's own prune_conformers_rmsd (rmsd_pruning.py:65-67, 131-144, 186-204) on ensembles large enough for them to run:
    final mask, every per-pass mask, which passes ran, cumulative key counts and the key set itself."""
    fx = load_large_prune(name)
    g = fx.g
    res = oracle.prune_heavy(fx.heavy, fx.thr, trace=True, row_parallel=row_parallel)
    assert [s["k"] for s in res["stats"]] == g["ks"].tolist()
    assert max(g["ks"]) >= (5000 if name == "G17" else 2000) and {1000, 500, 200} <= set(g["ks"].tolist())
    assert np.array_equal(np.packbits(res["mask"]), g["mask_bits"]), (int(res["mask"].sum()), int(np.unpackbits(g["mask_bits"])[:fx.n].sum()))
    assert np.array_equal(np.packbits(res["pass_masks"], axis=1), g["pass_mask_bits"])
    assert np.cumsum([s["new_keys"] for s in res["stats"]]).tolist() == g["pass_nkeys"].tolist()
    keys = np.array(sorted(set(map(tuple, res["keys"].tolist()))), dtype=np.int64).reshape(-1, 2)
    assert np.array_equal(keys, g["keys"].astype(np.int64))
    # N is divisible by no k that ran: the last chunk took a remainder in every pass (rmsd_pruning.py:141-142)
    assert all(fx.n % int(k) for k in g["ks"] if k > 1)
    if not row_parallel:          # the call the reference's callers make: (structures[mask], mask) from ALL atoms + atomnos
        pruned, mask = oracle.prune_conformers_rmsd(fx.structures, fx.atomnos, fx.thr)
        assert np.array_equal(np.packbits(mask), g["mask_bits"]) and np.array_equal(pruned, fx.structures[mask])
        mr, mm = oracle.prune_margins(fx.heavy, fx.thr, 0)
        assert mr > 1e-6 and mm > 1e-6, "guard band (SURVEY 8d): an evaluated pair lies within 1e-6 of a threshold"


def test_g3_cache_free_mode_differs(oracle):
    """mode=1 (cache-free) is the chemically intended result, NOT the reference's."""
    g = load_golden("G3_prune")
    structures, atomnos = g["structures1"], g["atomnos1"]
    _, m0 = oracle.prune_conformers_rmsd(structures, atomnos, 0.5, mode=0)
    _, m1 = oracle.prune_conformers_rmsd(structures, atomnos, 0.5, mode=1)
    assert m1.sum() < m0.sum()
    assert m1.sum() == 30          # 600 poses = 30 parents x 20 children


def test_g4_greedy_filter(oracle):
    g = load_golden("G4_rmsd_similarity")
    for c in range(int(g["n_cases"])):
        poses, ref = g[f"poses{c}"], g[f"accepted{c}"]
        assert np.array_equal(oracle.greedy_group_filter(poses, 1.0), ref)
        kept = [p for p, a in zip(poses, ref) if a]
        # the last pose against everything accepted before it, through the plain function
        assert oracle._rmsd_similarity(poses[-1], np.array(kept[:-1] if ref[-1] else kept), 1.0) == (not ref[-1])


def test_g5_rotations(oracle):
    g = load_golden("G5_rotations")
    for p, a, ref in zip(g["ptr"], g["ang"], g["rot_mat_from_pointer"]):
        assert np.allclose(oracle.rot_mat_from_pointer(p, a), ref, atol=1e-13, rtol=0)
    for q, ref in zip(g["quat"], g["quat_to_mat"]):
        assert np.allclose(oracle.quaternion_to_rotation_matrix(q), ref, atol=1e-14, rtol=0)
    for r, t, ref in zip(g["avp_ref"], g["avp_tgt"], g["align_vec_pair"]):
        got = oracle.align_vec_pair(r, t)
        # the rotation is pinned on the two vectors it aligns (the third axis is free when rank(B) = 2)
        assert np.allclose(got @ t.T, ref @ t.T, atol=1e-9, rtol=0)
        assert abs(np.linalg.det(got) - 1) < 1e-12
    for a, b, ref in zip(g["va_v1"], g["va_v2"], g["vec_angle"]):
        assert abs(oracle.vec_angle(a, b) - ref) < 1e-6   # acos near +-1 amplifies last-bit differences
    out = oracle.transform_coords(g["tc_coords"], g["tc_rot"], g["tc_pos"])
    assert np.allclose(out, g["transform_coords"], atol=1e-13, rtol=0)


def test_g7_csearch_rotations(oracle):
    """rotate_dihedral / torsion_comp_check / the candidate loop of torsion_module.py:463-500."""
    g = load_golden("G7_csearch")
    for c in range(int(g["n_cases"])):
        coords, torsions, masks, angles = g[f"coords{c}"], g[f"torsions{c}"], g[f"masks{c}"], g[f"angles{c}"]
        out, rb, margin = oracle.csearch_rotate(coords, torsions, masks, angles, 1.5, 0, return_margin=True)
        assert margin > 1e-9                                      # guard band: no distance sits on the threshold
        assert np.array_equal(rb, g[f"rotated_bonds{c}"])
        assert np.abs(out - g[f"out{c}"]).max() < 1e-9
        # single calls
        t0 = next(t for t in range(len(torsions)) if angles[0][t] != 0)
        one = oracle.rotate_dihedral(coords, torsions[t0], float(angles[0][t0]), masks[t0])
        assert oracle.torsion_comp_check(one, torsions[t0], masks[t0], 1.5) == int(g[f"first_checks{c}"][0])
    # part B: the reference's whole random_csearch (shuffled n-fold angle tables, n_out / max_tries selection, :505-511)
    for b in range(int(g["b_n"])):
        angles, n_out, max_tries = g[f"b_angles{b}"], int(g[f"b_n_out{b}"]), int(g[f"b_max_tries{b}"])
        out, rb = oracle.csearch_rotate(g["b_coords"], g["b_torsions"], g["b_masks"], angles.astype(np.int32), 1.5, 0)
        kept = []
        for a in range(len(angles)):
            if rb[a] != 0:
                kept.append(a)
                if len(kept) == n_out or a == max_tries:
                    break
        assert len(kept) == len(g[f"b_out{b}"]) and np.abs(out[kept] - g[f"b_out{b}"]).max() < 1e-9


def test_g6_g8_torsion_fingerprint_pruning(oracle):
    """fingerprints / tfd_similarity (G6) and prune_conformers_tfd end to end (G8, the reference's own
    function with networkx): the oracle's pair search + the product's host-side graph step give the reference's mask."""
    pytest.importorskip("networkx")
    from tscode_amd.numba_functions import _tfd_schedule          # the product's host-side schedule + graph step (pure Python)
    g6 = load_golden("G6_tfd")
    fp = oracle.torsion_fingerprints(g6["coords"], g6["quadruplets"])
    assert fp.dtype == np.float32 and np.abs(fp - g6["fingerprints"]).max() < 2e-5
    sims = [[oracle.tfd_similarity(g6["fingerprints"][i], g6["fingerprints"][j], t) for j in range(10)] for i in range(10) for t in (10, 90, 300)]
    assert np.array_equal(np.array(sims), g6["similarity"].astype(bool))
    g = load_golden("G8_tfd_prune")
    for c in range(int(g["n_cases"])):
        structures, thresh = g[f"structures{c}"], float(g[f"thresh{c}"])
        tf = oracle.torsion_fingerprints(structures, g["quadruplets"])
        assert np.abs(tf - g[f"tf_mat{c}"]).max() < 2e-5
        margins = []

        def first_similar(tf_mat, d, k, num_active, th):         # the pair search answered by the CPU oracle
            first, m = oracle.tfd_first_similar(tf_mat, d, k, num_active, th, return_margin=True)
            margins.append(m)
            return first
        pruned, mask = _tfd_schedule(structures, tf, thresh, False, first_similar)
        assert min(margins) > 1e-6                                # no fingerprint sum sits on the threshold
        assert np.array_equal(mask, g[f"mask{c}"]), (c, mask.sum(), g[f"mask{c}"].sum())
        assert np.array_equal(pruned, structures[mask])


def test_g9_moments_and_scores(oracle):
    """the oracle against the reference's get_inertia_moments / get_moi_similarity_matches /
    _score_embed_poses (G9)."""
    g = load_golden("G9_moi_scores")
    for c in range(int(g["n_cases"])):
        mo = oracle.inertia_moments(g[f"structures{c}"], g[f"masses{c}"])
        assert (np.abs(mo - g[f"moments{c}"]) / np.abs(g[f"moments{c}"])).max() < 1e-12
        first, margin = oracle.moi_first_similar(g[f"moments{c}"], 1e-2, return_margin=True)
        ref = np.full(len(mo), -1)
        ref[g[f"matches{c}"][:, 0]] = g[f"matches{c}"][:, 1]
        assert margin > 1e-9 and np.array_equal(first, ref)
    sc, err = oracle.embed_scores(g["sc_structures"], g["sc_indices"], g["sc_distances"])
    assert np.abs(sc - g["scores"]).max() < 1e-6
    thr = float(g["fitness_threshold"])
    assert np.abs(err - thr).min() > 1e-9                      # guard band of the reference's own fitness_check verdicts
    assert np.array_equal(err < thr, g["fitness_ok"])
    _, err_none = oracle.embed_scores(g["sc_structures"], g["sc_indices"], g["fitness_none"])       # targets of None are skipped (:552)
    assert np.array_equal(err_none < thr, g["fitness_ok_none"])


def test_tfd_single_match_chunks_need_no_graph():
    """The shortcut of _tfd_reject_matches for chunks with one match against the graph step itself (networkx) on the same input."""
    from tscode_amd.numba_functions import _tfd_reject_matches, _tfd_reject_graph
    rng = np.random.default_rng(5)
    n, d, k = 5000, 10, 500
    first = np.full(n, -1, dtype=np.int32)
    for c in rng.choice(k, size=300, replace=False):              # one match (i, j > i) in each of 300 chunks ...
        i = c * d + rng.integers(0, d - 1)
        first[i] = rng.integers(i + 1, (c + 1) * d)
    for c in rng.choice(k, size=40, replace=False):               # ... and a few chunks with several
        if np.all(first[c * d:(c + 1) * d] < 0):
            first[c * d], first[c * d + 3] = c * d + 5, c * d + 7
    a, b = np.ones(n, dtype=bool), np.ones(n, dtype=bool)
    _tfd_reject_matches(first, d, k, a)
    rows = np.flatnonzero(first >= 0)
    _tfd_reject_graph(first, d, k, b, [rows[rows // d == c] for c in np.unique(rows // d)])
    assert np.array_equal(a, b) and a.sum() < n - 300


def test_host_graph_step_replays_cpython_and_networkx():
    """tsc_host_graph_step (tscode_amd/csrc/host_order.hpp: CPython's set / tuple-hash / dict orders and networkx's traversals
    re-played on plain arrays) against the reference's own expression on the real objects: the package's start-up checks (small
    graphs, and the one beyond 50 000 entries where CPython grows its tables by 2x instead of 4x), and whole passes of
    _tfd_reject_matches through both paths."""
    pytest.importorskip("networkx")
    import tscode_amd.numba_functions as nf
    assert nf._host_graph_step_ok() and nf._host_graph_step_ok(big=True)
    rng = np.random.default_rng(17)
    for n, d, k, density in ((5000, 10, 500, 0.5), (5000, 250, 20, 0.7), (30000, 30000, 1, 0.6), (12000, 6000, 2, 0.3), (999, 7, 142, 0.9)):
        first = np.full(n, -1, dtype=np.int32)
        for c in range(k):
            lo = c * d
            hi = n if c == k - 1 else lo + d
            for i in range(lo, hi - 1):
                if rng.random() < density:
                    first[i] = rng.integers(i + 1, min(hi, i + 1 + int(rng.choice([1, 2, 5, 50, hi]))))
        a, b = np.ones(n, dtype=bool), np.ones(n, dtype=bool)
        nf._tfd_reject_matches(first, d, k, a)                      # library path (the checks above passed)
        saved = dict(nf._HOST_GRAPH_STEP)
        nf._HOST_GRAPH_STEP.update(small=False, big=False)
        try:
            nf._tfd_reject_matches(first, d, k, b)                  # Python path: the real set / networkx objects
        finally:
            nf._HOST_GRAPH_STEP.clear()
            nf._HOST_GRAPH_STEP.update(saved)
        assert np.array_equal(a, b), (n, d, k, int(a.sum()), int(b.sum()))
        assert a.sum() < n


def test_cluster_heads_shortcut_matches_networkx():
    """The member of a cluster that survives TFD / MOI pruning is tuple(subgraph.nodes)[0] (numba_functions.py:209-214): the
    shortcut that avoids one subgraph view per component must give the same member as networkx itself, on graphs built the
    way the pruning builds them (a set of (i, j) tuples) -- and the package's own start-up check must have accepted it here."""
    import networkx as nx
    from tscode_amd.numba_functions import _cluster_heads_fast, _cluster_heads_reference, _fast_cluster_heads_ok
    assert _fast_cluster_heads_ok()
    rng = np.random.default_rng(11)
    for n, m in ((3, 2), (30, 12), (500, 700), (3000, 2500), (3000, 20000)):
        matches = set()
        for a, b in rng.integers(0, n, size=(m, 2)).tolist():
            if a != b:
                matches.add((min(a, b), max(a, b)))
        g = nx.Graph(matches)
        ref = {frozenset(mem): h for mem, h in _cluster_heads_reference(g)}
        assert {frozenset(mem): h for mem, h in _cluster_heads_fast(g)} == ref


# ----------------------------------------------------------------------------- G10-G12: the embed loops
def test_g10_embed_helpers(oracle):
    """rotation_matrix_from_vectors (row a4), polygonize, cartesian_product, get_embed, rotate_dihedral against the reference's
    own outputs: the oracle AND the product's host-side mirrors (tscode_amd.algebra / utils are plain NumPy)."""
    import tscode_amd.algebra as alg
    import tscode_amd.utils as ut
    g = load_golden("G10_embed_helpers")
    for a, b, ref in zip(g["rmfv_v1"], g["rmfv_v2"], g["rmfv_out"]):
        assert np.abs(oracle.rotation_matrix_from_vectors(a, b) - ref).max() < 1e-12
        assert np.abs(alg.rotation_matrix_from_vectors(a, b) - ref).max() < 1e-12
    assert np.array_equal(g["rmfv_out"][0], np.eye(3))                                         # parallel: the exact-zero branch
    assert np.abs(g["rmfv_out"][1] - np.diag([-1.0, -1.0, 1.0])).max() < 1e-15                  # antiparallel: 180 degrees about z
    for ln, ref in list(zip(g["poly2_lengths"], g["poly2_out"])) + list(zip(g["poly3_lengths"], g["poly3_out"])):
        assert np.abs(oracle.polygonize(ln) - ref).max() < 1e-14 and np.abs(ut.polygonize(ln) - ref).max() < 1e-14
    for ln, raises in zip(g["poly3_bad"], g["poly3_bad_raises"]):
        assert raises
        with pytest.raises(ValueError):
            oracle.polygonize(ln)
        with pytest.raises(ut.TriangleError):
            ut.polygonize(ln)
    for i in range(int(g["cp_n"])):
        sizes = g[f"cp_sizes{i}"]
        assert np.array_equal(oracle.cartesian_product(*sizes), g[f"cp_out{i}"])
        assert np.array_equal(ut.cartesian_product(*[np.arange(s) for s in sizes]), g[f"cp_out{i}"])
    for k in range(int(g["ge_n"])):
        nm = int(g[f"ge{k}_n_mols"])
        frags = [g[f"ge{k}_coords{m}"] for m in range(nm)]
        rot = np.array([g[f"ge{k}_rot{m}"] for m in range(nm)])[None]
        pos = np.array([g[f"ge{k}_pos{m}"] for m in range(nm)])[None]
        out = oracle.transform_batch(frags, g[f"ge{k}_conf_ids"][None], rot, pos)[0]
        assert np.abs(out - g[f"ge{k}_out"]).max() < 1e-13
    for a, ref in zip(g["rd_angles"], g["rd_out_mask"]):
        assert np.abs(oracle.rotate_dihedral(g["rd_coords"], g["rd_dihedral"], float(a), g["rd_mask"]) - ref).max() < 1e-12
    moved = np.zeros(len(g["rd_coords"]), dtype=bool)
    moved[g["rd_moved"]] = True
    first = np.zeros(len(g["rd_coords"]), dtype=bool)
    first[g["rd_dihedral"][0]] = True
    for a, ref_i, ref_f in zip(g["rd_angles"], g["rd_out_indices"], g["rd_out_first"]):
        assert np.abs(oracle.rotate_dihedral(g["rd_coords"], g["rd_dihedral"], float(a), moved) - ref_i).max() < 1e-12
        assert np.abs(oracle.rotate_dihedral(g["rd_coords"], g["rd_dihedral"], float(a), first) - ref_f).max() < 1e-12


def _string_case(g, k):
    return dict(coords1=g[f"coords0_{k}"], coords2=g[f"coords1_{k}"], centers1=g[f"centers0_{k}"], orb_vecs1=g[f"orb_vecs0_{k}"],
                centers2=g[f"centers1_{k}"], orb_vecs2=g[f"orb_vecs1_{k}"], angles=g[f"angles_{k}"])


def test_g11_string_embed_c1(oracle):
    """BASELINE config C1: the reference's string_embed on its own CH3Cl + HCOOH (and variants): every candidate pose, every
    clash verdict, every fingerprint and the kept set, from the oracle's restatement of embeds.py:91-120."""
    g = load_golden("G11_string_embed")
    assert int(g["n_cases"]) >= 4 and len(g["candidates_0"]) == 72 and g["atomnos0_0"].tolist() == [6, 1, 1, 1, 17]
    for k in range(int(g["n_cases"])):
        cands, ok, kept, margin = oracle.string_embed(**_string_case(g, k), clash_thresh=float(g[f"clash_thresh_{k}"]),
                                                      quadruplets=g[f"quadruplets_{k}"], return_margin=True)
        assert cands.shape == g[f"candidates_{k}"].shape and np.abs(cands - g[f"candidates_{k}"]).max() < 1e-9
        assert oracle.clash_margin(g[f"candidates_{k}"], g[f"ids_{k}"], float(g[f"clash_thresh_{k}"])) > 1e-9
        assert np.array_equal(ok, g[f"clash_ok_{k}"])
        fp = oracle.torsion_fingerprints(g[f"candidates_{k}"][g[f"fp_index_{k}"]], g[f"quadruplets_{k}"])
        # Poses one 10-degree step apart differ by 10 degrees in one torsion: their sums sit ON is_new_structure's threshold of
        # 10 and `sum < 10` is decided by the float32 roundings of the angles.  The sum itself is exact (a few float32 values
        # added in float64), so verdicts are reproducible iff the fingerprints are bit-identical: required here, with the
        # distance of every angle from a float32 rounding tie as the guard band.
        assert np.array_equal(fp, g[f"fingerprints_{k}"]) and margin >= 0.0
        assert oracle.torsion_rounding_margin(g[f"candidates_{k}"][g[f"fp_index_{k}"]], g[f"quadruplets_{k}"]) > 1e-12
        assert np.array_equal(kept, g[f"kept_{k}"]), (k, kept.sum(), g[f"kept_{k}"].sum())
        assert np.abs(cands[kept] - g[f"poses_{k}"]).max() < 1e-9


def _cyclical_case(g, k):
    coords = [g[f"coords{m}_{k}"] for m in range(2)]
    reactive = [g[f"reactive_indices{m}_{k}"] for m in range(2)]
    pivots = [[(g[f"pivot_vec{m}_{c}_{k}"], g[f"pivot_mean{m}_{c}_{k}"], g[f"pivot_cumnums{m}_{c}_{k}"]) for c in range(len(coords[m]))] for m in range(2)]
    return coords, reactive, pivots


def test_g12_cyclical_embed(oracle):
    """The reference's cyclical_embed (rigid shortcut and general loop, tests/cyclical.txt's molecules and variants) against
    the oracle's restatement: candidates, groups, clash verdicts, greedy similarity verdicts, constrained indices."""
    g = load_golden("G12_cyclical_embed")
    for k in range(int(g["n_cases"])):
        coords, reactive, pivots = _cyclical_case(g, k)
        cands, group_of, ok, kept, gids = oracle.cyclical_embed(coords, reactive, pivots, g[f"angles_{k}"], float(g[f"clash_thresh_{k}"]),
                                                                rigid_shortcut=bool(g[f"rigid_{k}"]))
        assert cands.shape == g[f"candidates_{k}"].shape and np.abs(cands - g[f"candidates_{k}"]).max() < 1e-9
        assert np.array_equal(group_of, g[f"group_of_{k}"]) and np.array_equal(gids, g[f"group_ids_{k}"])
        assert oracle.clash_margin(g[f"candidates_{k}"], g[f"ids_{k}"], float(g[f"clash_thresh_{k}"])) > 1e-9
        assert np.array_equal(ok, g[f"clash_ok_{k}"])
        assert np.array_equal(kept, g[f"kept_{k}"]), (k, kept.sum(), g[f"kept_{k}"].sum())
        assert np.array_equal(gids[group_of[kept]], g[f"constrained_indices_{k}"])


def test_g15_rotate_dihedral_fractional_angles(oracle):
    """The reference's rotate_dihedral (tscode/utils.py:389-414) with fractional angles, as tscode/torsion_module.py:984-1005 calls it:
    single rotations (mask given / first atom only) and the rotate - look - rotate back trail of the correction search."""
    g = load_golden("G15_rotate_dihedral_fractional")
    first = np.zeros(len(g["coords"]), np.uint8)
    first[g["dihedral"][0]] = 1
    for a, rm, rf in zip(g["angles"], g["out_mask"], g["out_first"]):
        assert np.abs(oracle.rotate_dihedral(g["coords"], g["dihedral"], float(a), g["mask"]) - rm).max() < 1e-12
        assert np.abs(oracle.rotate_dihedral(g["coords"], g["dihedral"], float(a), first) - rf).max() < 1e-12
    seq, q = g["coords"].copy(), 0
    for a in g["angles"][:4]:
        for sign in (1.0, -1.0):
            seq = oracle.rotate_dihedral(seq, g["dihedral"], sign * float(a), g["mask"])
            assert np.abs(seq - g["trail"][q]).max() < 1e-12
            q += 1


def _cyclical_case3(g, k):
    coords = [g[f"coords{m}_{k}"] for m in range(3)]
    reactive = [g[f"reactive_indices{m}_{k}"] for m in range(3)]
    pivots = [[(g[f"pivot_vec{m}_{c}_{k}"], g[f"pivot_mean{m}_{c}_{k}"], g[f"pivot_cumnums{m}_{c}_{k}"]) for c in range(len(coords[m]))] for m in range(3)]
    cumnums = [g[f"reactive_cumnums{m}_{k}"] for m in range(3)]
    return coords, reactive, pivots, cumnums


def test_g18_cyclical_embed_trimolecular(oracle):
    """The reference's cyclical_embed for THREE molecules (tests/trimolecular.txt's molecules and variants), run with `vec_angle` padding the
    2-vectors of _get_directions with z = 0 -- as shipped that call is undefined (embeds.py:297-299 / algebra.py:87), so this fixture is
    "reference patched", unpinned where the reference is undefined -- against the oracle's restatement of _get_directions,
    _adjust_directions (the carry-over of `directions` from orientation to orientation, conformer 0 for the reactive atoms) and the pose loop."""
    g = load_golden("G18_cyclical_embed_trimolecular")
    for k in range(int(g["n_cases"])):
        coords, reactive, pivots, cumnums = _cyclical_case3(g, k)
        cands, group_of, ok, kept, gids = oracle.cyclical_embed3(coords, reactive, pivots, cumnums, g[f"angles_{k}"], float(g[f"clash_thresh_{k}"]))
        assert np.array_equal(group_of, g[f"group_of_{k}"]) and np.array_equal(gids, g[f"group_ids_{k}"])
        assert cands.shape == g[f"candidates_{k}"].shape and np.abs(cands - g[f"candidates_{k}"]).max() < 1e-9
        assert oracle.clash_margin(g[f"candidates_{k}"], g[f"ids_{k}"], float(g[f"clash_thresh_{k}"])) > 1e-9
        assert np.array_equal(ok, g[f"clash_ok_{k}"])
        assert np.array_equal(kept, g[f"kept_{k}"]), (k, kept.sum(), g[f"kept_{k}"].sum())
        assert np.abs(cands[kept] - g[f"poses_{k}"]).max() < 1e-9
        assert np.array_equal(gids[group_of[kept]], g[f"constrained_indices_{k}"])
