"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Bars (BASELINE.json north_star): surviving-index set and clash masks bit-exact; RMSD / max-deviation /
distance values within 1e-5 abs in fp64 (they agree to ~1e-12; the asserted tolerance is 1e-9 so that
a regression shows long before the stated bar).
"""

import numpy as np
import pytest

from conftest import LARGE_PRUNE_CASES, load_golden, load_large_prune

pytestmark = pytest.mark.gpu

VAL_TOL = 1e-9          # asserted; the stated bar is 1e-5 abs


@pytest.fixture(scope="module")
def eng():
    import tscode_amd
    return tscode_amd.get_engine(0)


# ----------------------------------------------------------------------------- K3 values
def test_rmsd_pairs_golden(eng):
    g = load_golden("G1_rmsd_and_max")
    for h in (3, 5, 9, 18, 30, 60, 120):
        p, q, ref = g[f"p_{h}"], g[f"q_{h}"], g[f"out_{h}"]
        n = len(p)
        pairs = np.stack([np.arange(n), np.arange(n) + n], axis=1)
        r, m = eng.rmsd_pairs(np.concatenate([p, q]), pairs)
        assert np.abs(r - ref[:, 0]).max() < VAL_TOL, (h, np.abs(r - ref[:, 0]).max())
        assert np.abs(m - ref[:, 1]).max() < VAL_TOL, (h, g[f"tags_{h}"][np.abs(m - ref[:, 1]).argmax()])


def test_rmsd_pairs_vs_oracle_random(eng, oracle):
    from tscode_amd.synthetic import make_config
    ens = make_config("C2", 400)
    heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
    rng = np.random.default_rng(0)
    pairs = rng.integers(0, len(heavy), size=(5000, 2))
    r, m = eng.rmsd_pairs(heavy, pairs)
    ro, mo = oracle.rmsd_pairs(heavy, pairs)
    assert np.abs(r - ro).max() < VAL_TOL and np.abs(m - mo).max() < VAL_TOL


def test_dropin_rmsd_functions(eng, oracle):
    import tscode_amd
    g = load_golden("G1_rmsd_and_max")
    p, q, ref = g["p_18"][10], g["q_18"][10], g["out_18"][10]
    r, m = tscode_amd.rmsd_and_max_numba(p, q)
    assert abs(r - ref[0]) < VAL_TOL and abs(m - ref[1]) < VAL_TOL
    g4 = load_golden("G4_rmsd_similarity")
    poses, acc = g4["poses1"], g4["accepted1"]
    kept = []
    for pose, a in zip(poses, acc):                       # embeds.py:715 greedy use, thr=1
        new = not tscode_amd._rmsd_similarity(pose, kept, rmsd_thr=1)
        assert new == bool(a)
        if new:
            kept.append(pose)


# ----------------------------------------------------------------------------- K2
def test_all_dists_golden(eng):
    g = load_golden("G2_clash")
    for k in range(int(g["ad_n"])):
        out = eng.all_dists(g[f"ad_a{k}"], g[f"ad_b{k}"])
        assert np.abs(out - g[f"ad_out{k}"]).max() < 1e-12


def test_clash_mask_golden(eng):
    g = load_golden("G2_clash")
    combos = g["cc_combos"]
    for k in range(int(g["cc_n"])):
        ids, coords = g[f"cc_ids{k}"], g[f"cc_coords{k}"]
        for ci, (thresh, mc) in enumerate(combos):
            mask, counts = eng.clash_mask(coords, ids, float(thresh), int(mc), return_counts=True)
            assert np.array_equal(mask, g[f"cc_out{k}"][ci].astype(bool)), (k, thresh, mc)
            assert np.array_equal(counts, g[f"cc_counts{k}"][0 if thresh == 1.4 else 1].sum(axis=1))


def test_count_clashes_golden(eng):
    import tscode_amd
    g = load_golden("G2_clash")
    for k in range(int(g["cl_n"])):
        c = g[f"cl_coords{k}"]
        assert tscode_amd.count_clashes(c) == int(g[f"cl_count{k}"])
        got = [tscode_amd.compenetration_check(c, max_clashes=mc) for mc in (0, 2, 10)]
        assert got == g[f"cl_check{k}"].tolist()


@pytest.mark.parametrize("cfg,n", [("C2", 10_000), ("C3", 20_000), ("C5", 3_000)])
def test_clash_mask_vs_oracle(eng, oracle, cfg, n):
    from tscode_amd.synthetic import make_config
    ens = make_config(cfg, n)
    poses = ens.poses()
    assert oracle.clash_margin(poses, ens.ids, 1.5) > 1e-9            # guard band
    for mc in (0, 2):
        mask, counts = eng.clash_mask(poses, ens.ids, 1.5, mc, return_counts=True)
        ref = oracle.compenetration_mask(poses, ens.ids, 1.5, mc)
        assert np.array_equal(mask, ref)
    assert 0.02 < mask.mean() < 0.98
    # drop-in single-pose call
    import tscode_amd
    for s in (0, 1, 2):
        assert tscode_amd.compenetration_check(poses[s], ens.ids, 1.5, 0) == oracle.compenetration_check(poses[s], ens.ids, 1.5, 0)


def test_clash_edge_shapes(eng, oracle):
    rng = np.random.default_rng(3)
    for ids in ((1, 1), (1, 70), (70, 1), (64, 65), (3, 2, 1), (1, 1, 1), (40, 90, 70)):
        n = sum(ids)
        coords = rng.normal(size=(37, n, 3)) * 2.5
        for mc in (0, 5):
            assert np.array_equal(eng.clash_mask(coords, ids, 1.5, mc), oracle.compenetration_mask(coords, ids, 1.5, mc)), ids
    coords = rng.normal(size=(9, 20, 3)) * 0.7
    assert np.array_equal(eng.clash_mask(coords, None, 0.5, 3), oracle.compenetration_mask(coords, None, 0.5, 3))
    assert eng.clash_mask(np.zeros((0, 5, 3)), (2, 3)).shape == (0,)


# ----------------------------------------------------------------------------- K1
def test_transform_batch_vs_oracle(eng, oracle):
    from tscode_amd import FragmentSet
    from tscode_amd.synthetic import make_config
    for cfg, n in (("C2", 3000), ("C5", 500)):
        ens = make_config(cfg, n)
        out = eng.transform_batch(FragmentSet(ens.frag_coords), ens.conf_idx, ens.rot, ens.pos)
        ref = oracle.transform_batch(ens.frag_coords, ens.conf_idx, ens.rot, ens.pos)
        assert out.shape == ref.shape and np.abs(out - ref).max() < 1e-12
        assert np.abs(out - ens.poses()).max() < 1e-12
    # several conformers per fragment, conformer picked per pose
    rng = np.random.default_rng(1)
    frags = [rng.normal(size=(4, 7, 3)), rng.normal(size=(3, 5, 3))]
    ci = np.stack([rng.integers(0, 4, 200), rng.integers(0, 3, 200)], axis=1).astype(np.int32)
    from tscode_amd.synthetic import quat_to_mat
    rot = quat_to_mat(rng.normal(size=(200, 2, 4)))
    pos = rng.normal(size=(200, 2, 3))
    out = eng.transform_batch(FragmentSet(frags), ci, rot, pos)
    assert np.abs(out - oracle.transform_batch(frags, ci, rot, pos)).max() < 1e-12


def test_dropin_embed_functions(eng):
    import tscode_amd
    g = load_golden("G5_rotations")
    out = tscode_amd.transform_coords(g["tc_coords"], g["tc_rot"], g["tc_pos"])
    assert np.abs(out - g["transform_coords"]).max() < 1e-13

    class Mol:                                           # duck-typed Hypermolecule (hypermolecule_class.py:171-184)
        def __init__(self, coords, rot, pos):
            self.atomcoords, self.rotation, self.position = coords, rot, pos
    rng = np.random.default_rng(2)
    m1 = Mol(rng.normal(size=(2, 6, 3)), np.eye(3), np.zeros(3))
    m2 = Mol(rng.normal(size=(3, 4, 3)), g["tc_rot"], np.array([1.0, 2.0, 3.0]))
    got = tscode_amd.get_embed((m1, m2), (1, 2))
    ref = np.concatenate([(m.rotation @ m.atomcoords[c].T).T + m.position for m, c in zip((m1, m2), (1, 2))])
    assert np.abs(got - ref).max() < 1e-13
    # string-embed pose parameters: reactive centres coincide (embeds.py:114)
    poses, rot, pos = tscode_amd.string_embed_poses(m1.atomcoords[0], m2.atomcoords[0], m1.atomcoords[0][2], m2.atomcoords[0][1],
                                                    np.array([0.3, -1.0, 0.2]), np.array([1.0, 0.1, -0.4]), range(0, 360, 10))
    assert poses.shape == (36, 10, 3)
    assert np.abs(poses[:, 6 + 1] - m1.atomcoords[0][2]).max() < 1e-12


# ----------------------------------------------------------------------------- prune
@pytest.mark.parametrize("spread, scale", [(0.05, 1.0), (0.4, 1.0), (0.4, 37.0), (3.0, 0.02), (0.0, 1.0)])
def test_matrix_core_screen_values_and_limit(eng, spread, scale):
    """Level 1 of the descriptor screen as v_mfma_f32_16x16x16_f16 computes it (csrc/mm.hpp: components rounded to float16, norms folded
    into the product): every value against the exact squared distance of the stored fp32 descriptors, within the error the limit allows
    for, and the property everything rests on -- no pair whose exact distance is within h thr^2 lies above the kernel's limit."""
    import ctypes as C

    from tscode_amd import _lib
    rng = np.random.default_rng(int(spread * 100) + int(scale * 7))
    n = 1000
    centres = rng.normal(size=(40, 16)) * 4.0
    D = (centres[rng.integers(0, 40, n)] + rng.normal(size=(n, 16)) * spread) * scale
    D = D.astype(np.float32)
    limit = 30 * 0.5 ** 2 * scale ** 2        # h thr^2 in the descriptors' units
    S = np.empty((2, 64, n), dtype=np.float32)
    lim_bits, sigma = C.c_int32(), C.c_float()
    _lib.check(eng.lib.tsc_screen_mm_values(eng._h, _lib.ptr(D), n, C.c_double(limit), _lib.ptr(S), C.byref(lim_bits), C.byref(sigma)))
    sg = float(sigma.value)
    dmax = float(np.abs(D).max())
    assert dmax == 0.0 or 32.0 <= dmax * sg < 64.0
    D64 = D.astype(np.float64)
    exact = np.stack([((D64[:64, None, f::2] - D64[None, :, f::2]) ** 2).sum(axis=2) for f in (0, 1)]) * sg * sg     # [fam][r][c]
    M = dmax * sg
    # what the limit allows for besides the descriptors' own storage rounding: both operands rounded to float16 (2^-11 of M' each, per
    # component), the norm pieces, the accumulation at 2^-20 of the term sum
    err = np.abs(S.astype(np.float64) - exact)
    delta = np.sqrt(8.0) * 2 * (2.0 ** -11 * M + 2.0 ** -25)
    allowed = 2 * np.sqrt(exact) * delta + delta ** 2 + 2 * 2.0 ** -14 + 2.0 ** -20 * 4 * 8 * 64.0 * 64.0
    assert (err <= allowed).all(), float((err / allowed).max())
    lim = np.array([lim_bits.value], dtype=np.int32).view(np.float32)[0]
    worst = np.maximum(S[0], S[1])
    inside = np.maximum(exact[0], exact[1]) <= limit * sg * sg
    assert (worst[inside].view(np.int32) <= lim_bits.value).all()
    assert float(lim) <= (np.sqrt(limit) * sg + 2 * np.sqrt(8.0) * 2 * 2.0 ** -11 * M * 1.01 + 1e-3) ** 2 + 0.3, (float(lim), limit * sg * sg)
    if spread == 0.4 and scale == 1.0:
        assert inside.sum() > 500 and (~inside).sum() > 10000    # (the case has pairs on both sides)


def test_prune_golden(eng, algo):
    import tscode_amd
    g = load_golden("G3_prune")
    for c in range(int(g["n_cases"])):
        structures, atomnos, thr = g[f"structures{c}"], g[f"atomnos{c}"], float(g[f"thr{c}"])
        pruned, mask = tscode_amd.prune_conformers_rmsd(structures, atomnos, thr)
        assert np.array_equal(mask, g[f"mask{c}"]), f"case {c}: {mask.sum()} vs {g[f'mask{c}'].sum()}"
        assert np.array_equal(pruned, structures[g[f"mask{c}"]])
        stats = tscode_amd.last_prune_stats()
        assert [s["k"] for s in stats] == g[f"ks{c}"].tolist()
        assert [s["n_active_after"] for s in stats] == g[f"pass_masks{c}"].sum(axis=1).tolist()
        assert np.cumsum([s["new_keys"] for s in stats]).tolist() == g[f"pass_nkeys{c}"].tolist()


SIEVE_TRIM_DEFAULT = 1   # the library's default of option sieve_trim; the fixture `algo` also runs the other setting


@pytest.fixture(params=[(0, 1, False, 1), (1, 1, False, 1), (2, 0, False, 1), (2, 0, True, 1), (2, 0, False, 0), (0, 1, False, 2), (2, 0, False, 3),
                        (2, 0, False, 4), (2, 0, False, 5), (2, 0, False, 6), (0, 1, False, 7), (2, 0, False, 8), (2, 0, False, 9), (2, 0, False, 10)],
                ids=["algo-auto", "algo-tile", "algo-sieve-global", "algo-sieve-other-screen", "algo-sieve-separate-apply", "algo-auto-ranks-from-memory",
                     "algo-sieve-culled", "algo-sieve-f32-stage1", "algo-sieve-culled-f32-stage1", "algo-sieve-vector-screen", "algo-auto-vector-screen-f32-stage1",
                     "algo-sieve-culled-vector-screen", "algo-sieve-16-row-matrix-core", "algo-sieve-16-row-matrix-core-separate-apply"])
def algo(request, eng):
    """Runs a test once per route through the prune: automatic choice (descriptor sieve whose pair kernel applies the verdicts
    tile by tile; passes with short chunks in the chunk-local kernel), register-tiled all-pairs (its passes are applied by
    k_apply_pass), descriptor sieve with every pass through the global path, the same with the screen's other instruction
    sequence (option sieve_trim flipped from its default), the same with k_apply_pass as a launch of its own (what a
    multi-rank pass does), the automatic choice with k_open_rows reading the scan-block prefix from memory (the path of
    ensembles beyond 4 M structures), the sieve with every pass of fewer than 64 chunks culled (sorted layout + bounding boxes:
    what the large passes of C4 / C5 run by default), and the last two again with stage 1 of the pair kernels reading the float32 copy
    of the coordinates (what runs of 128 MB of heavy atoms and more do by default).  Since round 5 the walked passes of the sieve
    screen on the matrix cores in runs of 150 000 structures and more (mm.hpp, option sieve_mm = 1, the default; 2 = always, what the
    fixture sets): every sieve route above takes that kernel -- fused and with
    its own apply launch, walked and culled (cull_mm.hpp), with stage 1 in float64 and on the float32 copy; the two "16-row" routes leave the choice to the library, which gives
    ensembles of this size the matrix-core screen on 16-row items (k_rmsd_sieve_mm16: what C3 runs); "other-screen" and the three
    "vector-screen" routes switch both off and run the packed-fp32 screen of sieve.hpp (what row tiles dealt to several ranks still use)."""
    eng.set_option("prune_algo", request.param[0])
    eng.set_option("sieve_mm", 0 if (request.param[2] or request.param[3] in (6, 7, 8)) else (1 if request.param[3] in (9, 10) else 2))
    eng.set_option("sieve_mm16", 0 if (request.param[2] or request.param[3] in (6, 7, 8)) else 1)
    eng.set_option("local_pass", request.param[1])
    if request.param[2]:
        eng.set_option("sieve_trim", 1 - SIEVE_TRIM_DEFAULT)
    eng.set_option("fused_apply", 1 if request.param[3] not in (0, 10) else 0)
    if request.param[3] == 2:
        eng.set_option("open_lds_blocks", 0)
    if request.param[3] in (3, 5, 8):      # every pass of fewer than 64 chunks laid out along the Morton curve, tile pairs skipped by bounding box (cull.hpp)
        eng.set_option("cull_min_pairs", 0)
        eng.set_option("cull", 2)
    if request.param[3] in (4, 5, 7):      # H of stage 1 from the float32 copy, its own rounding bound (sieve.hpp: pair_stage1)
        eng.set_option("stage1_f32", 2)
    yield request.param[0]
    eng.set_option("sieve_mm", 1)
    eng.set_option("sieve_mm16", 1)
    eng.set_option("stage1_f32", 1)
    eng.set_option("cull_min_pairs", 2.0e9)
    eng.set_option("cull", 1)
    eng.set_option("prune_algo", 0)
    eng.set_option("local_pass", 1)
    eng.set_option("sieve_trim", SIEVE_TRIM_DEFAULT)
    eng.set_option("fused_apply", 1)
    eng.set_option("open_lds_blocks", 2 ** 30)


@pytest.mark.parametrize("name", LARGE_PRUNE_CASES)
def test_prune_large_golden(eng, algo, name):
    """G16a / G16b / G17: the reference's own run of prune_conformers_rmsd on 40 023 / 41 999 / 104 999 structures -- the passes
    k = 5000 (G17), 2000, 1000, 500, 200 of the headline's schedule and their cache-key collisions with the coarse passes
    (rmsd_pruning.py:65-67, 131-144, 186-204) -- on every route through the prune: final mask, which passes ran, the active
    count after every pass, cumulative key counts."""
    import tscode_amd
    fx = load_large_prune(name)
    g = fx.g
    pruned, mask = tscode_amd.prune_conformers_rmsd(fx.structures, fx.atomnos, fx.thr)
    assert np.array_equal(np.packbits(mask), g["mask_bits"]), (int(mask.sum()), int(np.unpackbits(g["mask_bits"])[:fx.n].sum()))
    assert np.array_equal(pruned, fx.structures[mask])
    stats = tscode_amd.last_prune_stats()
    assert [s["k"] for s in stats] == g["ks"].tolist()
    assert [s["n_active_after"] for s in stats] == [int(np.unpackbits(b)[:fx.n].sum()) for b in g["pass_mask_bits"]]
    assert np.cumsum([s["new_keys"] for s in stats]).tolist() == g["pass_nkeys"].tolist()
    assert all(s["algo"] in (1, 2, 3) and (algo == 0 or s["algo"] == algo) for s in stats)


@pytest.mark.parametrize("mm", [1, 2], ids=["automatic-16-row-matrix-core", "64-row-matrix-core"])
@pytest.mark.parametrize("name", LARGE_PRUNE_CASES)
def test_prune_large_golden_every_pass_mask(eng, name, mm):
    """The same runs pass by pass (the stepping API): the mask after EVERY pass against the mask the reference's own
    _similarity_mask_rmsd_group returned for it -- with the kernel runs of this size take (the matrix-core screen on 16-row items) and with
    the 64-row matrix-core kernels of large runs (csrc/mm.hpp; sieve_mm = 2 forces them)."""
    import ctypes as C

    from tscode_amd import _lib
    eng.set_option("sieve_mm", mm)
    fx = load_large_prune(name)
    g = fx.g
    lib = eng.lib
    d_heavy = C.c_void_p()
    _lib.check(lib.tsc_malloc(eng._h, fx.heavy.nbytes, C.byref(d_heavy)))
    _lib.check(lib.tsc_memcpy_h2d(eng._h, d_heavy, _lib.ptr(fx.heavy), fx.heavy.nbytes))
    st = eng.prune_stepper(d_heavy.value, fx.n, fx.heavy.shape[1], fx.thr, 0)
    after = {}
    try:
        while True:
            k = st.next_pass()
            if k == 0:
                break
            st.pass_local(0, 1)
            st.pass_finish()
            m = np.empty(fx.n, dtype=np.uint8)
            _lib.check(lib.tsc_memcpy_d2h(eng._h, _lib.ptr(m), C.c_void_p(st.mask_ptr()), m.nbytes))
            after[int(k)] = np.packbits(m.astype(bool))
        ran = [s["k"] for s in st.stats()]
    finally:
        st.close()
        _lib.check(lib.tsc_free(eng._h, d_heavy))
        eng.set_option("sieve_mm", 1)
    assert ran == g["ks"].tolist()
    for k, bits in zip(ran, g["pass_mask_bits"]):
        assert np.array_equal(after[k], bits), f"{name}: the mask after the k = {k} pass differs from the reference's"
    assert np.array_equal(after[1], g["mask_bits"])


@pytest.mark.parametrize("mode", [0, 1])
def test_prune_c2_vs_oracle(eng, oracle, mode, algo):
    from tscode_amd.synthetic import make_config
    ens = make_config("C2")
    poses = ens.poses()
    poses = poses[oracle.compenetration_mask(poses, ens.ids, 1.5, 0)]
    heavy = np.ascontiguousarray(poses[:, ens.atomnos != 1])
    mr, mm = oracle.prune_margins(heavy, 0.5, mode)
    assert mr > 1e-6 and mm > 1e-6, "guard band violated: re-draw the ensemble (SURVEY 8d)"
    ref = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True)
    mask, stats = eng.prune_heavy(heavy, 0.5, mode)
    assert np.array_equal(mask, ref["mask"]), (mask.sum(), ref["mask"].sum())
    assert [s["k"] for s in stats] == [s["k"] for s in ref["stats"]]
    for s, r in zip(stats, ref["stats"]):
        assert s["n_active_after"] == r["n_active_after"]
        assert s["pairs_evaluated"] == r["pairs_evaluated"]      # the reference's sequential work, reproduced exactly
        assert s["new_keys"] == r["new_keys"]
        assert max(s["pairs_computed"], s["pairs_screened"]) >= s["pairs_evaluated"]   # the GPU looks at a superset
        assert s["algo"] in (1, 2, 3) and (algo == 0 or s["algo"] == algo)      # 3 = chunk-local kernel (automatic choice only)
    print(f"C2 mode {mode}: {len(heavy)} -> {mask.sum()}; margins rmsd {mr:.2e} maxdev {mm:.2e}")


def test_prune_edge_cases(eng, oracle, algo):
    rng = np.random.default_rng(7)
    cases = []
    one = rng.normal(size=(1, 5, 3)) * 3
    cases.append(one)                                                   # N = 1
    cases.append(np.concatenate([one, one]))                            # N = 2 identical
    cases.append(np.repeat(rng.normal(size=(1, 12, 3)) * 3, 300, axis=0))    # all identical: every row finds j = i+1
    cases.append(rng.normal(size=(19, 1, 3)) * 3)                       # h = 1
    cases.append(rng.normal(size=(450, 32, 3)) * 3)                     # h = 32 (largest single-tile size), nothing similar
    base = rng.normal(size=(50, 7, 3)) * 3
    cases.append((base[:, None] + rng.normal(size=(50, 9, 7, 3)) * 0.02).reshape(-1, 7, 3))   # clusters in order, N = 450
    cases.append(np.concatenate([base] * 13)[rng.permutation(650)])     # exact duplicates, shuffled, N = 650 (k = 20.. passes)
    for heavy in cases:
        heavy = np.ascontiguousarray(heavy)
        for mode in (0, 1):
            ref = oracle.prune_heavy(heavy, 0.5, mode=mode)
            mask, stats = eng.prune_heavy(heavy, 0.5, mode)
            assert np.array_equal(mask, ref["mask"]), (heavy.shape, mode, mask.sum(), ref["mask"].sum())
            assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
    import tscode_amd
    assert tscode_amd.prune_conformers_rmsd(np.zeros((0, 4, 3)), np.array([6, 6, 1, 1]))[1].shape == (0,)
    if algo == 1:
        with pytest.raises(Exception):
            eng.prune_heavy(np.zeros((10, 33, 3)))                       # register-tiled kernel: h > 32 refused loudly


def test_prune_many_heavy_atoms(eng, oracle):
    """More than 32 heavy atoms (BASELINE config 5 has 120): the sieve kernel takes any size."""
    from tscode_amd.synthetic import make_config
    ens = make_config("C5", 2500)
    poses = ens.poses()
    poses = poses[oracle.compenetration_mask(poses, ens.ids, 1.5, 0)]
    heavy = np.ascontiguousarray(poses[:, ens.atomnos != 1])
    assert heavy.shape[1] == 120
    for mode in (0, 1):
        ref = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True)
        mask, stats = eng.prune_heavy(heavy, 0.5, mode)
        assert np.array_equal(mask, ref["mask"]), (mode, mask.sum(), ref["mask"].sum())
        assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
    rng = np.random.default_rng(11)
    for h in (33, 47, 64, 257):
        base = rng.normal(size=(40, h, 3)) * 4
        heavy = np.ascontiguousarray((base[:, None] + rng.normal(size=(40, 6, h, 3)) * 0.03).reshape(-1, h, 3)[rng.permutation(240)])
        ref = oracle.prune_heavy(heavy, 0.5, mode=0)
        mask, _ = eng.prune_heavy(heavy, 0.5, 0)
        assert np.array_equal(mask, ref["mask"]), h


@pytest.mark.parametrize("mm", [1, 2], ids=["automatic-16-row-matrix-core", "64-row-matrix-core"])
def test_prune_sharded_rows_equal_single(eng, oracle, mm):
    """The stepping API with the row tiles of each pass dealt to 3 'ranks' (run one after the other on
    this GPU, merging through the atomicMin target) gives the same mask as the one-shot call -- also with the matrix-core
    kernel, which deals groups of 64 rows (sieve_mm = 2)."""
    import ctypes as C

    eng.set_option("sieve_mm", mm)

    from tscode_amd import _lib
    from tscode_amd.synthetic import make_config
    ens = make_config("C2", 6000)
    heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
    ref_mask, ref_stats = eng.prune_heavy(heavy, 0.5, 0)
    lib = eng.lib
    d_heavy = C.c_void_p()
    _lib.check(lib.tsc_malloc(eng._h, heavy.nbytes, C.byref(d_heavy)))
    _lib.check(lib.tsc_memcpy_h2d(eng._h, d_heavy, _lib.ptr(heavy), heavy.nbytes))
    # three steppers (one per rank) over the same heavy array; after each pass the best[]
    # arrays are min-merged on the host and written back, as the all-reduce(MIN) would do.
    steppers = [eng.prune_stepper(d_heavy.value, len(heavy), heavy.shape[1], 0.5, 0) for _ in range(3)]
    while True:
        ks_now = [s.next_pass() for s in steppers]
        assert len(set(ks_now)) == 1
        if ks_now[0] == 0:
            break
        bests = []
        for rank, s in enumerate(steppers):
            s.pass_local(rank, 3)
            p, n = s.best_ptr()
            b = np.empty(n, dtype=np.int32)
            _lib.check(lib.tsc_memcpy_d2h(eng._h, _lib.ptr(b), C.c_void_p(p), b.nbytes))
            bests.append(b)
        merged = np.minimum.reduce(bests)
        for s in steppers:
            p, n = s.best_ptr()
            _lib.check(lib.tsc_memcpy_h2d(eng._h, C.c_void_p(p), _lib.ptr(merged), merged.nbytes))
            s.pass_finish()
    masks = []
    for s in steppers:
        m = np.empty(len(heavy), dtype=np.uint8)
        _lib.check(lib.tsc_memcpy_d2h(eng._h, _lib.ptr(m), C.c_void_p(s.mask_ptr()), m.nbytes))
        masks.append(m.astype(bool))
        assert [x["n_active_after"] for x in s.stats()] == [x["n_active_after"] for x in ref_stats]
        s.close()
    for m in masks:
        assert np.array_equal(m, ref_mask)
    # ONE stepper standing in for all three ranks: rank 0 opens the pass, the others' rows follow (tsc_prune_pass_rows)
    s = eng.prune_stepper(d_heavy.value, len(heavy), heavy.shape[1], 0.5, 0)
    while s.next_pass() != 0:
        s.pass_local(0, 3)
        s.pass_rows(2, 3)
        s.pass_rows(1, 3)
        s.pass_rows(2, 3)                                 # a share twice: the same minimum
        s.pass_finish()
    m = np.empty(len(heavy), dtype=np.uint8)
    _lib.check(lib.tsc_memcpy_d2h(eng._h, _lib.ptr(m), C.c_void_p(s.mask_ptr()), m.nbytes))
    assert np.array_equal(m.astype(bool), ref_mask)
    assert [x["pairs_evaluated"] for x in s.stats()] == [x["pairs_evaluated"] for x in ref_stats]
    s.close()
    _lib.check(lib.tsc_free(eng._h, d_heavy))
    eng.set_option("sieve_mm", 1)


def test_non_finite_coordinates_have_a_defined_outcome(eng, oracle):
    """NaN / infinite coordinates: the reference's np.linalg.svd raises LinAlgError on such a pair (rmsd_pruning.py:19).
    The library neither hangs nor faults: a non-finite structure is similar to nothing (every comparison of :75 with a NaN is false)
    and is kept, every finite structure gets the verdict it gets from the oracle on the same input, the run reports that it met
    one (tsc_pass_stats.nonfinite_input) -- and the Python drop-in turns that report into the reference's exception.  The clash
    mask treats a NaN distance as 'not below the threshold', exactly as count(D < thresh) does (numba_functions.py:77-103)."""
    import tscode_amd
    from tscode_amd.synthetic import make_config
    ens = make_config("C2", 6000)
    poses = ens.poses()
    heavy = np.ascontiguousarray(poses[:, ens.atomnos != 1])
    clean_mask, clean_stats = eng.prune_heavy(heavy, 0.5, 0)
    assert all(s["nonfinite_input"] == 0 for s in clean_stats)
    bad = heavy.copy()
    bad[17, 3, 1] = np.nan
    bad[2500] = np.inf
    bad[5999, 0, 0] = -np.inf
    for mode in (0, 1):
        ref = oracle.prune_heavy(bad, 0.5, mode=mode)
        for local_pass in (1, 0):
            eng.set_option("local_pass", local_pass)
            try:
                mask, stats = eng.prune_heavy(bad, 0.5, mode)
            finally:
                eng.set_option("local_pass", 1)
            assert mask[[17, 2500, 5999]].all() and np.array_equal(mask, ref["mask"]), (mode, local_pass)
            assert all(s["nonfinite_input"] == 1 for s in stats)
            assert [s["n_active_after"] for s in stats] == [s["n_active_after"] for s in ref["stats"]]
    all_atoms = poses.copy()
    all_atoms[17, np.flatnonzero(ens.atomnos != 1)[3], 1] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        tscode_amd.prune_conformers_rmsd(all_atoms, ens.atomnos, 0.5)
    h_only = poses.copy()
    h_only[17, np.flatnonzero(ens.atomnos == 1)[0], 1] = np.nan            # a hydrogen: never enters the prune (:178-179)
    kept, m = tscode_amd.prune_conformers_rmsd(h_only, ens.atomnos, 0.5)
    assert np.array_equal(m, clean_mask)
    # clash mask
    cp = poses[:3000].copy()
    cp[5, 2, 0] = np.nan
    cp[77] = np.inf
    cp[1234, -1, 2] = -np.inf
    for max_clashes in (0, 2):
        got = tscode_amd.compenetration_mask(cp, ens.ids, 1.5, max_clashes)
        assert np.array_equal(got, oracle.compenetration_mask(cp, ens.ids, 1.5, max_clashes)), max_clashes


@pytest.mark.parametrize("mm", [1, 2], ids=["automatic", "64-row-matrix-core"])
@pytest.mark.parametrize("world,n_poses,tile_block,det", [(3, 20_000, 256, 1), (8, 30_000, 256, 1), (5, 9_000, 16, 1), (4, 12_000, 1, 1), (3, 20_000, 256, 0)])
def test_culled_row_tiles_dealt_to_emulated_ranks(eng, oracle, world, n_poses, tile_block, det, mm):
    """Passes sharded by ROW TILES with every pass culled (sorted layout + bounding boxes, cull.hpp): `world` prune runs over one array stand
    in for the ranks, rank r takes its runs of `cull_tile_block` consecutive tiles of the sorted layout, best[] is min-merged with torch as the
    all-reduce(MIN) would.  Each unordered pair of a pass is visited by exactly one rank -- the layouts of the ranks must be bit-identical for
    that -- so every rank's mask and active counts must be the oracle's.  det = 0: runs created WITHOUT "deterministic_basis" (each with an
    atomic-sum basis of its own: their layouts differ) -- the library must then not cull a pass dealt by row tiles: it walks
    the pass in index order on every rank, and the results are the oracle's all the same."""
    import torch

    from tscode_amd.synthetic import make_config
    ens = make_config("C2", n_poses)
    heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
    ref = oracle.prune_heavy(heavy, 0.5, mode=0, row_parallel=True)
    dev = torch.device("cuda:0")
    d_heavy = torch.from_numpy(heavy).to(dev)
    eng.set_option("cull_min_pairs", 0)
    eng.set_option("cull", 2)
    eng.set_option("local_pass", 0)
    eng.set_option("cull_tile_block", tile_block)
    eng.set_option("deterministic_basis", det)
    eng.set_option("sieve_mm", mm)      # (2: the matrix-core kernels, which deal groups of 64 rows -- runs of tile_block / 4 of them)
    try:
        sts = [eng.prune_stepper(d_heavy, len(heavy), heavy.shape[1], 0.5, 0) for _ in range(world)]
        bests = [torch.empty(len(heavy), dtype=torch.int32, device=dev) for _ in range(world)]
        for s, b in zip(sts, bests):
            s.use_best_buffer(b)
        while True:
            ks = {s.next_pass() for s in sts}
            assert len(ks) == 1
            if ks.pop() == 0:
                break
            for r, s in enumerate(sts):
                s.pass_local(r, world)
            eng.synchronize()
            merged = torch.stack(bests).amin(0)
            for b in bests:
                b.copy_(merged)
            torch.cuda.synchronize()
            for s in sts:
                s.pass_finish()
        keep = torch.empty(len(heavy), dtype=torch.uint8, device=dev)
        for s in sts:
            s.copy_mask(keep)
            eng.synchronize()
            assert np.array_equal(keep.cpu().numpy().astype(bool), ref["mask"])
            assert [x["n_active_after"] for x in s.stats()] == [x["n_active_after"] for x in ref["stats"]]
            s.close()
    finally:
        eng.set_option("cull_min_pairs", 2.0e9)
        eng.set_option("cull", 1)
        eng.set_option("local_pass", 1)
        eng.set_option("cull_tile_block", 256)
        eng.set_option("deterministic_basis", 0)
        eng.set_option("sieve_mm", 1)


@pytest.mark.parametrize("world,min_chunks,n_poses,mode,cull", [(2, 4, 12_000, 0, 0), (3, 1, 12_000, 0, 0), (8, 4, 40_000, 0, 0), (3, 4, 9_000, 1, 0), (5, 2, 700, 0, 0),
                                                                (2, 4, 12_000, 0, 1), (3, 1, 20_000, 0, 1), (4, 2, 9_000, 1, 1)])
def test_partitioned_passes_emulated_ranks(eng, oracle, world, min_chunks, n_poses, mode, cull):
    """Rank-partitioned passes (tsc_prune_pass_range / tsc_prune_pass_merge): `world` prune runs over the same heavy-atom array stand
    in for the ranks, each takes the chunks that start inside its block of the structure axis, and the all-reduce(SUM) of the
    exchange buffers is done here with torch.  Every rank must end with the oracle's mask, and the per-pass statistics -- active
    counts and the reference's own pair-evaluation counts -- must be the single-run ones on every rank."""
    import torch

    from tscode_amd.engine import PruneStepper
    from tscode_amd.synthetic import make_config
    ens = make_config("C2", n_poses)
    heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
    ref = oracle.prune_heavy(heavy, 0.5, mode=mode)
    one_mask, one_stats = eng.prune_heavy(heavy, 0.5, mode)
    assert np.array_equal(one_mask, ref["mask"])
    if cull:        # every pass of fewer than 64 chunks that does not fit the chunk-local kernel: sorted layout + bounding boxes inside each rank's chunks
        eng.set_option("cull", 2)
        eng.set_option("cull_min_pairs", 0)
    try:
        _partitioned_emulation(eng, torch, PruneStepper, heavy, ref, one_stats, world, min_chunks, mode)
    finally:
        eng.set_option("cull", 1)
        eng.set_option("cull_min_pairs", 2.0e9)


def _partitioned_emulation(eng, torch, PruneStepper, heavy, ref, one_stats, world, min_chunks, mode):
    dev = torch.device("cuda:0")
    d_heavy = torch.from_numpy(heavy).to(dev)
    n, h = heavy.shape[0], heavy.shape[1]
    words = PruneStepper.exchange_words(eng.lib, n, mode)
    steppers, exch = [], []
    for r in range(world):
        st = eng.prune_stepper(d_heavy, n, h, 0.5, mode)
        ex = torch.zeros(words, dtype=torch.int64, device=dev)
        st.set_partition(r, world, min_chunks, ex)
        steppers.append(st), exch.append(ex)
    per_pass = n // 64 + 48
    n_part = n_views = 0
    while True:
        ks = {st.next_pass() for st in steppers}
        assert len(ks) == 1
        k = ks.pop()
        if k == 0:
            break
        flags = {st.pass_partitioned() for st in steppers}
        assert len(flags) == 1 and flags.pop() == (k >= min_chunks * world)
        if k >= min_chunks * world:
            for st in steppers:
                st.pass_range()
            eng.synchronize()
            total = torch.stack([e[:per_pass] for e in exch]).sum(0)
            # the ranks' removed rows are disjoint: no bit may be set twice
            for a in range(world):
                for b in range(a + 1, world):
                    assert not bool((exch[a][:per_pass - 8] & exch[b][:per_pass - 8]).any()), (k, a, b)
            for e, st in zip(exch, steppers):
                e[:per_pass].copy_(total)
                torch.cuda.synchronize()
                st.pass_merge()
            n_part += 1
            continue
        ranges = {st.views_range() for st in steppers}
        assert len(ranges) == 1
        off, w = ranges.pop()
        if w:
            torch.cuda.synchronize()
            eng.synchronize()
            total = torch.stack([e[off:off + w] for e in exch]).sum(0)
            for e in exch:
                e[off:off + w].copy_(total)
            torch.cuda.synchronize()
            n_views += 1
        for st in steppers:
            st.views_merged()
            st.pass_local(0, 1)         # every rank runs the remaining passes whole (their sharding by row tiles is tested elsewhere)
            st.pass_finish()
    assert n_part >= 1 and n_views == (1 if mode == 0 else 0)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    for r, st in enumerate(steppers):
        st.copy_mask(keep)
        stats = st.stats()
        assert np.array_equal(keep.cpu().numpy().astype(bool), ref["mask"]), f"rank {r}"
        for a, b, c in zip(stats, one_stats, ref["stats"]):
            assert (a["k"], a["n_active_before"], a["n_active_after"]) == (b["k"], b["n_active_before"], b["n_active_after"]), (r, a, b)
            assert a["pairs_evaluated"] == c["pairs_evaluated"] == b["pairs_evaluated"], (r, a["k"], a["pairs_evaluated"], c["pairs_evaluated"])
            assert a["new_keys"] == b["new_keys"]
        assert len(stats) == len(one_stats)
        st.close()


def test_pipeline_c2_vs_oracle(eng, oracle):
    import torch

    from tscode_amd import FragmentSet
    from tscode_amd.synthetic import make_config
    ens = make_config("C2")
    fs = FragmentSet(ens.frag_coords)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_frags, d_ci, d_rot, d_pos = t(fs.flat), t(ens.conf_idx), t(ens.rot), t(ens.pos)
    n, na = ens.n_poses, ens.n_atoms
    clash = torch.empty(n, dtype=torch.uint8, device=dev)
    structures = torch.empty((n, na, 3), dtype=torch.float64, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    heavy_idx = np.flatnonzero(ens.atomnos != 1).astype(np.int32)
    torch.cuda.synchronize()
    res = eng.pipeline_dev(fs, d_frags, d_ci, d_rot, d_pos, n, heavy_idx, 1.5, 0, 0.5, 0, clash, structures, keep)
    poses = ens.poses()
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    assert np.array_equal(clash.cpu().numpy().astype(bool), cm)
    assert res["n_pass"] == cm.sum()
    got = structures[:res["n_pass"]].cpu().numpy()
    assert np.abs(got - poses[cm]).max() < 1e-12
    _, mask = oracle.prune_conformers_rmsd(poses[cm], ens.atomnos, 0.5)
    assert np.array_equal(keep[:res["n_pass"]].cpu().numpy().astype(bool), mask)
    assert res["n_keep"] == mask.sum()
    print("C2 pipeline:", res["n_pass"], "pass the clash check,", res["n_keep"], "survive;", res["ms"])
    # the descriptor basis taken from the filtered structures (on the main stream) instead of the unfiltered sample built
    # beside the clash kernel: a different basis, the same verdicts
    keep2 = torch.empty(n, dtype=torch.uint8, device=dev)
    for opt in ("early_basis", "fuse_descriptors"):     # (the latter: descriptors by k_descriptors instead of the embedding kernel)
        eng.set_option(opt, 0)
        res2 = eng.pipeline_dev(fs, d_frags, d_ci, d_rot, d_pos, n, heavy_idx, 1.5, 0, 0.5, 0, clash, structures, keep2)
        eng.set_option(opt, 1)
        assert res2["n_keep"] == res["n_keep"] and torch.equal(keep2[:res["n_pass"]], keep[:res["n_pass"]]), opt
        assert np.abs(structures[:res["n_pass"]].cpu().numpy() - poses[cm]).max() < 1e-12
    # every pose clashes: the side stream's work is joined although no prune follows
    res3 = eng.pipeline_dev(fs, d_frags, d_ci, d_rot, d_pos, n, heavy_idx, 50.0, 0, 0.5, 0, clash, structures, keep2)
    assert res3["n_pass"] == 0 and res3["n_keep"] == 0


def test_dropin_prune_gathers_heavy_atoms_on_the_device(eng, oracle):
    """tscode_amd.prune_conformers_rmsd on C2's survivors of the clash check (7 290 all-atom structures: above the size from which
    the heavy-atom gather of rmsd_pruning.py:178-179 runs on the device) against the oracle; the same call on a
    non-contiguous view and on float32 input (the host-gather branch) returns the same."""
    import tscode_amd
    from tscode_amd.synthetic import make_config
    ens = make_config("C2")
    poses = ens.poses()
    poses = poses[oracle.compenetration_mask(poses, ens.ids, 1.5, 0)]
    _, want = oracle.prune_conformers_rmsd(poses, ens.atomnos, 0.5)
    kept, mask = tscode_amd.prune_conformers_rmsd(poses, ens.atomnos, 0.5)
    assert np.array_equal(mask, want) and np.array_equal(kept, poses[want])
    assert [s["pairs_evaluated"] for s in tscode_amd.rmsd_pruning.last_prune_stats()]
    wide = np.zeros((len(poses), poses.shape[1] + 3, 3))
    wide[:, :poses.shape[1]] = poses
    view = wide[:, :poses.shape[1]]                       # not C-contiguous
    assert not view.flags.c_contiguous
    assert np.array_equal(tscode_amd.prune_conformers_rmsd(view, ens.atomnos, 0.5)[1], want)
    small = poses[:1500]                                  # below the size threshold: host gather
    assert np.array_equal(tscode_amd.prune_conformers_rmsd(small, ens.atomnos, 0.5)[1], oracle.prune_conformers_rmsd(small, ens.atomnos, 0.5)[1])


# ----------------------------------------------------------------------------- full size, by properties
def test_full_size_properties_c3(eng):
    """BASELINE config 3 (100k x 50) at full size, checked through size-independent properties:
    determinism, mode-1 idempotence (survivors of the cache-free prune are pairwise dissimilar, so pruning
    them again keeps all of them) and survivor-set nesting (every pass only removes)."""
    import torch

    from tscode_amd import FragmentSet
    from tscode_amd.synthetic import make_config
    ens = make_config("C3")
    fs = FragmentSet(ens.frag_coords)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_frags, d_ci, d_rot, d_pos = t(fs.flat), t(ens.conf_idx), t(ens.rot), t(ens.pos)
    n, na = ens.n_poses, ens.n_atoms
    heavy_idx = np.flatnonzero(ens.atomnos != 1).astype(np.int32)
    out = {}
    for mode in (0, 1):
        runs = []
        for rep in range(2):
            clash = torch.empty(n, dtype=torch.uint8, device=dev)
            structures = torch.empty((n, na, 3), dtype=torch.float64, device=dev)
            keep = torch.zeros(n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            res = eng.pipeline_dev(fs, d_frags, d_ci, d_rot, d_pos, n, heavy_idx, 1.5, 0, 0.5, mode, clash, structures, keep)
            runs.append((clash.cpu().numpy(), keep.cpu().numpy(), res))
        assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])   # deterministic
        res = runs[0][2]
        assert res["n_keep"] == int(runs[0][1][:res["n_pass"]].sum())
        acts = [s["n_active_after"] for s in res["stats"]]
        assert all(a >= b for a, b in zip([res["n_pass"]] + acts, acts))
        out[mode] = (structures[:res["n_pass"]], keep[:res["n_pass"]].cpu().numpy().astype(bool), res)
        print(f"C3 mode {mode}: pass {res['n_pass']}, keep {res['n_keep']}, ms {res['ms']}")
    # idempotence of the cache-free prune
    structures, keep1, res1 = out[1]
    surv = structures[torch.from_numpy(np.flatnonzero(keep1)).to(dev)][:, torch.from_numpy(heavy_idx.astype(np.int64)).to(dev)].contiguous()
    mask2 = torch.empty(len(surv), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.prune_heavy_dev(surv, len(surv), len(heavy_idx), 0.5, 1, mask2)
    eng.synchronize()
    assert int(mask2.sum().item()) == len(surv)
    assert out[1][2]["n_keep"] < out[0][2]["n_keep"]          # the reference-exact mode keeps more


@pytest.mark.parametrize("cfg", ["C3", "C5", "C4"])
def test_full_size_against_the_recorded_oracle(eng, cfg):
    """BASELINE configs 3, 5 and 4 at FULL size (100k x 50, 500k x 200, 1M x 50) against what the CPU oracle gave on the same
    seeded inputs (tests/golden/expected_full.json, generated by gen_expected_full.py / tools/record_c4.py: 50 s, 8 min and
    48 min of CPU): clash mask, survivor mask (SHA-256 of the packed bits), and the reference's pair-evaluation count of
    every pass."""
    import hashlib
    import json
    import os

    import torch

    from tscode_amd.pipeline import DevicePipeline
    from tscode_amd.synthetic import make_config
    ens = make_config(cfg)
    exp = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expected_full.json")))[f"{cfg}:{ens.n_poses}:mode0"]
    pipe = DevicePipeline(ens, device_index=0, mode=0)
    res = pipe.step()
    torch.cuda.synchronize()
    sha = lambda bits: hashlib.sha256(np.packbits(bits.astype(bool)).tobytes()).hexdigest()[:16]
    assert (res["n_pass"], res["n_keep"]) == (exp["n_pass"], exp["n_keep"])
    assert sha(pipe.d_clash.cpu().numpy()) == exp["clash_sha256_16"]
    assert sha(pipe.h_keep[:res["n_pass"]].numpy()) == exp["keep_sha256_16"]
    assert [s["pairs_evaluated"] for s in res["stats"]] == [p["pairs_evaluated"] for p in exp["passes"]]
    assert [s["n_active_after"] for s in res["stats"]] == [p["active_after"] for p in exp["passes"]]
    del pipe
    torch.cuda.empty_cache()


def test_concurrent_processes_share_the_gpu():
    """TSCoDe's multiembed runs the path from several processes at once.  Three processes step the C2 pipeline
    1000 times each on this one GPU; every step of every process must give the first step's survivor mask and evaluation
    counts, which equal the recorded oracle result (tools/soak.py).  Contention reorders workgroups: this is the test that
    found the chunk-local kernel reading a mask that workgroups of the same launch were already clearing."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "soak.py"), "C2", "1000"], cwd=root, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for _ in range(3)]
    for pr in procs:
        out, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, (out[-500:], err[-1500:])
        assert "1000 steps, 0 differ" in out and "recorded oracle result: True" in out, out[-500:]


class _SoloDist:
    """torch.distributed stand-in for a world of one rank: collectives are identities."""

    class ReduceOp:
        SUM, MIN = "sum", "min"

    @staticmethod
    def get_backend(group=None):
        return "solo"

    @staticmethod
    def all_reduce(t, op=None, group=None):
        return None

    @staticmethod
    def all_gather_into_tensor(out, inp, group=None):
        out[:inp.numel()].copy_(inp)


def test_shard_backend_world1_equals_one_shot(eng, oracle):
    """The GPU backend of the multi-rank protocol (HipShardBackend + sharded_step), driven with a single
    rank, gives the verdicts of the one-call pipeline and of the oracle."""
    from tscode_amd.pipeline import HipShardBackend, sharded_step
    from tscode_amd.synthetic import make_config
    ens = make_config("C2", 5000)
    be = HipShardBackend(ens, 0, 0, 1, 1.5, 0, 0.5, 0)
    res = sharded_step(be, 0, 1, _SoloDist)
    be.torch.cuda.synchronize()
    poses = ens.poses()
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    _, mask = oracle.prune_conformers_rmsd(poses[cm], ens.atomnos, 0.5)
    assert res["n_pass"] == cm.sum() and res["n_keep"] == mask.sum()
    assert np.array_equal(be.keep[:res["n_pass"]].cpu().numpy().astype(bool), mask)
    assert np.abs(be.structures[:res["n_pass"]].cpu().numpy() - poses[cm]).max() < 1e-12


def test_pipelines_of_both_kinds_interleave_in_one_process(eng, oracle):
    """A sharded pipeline built FIRST, then a one-call pipeline, then steps of both in turn with drop-in calls (the shared
    get_engine() context) in between: every pipeline owns its library context and stream, so nobody can switch the stream
    another one's kernels and collectives are ordered on."""
    import tscode_amd
    from tscode_amd.pipeline import DevicePipeline
    from tscode_amd.synthetic import make_config
    ens = make_config("C2", 6000)
    poses = ens.poses()
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    _, mask = oracle.prune_conformers_rmsd(poses[cm], ens.atomnos, 0.5)
    import torch.distributed as dist
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        sharded = DevicePipeline(ens, device_index=0, rank=0, world=1, mode=0, force_sharded=True)
        single = DevicePipeline(ens, device_index=0, mode=0)
        assert sharded.engine is not single.engine and single.engine is not eng
        for it in range(6):
            a = sharded.step()
            kept, m2 = tscode_amd.prune_conformers_rmsd(poses[cm][:1500], ens.atomnos, 0.5)       # a drop-in call on the shared context
            b = single.step()
            sharded.torch.cuda.synchronize()
            assert a["n_pass"] == b["n_pass"] == cm.sum() and a["n_keep"] == b["n_keep"] == mask.sum(), it
            assert np.array_equal(sharded.h_keep[:a["n_pass"]].numpy().astype(bool), mask)
            assert np.array_equal(single.h_keep[:b["n_pass"]].numpy().astype(bool), mask)
    finally:
        if created:
            dist.destroy_process_group()


def test_greedy_group_filter(eng, oracle):
    """the per-group greedy _rmsd_similarity filter of the embed loops, against G4 and the oracle."""
    import tscode_amd
    g = load_golden("G4_rmsd_similarity")
    for c in range(int(g["n_cases"])):
        poses, ref = g[f"poses{c}"], g[f"accepted{c}"]
        assert np.array_equal(tscode_amd.filter_angular_groups(poses, [len(poses)], 1.0), ref)
    # many ragged groups at once, against the oracle
    from tscode_amd.synthetic import make_ensemble
    ens = make_ensemble(3000, (10, 12, 9), seed=77, children=6, sigma_t=0.3, sigma_rot_deg=8.0)
    poses = ens.poses()
    rng = np.random.default_rng(5)
    sizes = []
    while sum(sizes) < len(poses):
        sizes.append(int(min(rng.integers(1, 217), len(poses) - sum(sizes))))
    got = tscode_amd.filter_angular_groups(poses, sizes, 1.0)
    off = np.concatenate([[0], np.cumsum(sizes)])
    want = np.concatenate([oracle.greedy_group_filter(poses[off[i]:off[i + 1]], 1.0) for i in range(len(sizes))])
    assert np.array_equal(got, want)
    assert 0.05 < got.mean() < 0.95
    assert tscode_amd.filter_angular_groups(np.zeros((0, 4, 3)), [], 1.0).shape == (0,)


def test_prune_random_small_ensembles(eng, oracle, algo):
    """Many small random ensembles (sizes that exercise 1-4 passes, ragged chunk remainders, 1..40 heavy atoms,
    loose and tight clusters, both modes): masks and the reference's pair-evaluation counts equal the oracle's."""
    rng = np.random.default_rng(2024)
    checked = 0
    for case in range(60):
        n = int(rng.choice([1, 2, 3, 19, 20, 21, 40, 41, 63, 64, 65, 100, 101, 127, 199, 200, 201, 257, 399, 400, 401, 450]))
        h = int(rng.integers(1, 41))
        if algo == 1 and h > 32:
            h = 32
        n_par = max(1, n // int(rng.integers(1, 9)))
        spread = float(rng.choice([0.01, 0.05, 0.15, 0.3]))
        base = rng.normal(size=(n_par, h, 3)) * float(rng.choice([1.0, 3.0, 8.0])) + rng.normal(size=(n_par, 1, 3)) * 2
        heavy = np.ascontiguousarray(base[rng.integers(0, n_par, size=n)] + rng.normal(size=(n, h, 3)) * spread)
        thr = float(rng.choice([0.25, 0.5, 1.0]))
        for mode in (0, 1):
            mr, mm = oracle.prune_margins(heavy, thr, mode)
            if min(mr, mm) < 1e-7:
                continue                      # a pair sits on a threshold: not a fair bit-exact case (guard band)
            ref = oracle.prune_heavy(heavy, thr, mode=mode)
            mask, stats = eng.prune_heavy(heavy, thr, mode)
            assert np.array_equal(mask, ref["mask"]), (case, n, h, thr, mode, int(mask.sum()), int(ref["mask"].sum()))
            assert [s["k"] for s in stats] == [s["k"] for s in ref["stats"]]
            assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
            checked += 1
    assert checked > 100


def test_prune_random_medium_ensembles(eng, oracle):
    """Three fixed cases of tests/soak_mm.py (random ensembles of a few thousand structures: several passes, chunks of every kind) through the
    64-row walked, the culled and the 16-row matrix-core kernels and the packed-fp32 one, stage 1 in float64 and float32: masks, the passes' k
    and the reference's pair-evaluation counts (rmsd_pruning.py:65-75) equal the oracle's.  The soak itself ran 38 such cases (profiles/r05_soak_mm.txt)."""
    import soak_mm
    rng = np.random.default_rng(77)
    checked = 0
    for _ in range(3):
        heavy, thr, what = soak_mm.make_case(rng, sizes=(1500, 2048, 3001, 6000))
        for mode in (0, 1):
            mr, mm = oracle.prune_margins(heavy, thr, mode)
            if min(mr, mm) < 1e-7:
                continue
            ref = oracle.prune_heavy(heavy, thr, mode=mode, row_parallel=True)
            res = soak_mm.run_routes(eng, heavy, thr, mode, ref)
            assert all(res.values()), (what, mode, res)
            checked += 1
    assert checked >= 4


def _blocks_of_near_duplicates(n_parents, children, h, seed):
    rng = np.random.default_rng(seed)
    parents = rng.normal(size=(n_parents, h, 3)) * 3.0
    return np.ascontiguousarray(np.repeat(parents, children, axis=0) + rng.normal(size=(n_parents * children, h, 3)) * 0.02)


_BIG_REF = {}


@pytest.mark.parametrize("lds_blocks", [2 ** 30, 128])
def test_prune_beyond_256_scan_blocks(eng, oracle, lds_blocks):
    """640 000 structures (313 scan blocks of 2048: more than one entry of the block prefix per thread of k_open_rows; with
    open_lds_blocks = 128 the prefix is read from memory instead of LDS), in runs of 200 near-duplicates so that the early
    passes remove almost everything and the oracle finishes in seconds.  Masks, schedule and evaluation counts equal the
    oracle's, in both modes."""
    heavy = _blocks_of_near_duplicates(3200, 200, 6, seed=77)
    eng.set_option("open_lds_blocks", lds_blocks)
    try:
        for mode in (0, 1):
            if mode not in _BIG_REF:              # (seconds of oracle time per mode: shared by the two parametrisations;
                # chunk-parallel: the first pass has 20 000 chunks, one parallel region each if rows were spread instead)
                _BIG_REF[mode] = oracle.prune_heavy(heavy, 0.5, mode=mode)
            ref = _BIG_REF[mode]
            mask, stats = eng.prune_heavy(heavy, 0.5, mode)
            assert np.array_equal(mask, ref["mask"]), (mode, int(mask.sum()), int(ref["mask"].sum()))
            assert [s["k"] for s in stats] == [s["k"] for s in ref["stats"]]
            assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
        assert int(mask.sum()) == 3200
    finally:
        eng.set_option("open_lds_blocks", 2 ** 30)


def test_prune_children_spread_around_the_threshold(eng, oracle, algo):
    """A mid-size ensemble whose children scatter AROUND the threshold (rotations of several degrees, 0.12 A shifts): many
    pairs the quartic tests cannot decide, long walks of the exact path, every large-pass code path of the pair kernels.
    Masks, pass schedule and the reference's pair-evaluation counts equal the oracle's (tools/stress_parity.py runs five
    such ensembles; this is its first)."""
    from tscode_amd.synthetic import make_ensemble
    ens = make_ensemble(12000, (25, 25), 7000, children=10, sigma_rot_deg=6.0, sigma_t=0.12, shell=(4.0, 9.0))
    poses = ens.poses()
    poses = poses[oracle.compenetration_mask(poses, ens.ids, 1.5, 0)]
    heavy = np.ascontiguousarray(poses[:, ens.atomnos != 1])
    for mode in (0, 1):
        mr, mm = oracle.prune_margins(heavy, 0.5, mode)
        assert min(mr, mm) > 1e-9, "a pair sits on a threshold: redraw the ensemble"
        ref = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True)
        mask, stats = eng.prune_heavy(heavy, 0.5, mode)
        assert np.array_equal(mask, ref["mask"]), (mode, int(mask.sum()), int(ref["mask"].sum()))
        assert [s["k"] for s in stats] == [s["k"] for s in ref["stats"]]
        assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
        assert sum(s["candidates"] for s in stats) > 100        # the explicit-rotation path really ran
    assert 0.05 < mask.mean() < 0.95


@pytest.mark.parametrize("offset", [100.0, 1000.0])
def test_prune_and_clash_far_from_the_origin(eng, oracle, offset, algo):
    """rmsd_and_max_numba never centres (rmsd_pruning.py:7-41) and prune_conformers_rmsd is also called on arbitrary xyz
    ensembles (operators.py:569, embedder.py:2027): structures 100 and 1000 A away from the origin.  Gp, Gq and the quartic's
    coefficients grow with the square of the offset, the descriptor norms with the offset, the clash kernel's fp32 band with the
    largest coordinate: the screens may decide less, the verdicts must stay the oracle's -- masks, pass schedule and the
    reference's own pair-evaluation counts."""
    from tscode_amd.synthetic import make_ensemble
    ens = make_ensemble(4000, (15, 15), 7100 + int(offset), children=8, sigma_rot_deg=2.0, sigma_t=0.08, shell=(4.0, 9.0))
    shift = np.array([0.6, -0.64, 0.48]) * offset
    poses = ens.poses() + shift
    # clash verdicts at a large coordinate scale (the packed-fp32 minimum's band is +-1e-3 A^2 at 1000 A)
    assert oracle.clash_margin(poses, ens.ids, 1.5) > 1e-9
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    assert np.array_equal(eng.clash_mask(poses, ens.ids, 1.5, 0), cm) and 0 < cm.sum() < len(cm)
    heavy = np.ascontiguousarray(poses[cm][:, ens.atomnos != 1])
    for mode in (0, 1):
        mr, mm = oracle.prune_margins(heavy, 0.5, mode)
        assert min(mr, mm) > 1e-7, "a pair sits on a threshold: redraw the ensemble"
        ref = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True)
        mask, stats = eng.prune_heavy(heavy, 0.5, mode)
        assert np.array_equal(mask, ref["mask"]), (offset, mode, int(mask.sum()), int(ref["mask"].sum()))
        assert [s["k"] for s in stats] == [s["k"] for s in ref["stats"]]
        assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
    assert 0.05 < ref["mask"].mean() < 0.95
    # Values of single pairs at that distance from the origin.  The rmsd is a stationary value of the rotation and agrees to
    # 1e-9 for every pair.  The maximum deviation is first order in the rotation: for two UNRELATED structures far from the
    # origin the optimum is nearly flat about the offset axis (the two largest eigenvalues of Horn's matrix, ~ h offset^2,
    # differ only by the structures' own extent), so any solver -- LAPACK in the reference, Jacobi in the oracle, the
    # quaternion path here -- returns that angle to about u h offset^2 / gap, times a lever arm of `offset`: 5e-9 was seen at
    # 100 A.  Pairs near the thresholds are near-duplicates (small rotation, wide gap): those are held to 1e-9.
    n_sub = 400
    ii, jj = np.triu_indices(n_sub, 1)
    pairs = np.stack([ii, jj], axis=1)
    r, m = eng.rmsd_pairs(heavy, pairs)
    ro, mo = oracle.rmsd_pairs(heavy, pairs)
    near = ro < 1.0
    assert near.sum() > 50
    assert np.abs(r - ro).max() < VAL_TOL and np.abs(m - mo)[near].max() < VAL_TOL
    assert np.abs(m - mo).max() < 1e-9 * (offset / 10.0) ** 2, np.abs(m - mo).max()        # 1e-7 at 100 A, 1e-5 (the stated bar) at 1000 A


@pytest.mark.parametrize("h", [512, 1024])
def test_prune_hundreds_of_heavy_atoms(eng, oracle, h):
    """h = 512 and 1024 heavy atoms per structure (the rounding bound of the quartic tests grows with h: rmsd.hpp,
    quartic_kappa; the descriptor build stages 12-24 KB per structure)."""
    rng = np.random.default_rng(h)
    n_par, n = 40, 320
    base = rng.normal(size=(n_par, h, 3)) * 6.0 + rng.normal(size=(n_par, 1, 3)) * 3
    which = rng.integers(0, n_par, size=n)
    spread = rng.choice([0.01, 0.03, 0.3], size=n)[:, None, None]
    heavy = np.ascontiguousarray(base[which] + rng.normal(size=(n, h, 3)) * spread)
    for mode in (0, 1):
        mr, mm = oracle.prune_margins(heavy, 0.5, mode)
        assert min(mr, mm) > 1e-7
        ref = oracle.prune_heavy(heavy, 0.5, mode=mode)
        mask, stats = eng.prune_heavy(heavy, 0.5, mode)
        assert np.array_equal(mask, ref["mask"]), (h, mode, int(mask.sum()), int(ref["mask"].sum()))
        assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
    assert n_par <= ref["mask"].sum() < n
    pairs = rng.integers(0, n, size=(500, 2))
    r, m = eng.rmsd_pairs(heavy, pairs)
    ro, mo = oracle.rmsd_pairs(heavy, pairs)
    assert np.abs(r - ro).max() < VAL_TOL and np.abs(m - mo).max() < VAL_TOL


def _pair_on_a_threshold(oracle, rng, h, target, which, delta):
    """Two structures whose rmsd (which = 0) or maxdev (which = 1) is `target + delta` to within 1e-13, found by bisection on
    the oracle's value: q = a rotation of p plus noise (rmsd) / plus one displaced atom (maxdev), scaled."""
    p = rng.normal(size=(h, 3)) * 2.5 + rng.normal(size=3)
    ang = rng.normal(size=3) * 0.02
    c, s_ = np.cos(ang), np.sin(ang)
    R = (np.array([[1, 0, 0], [0, c[0], -s_[0]], [0, s_[0], c[0]]]) @ np.array([[c[1], 0, s_[1]], [0, 1, 0], [-s_[1], 0, c[1]]])
         @ np.array([[c[2], -s_[2], 0], [s_[2], c[2], 0], [0, 0, 1]]))
    noise = rng.normal(size=(h, 3))
    if which == 1:
        noise *= 0.02
        noise[h // 3] = rng.normal(size=3) * 3.0          # one atom carries the maximum deviation, the rmsd stays small
    base = p @ R.T

    def val(t):
        return oracle.rmsd_and_max_numba(p, base + t * noise)[which]
    lo, hi = 0.0, 1.0
    while val(hi) < target + delta:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if val(mid) < target + delta:
            lo = mid
        else:
            hi = mid
        if hi - lo < 1e-16 * max(hi, 1.0):
            break
    q = base + hi * noise
    return p, q


def test_pairs_inside_the_guard_band(eng, oracle):
    """What happens INSIDE the band the bit-exact tests exclude: pairs built to sit 1e-9 .. 1e-7 from rmsd = thr and from
    maxdev = 2 thr, on both sides.  Three different eigen-solvers stand behind the compare (LAPACK gesdd in the reference,
    Jacobi in the oracle, Newton + adjugate quaternion on the GPU), so: (1) the values agree to 1e-9 (they agree to ~1e-13);
    (2) the prune's verdict on the two-structure ensemble {p, q} equals the oracle's wherever the oracle's value is at least
    1e-9 away from the threshold -- i.e. a verdict can only differ for a pair closer to a threshold than the asserted value
    tolerance; (3) how far the values really are apart is printed."""
    rng = np.random.default_rng(4242)
    thr = 0.5
    worst = 0.0
    n_checked = 0
    for h in (9, 30, 120):
        for which, target in ((0, thr), (1, 2 * thr)):
            for delta in (1e-7, -1e-7, 1e-8, -1e-8, 3e-9, -3e-9, 1e-9, -1e-9, 1e-10, -1e-10):
                for rep in range(3):
                    p, q = _pair_on_a_threshold(oracle, rng, h, target, which, delta)
                    heavy = np.ascontiguousarray(np.stack([p, q]))
                    ro, mo = oracle.rmsd_and_max_numba(p, q)
                    r, m = eng.rmsd_pairs(heavy, [[0, 1]])
                    assert abs(r[0] - ro) < VAL_TOL and abs(m[0] - mo) < VAL_TOL
                    worst = max(worst, abs(r[0] - ro), abs(m[0] - mo))
                    dist = min(abs(ro - thr), abs(mo - 2 * thr))          # the oracle's distance from the nearer threshold
                    assert dist < 2e-7
                    ref = oracle.prune_heavy(heavy, thr, mode=0)["mask"]
                    for algo_opt in (0, 1):
                        if algo_opt == 1 and h > 32:
                            continue
                        eng.set_option("prune_algo", algo_opt)
                        mask, _ = eng.prune_heavy(heavy, thr, 0)
                        eng.set_option("prune_algo", 0)
                        if dist >= 1e-9:
                            assert np.array_equal(mask, ref), (h, which, delta, ro, mo)
                            n_checked += 1
                        else:                                              # inside the value tolerance: either verdict is legitimate
                            assert mask[1] and (mask[0] or not mask[0])
    assert n_checked > 200
    print(f"largest |GPU - oracle| over the in-band pairs: {worst:.2e}")


def test_prune_when_descriptors_cannot_separate(eng, oracle):
    """Worst case for the sieve: two rigid bodies, A fixed and B rotated about the origin, with the atoms ordered so that
    every descriptor pair (a, a + h/2) lies inside one body.  Atom norms and those pair distances (both descriptor
    families) are then identical for all structures, nothing can be screened out, and every pair needs H.
    The verdicts must still be the oracle's."""
    from tscode_amd.synthetic import quat_to_mat
    rng = np.random.default_rng(99)
    A, B = rng.normal(size=(6, 3)) * 3, rng.normal(size=(6, 3)) * 3
    n_par = 60
    rots = quat_to_mat(rng.normal(size=(n_par, 4)))
    which = rng.integers(0, n_par, size=400)
    jitter = quat_to_mat(np.concatenate([np.ones((400, 1)), rng.normal(size=(400, 3)) * 0.004], axis=1))
    Bs = np.einsum("nij,njk,ak->nai", rots[which], jitter, B)            # B rotated about the origin: norms unchanged
    heavy = np.empty((400, 12, 3))
    heavy[:, [0, 1, 2, 6, 7, 8]] = A                                     # pairs (0,6) (1,7) (2,8) inside A
    heavy[:, [3, 4, 5, 9, 10, 11]] = Bs                                  # pairs (3,9) (4,10) (5,11) inside B
    heavy = np.ascontiguousarray(heavy)
    for mode in (0, 1):
        ref = oracle.prune_heavy(heavy, 0.5, mode=mode)
        mask, stats = eng.prune_heavy(heavy, 0.5, mode)
        assert np.array_equal(mask, ref["mask"])
        assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
        # nothing is screened out: H is formed for every pair the reference evaluates (the kernels may stop a row's OTHER candidates once
        # it has its first similar column: those are screened and never formed)
        assert sum(s["pairs_computed"] for s in stats) >= sum(s["pairs_evaluated"] for s in stats)
        assert sum(s["pairs_screened"] for s in stats) >= sum(s["pairs_evaluated"] for s in stats)
    assert 30 < ref["mask"].sum() < 400


def test_automatic_choice_takes_the_all_pairs_kernel_when_the_screen_separates_nothing(eng, oracle):
    """40 000 structures whose descriptors all coincide (synthetic.make_unscreenable): from the spread of the sample's descriptors
    the automatic choice (prune_algo 0) sees that the screen would let every pair through and runs the register-tiled all-pairs
    kernel instead -- in the cache-free mode, in a prune of its own (one synchronisation after the basis) and inside the pipeline (the
    spread arrives with the side chain).  The reference-exact mode and an ordinary ensemble keep the sieve.  Verdicts and evaluation counts equal the oracle's either way."""
    import torch

    from tscode_amd.pipeline import DevicePipeline
    from tscode_amd.synthetic import make_config, make_unscreenable
    heavy = make_unscreenable(40_000)
    for mode in (1, 0):
        ref = oracle.prune_heavy(heavy, 0.5, mode=mode, row_parallel=True)
        mask, stats = eng.prune_heavy(heavy, 0.5, mode)
        assert np.array_equal(mask, ref["mask"]), mode
        assert [s["pairs_evaluated"] for s in stats] == [s["pairs_evaluated"] for s in ref["stats"]]
        if mode == 1:
            assert {s["algo"] for s in stats} == {1}, [s["algo"] for s in stats]       # every pass by k_rmsd_tile
        else:                                                                           # reference-exact mode: the cache leaves few pairs, the sieve stays
            assert 1 not in {s["algo"] for s in stats}
    ens = make_config("C2", 40_000)                                                    # an ordinary ensemble: the sieve stays
    ordinary = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
    _, stats = eng.prune_heavy(ordinary, 0.5, 0)
    assert 1 not in {s["algo"] for s in stats}
    # inside the pipeline: fragment 0 is the 30-atom composite itself, with 4 000 such structures as its conformers (the poses pick one
    # each: many exact repeats), fragment 1 a lone hydrogen far away -- every pose passes the clash check, the heavy atoms of a pose
    # are one of the unscreenable structures
    from tscode_amd.synthetic import Ensemble
    n_poses = 40_000
    rng = np.random.default_rng(5)
    confs = make_unscreenable(4000, seed=7)
    ens = Ensemble(frag_coords=[confs, np.array([[[50.0, 0.0, 0.0]]])], conf_idx=np.stack([rng.integers(0, 4000, n_poses), np.zeros(n_poses, np.int64)], 1).astype(np.int32),
                   rot=np.ascontiguousarray(np.broadcast_to(np.eye(3), (n_poses, 2, 3, 3))), pos=np.zeros((n_poses, 2, 3)), ids=np.array([30, 1]),
                   atomnos=np.array([6] * 30 + [1]), seed=5)
    pipe = DevicePipeline(ens, device_index=0, mode=1)
    res = pipe.step()
    torch.cuda.synchronize()
    poses = ens.poses()
    ref = oracle.prune_heavy(np.ascontiguousarray(poses[:, :30]), 0.5, mode=1, row_parallel=True)
    assert res["n_pass"] == n_poses and np.array_equal(pipe.h_keep[:n_poses].numpy().astype(bool), ref["mask"])
    assert {s["algo"] for s in res["stats"]} == {1}
    assert [s["pairs_evaluated"] for s in res["stats"]] == [s["pairs_evaluated"] for s in ref["stats"]]


def test_clash_fp32_band_falls_back_to_fp64(eng, oracle):
    """Verdict-only clash masks use a packed-fp32 minimum with a rigorous band; poses whose closest inter-fragment
    distance sits within 1e-9 .. 1e-4 of the threshold must come out exactly like the fp64 reference."""
    rng = np.random.default_rng(77)
    n_a, n_b = 9, 7
    poses, expect = [], []
    for eps in (0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3):
        for rep in range(8):
            a = rng.normal(size=(n_a, 3)) * 2.0
            b = rng.normal(size=(n_b, 3)) * 2.0 + np.array([30.0, 0.0, 0.0])        # far away: no clash
            # move fragment b so that its atom 0 sits at distance 1.5 + eps from atom 3 of a, along a random direction
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            b = b - b[0] + a[3] + d * (1.5 + eps)
            shift = rng.normal(size=3) * 20.0                                         # large coordinates widen the band
            poses.append(np.concatenate([a, b]) + shift)
    poses = np.ascontiguousarray(np.array(poses))
    ids = np.array([n_a, n_b], dtype=np.int32)
    ref = oracle.compenetration_mask(poses, ids, 1.5, 0)
    for fp32 in (1, 0):
        eng.set_option("clash_fp32", fp32)
        got = eng.clash_mask(poses, ids, 1.5, 0)
        assert np.array_equal(got.astype(bool), ref.astype(bool)), fp32
    eng.set_option("clash_fp32", 1)
    assert 0 < ref.sum() < len(ref)


def test_fused_clash_one_pose_per_lane(eng, oracle):
    """k_clash_lanes (the fused embed + clash verdicts of two-fragment ensembles: one pose per lane, the smaller fragment in registers)
    against the oracle's embed + compenetration_check and against k_clash (option clash_lanes 0): fragment sizes on both sides of the
    register tiers (8 / 16 / 24 / 26 / 32 atoms; 33: the kernel does not take it), either fragment the smaller one, several conformers
    per fragment, pose counts that are no multiple of 64, poses whose closest distance sits 1e-12 .. 1e-3 from the threshold (the
    fp32 band's fp64 recount) and ensembles far from the origin (no usable band at all)."""
    import torch

    from tscode_amd.engine import FragmentSet
    rng = np.random.default_rng(4)
    dev = torch.device("cuda:0")

    def run(frag_coords, ci, rot, pos, lanes):
        fs = FragmentSet(frag_coords)
        eng.set_option("clash_lanes", lanes)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_mask = torch.zeros(len(ci), dtype=torch.uint8, device=dev)
        eng.embed_clash_mask_dev(fs, t(fs.flat), t(ci.astype(np.int32)), t(rot), t(pos), len(ci), 1.5, 0, d_mask)
        eng.synchronize()
        return d_mask.cpu().numpy().astype(bool)

    def rotations(n):
        q = rng.normal(size=(n, 4))
        from tscode_amd.synthetic import quat_to_mat
        return quat_to_mat(q)

    try:
        for n_a, n_b, n_conf, n_poses, offset in ((1, 1, 1, 1, 0.0), (2, 9, 1, 63, 0.0), (8, 7, 3, 65, 0.0), (9, 16, 2, 1000, 0.0), (17, 40, 1, 700, 0.0),
                                                  (24, 25, 1, 3000, 0.0), (25, 25, 4, 20000, 0.0), (26, 31, 1, 500, 0.0), (32, 33, 2, 1500, 0.0),
                                                  (33, 33, 1, 300, 0.0), (25, 25, 1, 2000, 1.0e5), (25, 25, 1, 500, 1.0e16)):
            fa = rng.normal(size=(n_conf, n_a, 3)) * 1.6
            fb = rng.normal(size=(n_conf, n_b, 3)) * 1.6
            ci = rng.integers(0, n_conf, size=(n_poses, 2))
            rot = np.stack([rotations(n_poses), rotations(n_poses)], axis=1)
            pos = np.stack([np.zeros((n_poses, 3)), rng.normal(size=(n_poses, 3)) * 3.5], axis=1) + offset
            ref_poses = oracle.transform_batch([fa, fb], ci.astype(np.int32), rot, pos)
            ref = oracle.compenetration_mask(ref_poses, np.array([n_a, n_b], np.int32), 1.5, 0).astype(bool)
            for lanes in (1, 0):
                got = run([fa, fb], ci, rot, pos, lanes)
                assert np.array_equal(got, ref), (n_a, n_b, n_conf, n_poses, offset, lanes, got.sum(), ref.sum())
            if offset == 0.0 and n_poses >= 500:
                assert 0 < ref.sum() < n_poses
        # inside the band: fragment b placed so that its atom 0 sits 1.5 + eps from atom 3 of a; everything else far away
        n_a, n_b = 25, 25
        fa = rng.normal(size=(1, n_a, 3)) * 2.0
        fb = rng.normal(size=(1, n_b, 3)) * 0.4 + np.array([9.0, 0.0, 0.0])      # a tight clump 9 A from its own atom 0
        fb[0, 0] = 0.0
        rot_l, pos_l = [], []
        for eps in (0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3):
            for rep in range(9):
                d = rng.normal(size=3)
                d /= np.linalg.norm(d)
                # rotate b so that its clump points AWAY from a (along d), then put its atom 0 at a[3] + d (1.5 + eps)
                from tscode_amd.algebra import rotation_matrix_from_vectors
                Rb = rotation_matrix_from_vectors(np.array([1.0, 0.0, 0.0]), d)
                shift = rng.normal(size=3) * 15.0
                rot_l.append(np.stack([np.eye(3), Rb]))
                pos_l.append(np.stack([shift, fa[0, 3] + d * (1.5 + eps) + shift]))
        rot, pos = np.array(rot_l), np.array(pos_l)
        ci = np.zeros((len(rot), 2), np.int64)
        ref_poses = oracle.transform_batch([fa, fb], ci.astype(np.int32), rot, pos)
        assert oracle.clash_margin(ref_poses, np.array([n_a, n_b], np.int32), 1.5) < 1e-9            # (this set is MEANT to sit on the threshold)
        ref = oracle.compenetration_mask(ref_poses, np.array([n_a, n_b], np.int32), 1.5, 0).astype(bool)
        for lanes in (1, 0):
            got = run([fa, fb], ci, rot, pos, lanes)
            # eps = 0 and +-1e-12 are decided by the last bits of the embedding (FMA contraction differs between compilers): compare from 1e-9 on
            assert np.array_equal(got[27:], ref[27:]), lanes
        assert 0 < ref[27:].sum() < len(ref) - 27
    finally:
        eng.set_option("clash_lanes", 1)


# ----------------------------------------------------------------------------- N3: csearch rotations
def test_csearch_rotations_golden(eng, oracle):
    """tscode/torsion_module.py:463-500 candidates against the reference-derived fixture G7 and the oracle."""
    import tscode_amd
    g = load_golden("G7_csearch")
    for c in range(int(g["n_cases"])):
        coords, torsions, masks, angles = g[f"coords{c}"], g[f"torsions{c}"], g[f"masks{c}"], g[f"angles{c}"]
        out, rb = eng.csearch_rotate(coords, torsions, masks, angles, 1.5, 0)
        assert np.array_equal(rb, g[f"rotated_bonds{c}"])                 # every verdict of every walk-back step
        assert np.abs(out - g[f"out{c}"]).max() < VAL_TOL
        # drop-in single calls
        t0 = next(t for t in range(len(torsions)) if angles[0][t] != 0)
        one = tscode_amd.rotate_dihedral(coords.copy(), torsions[t0], int(angles[0][t0]), mask=masks[t0].astype(bool))
        assert np.abs(one - oracle.rotate_dihedral(coords, torsions[t0], float(angles[0][t0]), masks[t0])).max() < VAL_TOL
        assert tscode_amd.torsion_comp_check(one, torsions[t0], masks[t0].astype(bool), 1.5) == int(g[f"first_checks{c}"][0])
        kept = tscode_amd.csearch_candidates(coords, torsions, masks, angles, n_out=7)
        assert kept.shape == (7, len(coords), 3) and np.abs(kept - g[f"out{c}"][np.flatnonzero(g[f"rotated_bonds{c}"])[:7]]).max() < VAL_TOL
    # part B: the reference's whole random_csearch -- shuffled n-fold angle tables, n_out / max_tries selection (:505-511)
    for b in range(int(g["b_n"])):
        got = tscode_amd.csearch_candidates(g["b_coords"], g["b_torsions"], g["b_masks"], g[f"b_angles{b}"], n_out=int(g[f"b_n_out{b}"]),
                                            max_tries=int(g[f"b_max_tries{b}"]))
        assert got.shape == g[f"b_out{b}"].shape and np.abs(got - g[f"b_out{b}"]).max() < VAL_TOL


def test_csearch_rotations_vs_oracle_large(eng, oracle):
    """A 200-atom trimolecular complex (the size of BASELINE config 5), 8 torsions, 3000 candidates."""
    rng = np.random.default_rng(21)
    from tscode_amd.synthetic import make_config
    ens = make_config("C5", 4)
    coords = ens.poses()[0]
    n = len(coords)
    # torsions along consecutive atoms of the first fragment; masks: everything after the bond inside that fragment
    n0 = ens.frag_coords[0].shape[1]
    centres = rng.choice(np.arange(2, n0 - 3), size=8, replace=False)
    torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres], dtype=np.int32)
    masks = np.zeros((8, n), dtype=np.uint8)
    for t, c in enumerate(centres):
        masks[t, c + 1:n0] = 1
    angles = rng.choice(np.array([0, 0, 60, 120, 180, -60, 25]), size=(3000, 8)).astype(np.int32)
    # (threshold 1.4: the synthetic fragments are walks of exactly 1.5 A steps, so 1.5 would sit on bonded distances)
    ref_out, ref_rb, margin = oracle.csearch_rotate(coords, torsions, masks, angles, 1.4, 0, return_margin=True)
    assert margin > 1e-9
    out, rb = eng.csearch_rotate(coords, torsions, masks, angles, 1.4, 0)
    assert np.array_equal(rb, ref_rb) and np.abs(out - ref_out).max() < VAL_TOL
    ok = eng.torsion_comp_check(out[:500], torsions[0], masks[0], 1.4, 0)
    assert ok.tolist() == [oracle.torsion_comp_check(o, torsions[0], masks[0], 1.4) for o in out[:500]]
    assert 0 < (rb == 0).sum() + (rb > 0).sum() and len(np.unique(rb)) > 2
    # a small rotating group (3 atoms: the wavefront is cut into slices of the fixed side), alone and next to a large one
    masks2 = masks.copy()
    masks2[0] = 0
    masks2[0, centres[0] + 1:centres[0] + 4] = 1
    masks2[3] = 0
    masks2[3, centres[3] + 1] = 1
    r2, rb2, margin = oracle.csearch_rotate(coords, torsions, masks2, angles[:1500], 1.4, 0, return_margin=True)
    assert margin > 1e-9
    o2, b2 = eng.csearch_rotate(coords, torsions, masks2, angles[:1500], 1.4, 0)
    assert np.array_equal(b2, rb2) and np.abs(o2 - r2).max() < VAL_TOL
    # a moved side of more than 64 atoms (several moved atoms per lane; the walk-back's look at the pair that clashed last covers
    # all of a lane's atoms), and a mask that turns the axis atom i2 with it (the walk-back then forms its matrix anew every step,
    # as the reference does, instead of keeping one)
    masks4 = masks.copy()
    masks4[1] = 0
    masks4[1, centres[1] + 1:n0 + 40] = 1                 # the rest of fragment 0 and 40 atoms of fragment 1: about 100 atoms turn
    masks4[5, centres[5]] = 1                             # i2 of torsion 5 turns with its moved side
    r4, rb4, margin = oracle.csearch_rotate(coords, torsions, masks4, angles[:1500], 1.4, 0, return_margin=True)
    assert margin > 1e-9 and int(masks4[1].sum()) > 64
    o4, b4 = eng.csearch_rotate(coords, torsions, masks4, angles[:1500], 1.4, 0)
    assert np.array_equal(b4, rb4) and np.abs(o4 - r4).max() < VAL_TOL
    # clashes allowed (the fp64 count path)
    r3, rb3 = oracle.csearch_rotate(coords, torsions, masks, angles[:800], 1.4, 3)
    o3, b3 = eng.csearch_rotate(coords, torsions, masks, angles[:800], 1.4, 3)
    assert np.array_equal(b3, rb3) and np.abs(o3 - r3).max() < VAL_TOL and not np.array_equal(rb3, ref_rb[:800])
    ok3 = eng.torsion_comp_check(out[:300], torsions[2], masks[2], 1.4, 2)
    assert ok3.tolist() == [oracle.torsion_comp_check(o, torsions[2], masks[2], 1.4, 2) for o in out[:300]]
    # thresholds a hair above / below the smallest moved-fixed distance of a structure: inside the fp32 band, decided in fp64
    x = out[7]
    mv, fx = masks[1].astype(bool), ~masks[1].astype(bool)
    fx[torsions[1][1]] = fx[torsions[1][2]] = False
    dmin = np.sqrt(((x[mv][:, None] - x[fx][None]) ** 2).sum(-1)).min()
    for thr, want in ((dmin * (1 + 1e-9), 0), (dmin * (1 - 1e-9), 1)):
        assert oracle.torsion_comp_check(x, torsions[1], masks[1], thr) == want
        assert eng.torsion_comp_check(x[None], torsions[1], masks[1], thr, 0).tolist() == [want]


# ----------------------------------------------------------------------------- N1: string-embed pose parameters
def test_string_embed_params_vs_oracle(eng, oracle):
    """tscode/embeds.py:98-116 for a whole embed in one launch: against the oracle, against the reference-derived
    matrices of G5, and by the invariant of :114 (the reactive centres coincide)."""
    import tscode_amd
    rng = np.random.default_rng(31)
    S = 257
    p1, p2, rv, mv = rng.normal(size=(4, S, 3)) * 2
    mv[0] = rv[0] * 1.7            # mol_vec parallel to ref_vec   -> antiparallel to -ref_vec: the 180-degree-about-z branch
    mv[1] = -rv[1] * 0.3           # mol_vec parallel to -ref_vec  -> identity branch
    cp = rng.integers(0, 5, size=(S, 2)).astype(np.int32)
    angles = np.arange(0, 360, 10).astype(np.float64)
    rot, pos, ci = eng.string_embed_params(p1, p2, rv, mv, cp, angles)
    ro, po, co = oracle.string_embed_params(p1, p2, rv, mv, cp, angles)
    assert np.array_equal(ci, co) and np.abs(rot - ro).max() < 1e-12 and np.abs(pos - po).max() < 1e-11
    A = len(angles)
    for s in (0, 1, 2, 100):
        for a in (0, 1, 17):
            R, t = rot[s * A + a, 1], pos[s * A + a, 1]
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and np.abs(R @ p2[s] + t - p1[s]).max() < 1e-11      # :114
            assert np.array_equal(rot[s * A + a, 0], np.eye(3)) and not pos[s * A + a, 0].any()
    # reference matrices: G5 holds rot_mat_from_pointer(ptr, ang) from the reference; with mol_vec = -ref_vec R0 is the identity
    g = load_golden("G5_rotations")
    ptr_, ang_ = g["ptr"], g["ang"]
    k = [i for i in range(len(ang_)) if ang_[i] != 0][:8]
    r2, _, _ = eng.string_embed_params(np.zeros((len(k), 3)), np.zeros((len(k), 3)), ptr_[k], -ptr_[k], np.zeros((len(k), 2), np.int32), ang_[k])
    for q, i in enumerate(k):
        assert np.abs(r2[q * len(k) + q, 1] - g["rot_mat_from_pointer"][i]).max() < 1e-12
    # the single-site drop-in still places the reactive centres on top of each other
    poses, r, t = tscode_amd.string_embed_poses(rng.normal(size=(6, 3)), rng.normal(size=(4, 3)), p1[5], p2[5], rv[5], mv[5], range(0, 360, 30))
    assert poses.shape == (12, 10, 3) and np.abs(np.einsum("nij,j->ni", r[:, 1], p2[5]) + t[:, 1] - p1[5]).max() < 1e-11


# ----------------------------------------------------------------------------- N2: torsion-fingerprint pruning
def test_tfd_pruning_golden(eng, oracle):
    """prune_conformers_tfd on the GPU against the reference's own results (G8: its function, networkx and all), plus the
    fingerprints and tfd_similarity against G6."""
    pytest.importorskip("networkx")
    import tscode_amd
    g6 = load_golden("G6_tfd")
    fp = tscode_amd._get_tf_mat(g6["coords"], g6["quadruplets"])
    assert fp.dtype == np.float32 and fp.shape == g6["fingerprints"].shape
    assert np.abs(fp - g6["fingerprints"]).max() < 2e-5 and (fp != g6["fingerprints"]).mean() < 0.01     # float32 of an fp64 angle
    assert np.array_equal(tscode_amd.get_torsion_fingerprint(g6["coords"][3], g6["quadruplets"]), fp[3])
    sims = [[tscode_amd.tfd_similarity(g6["fingerprints"][i], g6["fingerprints"][j], thresh=t) for j in range(10)] for i in range(4) for t in (10, 90, 300)]
    assert np.array_equal(np.array(sims), g6["similarity"][:12].astype(bool))
    g = load_golden("G8_tfd_prune")
    for c in range(int(g["n_cases"])):
        structures, thresh = g[f"structures{c}"], float(g[f"thresh{c}"])
        tf = eng.torsion_fingerprints(structures, g["quadruplets"])
        assert np.abs(tf - g[f"tf_mat{c}"]).max() < 2e-5
        # every pass of the pair search against the oracle on the reference's fingerprints
        n = len(structures)
        for k in (1, 2, 5, 10):
            if k == 1 or 5 * k < n:
                first_o, margin = oracle.tfd_first_similar(g[f"tf_mat{c}"], n // k, k, n - 3, thresh, return_margin=True)
                assert margin > 1e-6
                assert np.array_equal(eng.tfd_first_similar(g[f"tf_mat{c}"], n // k, k, n - 3, thresh), first_o), (c, k)
        pruned, mask = tscode_amd.prune_conformers_tfd(structures, g["quadruplets"], thresh=thresh)
        assert np.array_equal(mask, g[f"mask{c}"]), (c, mask.sum(), g[f"mask{c}"].sum())
        assert np.array_equal(pruned, structures[mask])


def test_cyclical_embed_params_vs_oracle(eng, oracle):
    """tscode/embeds.py:676-713 per (pose, molecule): against the oracle (align_vec_pair pinned by G5) and by the invariants
    of the construction."""
    import tscode_amd
    rng = np.random.default_rng(41)
    n = 3000
    st, en, di, pv, mp, r0, r1 = rng.normal(size=(7, n, 3)) * 2
    nr = rng.integers(1, 3, size=n).astype(np.int32)
    ang = rng.choice(np.arange(0, 360, 10), size=n).astype(np.float64)
    mp[5] = r0[5]; nr[5] = 1                                  # meanpoint == the reactive atom: the :677 fallback
    rot, pos = eng.cyclical_embed_params(st, en, di, pv, mp, r0, r1, nr, ang)
    ro, po = oracle.cyclical_embed_params(st, en, di, pv, mp, r0, r1, nr, ang)
    assert np.abs(rot - ro).max() < 1e-9 and np.abs(pos - po).max() < 1e-8
    assert np.abs(np.einsum("nij,nkj->nik", rot, rot) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(rot) - 1).max() < 1e-12
    apm = np.where((nr == 2)[:, None], (r0 + r1) / 2, r0)
    fixed = np.einsum("nij,nj->ni", rot, apm) + pos           # the reactive atoms' mean point does not depend on the angle ...
    rot0, pos0 = tscode_amd.cyclical_embed_params(st, en, di, pv, mp, r0, r1, nr, np.zeros(n))
    assert np.abs(fixed - (np.einsum("nij,nj->ni", rot0, apm) + pos0)).max() < 1e-9
    assert np.abs(np.einsum("nij,nj->ni", rot0, mp) + pos0 - (st + en) / 2).max() < 1e-9      # ... and at angle 0 the pivot's mean point sits on the side's
    # G5: align_vec_pair of the reference itself (angle 0, one reactive atom at the origin => rotation == align_vec_pair(ref, tgt))
    g = load_golden("G5_rotations")
    ref_v, tgt_v, avp = g["avp_ref"][2:], g["avp_tgt"][2:], g["align_vec_pair"][2:]
    m = len(ref_v)
    z = np.zeros((m, 3))
    r5, _ = eng.cyclical_embed_params(z, ref_v[:, 0], ref_v[:, 1], tgt_v[:, 0], tgt_v[:, 1], z, z, np.ones(m, np.int32), np.zeros(m))
    assert np.abs(r5 - avp).max() < 1e-9


# ----------------------------------------------------------------------------- N4: moments of inertia, embed scores
def test_moi_and_scores_golden(eng, oracle):
    """get_inertia_moments / get_moi_similarity_matches / the graph step of prune_by_moment_of_inertia / _score_embed_poses /
    fitness_check against the reference-derived fixture G9."""
    pytest.importorskip("networkx")
    import tscode_amd
    g = load_golden("G9_moi_scores")
    for c in range(int(g["n_cases"])):
        structures, masses = g[f"structures{c}"], g[f"masses{c}"]
        mo = eng.inertia_moments(structures, masses)
        assert (np.abs(mo - g[f"moments{c}"]) / np.abs(g[f"moments{c}"])).max() < 1e-12
        _, margin = oracle.moi_first_similar(g[f"moments{c}"], 1e-2, return_margin=True)
        assert margin > 1e-9
        matches = tscode_amd.get_moi_similarity_matches(structures, masses, max_deviation=1e-2)
        assert matches == [tuple(m) for m in g[f"matches{c}"].tolist()]
        # prune_by_moment_of_inertia: the reference's own function (no hydrogens in these cases), masses given per atom and
        # from the built-in table (the fixture's elements are in it)
        atomnos = g[f"atomnos{c}"]
        pruned, mask = tscode_amd.prune_by_moment_of_inertia(structures, atomnos, max_deviation=1e-2, masses=masses)
        assert np.array_equal(mask, g[f"mask{c}"]) and np.array_equal(pruned, structures[mask])
        import tscode_amd.optimization_methods as om
        om._WARNED_MASSES = False
        with pytest.warns(UserWarning, match="built-in table of standard atomic weights"):      # the default table is announced as unverified
            _, mask_t = tscode_amd.prune_by_moment_of_inertia(structures, atomnos, max_deviation=1e-2)
        assert np.array_equal(mask_t, g[f"mask{c}"])
        assert np.abs(tscode_amd.get_inertia_moments(structures[2], masses) - g[f"moments{c}"][2]).max() < 1e-9 * np.abs(g[f"moments{c}"][2]).max()
    _, mask_h = tscode_amd.prune_by_moment_of_inertia(g["h_structures"], g["h_atomnos"], max_deviation=1e-2, masses=g["h_masses"])
    assert np.array_equal(mask_h, g["h_mask"])                                # hydrogens are dropped before the moments (:335-336)
    sc = tscode_amd._score_embed_poses(g["sc_structures"], g["sc_indices"], g["sc_distances"])
    assert sc.dtype == np.float32 and np.abs(sc - g["scores"]).max() < 1e-5
    so, eo = oracle.embed_scores(g["sc_structures"], g["sc_indices"], g["sc_distances"])
    _, err = eng.embed_scores(g["sc_structures"], g["sc_indices"], g["sc_distances"])
    assert np.abs(err - eo).max() < 1e-12 and np.array_equal(sc, so)
    thr = float(g["fitness_threshold"])
    assert np.array_equal(tscode_amd.fitness_mask(g["sc_structures"], g["sc_indices"], g["sc_distances"], thr), g["fitness_ok"])
    none_targets = [[None if np.isnan(t) else float(t) for t in row] for row in g["fitness_none"]]
    assert np.array_equal(tscode_amd.fitness_mask(g["sc_structures"], g["sc_indices"], none_targets, thr), g["fitness_ok_none"])
    assert tscode_amd.fitness_check(g["sc_structures"][0], g["sc_indices"][0], list(g["sc_distances"][0]), thr) == bool(g["fitness_ok"][0])


def test_rotate_dihedral_fractional_angles(eng, oracle):
    """G15: the reference's rotate_dihedral (tscode/utils.py:389-414) with the fractional angles tscode/torsion_module.py:984-1005 passes
    -- the drop-in (one structure, in place, returns its argument), the batched form (the whole angle list in ONE call) and the
    rotate - look - rotate back trail of the correction search; and against the oracle on a larger random batch."""
    import tscode_amd
    g = load_golden("G15_rotate_dihedral_fractional")
    coords, dih, mask = g["coords"], g["dihedral"], g["mask"].astype(bool)
    for a, rm, rf in zip(g["angles"], g["out_mask"], g["out_first"]):
        c = coords.copy()
        assert tscode_amd.rotate_dihedral(c, dih, float(a), mask=mask) is c and np.abs(c - rm).max() < VAL_TOL
        assert np.abs(tscode_amd.rotate_dihedral(coords.copy(), dih, float(a)) - rf).max() < VAL_TOL
    batch = tscode_amd.rotate_dihedral_batch(np.repeat(coords[None], len(g["angles"]), axis=0), dih, g["angles"], mask)
    assert np.abs(batch - g["out_mask"]).max() < VAL_TOL
    seq, q = coords.copy(), 0
    for a in g["angles"][:4]:
        for sign in (1.0, -1.0):
            seq = tscode_amd.rotate_dihedral(seq, dih, sign * float(a), mask=mask)
            assert np.abs(seq - g["trail"][q]).max() < VAL_TOL
            q += 1
    rng = np.random.default_rng(15)
    big = rng.normal(size=(3000, 37, 3)) * 4
    m = rng.random(37) < 0.4
    ang = rng.uniform(-360, 360, size=3000)
    got = eng.rotate_dihedral_batch(big, [4, 9, 20, 30], m, ang)
    for s in rng.integers(0, 3000, size=40):
        assert np.abs(got[s] - oracle.rotate_dihedral(big[s], [4, 9, 20, 30], float(ang[s]), m.astype(np.uint8))).max() < VAL_TOL
    assert np.array_equal(got[:, ~m], big[:, ~m])                       # the other atoms are copied, bit for bit
    assert eng.rotate_dihedral_batch(np.zeros((0, 5, 3)), [0, 1, 2, 3], np.ones(5, bool), []).shape == (0, 5, 3)


def test_adjacent_rows_edge_cases(eng, oracle):
    """Empty and degenerate inputs of the next-row entry points (they must not fault and must mirror the reference's shapes)."""
    import tscode_amd
    quads = np.array([[0, 1, 2, 3]])
    # N2: one structure, two identical structures, no quadruplets
    one = np.random.default_rng(0).normal(size=(1, 5, 3))
    kept, mask = tscode_amd.prune_conformers_tfd(one, quads)
    assert mask.tolist() == [True] and kept.shape == (1, 5, 3)
    two = np.concatenate([one, one])
    _, mask = tscode_amd.prune_conformers_tfd(two, quads)
    assert mask.sum() == 1                                        # one of two identical structures goes
    assert eng.tfd_first_similar(np.zeros((0, 3), np.float32), 1, 1, 0, 10.0).shape == (0,)
    assert tscode_amd._get_tf_mat(one, np.zeros((0, 4), np.int32)).shape == (1, 0)
    # N3: no candidates; all-zero angle sets leave the structure untouched and rotate nothing
    coords = one[0]
    mask5 = np.array([0, 0, 0, 1, 1], dtype=np.uint8)
    out, rb = eng.csearch_rotate(coords, [(0, 1, 2, 3)], [mask5], np.zeros((0, 1), np.int32))
    assert out.shape == (0, 5, 3) and rb.shape == (0,)
    out, rb = eng.csearch_rotate(coords, [(0, 1, 2, 3)], [mask5], np.zeros((3, 1), np.int32))
    assert np.array_equal(out, np.broadcast_to(coords, (3, 5, 3))) and rb.tolist() == [0, 0, 0]
    assert tscode_amd.csearch_candidates(coords, [(0, 1, 2, 3)], [mask5], np.zeros((3, 1), np.int32)).shape[0] == 0
    # a negative angle that clashes is never walked back (angle // 5 < 0) and does not count as rotated
    clash = np.array([[0.0, 0, 0], [1.5, 0, 0], [2.2, 1.2, 0], [3.5, 1.4, 0.3], [0.3, 0.9, 0.2]])
    for ang in (-170, 170, -40, 40):
        o_ref, rb_ref = oracle.csearch_rotate(clash, [(0, 1, 2, 3)], [mask5], [[ang]], 1.5, 0)
        o_gpu, rb_gpu = eng.csearch_rotate(clash, [(0, 1, 2, 3)], [mask5], [[ang]], 1.5, 0)
        assert np.array_equal(rb_ref, rb_gpu) and np.abs(o_ref - o_gpu).max() < VAL_TOL
    # N4: a single structure has no match; constraints with a missing target are skipped
    assert tscode_amd.get_moi_similarity_matches(one, np.ones(5)) == []
    assert tscode_amd.fitness_check(coords, [(0, 1), (2, 3)], [None, float(np.linalg.norm(coords[2] - coords[3]))], 1e-9)
    # N1: no sites
    r, p, c = eng.string_embed_params(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 2), np.int32), [0.0, 10.0])
    assert r.shape == (0, 2, 3, 3) and p.shape == (0, 2, 3) and c.shape == (0, 2)


# ----------------------------------------------------------------------------- N1: the embed loops as drivers (G10-G12, config C1)
def test_embed_helpers_golden(eng, oracle):
    """Row a4 on the device and the other helpers of the embed loops against the reference's own outputs (G10)."""
    import types
    import tscode_amd
    g = load_golden("G10_embed_helpers")
    # rotation_matrix_from_vectors inside k_string_embed_params: angle 0, so R = R0 = rmfv(mol_vec, -ref_vec) (embeds.py:108)
    v1, v2 = g["rmfv_v1"], g["rmfv_v2"]
    zero = np.zeros_like(v1)
    rot, pos, _ = eng.string_embed_params(zero, zero, -v2, v1, np.zeros((len(v1), 2), np.int32), [0.0])
    assert np.abs(rot[:, 1] - g["rmfv_out"]).max() < 1e-12 and np.array_equal(rot[0, 1], np.eye(3))
    assert np.abs(rot[1, 1] - np.diag([-1.0, -1.0, 1.0])).max() < 1e-15 and np.abs(pos).max() == 0
    for a, b, ref in zip(v1, v2, g["rmfv_out"]):
        assert np.abs(tscode_amd.rotation_matrix_from_vectors(a, b) - ref).max() < 1e-12
    # get_embed on duck-typed molecules
    for k in range(int(g["ge_n"])):
        nm = int(g[f"ge{k}_n_mols"])
        mols = [types.SimpleNamespace(atomcoords=g[f"ge{k}_coords{m}"], rotation=g[f"ge{k}_rot{m}"], position=g[f"ge{k}_pos{m}"]) for m in range(nm)]
        assert np.abs(tscode_amd.get_embed(mols, g[f"ge{k}_conf_ids"]) - g[f"ge{k}_out"]).max() < 1e-12
    # rotate_dihedral: mask, indices_to_be_moved, neither
    for a, rm, ri, rf in zip(g["rd_angles"], g["rd_out_mask"], g["rd_out_indices"], g["rd_out_first"]):
        assert np.abs(tscode_amd.rotate_dihedral(g["rd_coords"].copy(), g["rd_dihedral"], int(a), mask=g["rd_mask"]) - rm).max() < VAL_TOL
        assert np.abs(tscode_amd.rotate_dihedral(g["rd_coords"].copy(), g["rd_dihedral"], int(a), indices_to_be_moved=list(g["rd_moved"])) - ri).max() < VAL_TOL
        assert np.abs(tscode_amd.rotate_dihedral(g["rd_coords"].copy(), g["rd_dihedral"], int(a)) - rf).max() < VAL_TOL


def test_tfd_greedy_filter_vs_oracle(eng, oracle):
    """is_new_structure over whole lists (embeds.py:47-69): clustered fingerprints, several blocks of 64, kept list in the hundreds."""
    rng = np.random.default_rng(77)
    # (4095 / 4096 / 4097 / 9000: the seams of the 4096-candidate super-blocks; T = 9 and 20: fingerprints too long for the register path and its
    # fp32 screen; 9000 x 3 with few parents: nearly every candidate is dropped by a structure kept in an earlier super-block; angles beyond
    # +-180 in the last case: differences above 360)
    for n, T, n_par, noise in ((1, 3, 1, 0.0), (63, 4, 9, 0.7), (64, 2, 64, 0.0), (65, 5, 20, 1.0), (700, 6, 150, 0.8), (5000, 8, 900, 0.6),
                               (4095, 6, 700, 0.7), (4096, 9, 500, 0.7), (4097, 20, 300, 0.5), (9000, 3, 40, 0.9), (9000, 6, 2500, 0.8)):
        parents = rng.uniform(-180, 180, size=(n_par, T))
        tf = (parents[rng.integers(0, n_par, size=n)] + rng.normal(size=(n, T)) * noise).astype(np.float32)
        tf = ((tf + 180) % 360 - 180).astype(np.float32)                  # wrap-around pairs (+179 vs -179) included
        if n == 9000 and T == 6:
            tf[::7] += 360.0                                              # the same angles a turn further: |difference| up to 540
        ref, margin = oracle.tfd_greedy_filter(tf, 10, return_margin=True)
        assert margin > 1e-6
        got = eng.tfd_greedy_filter(tf, 10.0)
        assert np.array_equal(got, ref), (n, T, got.sum(), ref.sum())
    assert eng.tfd_greedy_filter(np.zeros((0, 4), np.float32)).shape == (0,)
    assert eng.tfd_greedy_filter(np.zeros((5, 0), np.float32)).tolist() == [True, False, False, False, False]   # no torsions: sum 0 < 10


def _string_case(g, k):
    return dict(coords1=g[f"coords0_{k}"], coords2=g[f"coords1_{k}"], centers1=g[f"centers0_{k}"], orb_vecs1=g[f"orb_vecs0_{k}"],
                centers2=g[f"centers1_{k}"], orb_vecs2=g[f"orb_vecs1_{k}"], angles=g[f"angles_{k}"])


def test_string_embed_c1_golden(eng, oracle):
    """BASELINE config C1: the reference's own string_embed on tests/CH3Cl.xyz + tests/HCOOH.xyz (case 0 is tests/string.txt) and
    variants, recorded in G11, against ONE call of the product driver: string_embed_params -> embed -> compenetration mask ->
    torsion fingerprints -> is_new_structure, all on the device."""
    import tscode_amd
    g = load_golden("G11_string_embed")
    for k in range(int(g["n_cases"])):
        case = _string_case(g, k)
        thresh, quads = float(g[f"clash_thresh_{k}"]), g[f"quadruplets_{k}"]
        poses, tr = tscode_amd.string_embed_batch(**case, clash_thresh=thresh, quadruplets=quads, return_trace=True)
        assert np.array_equal(tr.clash_ok, g[f"clash_ok_{k}"])                               # every compenetration_check verdict
        assert np.array_equal(tr.kept, g[f"kept_{k}"]), (k, tr.kept.sum(), g[f"kept_{k}"].sum())   # every is_new_structure verdict
        assert poses.shape == g[f"poses_{k}"].shape and np.abs(poses - g[f"poses_{k}"]).max() < VAL_TOL
        # the stages one by one: every candidate pose, and the fingerprints bit for bit (the verdicts at `sum == 10` hang on them)
        c1, c2, a1, a2 = tr.sites.T
        rot, pos, ci = tscode_amd.string_embed_params(case["centers1"][c1, a1], case["centers2"][c2, a2], case["orb_vecs1"][c1, a1],
                                                      case["orb_vecs2"][c2, a2], np.stack([c1, c2], 1), case["angles"])
        cands = tscode_amd.embed_batch([case["coords1"], case["coords2"]], ci, rot, pos)
        assert np.abs(cands - g[f"candidates_{k}"]).max() < VAL_TOL
        assert np.array_equal(tscode_amd.compenetration_mask(cands, g[f"ids_{k}"], thresh, 0), g[f"clash_ok_{k}"])
        fp = tscode_amd._get_tf_mat(cands[g[f"fp_index_{k}"]], quads)
        assert np.array_equal(fp, g[f"fingerprints_{k}"])
        assert np.array_equal(eng.tfd_greedy_filter(fp, 10.0), g[f"kept_{k}"][g[f"clash_ok_{k}"]])
    # nothing passes the clash check: an empty result where the reference raises ZeroCandidatesError
    poses, tr = tscode_amd.string_embed_batch(**_string_case(g, 0), clash_thresh=9.0, quadruplets=g["quadruplets_0"], return_trace=True)
    assert poses.shape == (0, 10, 3) and not tr.clash_ok.any() and not tr.kept.any()


def _cyclical_case(g, k):
    coords = [g[f"coords{m}_{k}"] for m in range(2)]
    return [dict(coords=coords[m], reactive_indices=g[f"reactive_indices{m}_{k}"],
                 pivots=[(g[f"pivot_vec{m}_{c}_{k}"], g[f"pivot_mean{m}_{c}_{k}"], g[f"pivot_cumnums{m}_{c}_{k}"]) for c in range(len(coords[m]))])
            for m in range(2)]


def test_cyclical_embed_golden(eng, oracle):
    """The reference's cyclical_embed (rigid shortcut and general loop; tests/cyclical.txt's molecules and variants), G12, against
    ONE call of the product driver: polygonize + group layout on the host, pose parameters -> embed -> compenetration mask ->
    greedy per-group _rmsd_similarity filter on the device."""
    import tscode_amd
    g = load_golden("G12_cyclical_embed")
    for k in range(int(g["n_cases"])):
        mols = _cyclical_case(g, k)
        poses, cons, tr = tscode_amd.cyclical_embed_batch(mols, g[f"angles_{k}"], clash_thresh=float(g[f"clash_thresh_{k}"]),
                                                          rigid_shortcut=bool(g[f"rigid_{k}"]), return_trace=True)
        assert np.array_equal(tr.group_of, g[f"group_of_{k}"])
        assert np.array_equal(np.array([grp[3] for grp in tr.groups]), g[f"group_ids_{k}"])
        assert np.array_equal(tr.clash_ok, g[f"clash_ok_{k}"])
        assert np.array_equal(tr.kept, g[f"kept_{k}"]), (k, tr.kept.sum(), g[f"kept_{k}"].sum())
        assert poses.shape == g[f"poses_{k}"].shape and np.abs(poses - g[f"poses_{k}"]).max() < VAL_TOL
        assert np.array_equal(cons, g[f"constrained_indices_{k}"])
    with pytest.raises(ValueError):
        tscode_amd.cyclical_embed_batch(_cyclical_case(g, 0) * 2, g["angles_0"])           # not two molecules


def test_cyclical_embed_trimolecular_golden(eng, oracle):
    """THREE molecules (tests/trimolecular.txt's CH3Cl + 2 HCOOH and variants), G18: the reference's own cyclical_embed run with the one change
    that makes it defined (vec_angle pads the 2-vectors of _get_directions with z = 0; embeds.py:297-299 / algebra.py:87) -- "reference
    patched", unpinned where the reference is undefined -- against ONE call of the product driver: triangle, directions and their
    adjustment on the host, pose parameters -> embed of three fragments -> compenetration mask -> greedy per-group filter on the device."""
    import tscode_amd
    g = load_golden("G18_cyclical_embed_trimolecular")
    for k in range(int(g["n_cases"])):
        mols = []
        for m in range(3):
            coords = g[f"coords{m}_{k}"]
            mols.append(dict(coords=coords, reactive_indices=g[f"reactive_indices{m}_{k}"], reactive_cumnums=g[f"reactive_cumnums{m}_{k}"],
                             pivots=[(g[f"pivot_vec{m}_{c}_{k}"], g[f"pivot_mean{m}_{c}_{k}"], g[f"pivot_cumnums{m}_{c}_{k}"]) for c in range(len(coords))]))
        poses, cons, tr = tscode_amd.cyclical_embed_batch(mols, g[f"angles_{k}"], clash_thresh=float(g[f"clash_thresh_{k}"]), return_trace=True)
        assert np.array_equal(tr.group_of, g[f"group_of_{k}"])
        assert np.array_equal(np.array([grp[3] for grp in tr.groups]), g[f"group_ids_{k}"])
        assert oracle.clash_margin(g[f"candidates_{k}"], g[f"ids_{k}"], float(g[f"clash_thresh_{k}"])) > 1e-9
        assert np.array_equal(tr.clash_ok, g[f"clash_ok_{k}"])
        assert np.array_equal(tr.kept, g[f"kept_{k}"]), (k, tr.kept.sum(), g[f"kept_{k}"].sum())
        assert poses.shape == g[f"poses_{k}"].shape and np.abs(poses - g[f"poses_{k}"]).max() < VAL_TOL
        assert np.array_equal(cons, g[f"constrained_indices_{k}"]) and cons.shape[1:] == (3, 2)


def test_cyclical_embed_wide_groups(eng, oracle):
    """STEPS = 36: (36 + 1)^2 = 1369 poses per (conformers, pivots, orientation) group -- beyond the 1024 the group filter took in round 2
    (ADVICE r2).  The product driver against the oracle's loop on the molecules of G12's first case."""
    import tscode_amd
    from tscode_amd.utils import cartesian_product
    g = load_golden("G12_cyclical_embed")
    mols = _cyclical_case(g, 0)
    steps = np.linspace(-40.0, 40.0, 37)
    angles = cartesian_product(steps, steps)                                                # embedder.py:714-715
    assert len(angles) == 1369
    thresh = float(g["clash_thresh_0"])
    poses, cons, tr = tscode_amd.cyclical_embed_batch(mols, angles, clash_thresh=thresh, rigid_shortcut=bool(g["rigid_0"]), return_trace=True)
    get = lambda m, k: m[k] if isinstance(m, dict) else getattr(m, k)
    cands, group_of, ok, kept, gids = oracle.cyclical_embed([np.asarray(get(m, "coords")) for m in mols], [np.atleast_1d(get(m, "reactive_indices")) for m in mols],
                                                            [get(m, "pivots") for m in mols], angles, thresh, rigid_shortcut=bool(g["rigid_0"]))
    assert np.array_equal(tr.group_of, group_of) and np.bincount(group_of).max() == 1369
    assert np.array_equal(tr.clash_ok, ok) and np.array_equal(tr.kept, kept), (tr.kept.sum(), kept.sum())
    assert np.abs(poses - cands[kept]).max() < VAL_TOL
    with pytest.raises(ValueError):
        tscode_amd.cyclical_embed_batch(mols, np.zeros((8193, 2)))                          # beyond the library's group size


def test_embed_dropins_with_the_reference_signatures(eng):
    """tscode_amd.embeds.string_embed(embedder) / cyclical_embed(embedder): what install() puts in place of the reference's
    functions.  Duck-typed Embedder and molecules carrying exactly the recorded inputs of G11 / G12 (the reference's own
    Hypermolecule data); tscode.embeds is a stand-in module offering the helper names the drop-ins take from the live one."""
    import sys
    import types
    import tscode_amd
    from tscode_amd import embeds as E
    g11, g12 = load_golden("G11_string_embed"), load_golden("G12_cyclical_embed")

    import dropin_reads

    class Zero(Exception):
        pass
    logs = []
    standin = dropin_reads.reference_module_standin(Zero)
    saved = sys.modules.get("tscode.embeds")
    sys.modules["tscode.embeds"] = standin
    try:
        for k in range(int(g11["n_cases"])):
            standin._get_quadruplets = lambda graph, k=k: g11[f"quadruplets_{k}"]
            emb = dropin_reads.duck_string_embedder(g11, k, logs)
            poses = E.string_embed(emb)
            assert poses.shape == g11[f"poses_{k}"].shape and np.abs(poses - g11[f"poses_{k}"]).max() < VAL_TOL
            assert np.array_equal(emb.constrained_indices, g11[f"constrained_indices_{k}"])
        emb.options.clash_thresh = 9.0
        with pytest.raises(Zero):
            E.string_embed(emb)
        for k in range(int(g12["n_cases"])):
            emb = dropin_reads.duck_cyclical_embedder(g12, k, logs)
            mols = emb.objects
            poses = E.cyclical_embed(emb)
            assert poses.shape == g12[f"poses_{k}"].shape and np.abs(poses - g12[f"poses_{k}"]).max() < VAL_TOL
            assert np.array_equal(emb.constrained_indices, g12[f"constrained_indices_{k}"])
        # three molecules with the switch on: the batch driver (G18: the reference's own code with vec_angle's plane vectors padded)
        g18 = load_golden("G18_cyclical_embed_trimolecular")
        E.TRIMOLECULAR = True
        try:
            emb3 = dropin_reads.duck_cyclical_embedder(g18, 0, logs, n_mols=3)
            poses3 = E.cyclical_embed(emb3)
            assert poses3.shape == g18["poses_0"].shape and np.abs(poses3 - g18["poses_0"]).max() < VAL_TOL
            assert np.array_equal(emb3.constrained_indices, g18["constrained_indices_0"])
        finally:
            E.TRIMOLECULAR = False
        # three molecules by default: handed to the reference's own function (recorded by install())
        done = tscode_amd.install()
        assert ("tscode.embeds", "string_embed") in done and ("tscode.embeds", "cyclical_embed") in done
        assert standin.cyclical_embed is E.cyclical_embed
        emb.objects = mols + [mols[0]]
        assert standin.cyclical_embed(emb) == "the reference's own function"
        tscode_amd.uninstall()
        assert standin.cyclical_embed is not E.cyclical_embed
    finally:
        tscode_amd.uninstall()
        if saved is None:
            sys.modules.pop("tscode.embeds", None)
        else:
            sys.modules["tscode.embeds"] = saved
    assert logs


def test_c5_chain_csearch_feeds_the_pipeline(eng, oracle):
    """BASELINE config 5 as a chain (reduced size): csearch rotations of fragment 0 on the device -> the kept candidates are
    that fragment's conformers -> embed -> clash mask -> prune, the candidate array never leaving the GPU -- against the same
    chain built from the oracle's pieces (orc_csearch_rotate, orc_transform_batch, clash mask, prune)."""
    from tscode_amd.pipeline import CsearchChain
    from tscode_amd.synthetic import make_config
    ens = make_config("C5", 5000)
    n0 = ens.frag_coords[0].shape[1]
    torsions, masks = CsearchChain.chain_torsions(n0, 6, seed=3)
    rng = np.random.default_rng(8)
    angles = rng.choice(np.array([0, 0, 60, 120, 180, 240, 300, 25]), size=(700, 6)).astype(np.int32)
    n_out = 200
    # (threshold 1.4: the synthetic fragments are walks of exactly 1.5 A steps, so 1.5 would sit on bonded distances)
    chain = CsearchChain(ens, torsions, masks, angles, n_out=n_out, thresh=1.4, seed=11)
    res = chain.step()
    chain.torch.cuda.synchronize()
    out, rb, margin = oracle.csearch_rotate(ens.frag_coords[0][0], torsions, masks, angles, 1.4, 0, return_margin=True)
    assert margin > 1e-9
    confs = out[rb != 0][:n_out]
    assert res["n_conformers"] == len(confs) == n_out
    draw = np.random.default_rng(11).integers(0, 2 ** 30, size=ens.n_poses)
    ci = ens.conf_idx.copy()
    ci[:, 0] = draw % len(confs)
    poses = oracle.transform_batch([confs] + [f for f in ens.frag_coords[1:]], ci, ens.rot, ens.pos)
    assert oracle.clash_margin(poses, ens.ids, 1.5) > 1e-9
    cm = oracle.compenetration_mask(poses, ens.ids, 1.5, 0)
    assert np.array_equal(chain.d_clash.cpu().numpy().astype(bool), cm) and res["n_pass"] == cm.sum() and 0 < cm.sum() < len(cm)
    assert np.abs(chain.d_structures[:res["n_pass"]].cpu().numpy() - poses[cm]).max() < VAL_TOL
    heavy = np.ascontiguousarray(poses[cm][:, ens.atomnos != 1])
    mr, mm = oracle.prune_margins(heavy, 0.5, 0)
    assert min(mr, mm) > 1e-9
    ref = oracle.prune_heavy(heavy, 0.5, mode=0)
    assert np.array_equal(chain.h_keep[:res["n_pass"]].numpy().astype(bool), ref["mask"]) and res["n_keep"] == ref["mask"].sum()
    assert [s["pairs_evaluated"] for s in res["stats"]] == [s["pairs_evaluated"] for s in ref["stats"]]
    # a second step on the resident buffers gives the same verdicts
    res2 = chain.step()
    chain.torch.cuda.synchronize()
    assert (res2["n_conformers"], res2["n_pass"], res2["n_keep"]) == (res["n_conformers"], res["n_pass"], res["n_keep"])


def test_run_sharded_c_entry_two_ranks_in_threads(oracle):
    """tsc_prune_run_sharded driven as a C host would drive it: two ranks in ONE process, each with its own library context, stream
    and prune run, each inside ONE call of the library that walks the whole pass schedule and calls back for the collectives; the
    callbacks of the two threads meet at a barrier and reduce the two device buffers (SUM of int64 / MIN of int32) as RCCL would.
    Every rank ends with the oracle's mask and evaluation counts; the exchange log names partitioned passes, the views and row-tile
    passes; a callback that fails aborts the run with its own exception; with a world of one the call needs no callback at all."""
    import threading

    import torch

    from tscode_amd._lib import XCHG_MIN_I32, XCHG_SUM_I64, TscodeHipError
    from tscode_amd.engine import Engine, PruneStepper
    from tscode_amd.synthetic import make_config
    ens = make_config("C2", 30_000)
    heavy = np.ascontiguousarray(ens.poses()[:, ens.atomnos != 1])
    ref = oracle.prune_heavy(heavy, 0.5, mode=0, row_parallel=True)
    n, h = heavy.shape[0], heavy.shape[1]
    dev = torch.device("cuda:0")
    world = 2
    engs = [Engine(0) for _ in range(world)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
    for e, s in zip(engs, streams):
        e.set_stream(s.cuda_stream)
        e.set_option("deterministic_basis", 1)       # the ranks deal the tiles of one sorted layout where a pass is culled
    d_heavy = torch.from_numpy(heavy).to(dev)
    words = PruneStepper.exchange_words(engs[0].lib, n, 0)
    exch = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(world)]
    bests = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(world)]
    barrier = threading.Barrier(world)
    slots, logs, errors = [None] * world, [None] * world, [None] * world

    def exchange_for(r):
        def exchange(kind, addr, count):
            buf = exch[r] if kind == XCHG_SUM_I64 else bests[r]
            off = (addr - buf.data_ptr()) // buf.element_size()
            slots[r] = buf[off:off + count]
            streams[r].synchronize()
            barrier.wait()
            if r == 0:                                    # "the collective": every rank's buffer <- the reduction over the ranks
                st = torch.stack(slots)
                red = st.sum(0) if kind == XCHG_SUM_I64 else st.amin(0)
                for s_ in slots:
                    s_.copy_(red)
                torch.cuda.synchronize()
            barrier.wait()
        return exchange

    def rank_main(r, fail_at=None):
        try:
            with torch.cuda.stream(streams[r]):
                st = engs[r].prune_stepper(d_heavy, n, h, 0.5, 0)
                st.use_best_buffer(bests[r])
                calls = [0]
                inner = exchange_for(r)

                def exchange(kind, addr, count):
                    calls[0] += 1
                    if fail_at is not None and calls[0] == fail_at:
                        raise RuntimeError("the host's collective failed")
                    inner(kind, addr, count)
                logs[r] = st.run_sharded(r, world, 4, 2_000_000, exch[r], exchange)
                keep = torch.empty(n, dtype=torch.uint8, device=dev)
                st.copy_mask(keep)
                streams[r].synchronize()
                slots[r] = None
                errors[r] = (keep.cpu().numpy().astype(bool), st.stats())
                st.close()
        except BaseException as exc:  # noqa: BLE001
            errors[r] = exc
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for r in range(world):
        assert not isinstance(errors[r], BaseException), errors[r]
        mask, stats = errors[r]
        assert np.array_equal(mask, ref["mask"]), (r, mask.sum(), ref["mask"].sum())
        assert [x["pairs_evaluated"] for x in stats] == [x["pairs_evaluated"] for x in ref["stats"]]
        assert [x["n_active_after"] for x in stats] == [x["n_active_after"] for x in ref["stats"]]
    assert logs[0] == logs[1] and len(logs[0]) >= 4
    kinds = {(k > 0, kind) for k, kind, _ in logs[0]}
    assert (True, XCHG_SUM_I64) in kinds and (False, XCHG_SUM_I64) in kinds and (True, XCHG_MIN_I32) in kinds, logs[0]
    # a callback that raises: the run is aborted (TSC_ERR_STATE inside) and the exception is the caller's
    barrier.reset()
    threads = [threading.Thread(target=rank_main, args=(r, 2 if r == 0 else None)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert isinstance(errors[0], RuntimeError) and "collective failed" in str(errors[0])
    assert isinstance(errors[1], BaseException)           # (its partner never arrived: the barrier broke)
    # a world of one: tsc_prune_run_replicated to the end, no callback, nothing logged
    with torch.cuda.stream(streams[0]):
        st = engs[0].prune_stepper(d_heavy, n, h, 0.5, 0)
        assert st.run_sharded(0, 1, 4, 2_000_000, None, None) == []
        keep = torch.empty(n, dtype=torch.uint8, device=dev)
        st.copy_mask(keep)
        streams[0].synchronize()
        assert np.array_equal(keep.cpu().numpy().astype(bool), ref["mask"])
        st.close()
        # ... and several ranks without an exchange function are refused
        st = engs[0].prune_stepper(d_heavy, n, h, 0.5, 0)
        with pytest.raises(TscodeHipError):
            st.run_sharded(0, 2, 4, 2_000_000, exch[0], None)
        st.close()


def test_embed_masked_does_not_pull_buffers_from_under_a_live_run(eng):
    """the prune run created right after tsc_embed_masked_dev BORROWS the context's descriptor buffers; a later embed on the
    same context that would have to regrow them fails loudly (TSC_ERR_STATE) while the run lives, and works once it is destroyed."""
    from tscode_amd._lib import TscodeHipError
    from tscode_amd.pipeline import HipShardBackend
    from tscode_amd.synthetic import make_config
    small, big = make_config("C2", 20_000), make_config("C2", 60_000)
    a = HipShardBackend(small, 0, 0, 1, 1.5, 0, 0.5, 0)
    b = HipShardBackend(big, 0, 0, 1, 1.5, 0, 0.5, 0)
    b.eng = a.eng                                          # both front halves on ONE library context
    b.stream = a.stream
    with a.stream_context():
        a.clash_block_into_all()
        n_a = int(a.embed_masked_all())
        st = a.make_stepper(n_a)                           # takes the descriptors the embed wrote: borrows the context's buffers
        b.clash_block_into_all()
        with pytest.raises(TscodeHipError) as err:
            b.embed_masked_all()                           # three times the poses: the buffers would have to grow
        assert "still read" in str(err.value)
        while st.next_pass():
            st.pass_local(0, 1)
            st.pass_finish()
        n_keep = st.stats()[-1]["n_active_after"]
        st.close()
        b.clash_block_into_all()
        assert int(b.embed_masked_all()) > n_a             # the run is gone: now it may
    assert 0 < n_keep <= n_a


@pytest.mark.parametrize("world,cfg,n_poses,min_pairs", [(2, "C3", 0, 0), (3, "C2", 0, 100_000), (4, "C3", 0, 0), (3, "C4", 0, 0), (2, "C5", 0, 0),
                                                         (3, "C5chain", 30_000, 0), (2, "C5chain", 0, 0), (1, "C2", 0, 100_000), (1, "C3", 0, 0)])
def test_sharded_protocol_on_gpu_ranks(oracle, world, cfg, n_poses, min_pairs):
    """The multi-rank protocol on the product backend (HIP kernels), several ranks sharing this box's one GPU: gloo as the
    transport (device tensors staged over the host), everything else as under RCCL -- pose blocks, all-gather of the
    survivors' heavy atoms, row tiles of the large passes dealt round-robin with all-reduce(MIN), small passes replicated.
    C3 with two and four ranks, C4 (1M x 50) with three and C5 (500k x 200 atoms, three fragments) with two against the recorded
    oracle results; C2 with three ranks and the
    sharding threshold lowered so that small passes are sharded too, against the oracle run here."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "sharded_gpu_worker.py"), cfg, str(n_poses), str(min_pairs)]
    # a world of ONE runs the same protocol over RCCL ("nccl"), the transport of a real node: its collectives on device tensors, nothing staged
    env = dict(os.environ, SHARD_BACKEND="nccl" if world == 1 else "gloo")
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got["world"] == world and got["ranks_agree"] and got["steps_that_differ"] == 0
    # the three forms of the front half (pose blocks + all-gather / every rank embeds all poses / clash verdicts per block, survivors
    # embedded everywhere) were timed, one was chosen by every
    # rank alike, and each of them, forced, gives the same survivors and evaluation counts
    assert got["forms_agree"]
    # every step above ran its pass loop INSIDE the library (tsc_prune_run_sharded, the host called back for the collectives only);
    # the same loop driven from the host call by call gives the same survivors, evaluation counts and sequence of exchanges
    assert got["loops_agree"] is True
    # ... and with the per-pass exchanges inside the library (areas the ranks map into each other over hipIpc*, one-shot all-reduce kernels) in
    # place of the torch.distributed collectives: identical survivors, evaluation counts and exchange sequence, no exchange timed out
    if world > 1:
        assert got["ipc_agrees"] is True, got["ipc_status"]
    if world > 1 and cfg in ("C3", "C4"):
        assert len(got["partitioned"]) >= 3 and len(got["exchanges"]) >= 2, (got["partitioned"], got["exchanges"])
    if world > 1:           # (a world of one does not time the forms: it has nothing to choose)
        assert got["front_tuning"]["chosen"] in ("shard", "replicate", "hybrid") and len(got["front_tuning"]["ms_per_step"]) == 3
    if cfg == "C5chain" and n_poses > 0:
        # config 5 as a chain over three ranks (the conformational search cut into blocks of the angle table, the kept candidates
        # all-gathered in table order) against the one-GPU chain on the same inputs, which test_c5_chain_... checks against the oracle
        import hashlib

        from tscode_amd.pipeline import CsearchChain
        from tscode_amd.synthetic import make_config
        ens = make_config("C5", n_poses)
        torsions, tmasks = CsearchChain.chain_torsions(ens.frag_coords[0].shape[1], 8, seed=5)
        angle_table = np.random.default_rng(6).choice(np.array([0, 0, 60, 120, 180, 240, 300, 25]), size=(max(200, n_poses // 25), 8)).astype(np.int32)
        one = CsearchChain(ens, torsions, tmasks, angle_table, thresh=1.4, seed=7)
        r1 = one.step()
        one.torch.cuda.synchronize()
        d1 = hashlib.sha256(np.packbits(one.h_keep[:r1["n_pass"]].numpy().astype(bool)).tobytes()).hexdigest()[:16]
        assert (got["n_conformers"], got["n_pass"], got["n_keep"], got["keep_sha256_16"]) == (r1["n_conformers"], r1["n_pass"], r1["n_keep"], d1)
        assert got["pairs_evaluated"] == [s["pairs_evaluated"] for s in r1["stats"]] and 0 < r1["n_conformers"] <= len(angle_table)
    if cfg in ("C3", "C4", "C5", "C5chain") and n_poses == 0:   # full size: against the recorded oracle run (1M x 50 with three ranks: 206 398 survivors)
        key = {"C3": "C3:100000:mode0", "C4": "C4:1000000:mode0", "C5": "C5:500000:mode0", "C5chain": "C5chain:500000:mode0"}[cfg]
        exp = json.load(open(os.path.join(root, "tests", "golden", "expected_full.json"))).get(key)
        if exp is None:
            pytest.skip(f"{key} is not recorded in tests/golden/expected_full.json (tools/record_c5chain.py)")
        assert (got["n_pass"], got["n_keep"], got["keep_sha256_16"]) == (exp["n_pass"], exp["n_keep"], exp["keep_sha256_16"])
        assert got["pairs_evaluated"] == [p["pairs_evaluated"] for p in exp["passes"]]
        assert exp.get("n_conformers", got.get("n_conformers")) == got.get("n_conformers")
    elif cfg != "C5chain":
        from tscode_amd.synthetic import make_config
        ens = make_config(cfg)
        poses = ens.poses()
        poses = poses[oracle.compenetration_mask(poses, ens.ids, 1.5, 0)]
        ref = oracle.prune_heavy(np.ascontiguousarray(poses[:, ens.atomnos != 1]), 0.5, mode=0, row_parallel=True)
        assert got["n_pass"] == len(poses) and got["n_keep"] == int(ref["mask"].sum())
        assert got["pairs_evaluated"] == [s["pairs_evaluated"] for s in ref["stats"]]
        assert len(got["global_path_passes"]) >= 3            # the lowered threshold really sharded passes
