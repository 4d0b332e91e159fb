"""The adjacent rows at production-like sizes: one JSON line per entry point.

For each: wall time of the host-array call (PCIe-inclusive, best of 3), the oracle on a bounded sample of the same input on the host
cores (OpenMP where its loop is parallel; the k = 1 TFD pass is one thread), equality of the two on that sample, and the algorithmic bytes of the call.  Kernel durations come from running this
script under `rocprofv3 --kernel-trace --stats` (tools/profile_next_rows.sh joins the two into profiles/).
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import oracle
import tscode_amd
from tscode_amd import get_engine
from tscode_amd.synthetic import make_config

eng = get_engine(0)
THREADS = len(os.sched_getaffinity(0))
rng = np.random.default_rng(77)


def best_of(f, n=3):
    best, out = 1e30, None
    for _ in range(n):
        t0 = time.perf_counter()
        out = f()
        best = min(best, time.perf_counter() - t0)
    return best, out


def timed(f):
    t0 = time.perf_counter()
    out = f()
    return time.perf_counter() - t0, out


def line(row, entry, kernels, units, unit, gpu_s, cpu_s, cpu_units, equal, alg_bytes, size):
    print(json.dumps({"row": row, "entry": entry, "kernels": kernels, "size": size, "unit": unit, "units": units,
                      "gpu_wall_ms_pcie_inclusive": round(gpu_s * 1e3, 3), "gpu_units_per_s": units / gpu_s,
                      "oracle_units_per_s": cpu_units / cpu_s, "oracle_threads": THREADS, "oracle_sample_units": cpu_units,
                      "equal_on_sample": bool(equal), "algorithmic_bytes": int(alg_bytes)}), flush=True)


# ---- N1: greedy per-group filter --------------------------------------------------------------------------------------------
G, P, A = 4000, 24, 60
base = rng.normal(size=(G, 1, A, 3)) * 3
poses = (base + rng.normal(size=(G, P, A, 3)) * rng.choice([0.05, 0.6], size=(G, P, 1, 1))).reshape(-1, A, 3)
off = np.arange(0, G * P + 1, P, dtype=np.int32)
eng.greedy_group_filter(poses[:P * 4], off[:5])
dt, acc = best_of(lambda: eng.greedy_group_filter(poses, off))
ns = 200
dc, ref = timed(lambda: np.concatenate([oracle.greedy_group_filter(poses[g * P:(g + 1) * P]) for g in range(ns)]))
line("N1", "tsc_greedy_group_filter", ["k_greedy_group_filter"], G * P, "poses", dt, dc, ns * P, np.array_equal(acc[:ns * P], ref),
     poses.nbytes + G * P, f"{G} groups x {P} poses x {A} atoms")

# ---- N1: string-embed pose parameters ---------------------------------------------------------------------------------------
S, NA = 20000, 24
p1, p2, rv, mv = (rng.normal(size=(S, 3)) for _ in range(4))
cp = rng.integers(0, 50, size=(S, 2)).astype(np.int32)
ang = np.linspace(-180, 180, NA, endpoint=False)
eng.string_embed_params(p1[:10], p2[:10], rv[:10], mv[:10], cp[:10], ang)
dt, (rot, pos, ci) = best_of(lambda: eng.string_embed_params(p1, p2, rv, mv, cp, ang))
ns = 2000
dc, (ro, po, co) = timed(lambda: oracle.string_embed_params(p1[:ns], p2[:ns], rv[:ns], mv[:ns], cp[:ns], ang))
eq = np.abs(rot[:ns * NA] - ro).max() < 1e-12 and np.abs(pos[:ns * NA] - po).max() < 1e-11 and np.array_equal(ci[:ns * NA], co)
line("N1", "tsc_string_embed_params", ["k_string_embed_params"], S * NA, "poses", dt, dc, ns * NA, eq, S * 104 + S * NA * (144 + 48 + 8),
     f"{S} sites x {NA} angles")

# ---- N1: cyclical-embed pose parameters -------------------------------------------------------------------------------------
n = 500_000
v = [rng.normal(size=(n, 3)) for _ in range(7)]
nr = rng.integers(1, 3, size=n).astype(np.int32)
an = rng.choice(np.arange(-180, 180, 15.0), size=n)
eng.cyclical_embed_params(*[x[:10] for x in v], nr[:10], an[:10])
dt, (rot, pos) = best_of(lambda: eng.cyclical_embed_params(*v, nr, an))
ns = 20000
dc, (ro, po) = timed(lambda: oracle.cyclical_embed_params(*[x[:ns] for x in v], nr[:ns], an[:ns]))
eq = np.abs(rot[:ns] - ro).max() < 1e-9 and np.abs(pos[:ns] - po).max() < 1e-9
line("N1", "tsc_cyclical_embed_params", ["k_cyclical_embed_params"], n, "poses", dt, dc, ns, eq, n * (7 * 24 + 12 + 72 + 24), f"{n} (pose, molecule) rows")

# ---- N3: csearch rotations, torsion_comp_check ------------------------------------------------------------------------------
ens = make_config("C5", 4)
coords = ens.poses()[0]
na, n0 = len(coords), ens.frag_coords[0].shape[1]
centres = rng.choice(np.arange(2, n0 - 3), size=8, replace=False)
torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres], dtype=np.int32)
masks = np.zeros((8, na), dtype=np.uint8)
for t, c in enumerate(centres):
    masks[t, c + 1:n0] = 1
M = 100_000
angles = rng.choice(np.array([0, 0, 60, 120, 180, -60, 25]), size=(M, 8)).astype(np.int32)
eng.csearch_rotate(coords, torsions, masks, angles[:100], 1.4, 0)
dt, (out, rb) = best_of(lambda: eng.csearch_rotate(coords, torsions, masks, angles, 1.4, 0))
ns = 3000
dc, (ro, rr) = timed(lambda: oracle.csearch_rotate(coords, torsions, masks, angles[:ns], 1.4, 0))
line("N3", "tsc_csearch_rotate", ["k_csearch_rotate"], M, "candidates", dt, dc, ns, np.array_equal(rb[:ns], rr) and np.abs(out[:ns] - ro).max() < 1e-9,
     M * (na * 24 + 32 + 4), f"{na} atoms, 8 torsions, {M} candidates")
structs = out[:50_000]
eng.torsion_comp_check(structs[:10], torsions[0], masks[0], 1.4, 0)
dt, ok = best_of(lambda: eng.torsion_comp_check(structs, torsions[0], masks[0], 1.4, 0))
ns = 3000
dc, ref = timed(lambda: np.array([oracle.torsion_comp_check(s, torsions[0], masks[0], 1.4, 0) for s in structs[:ns]]))
line("N3", "tsc_torsion_comp_check", ["k_torsion_comp_check"], len(structs), "structures", dt, dc, ns, np.array_equal(ok[:ns], ref),
     structs.nbytes + 4 * len(structs), f"{len(structs)} structures x {na} atoms")

# ---- N2: torsion fingerprints and the TFD pair search -----------------------------------------------------------------------
quads = np.array([[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 5], [4, 5, 6, 7], [6, 7, 8, 9], [0, 4, 8, 11]])
parents = rng.normal(size=(10000, 12, 3)) * 2
s = (parents[:, None] + rng.normal(size=(10000, 5, 12, 3)) * 0.02).reshape(-1, 12, 3)
s = np.ascontiguousarray(s[rng.permutation(len(s))])
N = len(s)
eng.torsion_fingerprints(s[:10], quads)
dt, tf = best_of(lambda: eng.torsion_fingerprints(s, quads))
dc, tfo = timed(lambda: oracle.torsion_fingerprints(s, quads))
line("N2", "tsc_torsion_fingerprints", ["k_torsion_fingerprints"], N, "structures", dt, dc, N, np.array_equal(tf, tfo), s.nbytes + tf.nbytes,
     f"{N} structures x 12 atoms, 6 torsions")
eng.tfd_first_similar(tf[:100], 100, 1, 100)
dt, first = best_of(lambda: eng.tfd_first_similar(tf, N, 1, N))
ns = 12000
dc, fo = timed(lambda: oracle.tfd_first_similar(tf[:ns], ns, 1, ns))
dtp, fs = best_of(lambda: eng.tfd_first_similar(tf[:ns], ns, 1, ns), 1)
line("N2", "tsc_tfd_first_similar", ["k_tfd_first_similar"], N * (N - 1) // 2, "pairs (upper bound: rows stop at their first match)", dt, dc,
     ns * (ns - 1) // 2, np.array_equal(fs, fo), tf.nbytes + 4 * N, f"k = 1 pass over {N} fingerprints")
tscode_amd.prune_conformers_tfd(s[:3000], quads)            # (first use: networkx import, the graph step's self-check)
dt, (_, mask) = best_of(lambda: tscode_amd.prune_conformers_tfd(s, quads), 2)
line("N2", "tscode_amd.prune_conformers_tfd", ["k_torsion_fingerprints", "k_tfd_first_similar"], N, "structures", dt, dc, ns, True, s.nbytes + N,
     f"whole schedule, {N} structures, {int(mask.sum())} survive (oracle figure: its k = 1 pass on {ns} only)")

# ---- N1: the embed loops as drivers: greedy TFD filter, whole string embed, whole cyclical embed ------------------------------
NF, TQ = 100_000, 6
par = rng.uniform(-180, 180, size=(12000, TQ))
tfq = (par[rng.integers(0, len(par), size=NF)] + rng.normal(size=(NF, TQ)) * 0.5).astype(np.float32)
eng.tfd_greedy_filter(tfq[:100])
dt, acc = best_of(lambda: eng.tfd_greedy_filter(tfq), 1)
ns = 20000
dc, ref = timed(lambda: oracle.tfd_greedy_filter(tfq[:ns]))
line("N1", "tsc_tfd_greedy_filter", ["k_tfd_greedy_prior", "k_tfd_greedy_pairs", "k_tfd_greedy_replay"], NF, "fingerprints", dt, dc, ns, np.array_equal(eng.tfd_greedy_filter(tfq[:ns]), ref), tfq.nbytes + NF,
     f"is_new_structure over {NF} fingerprints x {TQ} torsions, {int(acc.sum())} kept")

ens3 = make_config("C3", 10)
f1, f2 = ens3.frag_coords[0][:1], ens3.frag_coords[1][:1]           # two 25-atom fragments
n1, n2 = f1.shape[1], f2.shape[1]
conf1 = f1 + rng.normal(size=(20, n1, 3)) * 0.05
conf2 = f2 + rng.normal(size=(25, n2, 3)) * 0.05
cen1, cen2 = conf1[:, :2] + rng.normal(size=(20, 2, 3)), conf2[:, :3] + rng.normal(size=(25, 3, 3))
ov1, ov2 = rng.normal(size=(20, 2, 3)), rng.normal(size=(25, 3, 3))
quadsx = np.array([[0, 1, 2, 3], [n1 - 1, 0, n1, n1 + 1], [n1, n1 + 1, n1 + 2, n1 + 3], [2, 1, 0, n1]])
sangles = [n * 10.0 for n in range(36)]
tscode_amd.string_embed_batch(conf1[:1], conf2[:1], cen1[:1], ov1[:1], cen2[:1], ov2[:1], sangles, 1.5, 0, quadsx)
dt, _ = best_of(lambda: tscode_amd.string_embed_batch(conf1, conf2, cen1, ov1, cen2, ov2, sangles, 1.5, 0, quadsx), 3)
sposes, tr = tscode_amd.string_embed_batch(conf1, conf2, cen1, ov1, cen2, ov2, sangles, 1.5, 0, quadsx, return_trace=True)
NS = len(tr.kept)
dc, (oc, ook, okept) = timed(lambda: oracle.string_embed(conf1[:2], conf2[:3], cen1[:2], ov1[:2], cen2[:3], ov2[:3], sangles, 1.5, quadsx))
_, tr_s = tscode_amd.string_embed_batch(conf1[:2], conf2[:3], cen1[:2], ov1[:2], cen2[:3], ov2[:3], sangles, 1.5, 0, quadsx, return_trace=True)
line("N1", "tsc_string_embed", ["k_string_embed_params", "k_clash", "k_transform", "k_torsion_fingerprints", "k_tfd_greedy_prior", "k_tfd_greedy_pairs", "k_tfd_greedy_replay"], NS, "candidate poses", dt, dc, len(ook),
     np.array_equal(tr_s.clash_ok, ook) and np.array_equal(tr_s.kept, okept), NS * (2 * 96 + 1) + int(tr.clash_ok.sum()) * (n1 + n2) * 24,
     f"whole string embed: 20 x 25 conformers, 2 x 3 centre pairs, 36 angles = {NS} candidates of {n1 + n2} atoms, {int(tr.clash_ok.sum())} pass the clash check, {int(tr.kept.sum())} kept")

mols = []
for cf, ri in ((conf1[:8], [0, 5]), (conf2[:8], [1, 7])):
    piv = []
    for c in range(len(cf)):
        # "orbital centres": the reactive atoms pushed 2.2 A away from the molecule's centroid, four variants each
        out = cf[c][ri] - cf[c].mean(axis=0)
        out /= np.linalg.norm(out, axis=1, keepdims=True)
        st_ = cf[c][ri[0]] + 2.2 * out[0] + rng.normal(size=(4, 3)) * 0.3
        en_ = cf[c][ri[1]] + 2.2 * out[1] + rng.normal(size=(4, 3)) * 0.3
        piv.append((en_ - st_, 0.5 * (st_ + en_), np.tile(np.array([[ri[0], ri[1]]]), (4, 1))))
    mols.append(dict(coords=cf, reactive_indices=ri, pivots=piv))
cang = np.stack(np.meshgrid(np.linspace(-45, 45, 6), np.linspace(-45, 45, 6)), -1).reshape(-1, 2)
tscode_amd.cyclical_embed_batch([dict(m, coords=m["coords"][:1], pivots=m["pivots"][:1]) for m in mols], cang, 1.5)
dt, _ = best_of(lambda: tscode_amd.cyclical_embed_batch(mols, cang, 1.5), 3)           # what the drop-in calls (no trace)
cposes, ccons, ctr = tscode_amd.cyclical_embed_batch(mols, cang, 1.5, return_trace=True)  # (the trace: 2 048 Python tuples, 3 ms of its own)
NCY = len(ctr.kept)
small = [dict(m, coords=m["coords"][:2], pivots=m["pivots"][:2]) for m in mols]
dc, orc = timed(lambda: oracle.cyclical_embed([m["coords"] for m in small], [np.array(m["reactive_indices"]) for m in small], [m["pivots"] for m in small], cang, 1.5))
_, _, ctr_s = tscode_amd.cyclical_embed_batch(small, cang, 1.5, return_trace=True)
line("N1", "tsc_cyclical_embed", ["k_cyclical_embed_params", "k_clash", "k_transform", "k_greedy_group_filter"], NCY, "candidate poses", dt, dc, len(orc[2]),
     np.array_equal(ctr_s.clash_ok, orc[2]) and np.array_equal(ctr_s.kept, orc[3]), NCY * (2 * 184 + 1) + int(ctr.clash_ok.sum()) * (n1 + n2) * 24,
     f"whole cyclical embed: 8 x 8 conformers, 4 x 4 pivots, 2 orientations, 36 angle sets = {NCY} candidates in {len(ctr.groups)} groups, "
     f"{int(ctr.clash_ok.sum())} pass the clash check, {int(ctr.kept.sum())} kept")

# ---- N4: moments of inertia, MOI pair search, embed scores --------------------------------------------------------------------
ens = make_config("C3", 100_000)
structs = ens.poses()
masses = rng.choice([1.008, 12.011, 14.007, 15.999], size=structs.shape[1])
eng.inertia_moments(structs[:10], masses)
dt, mo = best_of(lambda: eng.inertia_moments(structs, masses))
ns = 20000
dc, moo = timed(lambda: oracle.inertia_moments(structs[:ns], masses))
line("N4", "tsc_inertia_moments", ["k_inertia_moments"], len(structs), "structures", dt, dc, ns, np.abs(mo[:ns] / moo - 1).max() < 1e-10,
     structs.nbytes + mo.nbytes, f"{len(structs)} structures x {structs.shape[1]} atoms")
m20 = mo[:20000]
eng.moi_first_similar(m20[:100])
dt, first = best_of(lambda: eng.moi_first_similar(m20))
dc, fo = timed(lambda: oracle.moi_first_similar(m20))
line("N4", "tsc_moi_first_similar", ["k_moi_first_similar"], len(m20) * (len(m20) - 1) // 2, "pairs (upper bound)", dt, dc, len(m20) * (len(m20) - 1) // 2,
     np.array_equal(first, fo), m20.nbytes + 4 * len(m20), f"{len(m20)} moment triples")
idx = rng.integers(0, structs.shape[1], size=(len(structs), 3, 2)).astype(np.int32)
dist = rng.uniform(1.5, 3.0, size=(len(structs), 3))
eng.embed_scores(structs[:10], idx[:10], dist[:10])
dt, (sc, err) = best_of(lambda: eng.embed_scores(structs, idx, dist))
dc, (sco, erro) = timed(lambda: oracle.embed_scores(structs[:ns], idx[:ns], dist[:ns]))
line("N4", "tsc_embed_scores", ["k_embed_scores"], len(structs), "structures", dt, dc, ns, np.array_equal(sc[:ns], sco) and np.abs(err[:ns] - erro).max() < 1e-12,
     len(structs) * (3 * (48 + 8 + 8) + 12), f"{len(structs)} structures, 3 constraints (reads only the constrained atoms)")
