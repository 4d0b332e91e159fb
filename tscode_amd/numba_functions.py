"""Drop-in mirror of the clash functions of tscode/numba_functions.py on the MI355X engine."""

from __future__ import annotations

import numpy as np

from .engine import get_engine

__all__ = ["compenetration_check", "count_clashes", "compenetration_mask", "prune_conformers_tfd", "_get_tf_mat",
           "get_torsion_fingerprint", "tfd_similarity"]

TFD_KS = (5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1)      # numba_functions.py:160-162


def compenetration_mask(coords, ids=None, thresh=1.5, max_clashes=0, return_counts=False):
    """Batched compenetration_check over poses: coords f64[N, n, 3] -> bool[N]
    (the loop of tscode/embedder.py:1243-1248 in one launch)."""
    return get_engine().clash_mask(coords, ids, float(thresh), int(max_clashes), return_counts)


def compenetration_check(coords, ids=None, thresh=1.5, max_clashes=0) -> int:
    """tscode/numba_functions.py:59-105: 1 if the pose has at most max_clashes inter-fragment
    distances below thresh (ids=None: at most max_clashes count_clashes), else 0."""
    coords = np.asarray(coords, dtype=np.float64)
    return int(get_engine().clash_mask(coords[None], ids, float(thresh), int(max_clashes))[0])


def count_clashes(coords) -> int:
    """tscode/numba_functions.py:49-56: ordered atom pairs with 0 < d < 0.5 (each clash counts twice)."""
    coords = np.asarray(coords, dtype=np.float64)
    _, counts = get_engine().clash_mask(coords[None], None, 0.5, 0, return_counts=True)
    return int(counts[0])


# ---- torsion-fingerprint pruning (tscode/numba_functions.py:142-264) --------------------------
def _get_tf_mat(structures, quadruplets):
    """tscode/numba_functions.py:233-240: float32 [N, n_quadruplets] dihedral angles in degrees."""
    return get_engine().torsion_fingerprints(structures, quadruplets)


def get_torsion_fingerprint(coords, quadruplets):
    """tscode/numba_functions.py:255-264 for one structure."""
    return get_engine().torsion_fingerprints(np.asarray(coords, dtype=np.float64)[None], quadruplets)[0]


def tfd_similarity(tfp1, tfp2, thresh=10) -> bool:
    """tscode/numba_functions.py:242-253: True iff the wrapped absolute differences of the two fingerprints sum to < thresh."""
    tf = np.stack([np.asarray(tfp1, dtype=np.float32), np.asarray(tfp2, dtype=np.float32)])
    return bool(get_engine().tfd_first_similar(tf, 2, 1, 2, thresh)[0] == 1)


def _tfd_reject_matches(first, d, k, final_mask):
    """tscode/numba_functions.py:181-226 after the pair search: `first[i]` = absolute index of the first similar j of row i
    (-1: none).  Per chunk the matches go into the same Python objects the reference builds -- a set filled in row order,
    nx.Graph(matches), connected components, `tuple(graph.nodes)[0]` as the member kept -- so that the choice, which
    depends on CPython's set order inside networkx, comes out the same as the reference's in the same interpreter."""
    rows = np.flatnonzero(first >= 0)
    if len(rows) == 0:
        return
    steps = np.minimum(rows // d, int(k) - 1)
    bounds = np.flatnonzero(np.diff(steps)) + 1
    # A chunk with ONE match (i, j) needs no graph: nx.Graph({(i, j)}) inserts i, then j, its only component is the whole
    # graph, and tuple(graph.nodes)[0] is i -- j is rejected.  With many small chunks (the early passes) that is nearly
    # every chunk, and building 10^4 two-node graphs was most of the call's time.
    starts = np.concatenate(([0], bounds))
    sizes = np.diff(np.concatenate((starts, [len(rows)])))
    single = starts[sizes == 1]
    final_mask[first[rows[single]]] = 0
    if len(single) == len(starts):
        return
    multi = sizes > 1
    if _host_graph_step_ok(big=int(sizes.max()) > 40000):
        # the graph step of all the other chunks in ONE library call: the same insertions into CPython's hash tables and
        # networkx's traversals, re-played on plain arrays (tscode_amd/csrc/host_order.hpp)
        sel = np.concatenate([rows[a:a + n] for a, n in zip(starts[multi].tolist(), sizes[multi].tolist())])
        chunk_ptr = np.concatenate(([0], np.cumsum(sizes[multi]))).astype(np.int64)
        chunk_step = steps[starts[multi]]
        chunk_off = (chunk_step.astype(np.int64) * int(d))
        n_total = len(final_mask)
        chunk_len = np.where(chunk_step == int(k) - 1, n_total - chunk_off, int(d)).astype(np.int64)
        off_per_match = np.repeat(chunk_off, sizes[multi])
        _host_graph_step(sel - off_per_match, first[sel].astype(np.int64) - off_per_match, chunk_ptr, chunk_off, chunk_len, final_mask)
        return
    _tfd_reject_graph(first, d, k, final_mask, [rows[a:a + n] for a, n in zip(starts[multi].tolist(), sizes[multi].tolist())])


def _host_graph_step(rel_i, rel_j, chunk_ptr, chunk_off, chunk_len, keep_mask):
    """tsc_host_graph_step: clears keep_mask (bool[N], in place) for every member of a cluster that is not the one the
    reference's graph step keeps."""
    import ctypes as C
    from . import _lib
    rel_i, rel_j, chunk_ptr, chunk_off, chunk_len = (np.ascontiguousarray(a, dtype=np.int64) for a in (rel_i, rel_j, chunk_ptr, chunk_off, chunk_len))
    keep = keep_mask.view(np.uint8)
    assert keep.flags.c_contiguous
    _lib.check(_lib.load().tsc_host_graph_step(_lib.ptr(rel_i), _lib.ptr(rel_j), _lib.ptr(chunk_ptr), _lib.ptr(chunk_off), _lib.ptr(chunk_len),
                                               C.c_int64(len(chunk_off)), C.c_int64(len(keep)), _lib.ptr(keep)))


_HOST_GRAPH_STEP = {}          # {"small": bool, "big": bool}: tsc_host_graph_step reproduces this interpreter + networkx or not


def _host_graph_step_ok(big=False):
    """Is the library's re-play of CPython's set order and networkx's traversals (host_order.hpp) exact HERE?  Checked once per
    process, on random match graphs of the shapes the prunings produce (rows with one match each, j > i), against the
    reference's own expression `tuple(G.subgraph(c).nodes)[0]` on the real objects; graphs beyond 50 000 entries (CPython grows
    its tables differently there) are checked when a chunk of that size first shows up.  Any difference (another interpreter,
    another networkx) switches the library path off for the process; the Python path builds the real objects and is right by
    construction."""
    key = "big" if big else "small"
    if key not in _HOST_GRAPH_STEP:
        import random
        shapes = ((60000, 52000),) if big else ((2, 1), (3, 2), (9, 6), (40, 30), (64, 50), (300, 220), (1000, 900), (2500, 1700))
        try:
            rnd = random.Random(20240)
            ok = True
            for n_nodes, n_rows in shapes:
                rows = sorted(rnd.sample(range(n_nodes - 1), min(n_rows, n_nodes - 1)))
                first = {i: rnd.randrange(i + 1, min(n_nodes, i + 1 + rnd.choice((1, 3, 40, n_nodes)))) for i in rows}
                keep_ref = np.ones(n_nodes, dtype=bool)
                _tfd_reject_graph_python(rows, first, 0, keep_ref)
                keep = np.ones(n_nodes, dtype=bool)
                ri = np.array(rows, dtype=np.int64)
                rj = np.array([first[i] for i in rows], dtype=np.int64)
                _host_graph_step(ri, rj, [0, len(ri)], [0], [n_nodes], keep)
                ok = ok and np.array_equal(keep, keep_ref)
            _HOST_GRAPH_STEP[key] = bool(ok)
        except Exception:
            _HOST_GRAPH_STEP[key] = False
    return _HOST_GRAPH_STEP[key] and (not big or _host_graph_step_ok(False))


def _tfd_reject_graph_python(rows, first, off, keep):
    """One chunk with the reference's own objects and expression (numba_functions.py:190, :209-214, :222-224)."""
    import networkx as nx
    matches = set()
    for i in rows:
        matches.add((i - off, int(first[i]) - off))
    g = nx.Graph(matches)
    for c in nx.connected_components(g):
        group = tuple(g.subgraph(c).nodes)
        for i in group[1:]:
            keep[i + off] = False


_FAST_CLUSTER_HEADS = None       # None: not checked yet; True / False: the shortcut below reproduces this networkx or not


def _cluster_heads_reference(g):
    """[(members, head)] exactly as the reference gets them (numba_functions.py:209-214): head = tuple(subgraph.nodes)[0]."""
    import networkx as nx
    out = []
    for c in nx.connected_components(g):
        group = tuple(g.subgraph(c).nodes)
        out.append((group, group[0]))
    return out


def _cluster_heads_fast(g):
    """The same heads without building a subgraph view per component (50 us each, most of a large run's host time).  A view
    iterates either the set of its nodes, rebuilt from the component in its own iteration order, or -- when that set holds
    at least half of the graph -- the graph's nodes in insertion order (networkx coreviews.FilterAtlas.__iter__); both are
    plain set / dict walks.  Whether this matches the installed networkx is CHECKED once on a random graph
    (_fast_cluster_heads_ok); if it does not, the reference's own expression is used."""
    import networkx as nx
    nodes, n_total = g._node, len(g)
    out = []
    for c in nx.connected_components(g):
        shown = set(n for n in c if n in nodes)
        if 2 * len(shown) < n_total:
            head = next(n for n in shown if n in nodes)
        else:
            head = next(n for n in nodes if n in shown)
        out.append((shown, head))
    return out


def _fast_cluster_heads_ok():
    global _FAST_CLUSTER_HEADS
    if _FAST_CLUSTER_HEADS is None:
        import random

        import networkx as nx
        rnd = random.Random(12345)
        ok = True
        for n_nodes, n_edges in ((2, 1), (5, 3), (40, 25), (300, 260), (300, 900), (2000, 1500)):
            edges = set()
            while len(edges) < n_edges:
                a, b = rnd.randrange(n_nodes), rnd.randrange(n_nodes)
                if a != b:
                    edges.add((min(a, b), max(a, b)))
            g = nx.Graph(edges)
            ref = {frozenset(m): h for m, h in _cluster_heads_reference(g)}
            try:
                fast = {frozenset(m): h for m, h in _cluster_heads_fast(g)}
            except Exception:
                fast = None
            ok = ok and fast == ref
        _FAST_CLUSTER_HEADS = ok
    return _FAST_CLUSTER_HEADS


def _cluster_heads(g):
    return _cluster_heads_fast(g) if _fast_cluster_heads_ok() else _cluster_heads_reference(g)


def _tfd_reject_graph(first, d, k, final_mask, chunks):
    """The graph step of tscode/numba_functions.py:181-226 for the rows (ascending) of each chunk in ``chunks``."""
    import networkx as nx
    for sel in chunks:
        off = d * int(min(sel[0] // d, int(k) - 1))
        matches = set()
        for i_abs in sel.tolist():
            matches.add((i_abs - off, int(first[i_abs]) - off))          # :190
        g = nx.Graph(matches)                                            # :209
        for members, head in _cluster_heads(g):                          # :210-214, "keep the first structure"
            for i in members:
                if i != head:
                    final_mask[i + off] = 0                              # :222-224


def _tfd_schedule(structures, tf_mat, thresh, verbose, first_similar):
    """The pass schedule of tscode/numba_functions.py:160-226 around a pair search `first_similar(tf_mat, d, k, num_active,
    thresh) -> int32[N]` (the engine's kernel in the product; the tests also drive it with the CPU oracle's)."""
    n = structures.shape[0]
    final_mask = np.ones(n, dtype=bool)
    for k in TFD_KS:
        num_active_str = int(np.count_nonzero(final_mask))
        if k == 1 or 5 * k < num_active_str:                              # :166
            d = int(n // k)                                               # :173
            if d == 0:
                continue
            if verbose:
                print(f"Working on subgroups with k={k} ({num_active_str} candidates left) {' ' * 10}", end="\r")
            first = first_similar(tf_mat, d, int(k), num_active_str, thresh)
            _tfd_reject_matches(first, d, int(k), final_mask)
    return structures[final_mask], final_mask


def prune_conformers_tfd(structures, quadruplets, thresh=10, verbose=False):
    """tscode/numba_functions.py:142-231.  Fingerprints and the O(N^2 / k) pair search of every pass run on the GPU (one
    launch each); the schedule, the gate `k == 1 or 5 k < active` (:166) and the graph step stay in Python.  The reference's
    cache_set only skips pairs it already found dissimilar, so it changes no result and is not kept.
    Returns (structures[mask], mask)."""
    structures = np.asarray(structures)
    eng = get_engine()
    tf_mat = eng.torsion_fingerprints(structures, quadruplets)
    return _tfd_schedule(structures, tf_mat, thresh, verbose, eng.tfd_first_similar)
