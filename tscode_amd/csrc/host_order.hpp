// host_order.hpp -- HOST code: which member of a cluster the reference's graph step keeps.
//
// prune_conformers_tfd (tscode/numba_functions.py:201-226) and prune_by_moment_of_inertia (tscode/optimization_methods.py:
// 341-358) turn the similar pairs of a chunk into a graph and keep, of every connected component, `tuple(subgraph.nodes)[0]`.
// That is not "the first structure": with networkx 3.x it is
//   * the first node in the ITERATION ORDER OF A PYTHON SET -- the set networkx's show_nodes filter builds from the component
//     (itself a set, filled in breadth-first order by _plain_bfs) -- whenever the component holds less than half of the graph's
//     nodes (coreviews.FilterAtlas.__iter__), and
//   * the component's first node in the graph's node-insertion order otherwise;
// the graph's node and adjacency orders in turn follow the iteration order of the SET OF TUPLES the matches were collected in.
// So the survivor is a function of CPython's hash-table layout.  The Python side of this package reproduces it by building the
// very same objects (tscode_amd/numba_functions.py), which is exact but costs microseconds per edge in the interpreter
// (0.36-0.56 s for 50 000 structures).  This file re-plays the same insertions on plain arrays: CPython's set
// (Objects/setobject.c: open addressing, LINEAR_PROBES = 9, perturbation shift 5, growth by 4x (2x beyond 50 000 entries) once
// fill * 5 >= mask * 3, re-insertion in table order), its tuple hash (Objects/tupleobject.c, the xxHash-style combination used
// since 3.8) with hash(int) = int, dict insertion order, networkx's add_edges_from / connected_components / _plain_bfs /
// subgraph.  The emulation is CHECKED against the real objects on random graphs when it is first used
// (numba_functions._host_graph_step_ok); if the interpreter or networkx at hand behaves differently, the Python path is used.
#pragma once
#include <cstdint>
#include <vector>

namespace tsc_host {

struct PySet {  // keys: an opaque 64-bit id (an int's value, or the index of a tuple); equal ids never re-enter
    std::vector<int64_t> key;
    std::vector<uint64_t> hash;
    std::vector<uint8_t> used;
    uint64_t mask = 7, fill = 0;
    PySet() : key(8), hash(8), used(8, 0) {}
    void insert_clean(std::vector<int64_t> &k, std::vector<uint64_t> &h, std::vector<uint8_t> &u, uint64_t m, int64_t kk, uint64_t hh) const {
        uint64_t perturb = hh, i = hh & m;
        for (;;) {
            if (!u[i]) break;
            if (i + 9 <= m) {
                bool found = false;
                for (uint64_t j = 1; j <= 9; ++j)
                    if (!u[i + j]) {
                        i += j, found = true;
                        break;
                    }
                if (found) break;
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & m;
        }
        k[i] = kk, h[i] = hh, u[i] = 1;
    }
    void resize(uint64_t minused) {
        uint64_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<int64_t> k(newsize);
        std::vector<uint64_t> h(newsize);
        std::vector<uint8_t> u(newsize, 0);
        for (uint64_t e = 0; e <= mask; ++e)
            if (used[e]) insert_clean(k, h, u, newsize - 1, key[e], hash[e]);
        key.swap(k), hash.swap(h), used.swap(u);
        mask = newsize - 1;
    }
    void add(int64_t kk, uint64_t hh) {  // set_add_entry for a key that is not in the set yet
        uint64_t perturb = hh, i = hh & mask;
        uint64_t slot = 0;
        for (;;) {
            uint64_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            bool found = false;
            do {
                if (!used[e]) {
                    slot = e, found = true;
                    break;
                }
                ++e;
            } while (probes--);
            if (found) break;
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
        key[slot] = kk, hash[slot] = hh, used[slot] = 1;
        ++fill;
        if (fill * 5 < mask * 3) return;
        resize(fill > 50000 ? fill * 2 : fill * 4);
    }
    template <typename F>
    void for_each(F f) const {  // iteration = table order
        for (uint64_t e = 0; e <= mask; ++e)
            if (used[e]) f(key[e]);
    }
};

inline uint64_t tuple2_hash(uint64_t a, uint64_t b) {  // hash((a, b)) for small non-negative ints (hash(int) = int)
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5;
    for (uint64_t lane : {a, b}) {
        acc += lane * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 2ULL ^ (P5 ^ 3527539ULL);
    if (acc == ~0ULL) return 1546275796ULL;
    return acc;
}

// One chunk: matches (mi[q], mj[q]) in the order they were added to the Python set (rows ascending), node ids relative to
// the chunk, < d.  Clears keep[off + node] for every node that is not its component's head.  scratch: int[d] filled with -1.
inline void graph_step_chunk(const int64_t *mi, const int64_t *mj, int64_t nm, int64_t off, uint8_t *keep, std::vector<int32_t> &index_of) {
    // matches = set(); matches.add((i, j)) ...                                        numba_functions.py:190
    PySet ms;
    for (int64_t q = 0; q < nm; ++q) ms.add(q, tuple2_hash(uint64_t(mi[q]), uint64_t(mj[q])));
    // nx.Graph(matches): add_edges_from walks the set; a node enters _node / _adj when first met, u before v
    std::vector<int64_t> nodes;
    std::vector<std::vector<int32_t>> adj;
    auto node = [&](int64_t v) -> int32_t {
        int32_t &ix = index_of[size_t(v)];
        if (ix < 0) {
            ix = int32_t(nodes.size());
            nodes.push_back(v);
            adj.emplace_back();
        }
        return ix;
    };
    ms.for_each([&](int64_t q) {
        const int32_t u = node(mi[q]), v = node(mj[q]);
        adj[size_t(u)].push_back(v);  // (a pair occurs once: row i has one first match; (i, j) and (j, i) cannot both occur, j > i)
        adj[size_t(v)].push_back(u);
    });
    const size_t n = nodes.size();
    std::vector<uint8_t> seen(n, 0);
    std::vector<int32_t> level, next;
    for (size_t s = 0; s < n; ++s) {  // connected_components: for v in G, _plain_bfs from every node not seen yet
        if (seen[s]) continue;
        PySet comp;  // seen = {source}; seen.add(w) in breadth-first order
        comp.add(nodes[s], uint64_t(nodes[s]));
        seen[s] = 1;
        size_t members = 1;
        level.assign(1, int32_t(s));
        while (!level.empty()) {
            next.clear();
            for (int32_t v : level)
                for (int32_t w : adj[size_t(v)])
                    if (!seen[size_t(w)]) {
                        seen[size_t(w)] = 1;
                        comp.add(nodes[size_t(w)], uint64_t(nodes[size_t(w)]));
                        next.push_back(w);
                        ++members;
                    }
            level.swap(next);
        }
        // G.subgraph(c): show_nodes(set(nbunch_iter(c))) -- a new set filled in c's iteration order
        int64_t head = nodes[s];  // at least half of the graph: the graph's own node order, whose first member of c is the source
        if (2 * members < n) {
            PySet shown;
            comp.for_each([&](int64_t v) { shown.add(v, uint64_t(v)); });
            bool first = true;
            shown.for_each([&](int64_t v) {
                if (first) head = v, first = false;
            });
        }
        comp.for_each([&](int64_t v) {
            if (v != head) keep[off + v] = 0;
        });
    }
    for (int64_t v : nodes) index_of[size_t(v)] = -1;
}

}  // namespace tsc_host
