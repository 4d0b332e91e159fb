// tfd.hpp -- SURVEY.md 8(f) N2: torsion fingerprints and the pair search of prune_conformers_tfd.
//
// tscode/numba_functions.py:142-231.  The fingerprint of a structure is the float32 vector of its dihedral angles over a
// list of atom quadruplets (:255-264, algebra.py:24-55); two structures are similar when the wrapped absolute differences
// sum to less than `thresh` degrees (:242-253).  A pass cuts the structure list into k chunks and, inside a chunk, every
// row i looks for the first j > i that is similar (:181-199).  Rows are independent and the reference's cache only skips
// pairs already found dissimilar (it cannot change a verdict), so a pass is one launch: one wavefront per row, 64 columns
// per step, ballot for the first hit.  What happens to the matches (networkx components, "first of the cluster") stays on
// the host, with the reference's own Python objects (tscode_amd/numba_functions.py).
#pragma once
#include "common.hpp"

namespace tsc {

// algebra.py:24-55 in fp64, result in degrees; fp contraction off: the fixture values were produced without FMA and the
// result is rounded to float32 right after
__device__ inline double dihedral_deg_dev(const double *__restrict__ p0, const double *__restrict__ p1, const double *__restrict__ p2,
                                          const double *__restrict__ p3) {
#pragma clang fp contract(off)
    double b0[3], b1[3], b2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) b0[k] = -1.0 * (p1[k] - p0[k]), b1[k] = p2[k] - p1[k], b2[k] = p3[k] - p2[k];
    const double n1 = sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) b1[k] /= n1;
    const double d0 = b0[0] * b1[0] + b0[1] * b1[1] + b0[2] * b1[2], d2 = b2[0] * b1[0] + b2[1] * b1[1] + b2[2] * b1[2];
    double v[3], w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = b0[k] - d0 * b1[k], w[k] = b2[k] - d2 * b1[k];
    const double x = v[0] * w[0] + v[1] * w[1] + v[2] * w[2];
    const double c0 = b1[1] * v[2] - b1[2] * v[1], c1 = b1[2] * v[0] - b1[0] * v[2], c2 = b1[0] * v[1] - b1[1] * v[0];
    const double y = c0 * w[0] + c1 * w[1] + c2 * w[2];
    return atan2(y, x) * (180.0 / 3.14159265358979323846);
}

// _get_tf_mat (:233-240): out f32[N][T]
inline __global__ __launch_bounds__(256) void k_torsion_fingerprints(const double *__restrict__ coords, int64_t N, int n, const int32_t *__restrict__ quads,
                                                               int T, float *__restrict__ out) {
    const int64_t total = N * T;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t s = e / T;
        const int t = int(e - s * T);
        const double *c = coords + s * n * 3;
        const int32_t *q = quads + 4 * t;
        out[e] = float(dihedral_deg_dev(c + 3 * q[0], c + 3 * q[1], c + 3 * q[2], c + 3 * q[3]));
    }
}

// One pass of the pair search (:171-199).  first[i] = absolute index of the first similar j > i inside i's chunk, -1 if
// none or if i lies in no chunk (the last chunk ends at num_active, :175-178).
inline __global__ __launch_bounds__(256) void k_tfd_first_similar(const float *__restrict__ tf, int64_t N, int T, int64_t d, int64_t k, int64_t num_active,
                                                            double thresh, int32_t *__restrict__ first) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); i < N; i += int64_t(gridDim.x) * 4) {
        int64_t step = i / d;
        if (step > k - 1) step = k - 1;
        const int64_t start = d * step;
        int64_t len = (step == k - 1) ? num_active - start : d;
        if (len < 0) len = 0;
        const int64_t i_rel = i - start;
        int32_t found = -1;
        if (i_rel < len) {
            const float *a = tf + i * T;
            for (int64_t j0 = i_rel + 1; j0 < len && found < 0; j0 += 64) {
                const int64_t j = j0 + lane;
                bool sim = false;
                if (j < len) {
                    const float *b = tf + (start + j) * T;
                    double sum = 0.0;
                    for (int t = 0; t < T; ++t) {
                        const float d32 = fabsf(a[t] - b[t]);          // float32, like the reference's arrays
                        double dd = double(d32);
                        if (d32 > 180.0f) dd -= 360.0;                  // the integer term makes the rest float64
                        sum += fabs(dd);
                    }
                    sim = sum < thresh;
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(sim);
                if (m) found = int32_t(start + j0 + (__ffsll((long long)m) - 1));
            }
        }
        if (lane == 0) first[i] = found;
    }
}

// tfd_similarity (:242-253) of two fingerprints: float32 difference, float64 after the wrap, summed in order
__device__ inline bool tfd_similar_dev(const float *__restrict__ a, const float *__restrict__ b, int T, double thresh) {
    double sum = 0.0;
    for (int t = 0; t < T; ++t) {
        const float d32 = fabsf(a[t] - b[t]);
        double dd = double(d32);
        if (d32 > 180.0f) dd -= 360.0;
        sum += fabs(dd);
    }
    return sum < thresh;
}

// is_new_structure of the string embed (tscode/embeds.py:47-69) over a whole ordered list: structure s is kept iff its
// fingerprint is not tfd-similar to the fingerprint of any structure KEPT before it.  The reference's "LRU" never evicts
// (`lru_cache = lru_cache[1:]` rebinds a local name, :66-67), so every kept fingerprint is compared for ever.
// The filter is sequential by definition, but nearly all of its work is not: the list is walked in super-blocks of TG_SUPER
// candidates, and per super-block
//   k_tfd_greedy_prior   (whole GPU) marks the candidates that are similar to a structure kept in an EARLIER super-block
//                        (candidates x slices of a compact copy of the kept fingerprints; a lane stops at its first hit);
//   k_tfd_greedy_pairs   (whole GPU) every pair INSIDE the super-block, whatever the greedy order will make of it: row c of a
//                        TG_SUPER x TG_SUPER bit matrix = the earlier candidates of the super-block c is similar to;
//   k_tfd_greedy_replay  (one wavefront) the greedy order itself, which is all that is sequential: lane l holds the kept bits of
//                        candidates 64 l .. 64 l + 63; a candidate is kept iff the prior pass left it alive and its row meets no
//                        kept bit.  Per block of 64 candidates the rows are in registers (the next block's already in flight),
//                        the test against the blocks before is 64 independent ballots, and what remains serial is a scalar
//                        chain over the block's own 64 x 64 corner.
// 100 000 fingerprints with 12 000 kept: 1.07 s in a single-workgroup kernel over the whole list, 45 ms with the verdicts of a
// super-block walked by one workgroup (round 2), a few milliseconds this way.
// kept_list i32[N] (device scratch) ends up holding the kept indices in order; *n_kept their number (zeroed by the caller).
constexpr int TG_SUPER = 4096;
constexpr int TG_WORDS = TG_SUPER / 64;
constexpr int TG_REG_T = 8;         // fingerprints up to this length sit in registers and are screened in fp32 first
static_assert(TG_WORDS == 64, "k_tfd_greedy_replay: one lane per word of the kept mask");

// tfd_similar_dev with the candidate's fingerprint in registers and a float32 screen in front: min(d, |360 - d|) is the
// reference's wrapped difference (d <= 180: itself; d > 180: |d - 360|), summed in float32 it is off by at most
// T (2^-16 + 2^-23 sum) -- 2^-16 for the rounding of 360 - d, the rest for the additions -- so a sum outside [lo32, hi32] has the
// verdict of the float64 sum; inside (or NaN) the float64 sum decides.  Longer fingerprints take the float64 sum directly.
struct TfdScreen {
    float lo32, hi32;
};
__device__ inline TfdScreen tfd_screen(int T, double thresh) {
    const double abs_err = T * 3.1e-5, rel = T * 2.4e-7;
    TfdScreen sc;
    sc.lo32 = float((thresh - abs_err) * (1.0 - rel) - 1e-6 * fabs(thresh));
    sc.hi32 = float((thresh + abs_err) / (1.0 - rel) + 1e-6 * fabs(thresh));
    return sc;
}
__device__ inline void tfd_load_reg(const float *__restrict__ a, int T, float (&ar)[TG_REG_T]) {
#pragma unroll
    for (int t = 0; t < TG_REG_T; ++t) ar[t] = t < T ? a[t] : 0.0f;
}
__device__ inline bool tfd_similar_reg(const float *__restrict__ a, const float (&ar)[TG_REG_T], const float *__restrict__ b, int T, double thresh,
                                       const TfdScreen sc) {
    if (T <= TG_REG_T) {
        float s32 = 0.0f;
#pragma unroll
        for (int t = 0; t < TG_REG_T; ++t)
            if (t < T) {
                const float d = fabsf(ar[t] - b[t]);
                s32 += fminf(d, fabsf(360.0f - d));
            }
        if (s32 < sc.lo32) return true;
        if (s32 > sc.hi32) return false;
    }
    return tfd_similar_dev(a, b, T, thresh);
}

// blockIdx.x: 256 candidates, one per thread; blockIdx.y: a slice of the kept list.  The kept fingerprints are read from `kept_fp`, the
// compact copy k_tfd_greedy_replay keeps ([n_kept][T], in the order they were kept): the address is the same for every lane, so the
// loads are scalar loads and the comparison's operands sit in scalar registers -- no LDS, no barrier, no indirection through kept_list
// in front of every comparison (the first version walked kept_list -> tf per comparison: 207 us per super-block, an LDS-tiled one 153)
inline __global__ __launch_bounds__(256) void k_tfd_greedy_prior(const float *__restrict__ tf, int64_t base, int n_cand, int T, double thresh,
                                                           const float *__restrict__ kept_fp, const int32_t *__restrict__ n_kept,
                                                           uint8_t *__restrict__ dead) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int nk = *n_kept;
    const int per = (nk + int(gridDim.y) - 1) / int(gridDim.y);
    const int k0 = min(nk, int(blockIdx.y) * per), k1 = min(nk, k0 + per);
    const float *a = tf + (base + min(c, n_cand - 1)) * T;
    float ar[TG_REG_T];
    tfd_load_reg(a, T, ar);
    const TfdScreen sc = tfd_screen(T, thresh);
    bool live = c < n_cand;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {  // four kept fingerprints at a time: their (scalar) loads are in flight together
        if (((k - k0) & 31) == 0) {
            if (live && dead[c]) live = false;  // another slice has settled this candidate (a hint: a stale read only costs work)
            if (!__any(live)) return;
        }
        const float *b = kept_fp + size_t(k) * T;
        const bool s0 = tfd_similar_reg(a, ar, b, T, thresh, sc), s1 = tfd_similar_reg(a, ar, b + T, T, thresh, sc);
        const bool s2 = tfd_similar_reg(a, ar, b + 2 * T, T, thresh, sc), s3 = tfd_similar_reg(a, ar, b + 3 * T, T, thresh, sc);
        if (live && (s0 || s1 || s2 || s3)) {
            dead[c] = 1;
            live = false;
        }
    }
    for (; k < k1; ++k)
        if (live && tfd_similar_reg(a, ar, kept_fp + size_t(k) * T, T, thresh, sc)) {
            dead[c] = 1;
            live = false;
        }
}

// sim[c][w] bit j: candidate c of the super-block is similar to its candidate 64 w + j, for 64 w + j < c (zero elsewhere: every
// word is written); nz[c / 64] bit c % 64: row c has a bit set at all (zeroed by the caller).  blockIdx.x = w (its 64 fingerprints
// in LDS), blockIdx.y * 256 + threadIdx.x = c.
inline __global__ __launch_bounds__(256) void k_tfd_greedy_pairs(const float *__restrict__ tf, int64_t base, int n_cand, int T, double thresh,
                                                           unsigned long long *__restrict__ sim, unsigned long long *__restrict__ nz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_tfd_raw[];
    float *s_tile = reinterpret_cast<float *>(s_tfd_raw);
    const int w = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    const int ncol = max(0, min(64, n_cand - 64 * w));
    const bool staged = 64 * T <= 12288;
    if (staged) {
        for (int e = threadIdx.x; e < ncol * T; e += 256) s_tile[e] = tf[(base + 64 * w) * T + e];
        __syncthreads();
    }
    if (c >= TG_SUPER) return;
    unsigned long long bits = 0ull;
    if (c < n_cand && 64 * w < c) {
        const float *a = tf + (base + c) * T;
        float ar[TG_REG_T];
        tfd_load_reg(a, T, ar);
        const TfdScreen sc = tfd_screen(T, thresh);
        const int jn = min(64, c - 64 * w);
        for (int j = 0; j < jn; ++j) {
            const float *b = staged ? s_tile + j * T : tf + (base + 64 * w + j) * T;
            if (tfd_similar_reg(a, ar, b, T, thresh, sc)) bits |= 1ull << j;
        }
    }
    sim[size_t(c) * TG_WORDS + w] = bits;
    if (bits) atomicOr(&nz[c >> 6], 1ull << (c & 63));
}

__device__ inline unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane(unsigned(v), l), hi = __builtin_amdgcn_readlane(unsigned(v >> 32), l);
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

inline __global__ __launch_bounds__(64) void k_tfd_greedy_replay(const unsigned long long *__restrict__ sim, const unsigned long long *__restrict__ nz,
                                                           int64_t base, int n_cand, const uint8_t *__restrict__ dead_in,
                                                           uint8_t *__restrict__ accepted, int32_t *__restrict__ kept_list, int32_t *__restrict__ n_kept,
                                                           int32_t *__restrict__ n_kept_before) {
    const int lane = threadIdx.x;
    const int n_blocks = (n_cand + 63) / 64;
    // dead bits of block `lane` (candidates past the end count as dead) and which of its rows have any bit set
    unsigned long long D = 0ull;
    for (int i = 0; i < 64; ++i) {
        const int c = 64 * lane + i;
        if (c >= n_cand || dead_in[c]) D |= 1ull << i;
    }
    const unsigned long long NZ = nz[lane];
    unsigned long long W = 0ull;  // kept bits of block `lane`
    unsigned long long row[2][64], corner[2];
#pragma unroll
    for (int i = 0; i < 64; ++i) row[0][i] = sim[size_t(i) * TG_WORDS + lane];
    corner[0] = sim[size_t(lane) * TG_WORDS + 0];
    int nk = *n_kept;
    if (lane == 0) *n_kept_before = nk;  // (k_tfd_greedy_keep copies the fingerprints of what this super-block adds)
    auto block = [&](const int B, const unsigned long long(&cur)[64], unsigned long long(&nxt)[64], const unsigned long long cw,
                     unsigned long long &cw_next) __attribute__((always_inline)) {
        if (B + 1 < n_blocks) {
#pragma unroll
            for (int i = 0; i < 64; ++i) nxt[i] = sim[(size_t(B + 1) * 64 + i) * TG_WORDS + lane];
            cw_next = sim[(size_t(B + 1) * 64 + lane) * TG_WORDS + (B + 1)];  // row 64 (B + 1) + lane, the word of its own block
        }
        const unsigned long long dead = readlane64(D, B);
        const unsigned long long need = readlane64(NZ, B) & ~dead;  // alive rows that are similar to anything before them
        // against the blocks before this one: W of lane B is still zero, so the block's own corner does not take part
        unsigned long long blocked = dead;
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if ((need >> i) & 1ull)
                if (__any((cur[i] & W) != 0ull)) blocked |= 1ull << i;
        // the block's own corner, in order: rows without a bit are kept as they are; a row's bits only name rows before it
        unsigned long long acc = ~dead & ~need;
        for (unsigned long long todo = need & ~blocked; todo; todo &= todo - 1ull) {
            const int i = __ffsll((long long)todo) - 1;
            if ((readlane64(cw, i) & acc) == 0ull) acc |= 1ull << i;
        }
        if (lane == B) W = acc;
        const int c = 64 * B + lane;
        if (c < n_cand) {
            const bool ok = (acc >> lane) & 1ull;
            accepted[base + c] = ok ? 1 : 0;
            if (ok) kept_list[nk + __popcll(acc & ((1ull << lane) - 1ull))] = int32_t(base + c);
        }
        nk += __popcll(acc);
    };
    for (int B = 0; B < n_blocks; B += 2) {
        block(B, row[0], row[1], corner[0], corner[1]);
        if (B + 1 < n_blocks) block(B + 1, row[1], row[0], corner[1], corner[0]);
    }
    if (lane == 0) *n_kept = nk;
}

// the fingerprints of the structures the last super-block kept, appended to the compact copy k_tfd_greedy_prior reads (one workgroup: a few
// hundred fingerprints; inside the one-wavefront replay kernel the same copy cost 20 us per super-block)
inline __global__ __launch_bounds__(256) void k_tfd_greedy_keep(const float *__restrict__ tf, int T, const int32_t *__restrict__ kept_list,
                                                          const int32_t *__restrict__ n_kept_before, const int32_t *__restrict__ n_kept,
                                                          float *__restrict__ kept_fp) {
    const int s0 = *n_kept_before, total = (*n_kept - s0) * T;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int slot = s0 + e / T, t = e - (e / T) * T;
        kept_fp[size_t(slot) * T + t] = tf[int64_t(kept_list[slot]) * T + t];
    }
}

// flags of a compacted list back onto the full index space: full[idx[r]] = part[r] (full is zeroed by the caller)
inline __global__ __launch_bounds__(256) void k_scatter_flags(const uint8_t *__restrict__ part, const int32_t *__restrict__ idx, const int32_t *__restrict__ n_dev,
                                                        uint8_t *__restrict__ full) {
    const int n = *n_dev;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) full[idx[r]] = part[r];
}

// offsets of candidate groups in the list of the candidates that passed a filter: out[g] = pos[group_off[g]] (pos = exclusive
// scan of the filter's mask), out[n_groups] = total
inline __global__ __launch_bounds__(256) void k_group_offsets_after_filter(const int32_t *__restrict__ group_off, int n_groups, const int32_t *__restrict__ pos,
                                                                     int64_t n, const int32_t *__restrict__ total, int32_t *__restrict__ out) {
    for (int g = blockIdx.x * 256 + threadIdx.x; g <= n_groups; g += gridDim.x * 256) out[g] = (g == n_groups || group_off[g] >= n) ? *total : pos[group_off[g]];
}

}  // namespace tsc
