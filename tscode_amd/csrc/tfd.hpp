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
__global__ __launch_bounds__(256) void k_torsion_fingerprints(const double *__restrict__ coords, int64_t N, int n, const int32_t *__restrict__ quads,
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
__global__ __launch_bounds__(256) void k_tfd_first_similar(const float *__restrict__ tf, int64_t N, int T, int64_t d, int64_t k, int64_t num_active,
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
// The filter is sequential by definition, but most of its work is not: the list is walked in super-blocks of TG_SUPER
// candidates, and per super-block
//   k_tfd_greedy_prior   (whole GPU) marks the candidates that are similar to a structure kept in an EARLIER super-block
//                        (candidates x slices of the kept list; a lane stops at its first hit);
//   k_tfd_greedy_block   (one workgroup: the verdicts inside a super-block depend on each other) walks it in blocks of 64:
//                        all 64 against what this super-block has kept so far (16 wavefronts, a slice of that list each),
//                        all pairs inside the block into a 64 x 64 bit matrix, then one scalar replay of the greedy order.
// 100 000 fingerprints with 12 000 kept: 1.07 s in a single-workgroup kernel over the whole list, a few milliseconds this way.
// kept_list i32[N] (device scratch) ends up holding the kept indices in order; *n_kept their number (zeroed by the caller).
constexpr int TG_THREADS = 1024;
constexpr int TG_SUPER = 4096;

__global__ __launch_bounds__(256) void k_tfd_greedy_prior(const float *__restrict__ tf, int64_t base, int n_cand, int T, double thresh,
                                                           const int32_t *__restrict__ kept_list, const int32_t *__restrict__ n_kept,
                                                           uint8_t *__restrict__ dead) {
    // blockIdx.x: 64 candidates; the 4 wavefronts of a block and blockIdx.y cut the kept list into gridDim.y * 4 slices
    const int lane = threadIdx.x & 63, c = blockIdx.x * 64 + lane;
    const int slice = blockIdx.y * 4 + (threadIdx.x >> 6), n_slices = gridDim.y * 4;
    const int nk = *n_kept;
    if (c >= n_cand) return;
    const float *a = tf + (base + c) * T;
    int it = 0;
    for (int k = slice; k < nk; k += n_slices, ++it) {
        if ((it & 15) == 15 && dead[c]) return;                  // another slice has settled this candidate (a hint: a stale read only costs work)
        if (tfd_similar_dev(a, tf + int64_t(kept_list[k]) * T, T, thresh)) {
            dead[c] = 1;
            return;
        }
    }
}

__global__ __launch_bounds__(TG_THREADS) void k_tfd_greedy_block(const float *__restrict__ tf, int64_t base, int n_cand, int T, double thresh,
                                                                  const uint8_t *__restrict__ dead_in, uint8_t *__restrict__ accepted,
                                                                  int32_t *__restrict__ kept_list, int32_t *__restrict__ n_kept) {
    __shared__ int s_dead[64];
    __shared__ int s_nk;
    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int nk0 = *n_kept;                                     // kept before this super-block: k_tfd_greedy_prior compared with those
    if (tid == 0) s_nk = nk0;
    __syncthreads();
    for (int b0 = 0; b0 < n_cand; b0 += 64) {
        const int nb = min(64, n_cand - b0);
        const int nk = s_nk;
        if (tid < 64) s_dead[tid] = (tid < nb) ? int(dead_in[b0 + tid]) : 1;
        __syncthreads();
        // (1) candidate `lane` against the structures this super-block has kept so far: slice, slice + 16, ...
        if (lane < nb && !s_dead[lane]) {
            const float *a = tf + (base + b0 + lane) * T;
            bool dead = false;
            for (int k = nk0 + slice; k < nk && !dead; k += TG_THREADS / 64) dead = tfd_similar_dev(a, tf + int64_t(kept_list[k]) * T, T, thresh);
            if (dead) s_dead[lane] = 1;
        }
        __syncthreads();
        // (2) + (3): wavefront 0
        if (slice == 0) {
            const bool dead = lane < nb ? s_dead[lane] != 0 : true;
            unsigned long long sim = 0ull;                      // bit j: candidate `lane` is similar to candidate j < lane of this block
            if (!dead) {
                const float *a = tf + (base + b0 + lane) * T;
                for (int j = 0; j < lane; ++j)
                    if (!s_dead[j] && tfd_similar_dev(a, tf + (base + b0 + j) * T, T, thresh)) sim |= 1ull << j;
            }
            unsigned long long acc = 0ull;
            for (int c = 0; c < nb; ++c) {
                const unsigned lo = __builtin_amdgcn_readlane(unsigned(sim), c), hi = __builtin_amdgcn_readlane(unsigned(sim >> 32), c);
                const unsigned long long m = (static_cast<unsigned long long>(hi) << 32) | lo;
                const int d = __builtin_amdgcn_readlane(int(dead), c);
                if (!d && (m & acc) == 0ull) acc |= 1ull << c;
            }
            if (lane < nb) {
                const bool ok = (acc >> lane) & 1ull;
                accepted[base + b0 + lane] = ok ? 1 : 0;
                if (ok) kept_list[nk + __popcll(acc & ((1ull << lane) - 1ull))] = int32_t(base + b0 + lane);
            }
            if (lane == 0) s_nk = nk + __popcll(acc);
        }
        __syncthreads();                                        // kept_list and s_nk of this block are visible to the next round
    }
    if (tid == 0) *n_kept = s_nk;
}

// flags of a compacted list back onto the full index space: full[idx[r]] = part[r] (full is zeroed by the caller)
__global__ __launch_bounds__(256) void k_scatter_flags(const uint8_t *__restrict__ part, const int32_t *__restrict__ idx, const int32_t *__restrict__ n_dev,
                                                        uint8_t *__restrict__ full) {
    const int n = *n_dev;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) full[idx[r]] = part[r];
}

// offsets of candidate groups in the list of the candidates that passed a filter: out[g] = pos[group_off[g]] (pos = exclusive
// scan of the filter's mask), out[n_groups] = total
__global__ __launch_bounds__(256) void k_group_offsets_after_filter(const int32_t *__restrict__ group_off, int n_groups, const int32_t *__restrict__ pos,
                                                                     int64_t n, const int32_t *__restrict__ total, int32_t *__restrict__ out) {
    for (int g = blockIdx.x * 256 + threadIdx.x; g <= n_groups; g += gridDim.x * 256) out[g] = (g == n_groups || group_off[g] >= n) ? *total : pos[group_off[g]];
}

}  // namespace tsc
