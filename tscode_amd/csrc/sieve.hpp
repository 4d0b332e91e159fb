// sieve.hpp -- K3 with a rotation-invariant descriptor sieve in front of the Kabsch evaluation.
//
// rmsd_and_max_numba (tscode/rmsd_pruning.py:6-41) rotates p onto q ABOUT THE ORIGIN (it never centres),
// and a rotation about the origin keeps every atom's distance from the origin.  With n_a(p) = |p_a|:
//
//     |p_a R - q_a| >= | n_a(p) - n_a(q) |            for every rotation R and atom a, hence
//     h * rmsd(p, q)^2 >= sum_a (n_a(p) - n_a(q))^2  >=  | Q^T (n(p) - n(q)) |^2
//
// for any matrix Q with orthonormal columns.  A pair whose right-hand side exceeds h * thr^2 cannot satisfy
// rmsd < thr (:75), so it is dropped without forming H = p^T q.  Q (KD columns) is taken along the leading
// principal axes of the norm vectors of the ensemble, where they differ most; the choice of Q only affects how
// many pairs are dropped, never the result.  Everything that survives goes through the same sign test and
// explicit-rotation path as the tile kernel (rmsd.hpp), so the verdicts are those of the reference.
//
// Structure of the pass kernel (one wavefront = 16 rows x one column segment, lane = column):
//   screen : per 64-column tile the lane holds its column's descriptor (KD doubles, coalesced load), the row's
//            descriptor is wave-uniform (one scalar load); 2*KD flops per pair; survivors are pushed to a
//            per-wavefront LDS queue (ballot + prefix popcount);
//   drain  : whenever the queue holds 64 pairs, lane l takes pair l: H from the two structures in memory
//            (they sit in L1/L2), sign test, exact path; atomicMin(best[row], column).
// Any number of heavy atoms is supported (no register-resident structure).
#pragma once
#include "common.hpp"
#include "rmsd.hpp"

namespace tsc {

constexpr int KD = 8;           // descriptor dimensions
constexpr int DESC_SAMPLE = 4096;  // structures used to estimate the principal axes
constexpr int DESC_MAX_ATOMS = 256;

// second-moment matrix of the (sampled) norm vectors, with a constant 1 appended:
// M[a][b] = sum_s n_a(s) n_b(s), a, b in [0, h]  (index h = the constant) -> mean and covariance on the host.
// Only the first hd (<= DESC_MAX_ATOMS) atoms of a structure enter the descriptor; any subset keeps the bound valid.
__global__ __launch_bounds__(256) void k_norm_moments(const double *__restrict__ heavy, int64_t n, int h_row, int h, int64_t stride_structs,
                                                       int n_samples, double *__restrict__ M) {
    extern __shared__ __attribute__((aligned(16))) double s_n[];  // [chunk][h + 1]
    const int m = h + 1;
    constexpr int CHUNK = 32;
    const int n_chunks = (n_samples + CHUNK - 1) / CHUNK;
    for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int s0 = ch * CHUNK, ns = min(CHUNK, n_samples - s0);
        for (int e = threadIdx.x; e < ns * m; e += blockDim.x) {
            int s = e / m, a = e - s * m;
            double v = 1.0;
            if (a < h) {
                const double *x = heavy + (int64_t(s0 + s) * stride_structs * h_row + a) * 3;
                v = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            }
            s_n[s * m + a] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
            int a = e / m, b = e - a * m;
            if (b < a) continue;
            double acc = 0.0;
            for (int s = 0; s < ns; ++s) acc += s_n[s * m + a] * s_n[s * m + b];
            atomicAdd(&M[e], acc);
        }
        __syncthreads();
    }
}

// D[i][k] = sum_a Q[k][a] * |x_ia|  and  G[i] = sum_a |x_ia|^2 for every structure (original index space)
__global__ __launch_bounds__(256) void k_descriptors(const double *__restrict__ heavy, int64_t n, int h, int hd, const double *__restrict__ Q,
                                                      double *__restrict__ D, double *__restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) double s_q[];  // [KD][hd]
    for (int e = threadIdx.x; e < KD * hd; e += blockDim.x) s_q[e] = Q[e];
    __syncthreads();
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *x = heavy + i * h * 3;
    double d[KD], g = 0.0;
#pragma unroll
    for (int k = 0; k < KD; ++k) d[k] = 0.0;
    for (int a = 0; a < h; ++a) {
        double n2 = x[3 * a] * x[3 * a] + x[3 * a + 1] * x[3 * a + 1] + x[3 * a + 2] * x[3 * a + 2];
        g += n2;
        if (a < hd) {
            double nr = sqrt(n2);
#pragma unroll
            for (int k = 0; k < KD; ++k) d[k] = fma(s_q[k * hd + a], nr, d[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < KD; ++k) D[i * KD + k] = d[k];
    G[i] = g;
}

// per pass: descriptors of the active structures in the two layouts the sieve reads
//   Dr[r][KD] (rows, scalar loads)   Dc[k][ld] (columns, lane = column)
__global__ __launch_bounds__(256) void k_compact_desc(const double *__restrict__ D, const int32_t *__restrict__ act, int n_active,
                                                       double *__restrict__ Dr, double *__restrict__ Dc, int64_t ld) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;  // e = k * n_active_pad + r would need a division; use 2D split below
    int r = e;
    if (r >= n_active) return;
    const double *src = D + int64_t(act[r]) * KD;
#pragma unroll
    for (int k = 0; k < KD; ++k) {
        double v = src[k];
        Dr[int64_t(r) * KD + k] = v;
        Dc[int64_t(k) * ld + r] = v;
    }
}

struct SieveArgs {
    long long ld;
    int n_active;
    int h;
    int n_tiles;
    int tile_begin;
    int tile_stride;
    int seg_cols;
    double thr, maxdev_thr;
    double half_h_thr2;   // h * thr^2 / 2
    double desc_limit;    // h * thr^2 * (1 + 1e-9): squared descriptor distance above which a pair is dropped
};

// H = p^T q and the sign test / exact path for one pair read from memory.  Returns true iff the pair is
// similar in the reference's sense (rmsd < thr and maxdev < 2 thr, rmsd_pruning.py:75); *exact_taken tells
// whether the explicit-rotation path ran.
__device__ inline bool pair_is_similar(const double *__restrict__ p, const double *__restrict__ q, int h, double Gp, double Gq,
                                       double half_h_thr2, double thr, double maxdev_thr, bool &exact_taken) {
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < h; ++a) {
        const double px = p[3 * a], py = p[3 * a + 1], pz = p[3 * a + 2];
        const double qx = q[3 * a], qy = q[3 * a + 1], qz = q[3 * a + 2];
        H[0] = fma(px, qx, H[0]), H[1] = fma(px, qy, H[1]), H[2] = fma(px, qz, H[2]);
        H[3] = fma(py, qx, H[3]), H[4] = fma(py, qy, H[4]), H[5] = fma(py, qz, H[5]);
        H[6] = fma(pz, qx, H[6]), H[7] = fma(pz, qy, H[7]), H[8] = fma(pz, qz, H[8]);
    }
    exact_taken = false;
    if (certainly_dissimilar(H, 0.5 * (Gp + Gq) - half_h_thr2)) return false;
    exact_taken = true;
    double rm, md;
    rmsd_and_max_pair(p, q, h, rm, md);
    return rm < thr && md < maxdev_thr;
}

template <int TI>
__global__ __launch_bounds__(256) void k_rmsd_sieve(const double *__restrict__ heavy, const int32_t *__restrict__ act,
                                                     const double *__restrict__ Gall, const double *__restrict__ Dr,
                                                     const double *__restrict__ Dc, const int32_t *__restrict__ cend,
                                                     int32_t *__restrict__ best, unsigned long long *__restrict__ counters, SieveArgs a) {
    static_assert(TI <= 16, "queue entries keep the row in 4 bits");
    __shared__ unsigned s_queue[4][128];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * 4 + wid;
    const int tile = a.tile_begin + slot * a.tile_stride;
    if (tile >= a.n_tiles) return;
    const int r0 = tile * TI;
    const int nrows = min(TI, a.n_active - r0);
    const int seg_lo = ((r0 + 1) & ~63) + int(blockIdx.y) * a.seg_cols;
    const int seg_hi = seg_lo + a.seg_cols;

    int my_cend = 0, my_best = 0;
    if (lane < nrows) {
        my_cend = cend[r0 + lane];
        my_best = __hip_atomic_load(&best[r0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool live0 = lane < nrows && my_cend > max(r0 + lane + 1, seg_lo) && my_best >= seg_lo;
    unsigned alive = unsigned(__ballot(live0));
    if (!alive) return;
    int cmax = live0 ? my_cend : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
    cmax = min(__builtin_amdgcn_readfirstlane(cmax), seg_hi);

    const int h3 = a.h * 3;
    unsigned *queue = s_queue[wid];
    int qn = 0;
    unsigned long long n_screened = 0, n_eval = 0, n_exact = 0;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // evaluate queue entries [base, base + cnt), cnt <= 64, one per lane
    auto drain = [&](int base, int cnt) {
        bool sim = false, exact = false;
        int t = 0, col = 0;
        if (lane < cnt) {
            const unsigned e = queue[base + lane];
            t = int(e >> 16);
            col = seg_lo + int(e & 0xffffu);
            const int r = r0 + t;
            const int64_t i = act[r], j = act[col];
            sim = pair_is_similar(heavy + i * h3, heavy + j * h3, a.h, Gall[i], Gall[j], a.half_h_thr2, a.thr, a.maxdev_thr, exact);
            if (sim) atomicMin(&best[r], col);
        }
        n_eval += cnt;
        n_exact += __popcll(__ballot(exact));
        unsigned long long sm = __ballot(sim);
        while (sm) {  // rows that found a similar column stop being screened (the reference returns there, :75-77)
            const int l = __ffsll((long long)sm) - 1;
            sm &= sm - 1;
            alive &= ~(1u << __builtin_amdgcn_readlane(t, l));
        }
    };

    for (int c0 = seg_lo; c0 < cmax && alive; c0 += 64) {
        const int col = c0 + lane;
        double dq[KD];
#pragma unroll
        for (int k = 0; k < KD; ++k) dq[k] = Dc[int64_t(k) * a.ld + col];
        for (int t = 0; t < nrows; ++t) {
            if (!((alive >> t) & 1u)) continue;
            const int r = r0 + t;
            const int ce = __builtin_amdgcn_readlane(my_cend, t);
            if (ce <= c0 || r >= c0 + 63) continue;
            const double *__restrict__ dr = Dr + int64_t(r) * KD;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < KD; ++k) {
                const double d = dr[k] - dq[k];
                s = fma(d, d, s);
            }
            const bool valid = col > r && col < ce;
            const bool pass = valid && !(s > a.desc_limit);
            n_screened += __popcll(__ballot(valid));
            const unsigned long long m = __ballot(pass);
            if (m) {
                if (pass) queue[qn + __popcll(m & lt_mask)] = (unsigned(t) << 16) | unsigned(col - seg_lo);
                qn += __popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (qn >= 64) {
                    drain(qn - 64, 64);
                    qn -= 64;
                }
            }
        }
    }
    if (qn > 0) drain(0, qn);
    if (lane == 0) {
        atomicAdd(&counters[0], n_eval);
        atomicAdd(&counters[1], n_exact);
        atomicAdd(&counters[2], n_screened);
    }
}

// Orthonormal basis (KD x h, row-major) of the dominant subspace of the covariance of the norm vectors:
// block power iteration with modified Gram-Schmidt.  Any orthonormal Q is valid for the bound; this one is
// merely good.  M is the (h+1) x (h+1) second-moment matrix of k_norm_moments (upper triangle filled).
inline void descriptor_basis(const std::vector<double> &M, int h, int n_samples, std::vector<double> &Q) {
    const int m = h + 1;
    std::vector<double> C(size_t(h) * h);
    const double inv = n_samples > 0 ? 1.0 / n_samples : 0.0;
    for (int a = 0; a < h; ++a)
        for (int b = a; b < h; ++b) {
            double mu_a = M[size_t(a) * m + h] * inv, mu_b = M[size_t(b) * m + h] * inv;
            double c = M[size_t(a) * m + b] * inv - mu_a * mu_b;
            C[size_t(a) * h + b] = C[size_t(b) * h + a] = c;
        }
    const int kd = KD;
    Q.assign(size_t(kd) * h, 0.0);
    // deterministic start: spread unit vectors + a small ramp so that no start vector is orthogonal to everything
    for (int k = 0; k < kd; ++k)
        for (int a = 0; a < h; ++a) Q[size_t(k) * h + a] = ((a % kd) == k ? 1.0 : 0.0) + 1e-3 * ((a * 7 + k * 13) % 11 - 5);
    std::vector<double> Z(size_t(kd) * h);
    auto orthonormalise = [&](std::vector<double> &V) {
        for (int k = 0; k < kd; ++k) {
            double *v = &V[size_t(k) * h];
            for (int rep = 0; rep < 2; ++rep)
                for (int j = 0; j < k; ++j) {
                    const double *u = &V[size_t(j) * h];
                    double d = 0;
                    for (int a = 0; a < h; ++a) d += u[a] * v[a];
                    for (int a = 0; a < h; ++a) v[a] -= d * u[a];
                }
            double nn = 0;
            for (int a = 0; a < h; ++a) nn += v[a] * v[a];
            nn = std::sqrt(nn);
            if (!(nn > 1e-200)) {  // degenerate direction: fall back to a unit vector not yet spanned
                for (int a = 0; a < h; ++a) v[a] = 0.0;
                if (k < h) v[k] = 1.0;
                for (int j = 0; j < k; ++j) {
                    const double *u = &V[size_t(j) * h];
                    double d = 0;
                    for (int a = 0; a < h; ++a) d += u[a] * v[a];
                    for (int a = 0; a < h; ++a) v[a] -= d * u[a];
                }
                nn = 0;
                for (int a = 0; a < h; ++a) nn += v[a] * v[a];
                nn = std::sqrt(nn);
                if (!(nn > 1e-200)) {  // h < KD: no direction left, a zero row keeps the bound valid
                    for (int a = 0; a < h; ++a) v[a] = 0.0;
                    continue;
                }
            }
            for (int a = 0; a < h; ++a) v[a] /= nn;
        }
    };
    orthonormalise(Q);
    for (int it = 0; it < 24; ++it) {
        for (int k = 0; k < kd; ++k)
            for (int a = 0; a < h; ++a) {
                double acc = 0;
                const double *crow = &C[size_t(a) * h];
                const double *q = &Q[size_t(k) * h];
                for (int b = 0; b < h; ++b) acc += crow[b] * q[b];
                Z[size_t(k) * h + a] = acc;
            }
        // keep the previous direction where C annihilates it (zero variance): the bound stays valid
        for (int k = 0; k < kd; ++k) {
            double nn = 0;
            for (int a = 0; a < h; ++a) nn += Z[size_t(k) * h + a] * Z[size_t(k) * h + a];
            if (!(nn > 1e-280))
                for (int a = 0; a < h; ++a) Z[size_t(k) * h + a] = Q[size_t(k) * h + a];
        }
        orthonormalise(Z);
        Q.swap(Z);
    }
}

}  // namespace tsc
