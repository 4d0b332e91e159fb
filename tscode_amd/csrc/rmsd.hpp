// rmsd.hpp -- K3: Kabsch-rotation RMSD (no centring) and the per-pass kernels of prune_conformers_rmsd.
//
// Reference: tscode/rmsd_pruning.py.  rmsd_and_max_numba(p, q) (:6-41) rotates p onto q with the optimal
// PROPER rotation about the origin (SVD of H = p^T q with the det sign fix, no centring) and returns the
// RMSD and the largest per-atom deviation.  Two formulations of that same optimum are used here:
//
//  * the FILTER (tile kernel, every pair): the optimum satisfies  h * rmsd^2 = Gp + Gq - 2 * lmax,  where
//    lmax = s1 + s2 + sign(det H) * s3 is the largest root of the quartic  P(l) = l^4 + c2 l^2 + c1 l + c0,
//    c2 = -2 |H|_F^2, c1 = -8 det H, c0 = |H|_F^4 - 4 |cof H|_F^2   (characteristic polynomial of Horn's
//    4x4 quaternion matrix).  With L = (Gp + Gq - h thr^2) / 2, rmsd >= thr  <=>  lmax <= L, and since all
//    four roots are real, L > lmax  <=>  P(L) > 0, P'(L) > 0, P''(L) > 0 (for L > 0).  A pair is REJECTED
//    only when the three values exceed a rounding-error bound, so a rejection is always right; every
//    other pair goes to
//    (a second test of the same quartic ACCEPTS near-duplicates outright, see pair_verdict)
//  * the EXACT path (explicit rotation): the quaternion of the optimal rotation is the top eigenvector of
//    Horn's matrix (Newton for the eigenvalue + adjugate column, cyclic Jacobi when that degenerates, fp64);
//    p is rotated, the residual is formed atom by atom and rmsd = sqrt(sum |d|^2 / h), maxdev = max |d| are
//    compared with the thresholds exactly as the reference does (:75).  Values agree with the reference's
//    LAPACK path to ~1e-13.
//
// No MFMA, no LDS tiles for q: a lane owns one column structure q_j in registers (<= 32 heavy atoms after
// padding), the row structure p_i is wavefront-uniform and is read through the scalar cache, so the
// inner loop is 9 v_fma_f64 per atom with one SGPR operand each.
#pragma once
#include "common.hpp"
#include "scan.hpp"

namespace tsc {

// ---------------------------------------------------------------------------------------------------
// exact path

// Cyclic Jacobi on a symmetric 4x4 (full storage, constant indices only so it lives in registers).
// Returns the eigenvector of the largest eigenvalue in q[4].
__device__ inline void top_eigvec4(double A[4][4], double q[4]) {
    double V[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dia += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        }
        if (!(off > 1e-30 * dia)) break;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int r = p + 1; r < 4; ++r) {
                double apq = A[p][r];
                if (apq != 0.0) {
                    double theta = (A[r][r] - A[p][p]) / (2.0 * apq);
                    double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                    t = theta < 0.0 ? -t : t;
                    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {  // columns p, r of A and V
                        double akp = A[k][p], akr = A[k][r];
                        A[k][p] = c * akp - s * akr;
                        A[k][r] = s * akp + c * akr;
                        double vkp = V[k][p], vkr = V[k][r];
                        V[k][p] = c * vkp - s * vkr;
                        V[k][r] = s * vkp + c * vkr;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {  // rows p, r of A
                        double apk = A[p][k], ark = A[r][k];
                        A[p][k] = c * apk - s * ark;
                        A[r][k] = s * apk + c * ark;
                    }
                }
            }
        }
    }
    double best = A[0][0];
    q[0] = V[0][0], q[1] = V[1][0], q[2] = V[2][0], q[3] = V[3][0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        bool b = A[k][k] > best;
        best = b ? A[k][k] : best;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = b ? V[i][k] : q[i];
    }
}

// The same cyclic Jacobi on matrices that live in memory (LDS): A[16], V[16] row-major, dynamic indices.  The register
// version above costs 64 VGPRs for its two matrices; a kernel that wants a small register footprint runs this one for
// the (rare) degenerate pairs, one lane at a time on a 256-byte area per wavefront.
__device__ inline void top_eigvec4_mem(double *A, double *V, double q[4]) {
#pragma nounroll
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
#pragma nounroll
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma nounroll
        for (int i = 0; i < 4; ++i) {
            dia += A[5 * i] * A[5 * i];
#pragma nounroll
            for (int j = i + 1; j < 4; ++j) off += A[4 * i + j] * A[4 * i + j];
        }
        if (!(off > 1e-30 * dia)) break;
#pragma nounroll
        for (int p = 0; p < 3; ++p)
#pragma nounroll
            for (int r = p + 1; r < 4; ++r) {
                const double apq = A[4 * p + r];
                if (apq != 0.0) {
                    const double theta = (A[5 * r] - A[5 * p]) / (2.0 * apq);
                    double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                    t = theta < 0.0 ? -t : t;
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma nounroll
                    for (int k = 0; k < 4; ++k) {  // columns p, r of A and V
                        const double akp = A[4 * k + p], akr = A[4 * k + r];
                        A[4 * k + p] = c * akp - s * akr;
                        A[4 * k + r] = s * akp + c * akr;
                        const double vkp = V[4 * k + p], vkr = V[4 * k + r];
                        V[4 * k + p] = c * vkp - s * vkr;
                        V[4 * k + r] = s * vkp + c * vkr;
                    }
#pragma nounroll
                    for (int k = 0; k < 4; ++k) {  // rows p, r of A
                        const double apk = A[4 * p + k], ark = A[4 * r + k];
                        A[4 * p + k] = c * apk - s * ark;
                        A[4 * r + k] = s * apk + c * ark;
                    }
                }
            }
    }
    int kb = 0;
#pragma nounroll
    for (int k = 1; k < 4; ++k)
        if (A[5 * k] > A[5 * kb]) kb = k;
    for (int i = 0; i < 4; ++i) q[i] = V[4 * i + kb];
}

// 3x3 determinant of the minor of the symmetric 4x4 A that drops row I and column J, with the cofactor sign.
template <int I, int J>
__device__ inline double cofactor4(const double (&A)[4][4]) {
    constexpr int r0 = (I == 0) ? 1 : 0, r1 = (I <= 1) ? 2 : 1, r2 = (I <= 2) ? 3 : 2;
    constexpr int c0 = (J == 0) ? 1 : 0, c1 = (J <= 1) ? 2 : 1, c2 = (J <= 2) ? 3 : 2;
    const double d = A[r0][c0] * (A[r1][c1] * A[r2][c2] - A[r1][c2] * A[r2][c1]) -
                     A[r0][c1] * (A[r1][c0] * A[r2][c2] - A[r1][c2] * A[r2][c0]) +
                     A[r0][c2] * (A[r1][c0] * A[r2][c1] - A[r1][c1] * A[r2][c0]);
    return ((I + J) & 1) ? -d : d;
}

// The explicit-rotation half of rmsd_and_max_numba (rmsd_pruning.py:19-39) given S = p^T q (:15) and the squared
// norms Gp, Gq.  The quaternion of the optimal proper rotation is the top eigenvector of Horn's matrix N(S):
//   fast path : largest root of the characteristic quartic by Newton from the upper bound (Gp+Gq)/2 (monotone), then
//               the eigenvector as the best-conditioned column of adj(N - l I);
//   fallback  : cyclic Jacobi, when the top eigenvalue is (nearly) degenerate and the adjugate collapses
//               (collinear / planar-through-origin / mirror-symmetric cases).
// A group of `lpp` consecutive lanes (a power of two, all converged, all holding the same S) may share one pair:
// lane `sub` of the group then takes atoms sub, sub + lpp, ... of the residual loop and the group reduces.
// Horn's matrix N(S), row-major into 16 doubles
__device__ inline void horn_matrix(const double S[9], double *M) {
    M[0] = S[0] + S[4] + S[8], M[5] = S[0] - S[4] - S[8], M[10] = -S[0] + S[4] - S[8], M[15] = -S[0] - S[4] + S[8];
    M[1] = M[4] = S[5] - S[7], M[2] = M[8] = S[6] - S[2], M[3] = M[12] = S[1] - S[3];
    M[6] = M[9] = S[1] + S[3], M[7] = M[13] = S[6] + S[2], M[11] = M[14] = S[5] + S[7];
}

// Fast way to the rotation quaternion e (not normalised): Newton for the top eigenvalue, then the best-conditioned column
// of adj(N - l I).  Returns false when the top eigenvalue is (nearly) degenerate and the adjugate collapses (collinear /
// planar-through-origin / mirror-symmetric cases): the caller then runs a Jacobi eigen-solver (top_eigvec4 / _mem).
__device__ inline bool rotation_quaternion_fast(const double S[9], double Gp, double Gq, double e[4]) {
    // characteristic quartic l^4 + c2 l^2 + c1 l + c0 (see the sign test below)
    const double F = S[0] * S[0] + S[1] * S[1] + S[2] * S[2] + S[3] * S[3] + S[4] * S[4] + S[5] * S[5] + S[6] * S[6] + S[7] * S[7] + S[8] * S[8];
    const double C0 = S[4] * S[8] - S[5] * S[7], C1 = S[5] * S[6] - S[3] * S[8], C2 = S[3] * S[7] - S[4] * S[6];
    const double C3 = S[2] * S[7] - S[1] * S[8], C4 = S[0] * S[8] - S[2] * S[6], C5 = S[1] * S[6] - S[0] * S[7];
    const double C6 = S[1] * S[5] - S[2] * S[4], C7 = S[2] * S[3] - S[0] * S[5], C8 = S[0] * S[4] - S[1] * S[3];
    const double det = S[0] * C0 + S[1] * C1 + S[2] * C2;
    const double CF = C0 * C0 + C1 * C1 + C2 * C2 + C3 * C3 + C4 * C4 + C5 * C5 + C6 * C6 + C7 * C7 + C8 * C8;
    const double c2 = -2.0 * F, c1 = -8.0 * det, c0 = F * F - 4.0 * CF;
    double lam = 0.5 * (Gp + Gq);
    if (!(lam > 0.0)) return false;
    for (int it = 0; it < 40; ++it) {
        const double l2 = lam * lam;
        const double P = l2 * l2 + c2 * l2 + c1 * lam + c0, P1 = 4.0 * l2 * lam + 2.0 * c2 * lam + c1;
        if (!(P1 > 0.0)) return false;
        const double dl = P / P1;
        lam -= dl;
        if (fabs(dl) <= 1e-15 * fabs(lam)) break;
    }
    double A[4][4];
    {
        double M[16];
        horn_matrix(S, M);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) A[i][j] = M[4 * i + j] - (i == j ? lam : 0.0);
    }
    const double d0 = cofactor4<0, 0>(A), d1 = cofactor4<1, 1>(A), d2 = cofactor4<2, 2>(A), d3 = cofactor4<3, 3>(A);
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) scale = fmax(scale, fabs(A[i][j]));
    const double a0 = fabs(d0), a1 = fabs(d1), a2 = fabs(d2), a3 = fabs(d3);
    const double best = fmax(fmax(a0, a1), fmax(a2, a3));
    if (!(best > 1e-4 * scale * scale * scale)) return false;  // adj(A) = c v v^T with |c| large enough: any big column is v
    if (a0 == best) {
        e[0] = d0, e[1] = cofactor4<0, 1>(A), e[2] = cofactor4<0, 2>(A), e[3] = cofactor4<0, 3>(A);
    } else if (a1 == best) {
        e[0] = cofactor4<1, 0>(A), e[1] = d1, e[2] = cofactor4<1, 2>(A), e[3] = cofactor4<1, 3>(A);
    } else if (a2 == best) {
        e[0] = cofactor4<2, 0>(A), e[1] = cofactor4<2, 1>(A), e[2] = d2, e[3] = cofactor4<2, 3>(A);
    } else {
        e[0] = cofactor4<3, 0>(A), e[1] = cofactor4<3, 1>(A), e[2] = cofactor4<3, 2>(A), e[3] = d3;
    }
    return true;
}

// rmsd and max deviation of p rotated by the quaternion e (w, x, y, z; any length) against q (rmsd_pruning.py:29-39); lpp
// lanes share the pair (atoms sub, sub + lpp, ...) and reduce with a butterfly
__device__ inline void residual_rmsd_maxdev(const double *__restrict__ p, const double *__restrict__ q, int h, const double e[4], double &rmsd,
                                            double &maxdev, int sub = 0, int lpp = 1) {
    const double nn = 1.0 / sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
    const double w = e[0] * nn, x = e[1] * nn, y = e[2] * nn, z = e[3] * nn;
    const double R00 = w * w + x * x - y * y - z * z, R01 = 2 * (x * y - w * z), R02 = 2 * (x * z + w * y);
    const double R10 = 2 * (x * y + w * z), R11 = w * w - x * x + y * y - z * z, R12 = 2 * (y * z - w * x);
    const double R20 = 2 * (x * z - w * y), R21 = 2 * (y * z + w * x), R22 = w * w - x * x - y * y + z * z;
    double ss = 0.0, mx = 0.0;
#pragma unroll 2
    for (int a = sub; a < h; a += lpp) {  // :29-39
        const double px = p[3 * a], py = p[3 * a + 1], pz = p[3 * a + 2];
        const double dx = R00 * px + R01 * py + R02 * pz - q[3 * a];
        const double dy = R10 * px + R11 * py + R12 * pz - q[3 * a + 1];
        const double dz = R20 * px + R21 * py + R22 * pz - q[3 * a + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        ss += d2;
        mx = d2 > mx ? d2 : mx;
    }
    for (int off = lpp >> 1; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        mx = fmax(mx, __shfl_xor(mx, off));
    }
    rmsd = sqrt(ss / double(h));
    maxdev = sqrt(mx);  // max_a sqrt(d2_a) == sqrt(max_a d2_a): sqrt is monotone
}

__device__ inline void exact_rmsd_maxdev(const double *__restrict__ p, const double *__restrict__ q, int h, const double S[9],
                                         double Gp, double Gq, double &rmsd, double &maxdev, int sub = 0, int lpp = 1) {
    double e[4];
    if (!rotation_quaternion_fast(S, Gp, Gq, e)) {
        double N[4][4], M[16];
        horn_matrix(S, M);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) N[i][j] = M[4 * i + j];
        top_eigvec4(N, e);
    }
    residual_rmsd_maxdev(p, q, h, e, rmsd, maxdev, sub, lpp);
}

// rmsd_and_max_numba (rmsd_pruning.py:6-41) for one pair; p, q point at h consecutive xyz triples.
__device__ inline void rmsd_and_max_pair(const double *__restrict__ p, const double *__restrict__ q, int h, double &rmsd,
                                         double &maxdev) {
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Gp = 0.0, Gq = 0.0;
    for (int a = 0; a < h; ++a) {  // :15 cov_mat = p.T @ q
        const double px = p[3 * a], py = p[3 * a + 1], pz = p[3 * a + 2];
        const double qx = q[3 * a], qy = q[3 * a + 1], qz = q[3 * a + 2];
        S[0] += px * qx, S[1] += px * qy, S[2] += px * qz;
        S[3] += py * qx, S[4] += py * qy, S[5] += py * qz;
        S[6] += pz * qx, S[7] += pz * qy, S[8] += pz * qz;
        Gp += px * px + py * py + pz * pz;
        Gq += qx * qx + qy * qy + qz * qz;
    }
    exact_rmsd_maxdev(p, q, h, S, Gp, Gq, rmsd, maxdev);
}

__global__ __launch_bounds__(256) void k_rmsd_pairs(const double *__restrict__ heavy, int h, const int32_t *__restrict__ pairs,
                                                     int64_t n_pairs, double *__restrict__ rmsd, double *__restrict__ maxdev) {
    for (int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; k < n_pairs; k += int64_t(gridDim.x) * blockDim.x) {
        double r, m;
        rmsd_and_max_pair(heavy + int64_t(pairs[2 * k]) * h * 3, heavy + int64_t(pairs[2 * k + 1]) * h * 3, h, r, m);
        rmsd[k] = r;
        maxdev[k] = m;
    }
}

// Statistics counters of a pass.  Thousands of wavefronts report at the end of a kernel; same-address atomics
// serialise at ~12 ns each (MI355X_MICROARCH.md, "fanin"), which for 10^4 reports is longer than the kernels
// themselves, so the counters are spread over 64 buckets on separate 128-byte lines and summed on the host.
constexpr int CNT_BUCKETS = 64;
enum { CNT_FORMED = 0, CNT_EXACT = 1, CNT_SCREENED = 2, CNT_EVALUATED = 3, CNT_REMOVED = 4, CNT_WORDS = 16 };
struct PassCounters {
    unsigned long long w[CNT_BUCKETS][CNT_WORDS];
};
__device__ inline void count_add(PassCounters *c, unsigned bucket, int what, unsigned long long v) {
    if (v) atomicAdd(&c->w[bucket & (CNT_BUCKETS - 1)][what], v);
}

// Device-resident control block of a prune run.  The host enqueues every pass that COULD run (20 k < N) without
// waiting for counts; k_pass_step evaluates the reference's gate `k == 1 or 20*k < count_nonzero(mask)`
// (rmsd_pruning.py:192) on the device and the kernels of a pass that is gated off return immediately.
struct PruneState {
    int n_active;  // count_nonzero(mask) after the last finished pass
    int pass_on;   // gate of the pass in flight
    int A;         // active structures entering the pass in flight (== n_active at its start)
    unsigned ticket;  // blocks of k_apply_pass that have finished (the last one closes the pass)
};
struct PassRecord {  // one per schedule slot, read back once at the end of the run
    long long k, n_before, n_after, formed, exact, screened, evaluated, removed;
    int on, algo;
};

// Initial state of a run: out_mask = ones (rmsd_pruning.py:182), empty cache (:183), zeroed bitmaps, records and counters.
__global__ __launch_bounds__(256) void k_init_run(int64_t n, uint8_t *__restrict__ mask, unsigned long long *__restrict__ mbit,
                                                   unsigned long long *__restrict__ dbit, int bit_words, int32_t *__restrict__ n_keys,
                                                   PruneState *__restrict__ st, PassRecord *__restrict__ rec, int n_rec,
                                                   PassCounters *__restrict__ cnt, int32_t *__restrict__ bsum, int n_blocks, int block_items,
                                                   unsigned *__restrict__ dmax_bits, unsigned *__restrict__ zero_words, int n_zero_words,
                                                   int32_t *__restrict__ act, int first_slot, long long first_k, int first_algo, int dbit_extra_words) {
    // first_slot >= 0: the first pass of the schedule is opened here as well (gate of rmsd_pruning.py:192 on the full count,
    // its record), which is all a k_pass_step launch would do at this point
    const int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    unsigned long long *m8 = reinterpret_cast<unsigned long long *>(mask);  // scratch blocks are 256-byte aligned
    for (int64_t e = tid; e < n / 8; e += stride) m8[e] = 0x0101010101010101ull;
    for (int64_t e = (n / 8) * 8 + tid; e < n; e += stride) mask[e] = 1;
    // every entry of the active list is a valid structure index from the start: the pair kernel gathers through
    // act[x] for x up to n without knowing the active count, and the first global pass may come after chunk-local ones
    for (int64_t e = tid; e < n; e += stride) act[e] = int32_t(e);
    for (int64_t e = tid; e < bit_words; e += stride) mbit[e] = 0, dbit[e] = 0;
    for (int64_t e = tid; e < dbit_extra_words; e += stride) dbit[bit_words + e] = 0;  // the summary bitmap behind dbit
    unsigned long long *c = &cnt->w[0][0];
    for (int64_t e = tid; e < CNT_BUCKETS * CNT_WORDS; e += stride) c[e] = 0;
    for (int64_t e = tid; e < n_zero_words; e += stride) zero_words[e] = 0;  // tickets of the chunk-local pass kernel
    char *r = reinterpret_cast<char *>(rec);
    for (int64_t e = tid; e < int64_t(n_rec) * int64_t(sizeof(PassRecord)); e += stride)
        if (int(e / int64_t(sizeof(PassRecord))) != first_slot) r[e] = 0;  // (the first pass's record is written whole below)
    // per-block counts of the mask for the exclusive scan of every pass; k_apply_pass keeps them current
    for (int64_t e = tid; e < n_blocks; e += stride) {
        const int64_t lo = e * block_items;
        bsum[e] = int32_t(lo >= n ? 0 : (n - lo < block_items ? n - lo : block_items));
    }
    if (tid == 0) {
        n_keys[0] = 0;
        if (dmax_bits) *dmax_bits = 0;  // running maximum of the descriptor build (sieve.hpp)
        st->n_active = int(n), st->pass_on = 0, st->A = int(n), st->ticket = 0;
        if (first_slot >= 0) {
            const int on = (first_k == 1 || 20 * first_k < (long long)n) ? 1 : 0;
            PassRecord &fr = rec[first_slot];
            fr.k = first_k, fr.n_before = fr.n_after = (long long)n, fr.on = on, fr.algo = first_algo;
            fr.formed = fr.exact = fr.screened = fr.evaluated = fr.removed = 0;
            st->pass_on = on;
        }
    }
}

// Closes the pass in slot `prev` (sums its counters, updates n_active) and opens the pass in slot `cur` (gate, A,
// zeroed counters and cache-view bitmap).  prev / cur = -1: nothing to close / open.  Runs in ONE block of 256
// threads: the last block of k_apply_pass (below) or, for the first pass of a run, k_pass_step.
struct StepArgs {
    int prev, cur;
    long long k_cur;
    int algo_cur;
    int bit_words;
    int prev_algo;  // >= 0: the kernel that really ran the closing pass (recorded in its PassRecord), -1: as opened
};

__device__ inline void pass_step_block(PruneState *__restrict__ st, PassCounters *__restrict__ cnt, PassRecord *__restrict__ rec, const StepArgs &sa,
                                       unsigned long long *__restrict__ dbit, unsigned long long *s_sum /* 32 words of LDS */) {
    const bool closing = sa.prev >= 0 && st->pass_on != 0;
    if (closing) {
        // 256 threads, two loads each (word t % 8 of buckets t / 8 and t / 8 + 32), all in flight together; the buckets
        // were written by other blocks' atomics, so they are read at the L2, not through this CU's cache
        static_assert(CNT_BUCKETS == 64 && CNT_REMOVED < 8, "layout of the parallel counter sum");
        const int w = threadIdx.x & 7, b0 = threadIdx.x >> 3;
        unsigned long long v = __hip_atomic_load(&cnt->w[b0][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                               __hip_atomic_load(&cnt->w[b0 + 32][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) < 8) s_sum[8 * (threadIdx.x >> 6) + w] = v;  // s_sum[4][8]: one partial per wavefront
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (closing) {
            PassRecord &r = rec[sa.prev];
            for (int w = 0; w < 8; ++w) s_sum[w] += s_sum[8 + w] + s_sum[16 + w] + s_sum[24 + w];
            r.formed = (long long)s_sum[CNT_FORMED], r.exact = (long long)s_sum[CNT_EXACT], r.screened = (long long)s_sum[CNT_SCREENED];
            r.evaluated = (long long)s_sum[CNT_EVALUATED], r.removed = (long long)s_sum[CNT_REMOVED];
            st->n_active -= int(s_sum[CNT_REMOVED]);
            r.n_after = st->n_active;
            if (sa.prev_algo >= 0) r.algo = sa.prev_algo;
        }
        int on = 0;
        if (sa.cur >= 0) {
            on = (sa.k_cur == 1 || 20 * sa.k_cur < (long long)st->n_active) ? 1 : 0;  // rmsd_pruning.py:192
            PassRecord &r = rec[sa.cur];
            r.k = sa.k_cur, r.n_before = st->n_active, r.n_after = st->n_active, r.on = on, r.algo = sa.algo_cur;
            r.formed = r.exact = r.screened = r.evaluated = r.removed = 0;
        }
        st->A = st->n_active;
        st->pass_on = on;
    }
    __syncthreads();
    unsigned long long *c = &cnt->w[0][0];
    for (int e = threadIdx.x; e < CNT_BUCKETS * CNT_WORDS; e += 256) c[e] = 0;
    if (sa.cur >= 0)
        for (int e = threadIdx.x; e < sa.bit_words; e += 256) dbit[e] = 0;
}

__global__ __launch_bounds__(256) void k_pass_step(PruneState *__restrict__ st, PassCounters *__restrict__ cnt, PassRecord *__restrict__ rec,
                                                    StepArgs sa, unsigned long long *__restrict__ dbit) {
    __shared__ unsigned long long s_sum[32];
    pass_step_block(st, cnt, rec, sa, dbit, s_sum);
}

// ---------------------------------------------------------------------------------------------------
// per-pass helper kernels (the pass sequence is in tscode_hip.hip, tsc_prune_pass_local / tsc_prune_pass_finish)

struct PassGeom {
    int n;   // structures (< 2^31)
    int k;   // chunks in this pass
    int cs;  // chunk size n // k (rmsd_pruning.py:136)
};

__device__ inline void chunk_of(const PassGeom &g, int64_t i, int64_t &first, int64_t &last) {
    int c = int(i) / g.cs;  // 32-bit division: n < 2^31
    if (c >= g.k) c = g.k - 1;
    first = int64_t(c) * g.cs;                        // :140
    last = (c == g.k - 1) ? g.n : first + g.cs;       // :141-144
}

// Cache view of one pass: a key (a, b) = (first, first + (j - i)) (:65) can be hit only where a is a chunk
// start of this pass and b lies inside that chunk; then it is hit by exactly the pairs with a + (j-i) == b.
// Cache view of a pass from the key list, shared by k_dbit_build and k_open_pass: key (a, b) sets bit b of dbit when a is a
// chunk start of this pass and b lies in that chunk, and the bit of b's 1024-bit block in the summary dsum (k_stop_scan
// walks only the non-empty blocks).  The summary of a run of 57k structures is ONE word: its bits are OR-ed across the
// wavefront first and sent by one lane, or every applicable key would queue on the same address.
__device__ inline void build_cache_view(const PassGeom &g, const int32_t *__restrict__ key_a, const int32_t *__restrict__ key_b, int nk,
                                        unsigned long long *__restrict__ dbit, unsigned long long *__restrict__ dsum) {
    const int lane = threadIdx.x & 63;
    const int stride = gridDim.x * blockDim.x;
    for (int q0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; q0 < nk; q0 += stride) {  // wave-uniform bounds
        const int q = q0 + lane;
        bool ok = false;
        int b = 0;
        if (q < nk) {
            const int a = key_a[q];
            b = key_b[q];
            const int c = a / g.cs;
            if (c * g.cs == a && c < g.k) {
                const int last = (c == g.k - 1) ? g.n : g.cs * (c + 1);
                ok = b < last;
            }
        }
        if (ok) atomicOr(&dbit[b >> 6], 1ull << (b & 63));
        for (unsigned long long left = __ballot(ok); left;) {
            const int word = __shfl(b >> 16, __ffsll((long long)left) - 1);
            const bool same = ok && (b >> 16) == word;
            unsigned long long bits = same ? 1ull << ((b >> 10) & 63) : 0ull;
            for (int off = 32; off > 0; off >>= 1) bits |= __shfl_xor(bits, off);
            const unsigned long long grp = __ballot(same);
            if (lane == __ffsll((long long)grp) - 1) atomicOr(&dsum[word], bits);
            left &= ~grp;
        }
    }
}

// Cache view of a pass on its own (the chunk-local pass kernel, local_pass.hpp, needs nothing else from k_open_pass):
// key (a, b) sets bit b when a is a chunk start of this pass and b lies in that chunk.
// It also takes the SNAPSHOT of the mask that the pass reads (mbit, one bit per structure): the workgroups of the chunk-local
// kernel remove rows from the byte mask while others, possibly of the same chunk, have not started yet -- every row of a
// pass must see the mask as it was when the pass began (rmsd_pruning.py:151-157), whatever order the workgroups run in.
__global__ __launch_bounds__(256) void k_dbit_build(PassGeom g, int use_cache, const int32_t *__restrict__ key_a, const int32_t *__restrict__ key_b,
                                                     const int32_t *__restrict__ n_keys, unsigned long long *__restrict__ dbit,
                                                     const PruneState *__restrict__ st, unsigned long long *__restrict__ dsum,
                                                     const uint8_t *__restrict__ mask, unsigned long long *__restrict__ mbit) {
    if (st->pass_on == 0) return;
    const int n64 = (g.n + 63) & ~63;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n64; t += gridDim.x * blockDim.x) {  // (wave-uniform bounds)
        const unsigned long long w = __builtin_amdgcn_ballot_w64(t < g.n && mask[t] != 0);
        if ((threadIdx.x & 63) == 0) mbit[t >> 6] = w;
    }
    if (use_cache) build_cache_view(g, key_a, key_b, *n_keys, dbit, dsum);
}

// Opens the data of a pass in one launch: ranks of the active structures, their index list and the mask as bits
// (the second phase of the exclusive scan, scan.hpp; k_apply_pass keeps the per-block counts current), and the
// cache view of the pass: key (a, b) sets bit b when a is a chunk start of this pass and b lies in that chunk.
// Grid: scan_grid_blocks(n) blocks of SCAN_THREADS.
__global__ __launch_bounds__(SCAN_THREADS) void k_open_pass(PassGeom g, int use_cache, const PruneState *__restrict__ st,
                                                             const uint8_t *__restrict__ mask, const int32_t *__restrict__ bsum,
                                                             int32_t *__restrict__ pos, int32_t *__restrict__ act_idx,
                                                             uint8_t *__restrict__ mbit_bytes, int32_t *__restrict__ total_out,
                                                             const int32_t *__restrict__ key_a, const int32_t *__restrict__ key_b,
                                                             const int32_t *__restrict__ n_keys, unsigned long long *__restrict__ dbit,
                                                             unsigned long long *__restrict__ dsum) {
    __shared__ int s_w[SCAN_THREADS / WAVE];
    __shared__ int s_o[SCAN_THREADS / WAVE];
    if (st->pass_on == 0) return;
    scan_write_block(mask, g.n, bsum, pos, act_idx, mbit_bytes, total_out, s_w, s_o);
    if (!use_cache) return;
    build_cache_view(g, key_a, key_b, *n_keys, dbit, dsum);
}

__device__ inline unsigned long long extract64(const unsigned long long *__restrict__ bits, int64_t start) {
    int64_t w = start >> 6;
    int sh = int(start & 63);
    unsigned long long lo = bits[w] >> sh;
    unsigned long long hi = sh ? (bits[w + 1] << (64 - sh)) : 0ull;
    return lo | hi;
}

// One 16-lane group per active row i (4 rows per wavefront): cend[r] = rank of the first active column j in (i, last) with
// (first + (j - i)) in the cache view (the row returns "not similar" there, :66-67), else rank of `last`.
// Columns of compacted rank in (r, cend[r]) are the ones the reference may still evaluate for row r.
// The same group also initialises best[r]; the block (16 rows = one row tile of the pair kernel) records the tile's largest
// stop column.
__global__ __launch_bounds__(256) void k_stop_scan(PassGeom g, int use_cache, const PruneState *__restrict__ st,
                                                    const int32_t *__restrict__ act_idx, const int32_t *__restrict__ pos,
                                                    const unsigned long long *__restrict__ mbit, const unsigned long long *__restrict__ dbit,
                                                    const unsigned long long *__restrict__ dsum, int32_t *__restrict__ cend,
                                                    int32_t *__restrict__ best, int32_t *__restrict__ tile_cmax,
                                                    const float *__restrict__ D, float *__restrict__ Dc) {
    // 16 lanes per row, 4 rows per wavefront: a lane tests 64 deltas at a time, a group 1024 per step
    // D (optional): the 16 descriptor components of the row's structure are copied to position r of Dc on the way (one
    // float per lane), so that the pair kernel reads rows and columns by position -- no gather through the active list
    // in front of every column tile (a dependent round trip per tile)
    if (st->pass_on == 0) return;
    const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
    const int r = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
    const bool mine = r < st->A;
    int64_t i = 0, first = 0, last = 0;
    float dval = 0.0f;
    if (mine) {
        i = act_idx[r];
        chunk_of(g, i, first, last);
        if (D) dval = D[i * 16 + sl];
    }
    int64_t found = last;
    if (use_cache) {
        // candidate deltas d = 1 .. len, i.e. cache-view positions P = first + d in [p_lo, p_hi]; the view is sparse (a
        // key applies to a pass only when its chunk start is one of this pass's), so the row walks the NON-EMPTY 1024-bit
        // blocks of it, found through the summary bitmap dsum, instead of every block of its chunk
        const int64_t len = mine ? last - i - 1 : 0;
        const int64_t p_lo = first + 1, p_hi = first + len, shift = i - first;  // mask position of P is P + shift
        int64_t B = p_lo >> 10;
        const int64_t B_hi = p_hi >> 10;
        bool scanning = len > 0;
        while (__ballot(scanning) != 0) {
            if (scanning) {  // next non-empty block at or after B (the lanes of a row agree)
                bool any = false;
                while (B <= B_hi) {
                    const unsigned long long sw = dsum[B >> 6] >> (B & 63);
                    if (sw) {
                        B += __ffsll((long long)sw) - 1;
                        any = B <= B_hi;
                        break;
                    }
                    B = ((B >> 6) + 1) << 6;
                }
                scanning = any;
            }
            unsigned long long w = 0;
            const int64_t p0 = (B * 16 + sl) * 64;  // first position of this lane's word of the block
            if (scanning && p0 <= p_hi && p0 + 63 >= p_lo) {
                w = dbit[B * 16 + sl] & extract64(mbit, p0 + shift);
                if (p0 < p_lo) w &= ~0ull << (p_lo - p0);
                if (p_hi - p0 < 63) w &= (2ull << (p_hi - p0)) - 1ull;
            }
            const unsigned hit = unsigned(__ballot(w != 0) >> (16 * sub)) & 0xffffu;  // this row's 16 lanes
            if (scanning && hit) {
                const int fl = __ffs(hit) - 1;
                const unsigned long long wl = __shfl(w, 16 * sub + fl);
                found = (B * 16 + fl) * 64 + (__ffsll((long long)wl) - 1) + shift;  // mask position of the first cached column
                scanning = false;
            } else {
                (void)__shfl(w, 16 * sub);  // keep the shuffle convergent for every lane
                ++B;
                if (B > B_hi) scanning = false;
            }
        }
    }
    if (mine && D) Dc[int64_t(r) * 16 + sl] = dval;
    int my_c = 0;
    if (mine && sl == 0) {
        my_c = pos[found];
        cend[r] = my_c;
        best[r] = INT_MAX;  // atomicMin target of the pair kernel: no similar column found yet
    }
    // largest stop column of the 16 rows of this block = one row tile of the pair kernel: lets a work item whose column
    // segment lies beyond it leave after two scalar loads
    __shared__ int s_cmax[4];
    for (int off = 32; off > 0; off >>= 1) my_c = max(my_c, __shfl_xor(my_c, off));
    if (lane == 0) s_cmax[threadIdx.x >> 6] = my_c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cmax[blockIdx.x] = max(max(s_cmax[0], s_cmax[1]), max(s_cmax[2], s_cmax[3]));
}

// Gather the active structures into the two layouts the tile kernel reads:
//   Xr[r][HP3]   row-major, zero-padded to HP atoms  (row structures: wavefront-uniform scalar reads)
//   Xc[d][ld]    coordinate-major                    (column structures: lane = column, coalesced)
//   G[r] = sum |x|^2
// One block moves 64 structures through an LDS tile so that both global sides are coalesced.
__global__ __launch_bounds__(256) void k_compact_coords(const double *__restrict__ heavy, int h, int hp3,
                                                         const int32_t *__restrict__ act_idx, const PruneState *__restrict__ st,
                                                         double *__restrict__ Xr, double *__restrict__ Xc, int64_t ld,
                                                         double *__restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) double s_tile[];  // [64][hp3 + 1]
    if (st->pass_on == 0) return;
    const int n_active = st->A;
    const int h3 = h * 3, pitch = hp3 + 1;
    const int r0 = blockIdx.x * 64;
    if (r0 >= n_active) return;
    const int nr = min(64, n_active - r0);
    for (int e = threadIdx.x; e < 64 * hp3; e += 256) {
        int rr = e / hp3, d = e - rr * hp3;
        double v = 0.0;
        if (rr < nr && d < h3) v = heavy[int64_t(act_idx[r0 + rr]) * h3 + d];
        s_tile[rr * pitch + d] = v;
        if (rr < nr) Xr[int64_t(r0 + rr) * hp3 + d] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * hp3; e += 256) {
        int d = e >> 6, rr = e & 63;
        Xc[int64_t(d) * ld + r0 + rr] = s_tile[rr * pitch + d];  // rows >= nr are zeros
    }
    if (threadIdx.x < 64) {
        double g = 0.0;
        for (int d = 0; d < h3; ++d) {
            double v = s_tile[threadIdx.x * pitch + d];
            g += v * v;
        }
        G[r0 + threadIdx.x] = g;
    }
}

// ---------------------------------------------------------------------------------------------------
// the tile kernel

struct TileArgs {
    long long ld;     // leading dimension of Xc (structures per coordinate row)
    int h;
    int tile_begin;   // first tile of this rank
    int tile_stride;  // world size (row tiles are dealt round-robin to ranks)
    int seg_cols;     // columns per grid.y segment (multiple of 64)
    double thr, maxdev_thr;
    double half_h_thr2;  // h * thr^2 / 2
};

// Fast tests on the characteristic quartic of Horn's matrix (described at the top of this file).
//   PAIR_DISSIMILAR  certainly rmsd >= thr:  P, P', P'' at L = (Gp + Gq - h thr^2) / 2 all exceed their rounding bounds, so
//                    L lies above the largest root l1 and h rmsd^2 = Gp + Gq - 2 l1 > h thr^2;
//   PAIR_SIMILAR     certainly similar (near-duplicates, the bulk of what a prune removes): with x = (Gp + Gq) / 2 - 2 thr^2 > 0,
//                    P''(x) > 0 and P'(x) > 0 put x above the largest root of P' (the roots of P'' are +-sqrt(|H|_F^2 / 3) and
//                    those of P' interlace them; the other region with both signs lies below the negative root of P''),
//                    hence above the second root l2 of P, and P(x) < 0 then puts it below the largest root l1 (all with
//                    rounding bounds), so h rmsd^2 = Gp + Gq - 2 l1 < 4 thr^2: rmsd < 2 thr / sqrt(h) <= thr
//                    for h >= 4, and maxdev <= sqrt(sum dev^2) = sqrt(h) rmsd < 2 thr -- both conditions of
//                    rmsd_pruning.py:75 hold without forming the rotation (needs maxdev_thr == 2 thr, as :95 sets it);
//   PAIR_UNDECIDED   everything else takes the explicit-rotation path.
enum { PAIR_DISSIMILAR = 0, PAIR_SIMILAR = 1, PAIR_UNDECIDED = 2 };

struct Quartic {
    double F, CF, c2, c1, c0;
};
__device__ inline Quartic horn_quartic(const double H[9]) {
    Quartic q;
    q.F = H[0] * H[0] + H[1] * H[1] + H[2] * H[2] + H[3] * H[3] + H[4] * H[4] + H[5] * H[5] + H[6] * H[6] + H[7] * H[7] + H[8] * H[8];
    const double C0 = H[4] * H[8] - H[5] * H[7], C1 = H[5] * H[6] - H[3] * H[8], C2 = H[3] * H[7] - H[4] * H[6];
    const double C3 = H[2] * H[7] - H[1] * H[8], C4 = H[0] * H[8] - H[2] * H[6], C5 = H[1] * H[6] - H[0] * H[7];
    const double C6 = H[1] * H[5] - H[2] * H[4], C7 = H[2] * H[3] - H[0] * H[5], C8 = H[0] * H[4] - H[1] * H[3];
    const double det = H[0] * C0 + H[1] * C1 + H[2] * C2;
    q.CF = C0 * C0 + C1 * C1 + C2 * C2 + C3 * C3 + C4 * C4 + C5 * C5 + C6 * C6 + C7 * C7 + C8 * C8;
    q.c2 = -2.0 * q.F, q.c1 = -8.0 * det, q.c0 = q.F * q.F - 4.0 * q.CF;
    return q;
}
// Rounding bound of the quartic tests: every comparison below is `value > kappa * (sum of the magnitudes of its terms)`.
// Derivation (u = 2^-53, gamma = h u, S^2 = Gp Gq, rho = ((Gp + Gq) / 2) / L >= S / L, L = the test point):
//   * H: an h-term fma chain per entry, |dH_ij| <= gamma sum_a |p_ai q_aj| <= gamma sqrt(sum p_ai^2 sum q_aj^2), so
//     |dH|_F <= gamma S, and |H|_F <= S (Cauchy-Schwarz);
//   * F = |H|_F^2:  dF <= 2 |H| |dH| <= 2 gamma S^2;  det: d det <= |cof H|_F |dH|_F <= gamma S^3;  CF = |cof H|_F^2:
//     dCF <= 2 |cof| 2 |H| |dH| <= 4 gamma S^4;  hence d(c2 L^2) <= 4 gamma S^2 L^2, d(c1 L) <= 8 gamma S^3 L,
//     d(c0) = 2 F dF + 4 dCF <= 20 gamma S^4: together <= 32 gamma rho^4 L^4 in P, and by the same steps
//     <= 4 gamma rho^3 (4 L^3) in P' and <= gamma rho^2 (6 L^2) in P'';
//   * the test point itself: Gp, Gq are 3h-term sums, |dL| <= 3 gamma (Gp + Gq) / 2 = 3 gamma rho L, which moves P by at most
//     |P'| |dL| <= 12 gamma rho L^4 (and P', P'' by less, relative to their term sums);
//   * evaluating the three polynomials: a few u times the term sums.
// All of it is below 64 h u rho^4 times the respective term sum (each sum contains L^4, 4 L^3, 6 L^2).  kappa is that
// bound, and never less than the 1e-12 the recorded runs were made with (which it exceeds only beyond about 140 heavy atoms
// or where the structures sit so close to the origin that L is small against Gp + Gq): a larger kappa only hands more pairs
// to the explicit-rotation path, it cannot change a verdict.  Structures far from the origin (the path never centres,
// rmsd_pruning.py:7-41) make Gp, Gq and L grow with the square of the offset while the margin P(L) ~ P'(l1) (L - l1) keeps
// L - l1 = h (rmsd^2 - thr^2) / 2: the tests decide less and less (at 10^4 A nothing) and everything takes the exact path.
constexpr double QUARTIC_KAPPA_MIN = 1e-12;
__device__ inline double quartic_kappa(int h, double half_sum, double L) {
    const double rho = half_sum / L, r2 = rho * rho;          // callers test L > 0 themselves
    const double k = 64.0 * 1.1102230246251565e-16 * double(h) * r2 * r2;
    return (k > QUARTIC_KAPPA_MIN) ? k : QUARTIC_KAPPA_MIN;  // (a NaN k -- L == 0 -- keeps the minimum; the L > 0 test fails then)
}

__device__ inline bool quartic_above_top_root(const Quartic &q, double L, double kappa) {
    const double L2 = L * L;
    const double t4 = L2 * L2, t2 = q.c2 * L2, t1 = q.c1 * L;
    const double P = t4 + t2 + t1 + q.c0;
    const double eP = kappa * (t4 + fabs(t2) + fabs(t1) + fabs(q.c0) + 4.0 * q.CF + q.F * q.F);
    const double P1 = 4.0 * L2 * L + 2.0 * q.c2 * L + q.c1;
    const double e1 = kappa * (4.0 * L2 * fabs(L) + 2.0 * fabs(q.c2 * L) + fabs(q.c1));
    const double P2 = 6.0 * L2 + q.c2;
    const double e2 = kappa * (6.0 * L2 + fabs(q.c2));
    return (L > 0.0) && (P > eP) && (P1 > e1) && (P2 > e2);
}
__device__ inline bool quartic_between_top_roots(const Quartic &q, double x, double kappa) {
    const double x2 = x * x;
    const double t4 = x2 * x2, t2 = q.c2 * x2, t1 = q.c1 * x;
    const double P = t4 + t2 + t1 + q.c0;
    const double eP = kappa * (t4 + fabs(t2) + fabs(t1) + fabs(q.c0) + 4.0 * q.CF + q.F * q.F);
    const double P1 = 4.0 * x2 * x + 2.0 * q.c2 * x + q.c1;
    const double e1 = kappa * (4.0 * x2 * fabs(x) + 2.0 * fabs(q.c2 * x) + fabs(q.c1));
    const double P2 = 6.0 * x2 + q.c2;
    const double e2 = kappa * (6.0 * x2 + fabs(q.c2));
    return (x > 0.0) && (P2 > e2) && (P1 > e1) && (P < -eP);
}

// half_sum = (Gp + Gq) / 2, L = half_sum - h thr^2 / 2, h = atoms per structure
__device__ inline bool certainly_dissimilar(const double H[9], double L, double half_sum, int h) {
    return quartic_above_top_root(horn_quartic(H), L, quartic_kappa(h, half_sum, L));
}

// half_sum = (Gp + Gq) / 2, half_h_thr2 = h thr^2 / 2, two_thr2 = 2 thr^2 (pass a negative two_thr2 to switch the
// near-duplicate test off: h < 4, or a maxdev threshold other than 2 thr)
__device__ inline int pair_verdict(const double H[9], double half_sum, double half_h_thr2, double two_thr2, int h) {
    const Quartic q = horn_quartic(H);
    const double L = half_sum - half_h_thr2;
    if (quartic_above_top_root(q, L, quartic_kappa(h, half_sum, L))) return PAIR_DISSIMILAR;
    const double x = half_sum - two_thr2;
    if (two_thr2 > 0.0 && quartic_between_top_roots(q, x, quartic_kappa(h, half_sum, x))) return PAIR_SIMILAR;
    return PAIR_UNDECIDED;
}

// One wavefront = one work item = (row tile of TI consecutive active rows) x (one column segment).
// Lane = column.  For every 64-column tile: load q_j into registers, then for each live row accumulate
// H = p_i^T q_j, run the sign test, remember survivors as per-lane bits; after the row loop the few
// survivors take the exact path and the smallest similar column of each row goes to best[] (atomicMin).
// A row stops being visited once it has a similar column to its left (the reference returns at the
// first similar column, rmsd_pruning.py:75-77).
template <int HP, int TI>
__global__ __launch_bounds__(256, 2) void k_rmsd_tile(const double *__restrict__ Xr, const double *__restrict__ Xc,
                                                       const double *__restrict__ G, const int32_t *__restrict__ cend,
                                                       int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                                       const PruneState *__restrict__ st, TileArgs a) {
    // Xr [n_active][HP*3] row structures; Xc [HP*3][ld] column structures; G [ld] squared norms;
    // cend [n_active] exclusive column bound of each row; best [n_active] atomicMin target (INT_MAX = none);
    // counters[0] pairs computed, counters[1] pairs sent to the exact path.
    static_assert(TI <= 32, "row liveness is a 32-bit lane mask");
    constexpr int HP3 = HP * 3;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (st->pass_on == 0) return;
    const int n_active = st->A;
    const int slot = blockIdx.x * 4 + wid;
    const int tile = a.tile_begin + slot * a.tile_stride;
    const int r0 = tile * TI;
    if (r0 >= n_active) return;
    const int nrows = min(TI, n_active - r0);
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

    unsigned long long n_computed = 0, n_cand = 0;
    for (int c0 = seg_lo; c0 < cmax && alive; c0 += 64) {
        const int col = c0 + lane;
        double q[HP3];
        {
            const double *xc = Xc + col;
#pragma unroll
            for (int d = 0; d < HP3; ++d, xc += a.ld) q[d] = *xc;
        }
        const double Gq = G[col];
        unsigned mycand = 0;  // bit t: this lane's column survived the sign test against row t
        unsigned candrows = 0;
        for (int t = 0; t < nrows; ++t) {
            if (!((alive >> t) & 1u)) continue;
            const int r = r0 + t;
            const int ce = __builtin_amdgcn_readlane(my_cend, t);
            if (ce <= c0 || r >= c0 + 63) continue;
            const double *__restrict__ pr = Xr + int64_t(r) * HP3;
            double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < HP; ++k) {
                const double px = pr[3 * k], py = pr[3 * k + 1], pz = pr[3 * k + 2];
                H[0] = fma(px, q[3 * k], H[0]), H[1] = fma(px, q[3 * k + 1], H[1]), H[2] = fma(px, q[3 * k + 2], H[2]);
                H[3] = fma(py, q[3 * k], H[3]), H[4] = fma(py, q[3 * k + 1], H[4]), H[5] = fma(py, q[3 * k + 2], H[5]);
                H[6] = fma(pz, q[3 * k], H[6]), H[7] = fma(pz, q[3 * k + 1], H[7]), H[8] = fma(pz, q[3 * k + 2], H[8]);
            }
            const double L = 0.5 * (G[r] + Gq) - a.half_h_thr2;
            const bool valid = col > r && col < ce;
            const bool cand = valid && !certainly_dissimilar(H, L, 0.5 * (G[r] + Gq), a.h);
            const unsigned long long vm = __ballot(valid);
            n_computed += __popcll(vm);
            if (__ballot(cand)) candrows |= 1u << t;
            mycand |= cand ? (1u << t) : 0u;
        }
        // exact path for the survivors of this column tile
        while (candrows) {
            const int t = __ffs(candrows) - 1;
            candrows &= candrows - 1;
            const int r = r0 + t;
            bool sim = false;
            if ((mycand >> t) & 1u) {
                double rm, md;
                rmsd_and_max_pair(Xr + int64_t(r) * HP3, Xr + int64_t(col) * HP3, a.h, rm, md);
                sim = rm < a.thr && md < a.maxdev_thr;  // rmsd_pruning.py:75
            }
            n_cand += __popcll(__ballot((mycand >> t) & 1u));
            const unsigned long long sm = __ballot(sim);
            if (sm) {
                if (lane == 0) atomicMin(&best[r], c0 + __ffsll((long long)sm) - 1);
                alive &= ~(1u << t);
            }
        }
    }
    if (lane == 0) {
        count_add(counters, unsigned(slot), CNT_FORMED, n_computed);
        count_add(counters, unsigned(slot), CNT_EXACT, n_cand);
    }
}

// Apply a finished pass: rows with a similar column are removed (:113) and leave one cache key each
// (:76, appended after the pass at :204); counts what the reference's sequential scan would have evaluated.
__global__ __launch_bounds__(256) void k_apply_pass(PassGeom g, PruneState *__restrict__ st, const int32_t *__restrict__ act_idx,
                                                     const int32_t *__restrict__ cend, const int32_t *__restrict__ best,
                                                     uint8_t *__restrict__ mask, int32_t *__restrict__ key_a,
                                                     int32_t *__restrict__ key_b, int32_t *__restrict__ n_keys,
                                                     PassCounters *__restrict__ cnt, int32_t *__restrict__ bsum, int block_items, PassRecord *__restrict__ rec,
                                                     StepArgs next, unsigned long long *__restrict__ dbit) {
    __shared__ unsigned long long s_sum[32];
    __shared__ int s_last;
    const bool pass_on = st->pass_on != 0;
    const int n_active = pass_on ? st->A : 0;  // a pass that is gated off has no rows; its blocks still take a ticket
    const int lane = threadIdx.x & 63;
    unsigned long long ev_total = 0, rm_total = 0;
    // a capped grid walks the rows tile by tile (block-uniform bounds: every wavefront keeps all its lanes for the ballots)
    for (int row0 = blockIdx.x * 256; row0 < n_active; row0 += gridDim.x * 256) {
        const int r = row0 + threadIdx.x;
        unsigned long long ev = 0;
        bool removed = false;
        int my_block = -1;
        int64_t first = 0, delta = 0;
        if (r < n_active) {
            const int b = best[r];
            if (b != INT_MAX) {
                const int64_t i = act_idx[r], j = act_idx[b];
                int64_t last;
                chunk_of(g, i, first, last);
                mask[i] = 0;
                my_block = int(i / block_items);
                delta = j - i;
                removed = true;
                ev = (unsigned long long)(b - r);  // columns r+1 .. b were evaluated
            } else {
                ev = (unsigned long long)(cend[r] - r - 1);  // every active column before the stop column
            }
        }
        // the scan's per-block counts follow the mask: one atomic per (wavefront, scan block) -- the removed rows of a
        // wavefront fall into one or two blocks, and thousands of single decrements of two cache lines would serialise
        for (unsigned long long left = __ballot(removed); left;) {
            const int l = __ffsll((long long)left) - 1;
            const int blk = __shfl(my_block, l);
            const unsigned long long same = __ballot(removed && my_block == blk);
            if (lane == l) atomicSub(&bsum[blk], __popcll(same));
            left &= ~same;
        }
        // one slot reservation per wavefront for the keys of its removed rows (order inside the cache is irrelevant)
        const unsigned long long rm = __ballot(removed);
        const int n_rm = __popcll(rm);
        int base = 0;
        if (n_rm) {
            if (lane == 0) base = atomicAdd(n_keys, n_rm);
            base = __shfl(base, 0);
            if (removed) {
                const int slot = base + __popcll(rm & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                key_a[slot] = int32_t(first);
                key_b[slot] = int32_t(first + delta);
            }
        }
        for (int off = 32; off > 0; off >>= 1) ev += __shfl_down(ev, off);
        ev_total += ev, rm_total += (unsigned long long)n_rm;
    }
    if (lane == 0) {
        const unsigned bucket = blockIdx.x * 4 + (threadIdx.x >> 6);
        count_add(cnt, bucket, CNT_EVALUATED, ev_total);
        count_add(cnt, bucket, CNT_REMOVED, rm_total);
    }
    // The last block to get here closes this pass and opens the next one (what a separate one-block launch would do).
    // The only data handed from block to block are the statistics buckets, and those are written by agent-scope
    // atomics and read with agent-scope (sc1) loads -- so no cache fence is needed (a __threadfence() per block costs
    // an L2 write-back each: measured 14 us per pass): every wave drains its atomics, the block's barrier, ONE lane's
    // ticket add, and the block whose add came last reads (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(&st->ticket, 1u);
        s_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    pass_step_block(st, cnt, rec, next, dbit, s_sum);
    if (threadIdx.x == 0) st->ticket = 0;
}

// End of a run: the pass records and (optionally) the survivor mask go to host-visible memory from ONE small launch instead
// of two copy commands.  `rec_host` and `mask_host` are pinned host allocations mapped into the device's address space.
__global__ __launch_bounds__(256) void k_export_run(const unsigned long long *__restrict__ rec, int rec_words, unsigned long long *__restrict__ rec_host,
                                                     const unsigned long long *__restrict__ mask, int64_t mask_words, const uint8_t *__restrict__ mask_bytes,
                                                     int64_t n, unsigned long long *__restrict__ mask_host) {
    const int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = tid; e < rec_words; e += stride) rec_host[e] = rec[e];
    if (mask_host) {
        for (int64_t e = tid; e < mask_words; e += stride) mask_host[e] = mask[e];
        uint8_t *tail = reinterpret_cast<uint8_t *>(mask_host);
        for (int64_t b = mask_words * 8 + tid; b < n; b += stride) tail[b] = mask_bytes[b];
    }
}

__global__ void k_fill_i32(int32_t *p, int64_t n, int32_t v) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) p[i] = v;
}

}  // namespace tsc
