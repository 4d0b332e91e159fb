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
#include "mm_record.hpp"
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

inline __global__ __launch_bounds__(256) void k_rmsd_pairs(const double *__restrict__ heavy, int h, const int32_t *__restrict__ pairs,
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
enum { CNT_FORMED = 0, CNT_EXACT = 1, CNT_SCREENED = 2, CNT_EVALUATED = 3, CNT_REMOVED = 4, CNT_WALK = 5 /* pairs inside the rows' ranges (k_open_rows, for k_cull_decide) */, CNT_WORDS = 16 };
struct PassCounters {
    unsigned long long w[CNT_BUCKETS][CNT_WORDS];
};
__device__ inline void count_add(PassCounters *c, unsigned bucket, int what, unsigned long long v) {
    if (v) atomicAdd(&c->w[bucket & (CNT_BUCKETS - 1)][what], v);
}

// Device-resident control block of a prune run.  The host enqueues every pass that COULD run (20 k < N) without
// waiting for counts; the kernel that closes a pass evaluates the reference's gate `k == 1 or 20*k < count_nonzero(mask)`
// (rmsd_pruning.py:192) for the next one on the device, and the kernels of a pass that is gated off return immediately.
struct PruneState {
    int n_active;  // count_nonzero(mask) after the last finished pass
    int pass_on;   // gate of the pass in flight
    int A;         // active structures entering the pass in flight (== n_active at its start)
    unsigned ticket;  // blocks of k_apply_pass that have finished (the last one closes the pass)
    int bitsel;    // which of the two bit copies of the mask the pass in flight READS (it clears the rows it removes in the other)
    int cull_on;   // a pass that MAY be culled (cull.hpp): 1 = the sorted layout + bounding boxes run it, 0 = the ordered walk does
                   // (k_cull_decide, from the rows' ranges: a cache that ends most rows early leaves the walk little to do)
    int row_lo;    // rank-partitioned pass (tsc_prune_pass_range): active rank of this rank's local row 0; 0 in every other pass.
                   // In such a pass A counts only the active structures of this rank's chunks and act / cend / best / Dc are
                   // indexed by LOCAL row (global active rank - row_lo): the pair kernel sees an ensemble of A rows
};
struct PassRecord {  // one per schedule slot, read back once at the end of the run
    long long k, n_before, n_after, formed, exact, screened, evaluated, removed;
    int on, algo;
};

// "Every unit of this pass has finished": two-level arrival counter (same-address atomics serialise at ~12 ns each, so
// thousands of arrivals go to PT_GROUPS counters on separate 128-byte lines and only the last of a group to the top).
constexpr int PT_GROUPS = 64;
struct PassTickets {  // zeroed by k_init_run and again by the wavefront that closes a pass
    unsigned group[PT_GROUPS][32];
    unsigned top;
    unsigned pad[31];
};
// one lane calls this for unit `unit` of `n_units`; true for the arrival that completes the pass
__device__ inline bool tickets_arrive(PassTickets *tk, unsigned unit, unsigned n_units, unsigned n_groups) {
    const unsigned grp = unit % n_groups;
    const unsigned in_group = (n_units - grp + n_groups - 1) / n_groups;
    if (atomicAdd(&tk->group[grp][0], 1u) != in_group - 1) return false;
    const unsigned groups = n_units < n_groups ? n_units : n_groups;
    return atomicAdd(&tk->top, 1u) == groups - 1;
}

// The similarity cache of the reference (rmsd_pruning.py:65-76, :204) as one bitmap PER FUTURE PASS.  A row removed in a
// pass leaves the key (a, b) = (first, first + (j - i)); a later pass hits that key only where `a` is the start of one of
// ITS chunks and `b` lies inside that chunk -- and then exactly for the pairs with first + (j - i) == b.  The schedule of
// a run is known when it starts, so the kernel that removes a row sets bit b in the view of every later pass the key
// applies to (a handful of integer divisions per removed row; few keys apply anywhere) and no pass has to walk the key
// list before it can start.  Behind the n-bit view of a pass sits its summary, one bit per 1024 view bits: the stop-column
// search walks only non-empty blocks.
constexpr int MAX_SLOTS = 18;
struct ViewPass {  // one view's pass: k chunks of cs = n // k structures; a / cs == (a * magic) >> shift for every a < 2^31
    int k, cs;
    unsigned magic;
    int shift;
};
struct CacheViews {
    unsigned long long *views;  // [count][stride] words
    long long stride;           // words per view: bit_words + summary words
    int bit_words;
    int n;                      // structures of the run
    int first, count;           // views first .. count - 1 belong to the passes after the one in flight
    const ViewPass *pass;       // [count]
};
// Division by an invariant: m = floor(2^(31+L) / d) + 1 with L = ceil(log2 d) divides every 31-bit dividend exactly
// (Granlund & Montgomery 1994, theorem 4.2 with N = 31); m < 2^32.
inline ViewPass view_pass(int n, int k) {
    ViewPass v;
    v.k = k, v.cs = n / k;
    int L = 0;
    while ((1ll << L) < (long long)v.cs) ++L;
    v.magic = unsigned(((1ull << (31 + L)) / (unsigned long long)v.cs) + 1ull);
    v.shift = 31 + L;
    return v;
}
// A whole wavefront calls this with one removed row per lane (or none): key (a, b) of lane `removed`.  The loop over the later
// passes is wave-uniform (their parameters come by scalar loads); the summary words, which every key of a chunk shares, take
// one atomic per wavefront and word instead of one per key (same-address atomics serialise at ~12 ns each).
__device__ inline void views_insert_wave(const CacheViews &cv, bool removed, int a, int b) {
    if (__ballot(removed) == 0) return;
    const int lane = threadIdx.x & 63;
    for (int j = cv.first; j < cv.count; ++j) {
        const ViewPass vp = cv.pass[j];
        bool ok = false;
        if (removed) {
            const int c = int(((unsigned long long)unsigned(a) * vp.magic) >> vp.shift);
            if (c * vp.cs == a && c < vp.k) ok = b < ((c == vp.k - 1) ? cv.n : vp.cs * (c + 1));
        }
        unsigned long long left = __ballot(ok);
        if (!left) continue;
        unsigned long long *v = cv.views + int64_t(j) * cv.stride;
        if (ok) atomicOr(&v[b >> 6], 1ull << (b & 63));
        while (left) {
            const int leader = __ffsll((long long)left) - 1;
            const int word = __shfl(b >> 16, leader);
            const bool same = ok && (b >> 16) == word;
            unsigned long long sb = same ? 1ull << ((b >> 10) & 63) : 0ull;
            for (int off = 32; off > 0; off >>= 1) sb |= __shfl_xor(sb, off);
            if (lane == leader) {
                unsigned long long *ds = v + cv.bit_words + word;
                if ((__hip_atomic_load(ds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & sb) != sb) atomicOr(ds, sb);
            }
            left &= ~__ballot(same);
        }
    }
}

// Initial state of a run: out_mask = ones (rmsd_pruning.py:182), empty cache (:183), zeroed records, counters, tickets.
struct InitArgs {
    int64_t n;
    uint8_t *mask;
    unsigned long long *bits;  // two copies of the mask as bits, bit_words each
    int bit_words;
    unsigned long long *views;  // cache views of every pass of the schedule + their summaries
    int64_t view_words;
    ViewPass *view_pass;        // device copy of the schedule
    int n_views;
    ViewPass sched[MAX_SLOTS];
    PruneState *st;
    PassRecord *rec;
    int n_rec;
    PassCounters *cnt;
    int32_t *bsum, *boff;       // per scan block: active structures in it / before it
    int n_blocks, block_items;
    unsigned *dmax_bits;
    unsigned *zero_words;       // arrival counters of the kernels that close a pass
    int64_t n_zero_words;
    int32_t *tile_done;         // arrivals per row tile of the fused pair kernel
    int64_t n_tile_done;
    int32_t *act;
    int first_slot;
    long long first_k;
    int first_algo;
};
inline __global__ __launch_bounds__(256) void k_init_run(InitArgs a) {
    // first_slot >= 0: the first pass of the schedule is opened here as well (gate of rmsd_pruning.py:192 on the full count,
    // its record), which is all a k_pass_step launch would do at this point
    const int64_t n = a.n;
    const int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    unsigned long long *m8 = reinterpret_cast<unsigned long long *>(a.mask);  // scratch blocks are 256-byte aligned
    for (int64_t e = tid; e < n / 8; e += stride) m8[e] = 0x0101010101010101ull;
    for (int64_t e = (n / 8) * 8 + tid; e < n; e += stride) a.mask[e] = 1;
    // every entry of the active list is a valid structure index from the start: the pair kernel gathers through
    // act[x] for x up to n without knowing the active count
    for (int64_t e = tid; e < n; e += stride) a.act[e] = int32_t(e);
    for (int64_t e = tid; e < a.bit_words; e += stride) {  // n ones, zeros behind them, in both copies
        const int64_t lo = e * 64;
        const unsigned long long w = lo + 64 <= n ? ~0ull : (lo >= n ? 0ull : ((1ull << (n - lo)) - 1ull));
        a.bits[e] = w, a.bits[a.bit_words + e] = w;
    }
    for (int64_t e = tid; e < a.view_words; e += stride) a.views[e] = 0;
    unsigned long long *c = &a.cnt->w[0][0];
    for (int64_t e = tid; e < CNT_BUCKETS * CNT_WORDS; e += stride) c[e] = 0;
    for (int64_t e = tid; e < a.n_zero_words; e += stride) a.zero_words[e] = 0;
    for (int64_t e = tid; e < a.n_tile_done; e += stride) a.tile_done[e] = 0;
    char *r = reinterpret_cast<char *>(a.rec);
    for (int64_t e = tid; e < int64_t(a.n_rec) * int64_t(sizeof(PassRecord)); e += stride)
        if (int(e / int64_t(sizeof(PassRecord))) != a.first_slot) r[e] = 0;  // (the first pass's record is written whole below)
    // per-block counts of the mask and their exclusive prefix: what ranks an active structure (k_open_rows); the kernels
    // that remove rows keep the counts current, the one that closes a pass the prefix
    for (int64_t e = tid; e <= a.n_blocks; e += stride) {
        const int64_t lo = e * a.block_items;
        if (e < a.n_blocks) a.bsum[e] = int32_t(lo >= n ? 0 : (n - lo < a.block_items ? n - lo : a.block_items));
        a.boff[e] = int32_t(lo >= n ? n : lo);
    }
    if (tid < a.n_views) a.view_pass[tid] = a.sched[tid];
    if (tid == 0) {
        if (a.dmax_bits) *a.dmax_bits = 0;  // running maximum of the descriptor build (sieve.hpp)
        PruneState *st = a.st;
        st->n_active = int(n), st->pass_on = 0, st->A = int(n), st->ticket = 0, st->bitsel = 0, st->row_lo = 0, st->cull_on = 0;
        if (a.first_slot >= 0) {
            const int on = (a.first_k == 1 || 20 * a.first_k < (long long)n) ? 1 : 0;
            PassRecord &fr = a.rec[a.first_slot];
            fr.k = a.first_k, fr.n_before = fr.n_after = (long long)n, fr.on = on, fr.algo = a.first_algo;
            fr.formed = fr.exact = fr.screened = fr.evaluated = fr.removed = 0;
            st->pass_on = on;
        }
    }
}

// Closes the pass in slot `prev` (sums its counters, updates n_active, re-ranks the scan blocks, flips the bit copy) and
// opens the pass in slot `cur` (gate, A, zeroed counters).  prev / cur = -1: nothing to close / open.  Runs in ONE
// WAVEFRONT: of the last unit of the closing pass to finish -- a row tile of the pair kernel, a block of k_apply_pass or of
// k_pass_chunks -- or, where no pass precedes, of k_pass_step.
struct StepArgs {
    int prev, cur;
    long long k_cur;
    int algo_cur;
    int prev_algo;  // >= 0: the kernel that really ran the closing pass (recorded in its PassRecord), -1: as opened
};
struct StepCtx {
    PruneState *st;
    PassCounters *cnt;
    PassRecord *rec;
    int32_t *bsum, *boff;
    int n_blocks;
    unsigned *tickets;  // every arrival counter of the run (PassTickets and the chunk-local kernel's: one per 128-byte line), zeroed on the way out
    int ticket_lines;
    unsigned long long *exch_tail;  // rank-partitioned pass only (else null): the last unit of this rank's share does not close the
                                    // pass -- it leaves this rank's five statistics here, behind the removed-row bits of the exchange
                                    // buffer, and k_pass_merge closes the pass after the ranks' buffers have been summed
};

__device__ inline void pass_step_wave(const StepCtx &sc, const StepArgs &sa) {
    PruneState *st = sc.st;
    const int lane = threadIdx.x & 63;
    static_assert(CNT_BUCKETS == 64 && CNT_REMOVED == 4, "one bucket per lane, five statistics");
    // the buckets and block counts were written by other CUs' atomics: read at the L2, not through this CU's cache.  All
    // loads of the step are issued before the first is used (one memory round trip, not seven)
    const int pass_on = st->pass_on, n_before = st->n_active, bitsel = st->bitsel;
    unsigned long long v0 = __hip_atomic_load(&sc.cnt->w[lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v1 = __hip_atomic_load(&sc.cnt->w[lane][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v2 = __hip_atomic_load(&sc.cnt->w[lane][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v3 = __hip_atomic_load(&sc.cnt->w[lane][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v4 = __hip_atomic_load(&sc.cnt->w[lane][4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int c_first = lane < sc.n_blocks ? __hip_atomic_load(&sc.bsum[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const bool closing = sa.prev >= 0 && pass_on != 0;
    unsigned long long sum[CNT_REMOVED + 1] = {v0, v1, v2, v3, v4};
    if (sc.exch_tail) {  // this rank's share of a rank-partitioned pass is done: statistics into the exchange buffer, counters cleared
#pragma unroll
        for (int w = 0; w <= CNT_REMOVED; ++w)
            for (int off = 32; off > 0; off >>= 1) sum[w] += __shfl_xor(sum[w], off);
        if (lane == 0 && pass_on != 0) {
#pragma unroll
            for (int w = 0; w <= CNT_REMOVED; ++w) sc.exch_tail[w] = sum[w];
        }
        for (int e = lane; e < CNT_BUCKETS * 8; e += 64) sc.cnt->w[e >> 3][e & 7] = 0;
        for (int e = lane; e < sc.ticket_lines; e += 64) sc.tickets[32 * e] = 0;
        return;
    }
    if (closing) {
#pragma unroll
        for (int w = 0; w <= CNT_REMOVED; ++w)
            for (int off = 32; off > 0; off >>= 1) sum[w] += __shfl_xor(sum[w], off);
        int run = 0;
        for (int b0 = 0; b0 < sc.n_blocks; b0 += 64) {
            const int b = b0 + lane;
            const int c = b0 == 0 ? c_first : (b < sc.n_blocks ? __hip_atomic_load(&sc.bsum[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
            int incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            if (b < sc.n_blocks) sc.boff[b] = run + incl - c;
            run += __shfl(incl, 63);
        }
        if (lane == 0) sc.boff[sc.n_blocks] = run;
    }
    if (lane == 0) {
        int n_active = n_before;
        if (closing) {
            PassRecord &r = sc.rec[sa.prev];
            r.formed = (long long)sum[CNT_FORMED], r.exact = (long long)sum[CNT_EXACT], r.screened = (long long)sum[CNT_SCREENED];
            r.evaluated = (long long)sum[CNT_EVALUATED], r.removed = (long long)sum[CNT_REMOVED];
            n_active -= int(sum[CNT_REMOVED]);
            st->n_active = n_active;
            r.n_after = n_active;
            if (sa.prev_algo >= 0) r.algo = sa.prev_algo;
            st->bitsel = bitsel ^ 1;  // the copy the closing pass cleared its removed rows in is the current one now
        }
        int on = 0;
        if (sa.cur >= 0) {
            on = (sa.k_cur == 1 || 20 * sa.k_cur < (long long)n_active) ? 1 : 0;  // rmsd_pruning.py:192
            PassRecord &r = sc.rec[sa.cur];
            r.k = sa.k_cur, r.n_before = n_active, r.n_after = n_active, r.on = on, r.algo = sa.algo_cur;
            r.formed = r.exact = r.screened = r.evaluated = r.removed = 0;
        }
        st->A = n_active;
        st->row_lo = 0;
        st->pass_on = on;
        st->ticket = 0;
    }
    // counters back to zero (only the words in use); the arrival counters: the first word of every 128-byte line
    for (int e = lane; e < CNT_BUCKETS * 8; e += 64) sc.cnt->w[e >> 3][e & 7] = 0;
    for (int e = lane; e < sc.ticket_lines; e += 64) sc.tickets[32 * e] = 0;
}

inline __global__ __launch_bounds__(64) void k_pass_step(StepCtx sc, StepArgs sa) { pass_step_wave(sc, sa); }

// ---------------------------------------------------------------------------------------------------
// per-pass helper kernels (the pass sequence is in prune.hip, tsc_prune_pass_local / tsc_prune_pass_finish)

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

__device__ inline unsigned long long extract64(const unsigned long long *__restrict__ bits, int64_t start) {
    int64_t w = start >> 6;
    int sh = int(start & 63);
    unsigned long long lo = bits[w] >> sh;
    unsigned long long hi = sh ? (bits[w + 1] << (64 - sh)) : 0ull;
    return lo | hi;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SCAN_BLOCK_WORDS = 32;  // a scan block (scan.hpp: SCAN_TILE = 2048 structures) as 64-bit words of the mask's bit copy
constexpr int DESC_WORDS = 16;        // floats per descriptor (sieve.hpp: DW)

// position of the k-th (0-based) set bit of w (k < popcount(w))
__device__ inline int select64(unsigned long long w, int k) {
    int p = 0;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const int c = __popcll((w >> p) & ((1ull << s) - 1ull));
        if (k >= c) k -= c, p += s;
    }
    return p;
}

// Everything a pass needs per ROW, in one launch with no kernel in front of it: ONE LANE per active row r, one wavefront per 64 rows
// = four row tiles of the pair kernel, four wavefronts per block.
//   * which structure is the r-th active one: the scan block from the prefix of the block counts (boff, kept current by
//     the kernel that closed the previous pass), the bit inside the block from the 32 words of the mask's bit copy --
//     an ordered compaction without a scan pass over the whole mask.  The 64 rows of a wavefront are consecutive ranks, so they lie
//     in one or two scan blocks: the wavefront stages a block's 32 words and their running popcounts in LDS ONCE (stage_block) and
//     every row of that block finds its word by search and its bit by select;
//   * cend[r] = rank of the first active column j in (i, last) with (first + (j - i)) in the cache view of this pass (the
//     row returns "not similar" there, :66-67), else rank of `last`.  Columns of compacted rank in (r, cend[r]) are the
//     ones the reference may still evaluate for row r.  Ranks of positions come from staged blocks too: the rows of a wavefront
//     share their chunk's end (one or two distinct targets), a cache hit lies a few positions behind its row;
//   * act[r], best[r] = none, the row's descriptor copied to position r of Dc (the pair kernel reads rows and columns by
//     position), the largest stop column of every tile;
//   * the other bit copy of the mask brought up to date (the rows this pass removes are cleared THERE, so that every row of
//     the pass sees the mask as it was when the pass began, rmsd_pruning.py:151-157, whatever order the tiles finish in).
// fused != 0: the pair kernel applies the verdicts of a row tile itself when the tile's last work item finishes, and the
// last tile closes the pass.  A tile without work (no row with columns to look at; beyond the active count; the pass gated
// off) has nothing to apply and arrives here.
// (Rounds 2 - 4 gave a row 16, then 4 lanes: at a million structures 30 000 wavefronts each waited out the same memory round trips and
// the kernel was bound by instruction issue -- 45 us per pass at C4, 12 at C3.  A lane per row is a quarter of the wavefronts and of
// the wave-instructions per row.)
constexpr int OPEN_LDS_BLOCKS = 2048;  // scan blocks whose prefix is staged in LDS (4 M structures); beyond: read from memory
constexpr int OPEN_ROWS = 64;          // rows per wavefront
constexpr int OPEN_LIST_CAP = 64 + 1024;  // a refill stops once it holds 64 entries; the block that takes it there adds 1024 at most
struct OpenArgs {
    int use_cache, fused, lds_cap;
    const unsigned long long *view;  // cache view of this pass and its summary (bit_words behind it)
    unsigned long long *bits;        // the two bit copies of the mask
    int bit_words;
    const int32_t *boff;
    int n_blocks, block_items;
    unsigned n_tiles;                // row tiles of the pass (arrival count of a fused pass)
    PassTickets *tickets;
    int32_t *rank_of;                // (optional) rank_of[structure] = its active rank: what a culled pass (cull.hpp) lays its sorted order out from
    _Float16 *Dh;                    // (optional) the float16 records of the matrix-core screen (mm.hpp), by position like Dc
    const unsigned *dmax_bits;       // ... and the largest |descriptor component| of the run that scales them
    unsigned long long *dbg;         // -DTSC_DBG_STAMPS builds only: 8 time stamps per wavefront (tools/stamps.py), else null
};
#ifdef TSC_DBG_STAMPS
#define TSC_OPEN_STAMP(i)                                                                                                     \
    do {                                                                                                                      \
        if (oa.dbg) {                                                                                                         \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                       \
            if ((threadIdx.x & 63) == 0) oa.dbg[size_t(blockIdx.x) * 32 + (threadIdx.x >> 6) * 8 + (i)] = wall_clock64();       \
        }                                                                                                                     \
    } while (0)
#else
#define TSC_OPEN_STAMP(i) do { } while (0)
#endif
inline __global__ __launch_bounds__(256) void k_open_rows(PassGeom g, OpenArgs oa, StepCtx sc, StepArgs next, int32_t *__restrict__ act,
                                                    int32_t *__restrict__ cend, int32_t *__restrict__ best, int32_t *__restrict__ tile_cmax,
                                                    const float *__restrict__ D, float *__restrict__ Dc) {
    static_assert(SCAN_BLOCK_WORDS == 32 && OPEN_ROWS == 64 && DESC_WORDS == 16, "one wavefront = 64 rows = four row tiles; a block's 32 words on 32 lanes");
    __shared__ int s_boff[OPEN_LDS_BLOCKS + 1];
    __shared__ unsigned long long s_w[4][SCAN_BLOCK_WORDS];   // per wavefront: the words of the scan block staged last ...
    __shared__ int s_pre[4][SCAN_BLOCK_WORDS];                // ... and the set bits below each of them
    __shared__ int s_list[4][OPEN_LIST_CAP];                  // per wavefront: deltas of the cache view's set bits inside its chunk, ascending
    const PruneState *st = sc.st;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (the first 256 entries of the prefix are requested together with the state block: one round trip for both)
    const bool in_lds = oa.n_blocks <= oa.lds_cap;
    const int boff_mine = (in_lds && int(threadIdx.x) <= oa.n_blocks) ? oa.boff[threadIdx.x] : 0;
    TSC_OPEN_STAMP(0);  // started (and the first loads have come back)
    // (row_lo / n_all differ from 0 / A only in a rank-partitioned pass: rows and ranks below are LOCAL except where the mask's
    // own ranking -- the prefix of the scan blocks -- is consulted)
    // One read of the state per WORKGROUP: in a fused pass that is gated off the last tile to arrive -- possibly a wavefront of this
    // very launch -- opens the NEXT pass and rewrites the state block, and a wavefront of this block that started late would then
    // see the next pass's gate while its siblings saw this one's (a barrier below is conditional on it)
    __shared__ int s_state[5];
    if (threadIdx.x == 0) s_state[0] = st->pass_on, s_state[1] = st->A, s_state[2] = st->bitsel, s_state[3] = st->row_lo, s_state[4] = st->n_active;
    __syncthreads();
    const int pass_on = s_state[0], A = s_state[1], sel = s_state[2], row_lo = s_state[3], n_all = s_state[4];
    const unsigned long long *X = oa.bits + size_t(sel) * oa.bit_words;
    const unsigned wave = blockIdx.x * 4 + wid;          // rows 64 wave .. 64 wave + 63, tiles 4 wave .. 4 wave + 3
    const int R0 = int(wave) * OPEN_ROWS;
    const unsigned tile = wave * 4 + unsigned(lane >> 4);
    if (pass_on) {
        unsigned long long *Xo = oa.bits + size_t(sel ^ 1) * oa.bit_words;
        for (int w = blockIdx.x * 256 + threadIdx.x; w < oa.bit_words; w += gridDim.x * 256) Xo[w] = X[w];
    }
    if (pass_on && in_lds && int(blockIdx.x) * 4 * OPEN_ROWS < A) {  // (block-uniform)
        if (int(threadIdx.x) <= oa.n_blocks) s_boff[threadIdx.x] = boff_mine;
        for (int e = threadIdx.x + 256; e <= oa.n_blocks; e += 256) s_boff[e] = oa.boff[e];
        __syncthreads();
    }
    if (wave * 4 >= oa.n_tiles) return;  // (padding of the last block)
    TSC_OPEN_STAMP(1);  // bit copy made, prefix staged
    int my_c = 0;        // stop column of this lane's row (0 for lanes without one: the tile maxima below)
    if (pass_on && R0 < A) {
        // The kernel is a chain of dependent memory round trips (a light pass is bound by it, not by work): loads that do not depend on
        // each other are issued TOGETHER -- (1) state + prefix, above; (2) the words of the rows' scan blocks; (3) descriptors, the cache
        // view's summary and the words of the stop positions' scan block; then the stores.
        auto before = [&](int b) { return in_lds ? s_boff[b] : oa.boff[b]; };
        unsigned long long *sw = s_w[wid];
        int *sp = s_pre[wid];
        auto load_block = [&](int b) -> unsigned long long {   // word `lane` of scan block b (zeros beyond the last block)
            return (lane < SCAN_BLOCK_WORDS && b < oa.n_blocks) ? X[size_t(b) * SCAN_BLOCK_WORDS + lane] : 0ull;
        };
        // a block's 32 words and the set bits below each of them -> this wavefront's LDS area
        auto stage_words = [&](unsigned long long w) {
            __builtin_amdgcn_wave_barrier();   // (everybody has finished with the block staged before)
            const int c = __popcll(w);
            int incl = c;
#pragma unroll
            for (int off = 1; off < SCAN_BLOCK_WORDS; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            if (lane < SCAN_BLOCK_WORDS) sw[lane] = w, sp[lane] = incl - c;
            __builtin_amdgcn_wave_barrier();
        };
        const int r_true = R0 + lane;
        const bool mine = r_true < A;
        const int r = mine ? r_true : A - 1;  // (idle lanes walk along with the last row: the loops below stay convergent)
        const int rg = r + row_lo;            // its rank among ALL active structures
        // scan block of rank rg: the last b with boff[b] <= rg  (boff[0] = 0, boff[n_blocks] = n_all > rg).  The wavefront's rows are
        // consecutive ranks: the block of its FIRST row by a 64-way search (the lanes test 64 boundaries at once, coarse then fine --
        // three trips to LDS for up to 262 144 blocks instead of one per halving), the others a boundary or two further on
        int lo;
        {
            const int rg0 = __builtin_amdgcn_readfirstlane(rg);
            int base = 0, span = oa.n_blocks;   // invariant: boff[base] <= rg0 < boff[base + span]
            while (span > 1) {
                const int step = (span + 63) >> 6;
                const int b = base + lane * step;
                const unsigned long long le = __ballot(b < base + span && before(b) <= rg0);   // lane 0 always (the invariant)
                const int q = __popcll(le) - 1;           // boundaries are ascending: the set lanes are 0 .. q
                base += q * step;
                span = min(step, oa.n_blocks - base);
            }
            lo = base;
            while (__ballot(lo + 1 < oa.n_blocks && before(lo + 1) <= rg))
                if (lo + 1 < oa.n_blocks && before(lo + 1) <= rg) ++lo;
        }
        const int rem = rg - before(lo);      // the row is the rem-th active structure of its scan block
        int pos = 0;
        {
            // the (one or two, on sparse masks more) distinct scan blocks of this wavefront's rows: the first and the last are requested
            // together
            const int b_first = __builtin_amdgcn_readfirstlane(lo), b_last = __builtin_amdgcn_readlane(lo, 63);
            const unsigned long long w_first = load_block(b_first);
            const unsigned long long w_last = b_last != b_first ? load_block(b_last) : 0ull;
            for (unsigned long long pending = __ballot(true); pending;) {
                const int b = __builtin_amdgcn_readlane(lo, __ffsll((long long)pending) - 1);
                stage_words(b == b_first ? w_first : (b == b_last ? w_last : load_block(b)));
                if (lo == b) {
                    int u = 0;                     // the last word with sp[u] <= rem
#pragma unroll
                    for (int s2 = SCAN_BLOCK_WORDS / 2; s2 > 0; s2 >>= 1)
                        if (sp[u + s2] <= rem) u += s2;
                    pos = 64 * u + select64(sw[u], rem - sp[u]);
                }
                pending &= ~__ballot(lo == b);
            }
        }
        const int64_t i = int64_t(lo) * oa.block_items + pos;
        int64_t first, last;
        chunk_of(g, i, first, last);
        TSC_OPEN_STAMP(2);  // the row's structure found
        // candidate deltas d = 1 .. len, i.e. cache-view positions P = first + d in [p_lo, p_hi]; the view is sparse (a
        // key applies to a pass only when its chunk start is one of this pass's), so the row walks the NON-EMPTY 1024-bit
        // blocks of it, found through the summary bitmap, instead of every block of its chunk
        const unsigned long long *dbit = oa.view, *dsum = oa.view + oa.bit_words;
        const int64_t len = mine ? last - i - 1 : 0;
        const int64_t p_lo = first + 1, p_hi = first + len, shift = i - first;  // mask position of P is P + shift
        const bool walk = oa.use_cache != 0 && len > 0;
        // one batch of requests: the descriptor, the first summary word of the cache view, the scan block of the chunk's end
        f32x4 dval[4];
        if (D) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dval[q] = *reinterpret_cast<const f32x4 *>(D + i * 16 + 4 * q);
        }
        const unsigned long long sum_first = walk ? dsum[(p_lo >> 10) >> 6] : 0ull;
        const int bl_first = __builtin_amdgcn_readfirstlane(int(last / oa.block_items));
        const unsigned long long wl_first = load_block(bl_first);
        // The stop column is the end of the chunk unless the cache view has a hit: the first delta d >= 1 whose key (first, first + d) is in the
        // view AND whose column i + d is active (rmsd_pruning.py:65-67).
        // Rows of ONE chunk (every wavefront of a late pass) look at the same view bits, shifted against the mask by their own position: the
        // wavefront lists the view's set bits once (in LDS, ascending, 64 or more at a time, walking the non-empty 1024-bit blocks through the
        // summary) and every row tests ITS mask bit behind each of them, eight at a time.  A hit comes after a few tests -- a row's
        // column is active more often than not -- where a walk word by word and row by row was as long as the unluckiest of the 64 rows
        // (the k = 1 pass of C3: 68 us for 21 000 rows).  Wavefronts whose rows lie in several chunks (early passes: short chunks, nearly empty
        // views) walk lane by lane, below.
        const bool one_chunk = __ballot(first != __shfl(first, 0)) == 0;
        int64_t found = last;
        if (oa.use_cache != 0 && one_chunk) {
            int *list = s_list[wid];
            const int64_t dmax = __shfl(len, 0);              // rows ascend: lane 0 has the longest range (and is always a row of the pass)
            const unsigned long long sum0 = (unsigned long long)__shfl((long long)sum_first, 0);   // (lanes without a range of their own loaded none)
            int64_t Bc = (first + 1) >> 10;
            const int64_t Bh = (first + dmax) >> 10;
            bool scanning = len > 0;
            while (__ballot(scanning)) {
                int n_list = 0;
                while (n_list < 64 && Bc <= Bh) {             // (wave-uniform: every lane holds the same first, dmax, Bc, n_list)
                    // up to 16 of the next non-empty blocks at once, four lanes per block and four words per lane: a sparse view (C3's
                    // last pass: 96 keys over 30 blocks) is listed in one round trip, not one per block
                    const unsigned long long sb = ((Bc >> 6) == ((p_lo >> 10) >> 6) ? sum0 : dsum[Bc >> 6]) >> (Bc & 63);   // bit t: block Bc + t
                    if (!sb) {
                        Bc = ((Bc >> 6) + 1) << 6;
                        continue;
                    }
                    const int j = lane >> 2, q = lane & 3;
                    int64_t Bj = -1;
                    if (j < __popcll(sb)) {
                        Bj = Bc + select64(sb, j);
                        if (Bj > Bh) Bj = -1;
                    }
                    unsigned long long w4[4] = {0ull, 0ull, 0ull, 0ull};
                    int c = 0;
                    if (Bj >= 0) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int64_t p0 = (Bj * 16 + 4 * q + u) * 64;
                            unsigned long long w = 0ull;
                            if (p0 <= first + dmax && p0 + 63 >= first + 1) {
                                w = dbit[p0 >> 6];
                                if (p0 < first + 1) w &= (first + 1 - p0 < 64) ? (~0ull << (first + 1 - p0)) : 0ull;
                                if (first + dmax - p0 < 63) w &= (2ull << (first + dmax - p0)) - 1ull;
                            }
                            w4[u] = w;
                            c += __popcll(w);
                        }
                    }
                    int incl = c;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int t = __shfl_up(incl, off);
                        if (lane >= off) incl += t;
                    }
                    // whole blocks, in order, as long as the list has room (the first always fits: a block holds 1024 keys at most)
                    const bool fits = Bj >= 0 && __shfl(incl, lane | 3) <= OPEN_LIST_CAP - n_list;
                    const int n_valid = __popcll(__ballot(Bj >= 0 && q == 3)), n_fit = __popcll(__ballot(fits && q == 3));
                    if (fits) {
                        int kk = n_list + incl - c;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int64_t p0 = (Bj * 16 + 4 * q + u) * 64;
                            for (unsigned long long w = w4[u]; w; w &= w - 1) list[kk++] = int(p0 + (__ffsll((long long)w) - 1) - first);
                        }
                    }
                    if (n_valid == 0) {
                        Bc = Bh + 1;                            // the summary's next blocks lie beyond the chunk
                    } else {
                        n_list += __shfl(incl, 4 * n_fit - 1);
                        Bc = __shfl(Bj, 4 * n_fit - 1) + 1;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                TSC_OPEN_STAMP(6);  // (the list refilled)
                if (n_list == 0) break;                        // the view has no further key inside the chunk: no row stops early
                int kidx = 0;
                while (__ballot(scanning && kidx < n_list)) {
                    if (scanning && kidx < n_list) {
                        int dd[8];
                        unsigned long long mw[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) dd[t] = kidx + t < n_list ? list[kidx + t] : INT_MAX;
#pragma unroll
                        for (int t = 0; t < 8; ++t) mw[t] = int64_t(dd[t]) <= len ? X[(i + dd[t]) >> 6] : 0ull;
                        int hit = -1;
#pragma unroll
                        for (int t = 7; t >= 0; --t)
                            if (int64_t(dd[t]) <= len && ((mw[t] >> ((i + dd[t]) & 63)) & 1ull)) hit = dd[t];
                        bool beyond = false;                   // the (ascending) list has left this row's range
#pragma unroll
                        for (int t = 0; t < 8; ++t) beyond = beyond || (dd[t] != INT_MAX && int64_t(dd[t]) > len);
                        if (hit >= 0) found = i + hit, scanning = false;
                        else if (beyond) scanning = false;
                        kidx += 8;
                    }
                }
                __builtin_amdgcn_wave_barrier();               // (before the list is refilled)
                TSC_OPEN_STAMP(7);  // (the list tested)
                // a row still scanning here has used up this list: on to the next non-empty blocks (none left: n_list = 0 above)
                if (Bc > Bh) {
                    // (the range is exhausted: rows that have not hit anything keep found = last)
                    break;
                }
            }
        }
        if (walk && !one_chunk) {
            int64_t B = p_lo >> 10;
            const int64_t B_hi = p_hi >> 10, S_first = (p_lo >> 10) >> 6;
            bool scanning = true;
            while (scanning) {
                bool any = false;            // next non-empty block at or after B
                while (B <= B_hi) {
                    const unsigned long long sb = ((B >> 6) == S_first ? sum_first : dsum[B >> 6]) >> (B & 63);
                    if (sb) {
                        B += __ffsll((long long)sb) - 1;
                        any = B <= B_hi;
                        break;
                    }
                    B = ((B >> 6) + 1) << 6;
                }
                if (!any) break;
                // the 16 words of the block: requested together (one round trip, not one per word), then the mask words behind the
                // non-empty ones; in ascending order the first cached column (mask position) ends the walk
                unsigned long long vw[16], xw[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t p0 = (B * 16 + u) * 64;  // first position of this word
                    unsigned long long w = (p0 <= p_hi && p0 + 63 >= p_lo) ? dbit[p0 >> 6] : 0ull;
                    if (p0 < p_lo) w &= (p_lo - p0 < 64) ? (~0ull << (p_lo - p0)) : 0ull;
                    if (p_hi - p0 < 63) w &= p_hi >= p0 ? ((2ull << (p_hi - p0)) - 1ull) : 0ull;
                    vw[u] = w;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) xw[u] = vw[u] ? extract64(X, (B * 16 + u) * 64 + shift) : 0ull;
#pragma unroll
                for (int u = 15; u >= 0; --u) {
                    const unsigned long long w = vw[u] & xw[u];
                    if (w) found = (B * 16 + u) * 64 + (__ffsll((long long)w) - 1) + shift, scanning = false;
                }
                ++B;
                if (B > B_hi) scanning = false;
            }
        }
        TSC_OPEN_STAMP(3);  // the cache view walked
        // rank of the stop position = active structures before it: the prefix of its scan block + the set bits of the block below it.
        // The rows of a wavefront share their targets' scan blocks (the chunk's end; a hit a few positions behind its row): one staging
        // per distinct block
        {
            const int bf = int(found / oa.block_items), off = int(found - int64_t(bf) * oa.block_items);
            for (unsigned long long pending = __ballot(true); pending;) {
                const int b = __builtin_amdgcn_readlane(bf, __ffsll((long long)pending) - 1);
                stage_words(b == bl_first ? wl_first : load_block(b));
                if (bf == b) {
                    const int u = off >> 6, below = off & 63;
                    const int cnt = sp[u] + __popcll(sw[u] & (below ? (~0ull >> (64 - below)) : 0ull));
                    my_c = (b < oa.n_blocks ? before(b) : n_all) + cnt - row_lo;
                }
                pending &= ~__ballot(bf == b);
            }
        }
        TSC_OPEN_STAMP(4);  // stop column's rank
        if (mine) {
            if (D) {
                if (Dc)   // (null where nothing reads the fp32 rows by position: the 16-row matrix-core kernel of a pass that cannot be culled)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(Dc + int64_t(r) * 16 + 4 * q) = dval[q];
                if (oa.Dh) {   // (made here every pass; copying them from records made once per run was measured: no faster, 80 B more per structure)
                    float dv[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) dv[4 * q] = dval[q].x, dv[4 * q + 1] = dval[q].y, dv[4 * q + 2] = dval[q].z, dv[4 * q + 3] = dval[q].w;
                    mm_write_record(dv, mm_scale(*oa.dmax_bits), oa.Dh + int64_t(r) * MM_REC_HALVES);
                }
            }
            act[r] = int32_t(i);
            cend[r] = my_c;
            best[r] = INT_MAX;  // atomicMin target of the pair kernel: no similar column found yet
            if (oa.rank_of) oa.rank_of[i] = r;
        }
        if (oa.rank_of) {  // a pass that may be culled: how many pairs lie inside the rows' ranges (what the ordered walk would look at)
            long long w = mine ? (long long)max(0, my_c - r - 1) : 0ll;
            for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off);
            if (lane == 0) count_add(sc.cnt, wave, CNT_WALK, (unsigned long long)w);
        }
        if (!mine) my_c = 0;
        TSC_OPEN_STAMP(5);  // everything written
    }
    // largest stop column of the 16 rows of every tile: lets a work item of the pair kernel whose column segment lies
    // beyond it leave after one scalar load (0: every work item of the tile leaves at its first test -- also for a pass that is
    // gated off and for tiles beyond the active count)
    int cmax = my_c;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
    const bool tile_lane = (lane & 15) == 0 && tile < oa.n_tiles;
    if (tile_lane) tile_cmax[tile] = cmax;
    // no work item of the pair kernel will find a column for this tile: it has nothing to apply and arrives here
    const bool dead = tile_lane && cmax <= ((int(tile) * 16 + 1) & ~63);
    if (!oa.fused || !__ballot(dead)) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool fin = dead && tickets_arrive(oa.tickets, tile, oa.n_tiles, PT_GROUPS);
    if (__ballot(fin)) pass_step_wave(sc, next);
}

// The verdicts of up to 64 rows of a finished pass, one row per lane (a whole wavefront calls this): rows with a similar
// column are removed (:113) -- mask byte, the bit in the copy the NEXT pass reads, the count of their scan block -- and
// leave one cache key each (:76, appended after the pass at :204), entered into the views of the later passes; counts what
// the reference's sequential scan would have evaluated.  best[] was written by other CUs' atomics: read at the L2.
struct ApplyArgs {
    PassGeom g;
    const int32_t *act, *cend;
    const int32_t *best;
    uint8_t *mask;
    unsigned long long *bits;
    int bit_words;
    int32_t *bsum;
    int block_items;
    CacheViews cv;
    unsigned long long *exch;  // rank-partitioned pass (else null): removed rows are only NOTED here, one bit each; mask, bit copy and
                               // block counts follow in k_pass_merge, from the sum of every rank's notes
};
__device__ inline void apply_wave_rows(const ApplyArgs &a, int sel, int r, bool valid, unsigned long long &ev_total, unsigned long long &rm_total) {
    const int lane = threadIdx.x & 63;
    unsigned long long ev = 0;
    bool removed = false;
    int my_block = -1;
    int64_t first = 0, delta = 0;
    if (valid) {
        const int b = __hip_atomic_load(&a.best[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != INT_MAX) {
            const int64_t i = a.act[r], j = a.act[b];
            int64_t last;
            chunk_of(a.g, i, first, last);
            if (a.exch) {
                atomicOr(&a.exch[i >> 6], 1ull << (i & 63));
            } else {
                a.mask[i] = 0;
                atomicAnd(&a.bits[size_t(sel ^ 1) * a.bit_words + (i >> 6)], ~(1ull << (i & 63)));
                my_block = int(i / a.block_items);
            }
            delta = j - i;
            removed = true;
            ev = (unsigned long long)(b - r);  // columns r+1 .. b were evaluated
        } else {
            ev = (unsigned long long)(a.cend[r] - r - 1);  // every active column before the stop column
        }
    }
    // the per-block counts follow the mask: one atomic per (wavefront, scan block) -- the removed rows of a wavefront fall
    // into one or two blocks, and thousands of single decrements of two cache lines would serialise
    for (unsigned long long left = __ballot(removed && my_block >= 0); left;) {
        const int l = __ffsll((long long)left) - 1;
        const int blk = __shfl(my_block, l);
        const unsigned long long same = __ballot(removed && my_block == blk);
        if (lane == l) atomicSub(&a.bsum[blk], __popcll(same));
        left &= ~same;
    }
    views_insert_wave(a.cv, removed, int(first), int(first + delta));
    const int n_rm = __popcll(__ballot(removed));
    for (int off = 32; off > 0; off >>= 1) ev += __shfl_down(ev, off);
    ev_total += ev, rm_total += (unsigned long long)n_rm;
}

// Gather the active structures into the two layouts the tile kernel reads:
//   Xr[r][HP3]   row-major, zero-padded to HP atoms  (row structures: wavefront-uniform scalar reads)
//   Xc[d][ld]    coordinate-major                    (column structures: lane = column, coalesced)
//   G[r] = sum |x|^2
// One block moves 64 structures through an LDS tile so that both global sides are coalesced.
inline __global__ __launch_bounds__(256) void k_compact_coords(const double *__restrict__ heavy, int h, int hp3,
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
// H FORMED FROM A FLOAT32 COPY of the coordinates (sieve.hpp, pair_H32: the first look at a pair that passed the screen reads half the
// bytes): p32 = p (1 + d), |d| <= 2^-24, products of two floats are exact in float64 and summed there, so
// |dH_ij| <= (2^-23 + 2^-48 + h u) sum_a |p_ai q_aj| and |dH|_F <= gamma32 S with gamma32 = 2.0001 * 2^-24 + h u.  The derivation above goes
// through with that gamma in the H terms (32 gamma32 rho^4) while the test point keeps its float64 bound (12 h u rho): below
// 64 gamma32 rho^4 together.  Such a test decides every pair whose margin is not tiny (|P(L)| is some 1e-3 .. 1e-1 of its term sum for
// the pairs a prune meets, kappa about 8e-6); the others are formed again from the float64 coordinates.
constexpr double QUARTIC_KAPPA_MIN = 1e-12;
__device__ inline double quartic_gamma64(int h) { return 1.1102230246251565e-16 * double(h); }
__device__ inline double quartic_gamma32(int h) { return 2.0001 * 5.9604644775390625e-08 + 1.1102230246251565e-16 * double(h); }
__device__ inline double quartic_kappa_g(double gamma, double half_sum, double L) {
    const double rho = half_sum / L, r2 = rho * rho;          // callers test L > 0 themselves
    const double k = 64.0 * gamma * r2 * r2;
    return (k > QUARTIC_KAPPA_MIN) ? k : QUARTIC_KAPPA_MIN;  // (a NaN k -- L == 0 -- keeps the minimum; the L > 0 test fails then)
}
__device__ inline double quartic_kappa(int h, double half_sum, double L) { return quartic_kappa_g(quartic_gamma64(h), half_sum, L); }

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
__device__ inline int pair_verdict_g(const double H[9], double half_sum, double half_h_thr2, double two_thr2, double gamma) {
    const Quartic q = horn_quartic(H);
    const double L = half_sum - half_h_thr2;
    if (quartic_above_top_root(q, L, quartic_kappa_g(gamma, half_sum, L))) return PAIR_DISSIMILAR;
    const double x = half_sum - two_thr2;
    if (two_thr2 > 0.0 && quartic_between_top_roots(q, x, quartic_kappa_g(gamma, half_sum, x))) return PAIR_SIMILAR;
    return PAIR_UNDECIDED;
}
__device__ inline int pair_verdict(const double H[9], double half_sum, double half_h_thr2, double two_thr2, int h) {
    return pair_verdict_g(H, half_sum, half_h_thr2, two_thr2, quartic_gamma64(h));
}

// One wavefront = one work item = (row tile of TI consecutive active rows) x (one column segment).
// Lane = column.  For every 64-column tile: load q_j into registers, then for each live row accumulate
// H = p_i^T q_j, run the sign test, remember survivors as per-lane bits; after the row loop the few
// survivors take the exact path and the smallest similar column of each row goes to best[] (atomicMin).
// A row stops being visited once it has a similar column to its left (the reference returns at the
// first similar column, rmsd_pruning.py:75-77).
template <int HP, int TI>
inline __global__ __launch_bounds__(256, 2) void k_rmsd_tile(const double *__restrict__ Xr, const double *__restrict__ Xc,
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

// Apply a finished pass as a launch of its own (the register-tiled kernel's passes, and passes whose rows were searched by
// several ranks: best[] is complete only after the exchange).  The sieve kernel of a single-rank run does this itself, tile
// by tile (sieve.hpp).
inline __global__ __launch_bounds__(256) void k_apply_pass(ApplyArgs a, StepCtx sc, StepArgs next) {
    __shared__ int s_last;
    PruneState *st = sc.st;
    const bool pass_on = st->pass_on != 0;
    const int n_active = pass_on ? st->A : 0;  // a pass that is gated off has no rows; its blocks still take a ticket
    const int sel = st->bitsel;
    const int lane = threadIdx.x & 63;
    unsigned long long ev_total = 0, rm_total = 0;
    // a capped grid walks the rows tile by tile (block-uniform bounds: every wavefront keeps all its lanes for the ballots)
    for (int row0 = blockIdx.x * 256; row0 < n_active; row0 += gridDim.x * 256) {
        const int r = row0 + threadIdx.x;
        apply_wave_rows(a, sel, r, r < n_active, ev_total, rm_total);
    }
    if (lane == 0) {
        const unsigned bucket = blockIdx.x * 4 + (threadIdx.x >> 6);
        count_add(sc.cnt, bucket, CNT_EVALUATED, ev_total);
        count_add(sc.cnt, bucket, CNT_REMOVED, rm_total);
    }
    // The last block to get here closes this pass and opens the next one (what a separate one-block launch would do).
    // The only data handed from block to block are the statistics buckets and block counts, and those are written by
    // agent-scope atomics and read with agent-scope (sc1) loads -- so no cache fence is needed (a __threadfence() per block
    // costs an L2 write-back each: measured 14 us per pass): every wave drains its atomics, the block's barrier, ONE lane's
    // ticket add, and the block whose add came last reads (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(&st->ticket, 1u);
        s_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) pass_step_wave(sc, next);
}

// ---------------------------------------------------------------------------------------------------
// Rank-partitioned passes (one process per GPU, tscode_amd/pipeline.py).  The chunks of a pass are independent
// (rmsd_pruning.py:139-157: every chunk reads the same input mask and the same cache), so while a pass has several chunks
// per rank each rank takes the chunks that START inside its block [n r / W, n (r + 1) / W) of the structure axis and runs
// the whole pass flow on them alone -- k_open_rows over its rows only, the pair kernel, the tile-wise apply.  What it learns
// is which rows it removed: one bit each in an exchange buffer (n / 64 words + the pass's five statistics).  The host sums
// the ranks' buffers (the bits are disjoint, so the sum is their union; NCCL / RCCL have no bitwise reduction) and
// k_pass_merge makes the pass's outcome the state of every rank: mask bytes, the bit copy the next pass reads, the scan
// blocks' counts and prefix, the record, the gate of rmsd_pruning.py:192 for the next pass and -- when that one is
// rank-partitioned too -- which of the active structures are this rank's rows in it.
// The cache keys of the removed rows stay with the rank that removed them: a key (a, b) can only be hit by a later pass in
// the chunk that starts at a (rmsd.hpp, CacheViews), and the chunk that starts at a belongs to the same rank in every
// rank-partitioned pass.  Before the first pass that is NOT partitioned this way the host sums the cache views of the
// remaining passes over the ranks once (disjoint bits again) and k_views_summaries rebuilds their summary words.

// Active structures before position pos (0 <= pos <= n) in the bit copy X: the prefix of its scan block + the set bits of
// the block below it.  One wavefront calls this.
__device__ inline int rank_below_wave(const int32_t *__restrict__ boff, const unsigned long long *__restrict__ X, int n_blocks, int n_active, int64_t pos) {
    const int lane = threadIdx.x & 63;
    const int bf = int(pos / (64 * SCAN_BLOCK_WORDS)), off = int(pos - int64_t(bf) * (64 * SCAN_BLOCK_WORDS));
    int cnt = 0;
    if (bf < n_blocks && lane < SCAN_BLOCK_WORDS) {
        const int below = off - 64 * lane;
        const unsigned long long m = below >= 64 ? ~0ull : (below > 0 ? (1ull << below) - 1ull : 0ull);
        cnt = m ? __popcll(X[size_t(bf) * SCAN_BLOCK_WORDS + lane] & m) : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    return (bf < n_blocks ? boff[bf] : n_active) + cnt;
}

// The rows of this rank in the OPEN pass: the active structures of [s_lo, s_hi) (whole chunks of that pass).  Needed as a
// launch of its own only where no k_pass_merge precedes the pass (the first pass of a run).
inline __global__ __launch_bounds__(64) void k_range_open(PruneState *__restrict__ st, const int32_t *__restrict__ boff, const unsigned long long *__restrict__ bits,
                                                    int bit_words, int n_blocks, int s_lo, int s_hi) {
    const unsigned long long *X = bits + size_t(st->bitsel) * bit_words;
    const int n_all = st->n_active;
    const int lo = rank_below_wave(boff, X, n_blocks, n_all, s_lo), hi = rank_below_wave(boff, X, n_blocks, n_all, s_hi);
    if ((threadIdx.x & 63) == 0) st->row_lo = lo, st->A = hi - lo;
}

struct MergeArgs {
    int n, bit_words, n_blocks;
    unsigned long long *bits;   // the two bit copies of the mask
    unsigned long long *exch;   // bit_words words of removed rows, then the five statistics: summed over the ranks by the host
    uint8_t *mask;
    int next_s_lo, next_s_hi;   // the next pass is rank-partitioned: this rank's structures [s_lo, s_hi) in it; -1: it is not
};
constexpr int MERGE_LDS_BLOCKS = 8192;  // scan blocks counted in LDS (16.7 M structures); beyond: every thread counts whole blocks from memory
inline __global__ __launch_bounds__(1024) void k_pass_merge(MergeArgs a, StepCtx sc, StepArgs sa) {
    __shared__ int s_cnt[MERGE_LDS_BLOCKS];
    __shared__ int s_part[16], s_run;
    PruneState *st = sc.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pass_on = st->pass_on, sel = st->bitsel, n_before = st->n_active;
    const bool closing = sa.prev >= 0 && pass_on != 0;
    const bool in_lds = a.n_blocks <= MERGE_LDS_BLOCKS;
    const unsigned long long *X = a.bits + size_t(sel) * a.bit_words;
    unsigned long long *Xo = a.bits + size_t(sel ^ 1) * a.bit_words;
    if (tid == 0) s_run = closing ? 0 : n_before;
    if (closing) {
        if (in_lds)
            for (int b = tid; b < a.n_blocks; b += 1024) s_cnt[b] = 0;
        __syncthreads();
        // the pass's outcome: the bit copy the next pass reads, the mask bytes of the removed rows, the scan blocks' counts
        for (int w = tid; w < a.bit_words; w += 1024) {
            const unsigned long long old = X[w];
            unsigned long long rm = a.exch[w];
            const unsigned long long now = old & ~rm;
            Xo[w] = now;
            if (in_lds && now) atomicAdd(&s_cnt[w / SCAN_BLOCK_WORDS], __popcll(now));  // (words behind the last block's are zero)
            if (rm) {
                a.exch[w] = 0;
                for (rm &= old; rm; rm &= rm - 1) a.mask[int64_t(w) * 64 + (__ffsll((long long)rm) - 1)] = 0;
            }
        }
        __syncthreads();  // (one block: its global and LDS writes are visible to all its threads behind the barrier)
        for (int b0 = 0; b0 < a.n_blocks; b0 += 1024) {
            const int b = b0 + tid;
            int c = 0;
            if (b < a.n_blocks) {
                if (in_lds) {
                    c = s_cnt[b];
                } else {
                    for (int u = 0; u < SCAN_BLOCK_WORDS; ++u)  // (the block list is padded beyond the last word of a bit copy)
                        c += b * SCAN_BLOCK_WORDS + u < a.bit_words ? __popcll(Xo[size_t(b) * SCAN_BLOCK_WORDS + u]) : 0;
                }
                sc.bsum[b] = c;
            }
            int incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            if (lane == 63) s_part[wv] = incl;
            __syncthreads();
            int base = s_run;
            for (int q = 0; q < wv; ++q) base += s_part[q];
            if (b < a.n_blocks) sc.boff[b] = base + incl - c;
            __syncthreads();
            if (tid == 1023) s_run = base + incl;
            __syncthreads();
        }
        if (tid == 0) sc.boff[a.n_blocks] = s_run;
    }
    __syncthreads();
    const int n_active = s_run;
    if (tid == 0) {
        if (closing) {
            PassRecord &r = sc.rec[sa.prev];
            r.formed = (long long)a.exch[a.bit_words + CNT_FORMED], r.exact = (long long)a.exch[a.bit_words + CNT_EXACT];
            r.screened = (long long)a.exch[a.bit_words + CNT_SCREENED], r.evaluated = (long long)a.exch[a.bit_words + CNT_EVALUATED];
            r.removed = (long long)(n_before - n_active);
            r.n_after = n_active;
            if (sa.prev_algo >= 0) r.algo = sa.prev_algo;
            st->n_active = n_active;
            st->bitsel = sel ^ 1;
        }
        for (int w = 0; w < 8; ++w) a.exch[a.bit_words + w] = 0;
        int on = 0;
        if (sa.cur >= 0) {
            on = (sa.k_cur == 1 || 20 * sa.k_cur < (long long)n_active) ? 1 : 0;  // rmsd_pruning.py:192
            PassRecord &r = sc.rec[sa.cur];
            r.k = sa.k_cur, r.n_before = n_active, r.n_after = n_active, r.on = on, r.algo = sa.algo_cur;
            r.formed = r.exact = r.screened = r.evaluated = r.removed = 0;
        }
        st->pass_on = on;
        st->ticket = 0;
        if (a.next_s_lo < 0) st->A = n_active, st->row_lo = 0;
    }
    if (a.next_s_lo >= 0 && wv == 0) {  // (boff and the new bit copy were written by this block, before the barriers above)
        const unsigned long long *Xn = closing ? Xo : X;
        const int lo = rank_below_wave(sc.boff, Xn, a.n_blocks, n_active, a.next_s_lo), hi = rank_below_wave(sc.boff, Xn, a.n_blocks, n_active, a.next_s_hi);
        if (lane == 0) st->row_lo = lo, st->A = hi - lo;
    }
}

// Summary words of `count` cache views (one bit per 1024 view bits, CacheViews) rebuilt from their bits: after the host has
// summed the views of the remaining passes over the ranks, the summed summary words mean nothing.
inline __global__ __launch_bounds__(256) void k_views_summaries(unsigned long long *__restrict__ views, long long stride, int bit_words, int dsum_words, int count) {
    // one thread per summary BIT (16 view words); the 64 bits of a summary word sit in one wavefront and meet by ballot
    const int per_view = dsum_words * 64;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < int64_t(count) * per_view; e += int64_t(gridDim.x) * blockDim.x) {
        const int v = int(e / per_view), sb = int(e - int64_t(v) * per_view);
        unsigned long long *bits = views + int64_t(v) * stride;
        const int64_t w0 = int64_t(sb) * 16;
        unsigned long long any = 0;
        if (w0 < bit_words) {
#pragma unroll
            for (int u = 0; u < 16; ++u) any |= (w0 + u < bit_words) ? bits[w0 + u] : 0ull;
        }
        const unsigned long long word = __ballot(any != 0);
        if ((threadIdx.x & 63) == 0) bits[bit_words + (sb >> 6)] = word;
    }
}

// End of a run: the pass records and (optionally) the survivor mask go to host-visible memory from ONE small launch instead
// of two copy commands.  `rec_host` and `mask_host` are pinned host allocations mapped into the device's address space.
inline __global__ __launch_bounds__(256) void k_export_run(const unsigned long long *__restrict__ rec, int rec_words, unsigned long long *__restrict__ rec_host,
                                                     const unsigned long long *__restrict__ mask, int64_t mask_words, const uint8_t *__restrict__ mask_bytes,
                                                     int64_t n, unsigned long long *__restrict__ mask_host, const unsigned *__restrict__ dmax_bits) {
    const int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = tid; e < rec_words; e += stride) rec_host[e] = rec[e];
    if (tid == 0) rec_host[rec_words] = dmax_bits ? (unsigned long long)*dmax_bits : 0ull;  // (behind the records: did the descriptor build meet a non-finite structure)
    if (mask_host) {
        for (int64_t e = tid; e < mask_words; e += stride) mask_host[e] = mask[e];
        uint8_t *tail = reinterpret_cast<uint8_t *>(mask_host);
        for (int64_t b = mask_words * 8 + tid; b < n; b += stride) tail[b] = mask_bytes[b];
    }
}

inline __global__ void k_fill_i32(int32_t *p, int64_t n, int32_t v) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) p[i] = v;
}

}  // namespace tsc
