// sieve.hpp -- K3 with a rotation-invariant descriptor sieve in front of the Kabsch evaluation.
//
// rmsd_and_max_numba (tscode/rmsd_pruning.py:6-41) rotates p onto q ABOUT THE ORIGIN (it never centres).
// Write dev_a = |p_a R - q_a| for the deviation of atom a after the optimal rotation R; h * rmsd^2 = sum_a dev_a^2.
// Two families of quantities are untouched by R, and each bounds the deviations from below:
//
//   family 0  atom norms:           n_a(x) = |x_a|                      dev_a           >= | n_a(p) - n_a(q) |
//   family 1  atom-pair distances:  d_ab(x) = |x_a - x_b|               dev_a + dev_b   >= | d_ab(p) - d_ab(q) |
//             over DISJOINT pairs (a, b = a + h/2), so that             dev_a^2 + dev_b^2 >= (d_ab(p) - d_ab(q))^2 / 2
//
// Hence, with f0(x) = (n_a)_a and f1(x) = (d_ab / sqrt 2)_(a,b):   h * rmsd(p,q)^2 >= |f(p) - f(q)|^2 >= |Q^T (f(p) - f(q))|^2
// for either family and any Q with orthonormal columns.  A pair for which one of the two right-hand sides exceeds
// h * thr^2 cannot have rmsd < thr (:75) and is dropped without forming H = p^T q.  Q (KD columns per family) spans the
// leading principal axes of the feature vectors of the ensemble; the choice of Q (and of the atom pairs) only
// changes how many pairs are dropped, never a verdict.  Everything that survives takes the same sign test and
// explicit-rotation path as the register-tiled kernel (rmsd.hpp).
//
// The screen runs in fp32 on mean-centred descriptors.  With M the largest |component| of the ensemble, every computed
// difference is within eta = 3 * 2^-24 * 2M of the exact one (rounding of both operands and of the subtraction), so
// s_exact >= s32 (1 - 2^-20) - 2 eta sqrt(KD s32); screen_limit32 turns h thr^2 into the fp32 limit above which that
// lower bound certainly exceeds h thr^2.  Pairs in the sliver between the two limits are simply not dropped.
//
// Pass kernel (k_rmsd_sieve): one wavefront = 16 rows x one column segment, lane = CPL columns of a tile.
//   screen : descriptors are read by position from a copy in active order that k_open_rows writes on its way (one float4
//            per lane of its 4 lanes per row), so no index load stands in front of a column tile; the two families of a
//            component sit side by side, so one v_pk_add_f32 + one v_pk_fma_f32 advance both
//            distances; the 16 row descriptors sit in LDS and are read as broadcasts; ONE compare per (row, tile) decides
//            whether any column is within the limit; survivors go to a per-wavefront LDS queue (ballot + prefix popcount);
//   drain  : stage 1, whenever the queue holds 64 pairs (or at the end, spread over several lanes per pair): H from the
//            two structures in memory, the quartic tests (reject / accept near-duplicates / undecided);
//            stage 2, the explicit rotation for the undecided, when 64 have gathered or at the end;
//            atomicMin(best[row], column);
//   apply  : (single-rank runs) the work item that finishes a row tile last removes the tile's rows that found a similar
//            column and enters their cache keys into the views of the later passes; the last tile of the pass closes it.
// Any number of heavy atoms is supported (no register-resident structure).
#pragma once
#include "common.hpp"
#include "embed_clash.hpp"
#include "rmsd.hpp"

namespace tsc {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KD = 8;              // descriptor dimensions per family
constexpr int NFAM = 2;            // feature families
constexpr int DW = KD * NFAM;      // doubles per structure descriptor
constexpr int DESC_SAMPLE = 4096;  // structures used to estimate the principal axes
constexpr int DESC_MAX_FEAT = 256; // features per family that enter the descriptor (any subset keeps the bound valid)
constexpr unsigned NONFINITE_BITS = 0x7fc00000u;  // what the running maximum of |descriptor| reads once a non-finite structure was seen

// floats per structure of the float32 copy of the heavy atoms (pair_stage1 below): x y z of four atoms per three float4, zero-padded
__host__ __device__ inline int heavy32_pitch(int h) { return 12 * ((h + 3) / 4); }

// feature a of family fam of the structure at x (h atoms, xyz triples)
__device__ inline double feature(const double *__restrict__ x, int h, int fam, int a) {
    if (fam == 0) return sqrt(x[3 * a] * x[3 * a] + x[3 * a + 1] * x[3 * a + 1] + x[3 * a + 2] * x[3 * a + 2]);
    const int b = a + h / 2;
    const double dx = x[3 * a] - x[3 * b], dy = x[3 * a + 1] - x[3 * b + 1], dz = x[3 * a + 2] - x[3 * b + 2];
    return sqrt(0.5 * (dx * dx + dy * dy + dz * dz));
}

inline int n_features(int h, int fam) { return std::min(fam == 0 ? h : h / 2, DESC_MAX_FEAT); }

// Second-moment matrix of the sampled feature vectors of one family, with a constant 1 appended:
// M[a][b] = sum_s f_a(s) f_b(s), a, b in [0, nf]  (index nf = the constant) -> mean and covariance on the host.
// blockIdx.y = family: both moment matrices come out of one launch.
// DETERMINISTIC: the sums are taken in a fixed order -- workgroup b of the MOM_BLOCKS of a family adds the chunks b, b + MOM_BLOCKS,
// ... into a partial matrix of its own (Mfam + b * m * m: every entry has one owner thread, no atomics) and the workgroup that
// finishes last adds the partials in index order.  The basis, the descriptors and with them the Morton order of a culled pass
// (cull.hpp) then come out bit for bit the same on every rank of a sharded run that feeds them the same sample: the ranks deal
// the tiles of that order among themselves, and orders that differed in one structure would let pairs go unvisited.
constexpr int MOM_BLOCKS = 32;
inline __global__ __launch_bounds__(256) void k_feature_moments(const double *__restrict__ heavy, int h, int nf0, int nf1, int64_t stride_structs,
                                                          int n_samples, double *__restrict__ M0, double *__restrict__ M1, unsigned *__restrict__ tickets) {
    // tickets == null: the fast form -- one workgroup per chunk of 32 samples, atomicAdd into the (zeroed) first partial matrix of the
    // family: the order of the additions, and with it the last bits of the basis, differ from run to run.  Any basis gives the same
    // verdicts; only a SHARDED run needs the same bits on every rank (option "deterministic_basis").
    // tickets[fam] (zero on entry): the workgroup of a family that finishes LAST adds the partial matrices up, in index order, into the first
    extern __shared__ __attribute__((aligned(16))) double s_n[];  // [chunk][nf + 1]
    __shared__ int s_last;
    const int fam = blockIdx.y, nf = fam == 0 ? nf0 : nf1;
    if (nf == 0) return;
    const int m = nf + 1;
    double *__restrict__ Mf = fam == 0 ? M0 : M1;
    constexpr int CHUNK = 32;
    const int n_chunks = (n_samples + CHUNK - 1) / CHUNK;
    if (!tickets) {
        for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
            const int s0 = ch * CHUNK, ns = min(CHUNK, n_samples - s0);
            for (int e = threadIdx.x; e < ns * m; e += blockDim.x) {
                int s = e / m, a = e - s * m;
                s_n[s * m + a] = (a < nf) ? feature(heavy + int64_t(s0 + s) * stride_structs * h * 3, h, fam, a) : 1.0;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
                int a = e / m, b = e - a * m;
                if (b < a) continue;
                double acc = 0.0;
                for (int s = 0; s < ns; ++s) acc += s_n[s * m + a] * s_n[s * m + b];
                atomicAdd(&Mf[e], acc);
            }
            __syncthreads();
        }
        return;
    }
    double *__restrict__ M = Mf + size_t(blockIdx.x) * m * m;
    bool first = true;
    for (int ch = blockIdx.x; ch < n_chunks || first; ch += gridDim.x) {
        const int s0 = ch * CHUNK, ns = ch < n_chunks ? min(CHUNK, n_samples - s0) : 0;   // (a workgroup without a chunk still writes zeros)
        for (int e = threadIdx.x; e < ns * m; e += blockDim.x) {
            int s = e / m, a = e - s * m;
            s_n[s * m + a] = (a < nf) ? feature(heavy + int64_t(s0 + s) * stride_structs * h * 3, h, fam, a) : 1.0;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
            int a = e / m, b = e - a * m;
            if (b < a) continue;
            double acc = first ? 0.0 : M[e];
            for (int s = 0; s < ns; ++s) acc += s_n[s * m + a] * s_n[s * m + b];
            M[e] = acc;
        }
        __syncthreads();
        first = false;
    }
    // Hand-off of the partial matrices to the workgroup that finishes last, in the form MI355X_MICROARCH.md lists as valid (inter-workgroup
    // visibility): every storing wavefront drains its stores, the workgroup meets, ONE lane releases at agent scope (write-back of the XCD's
    // L2) and -- behind an explicit wait the compiler cannot drop: its pass removes the one after buffer_wbl2 when it believes the scoreboard
    // empty, and the ticket could then overtake the write-back -- takes the ticket; the last workgroup acquires before it loads.  (Rounds
    // 2 - 3 had __threadfence() on both sides and no explicit wait; no wrong basis was ever traced to it.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = (atomicAdd(&tickets[fam], 1u) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
        int a = e / m, b = e - a * m;
        if (b < a) continue;
        // (this workgroup has not touched the other workgroups' partials before: the loads miss its CU's cache and come from the L2,
        // where the writers' stores are since their fences; all of an entry's partials in flight at once, added in index order)
        double v[MOM_BLOCKS], acc = 0.0;
#pragma unroll
        for (int u = 0; u < MOM_BLOCKS; ++u) v[u] = __builtin_nontemporal_load(&Mf[size_t(u) * m * m + e]);   // one round trip
#pragma unroll
        for (int u = 0; u < MOM_BLOCKS; ++u) acc += v[u];
        Mf[e] = acc;   // (entry e of partial 0 is read by this thread alone)
    }
}

// fp32 limit of the screen for an exact limit `limit` = h thr^2, from the largest descriptor magnitude dmax: every computed
// difference of two components is within eta = 3 * 2^-24 * 2 dmax of the exact one, the 8-term sum of squares carries at most
// 2^-20 relative error, so s_exact >= s32 (1 - 2^-20) - 2 eta sqrt(KD s32); the limit returned is the smallest s32 above which
// that lower bound certainly exceeds `limit`.  A NaN / inf dmax gives a limit that drops nothing.
__device__ inline float screen_limit32(float dmaxf, double limit) {
    const double dmax = (dmaxf >= 0.0f && dmaxf < 3.0e38f) ? double(dmaxf) : 3.0e38;
    const double eta = 3.0 * 5.9604644775390625e-08 * 2.0 * dmax;               // 3 * 2^-24 * 2 dmax
    const double a = 1.0 - 9.5367431640625e-07, b = 2.0 * eta * sqrt(double(KD));  // 1 - 2^-20
    const double x = (b + sqrt(b * b + 4.0 * a * limit)) / (2.0 * a);
    const double l32 = x * x * (1.0 + 1e-6) + 1e-30;
    float f = float(l32);
    if (double(f) < l32) f = __uint_as_float(__float_as_uint(f) + 1u);  // next float up (f > 0)
    return (l32 < 3.0e38) ? f : 3.4e38f;
}

// The same limit for the screen as the pair kernel computes it: S = fl(fl(|a|^2 + |b|^2) - 2 fl(a.b)) on the stored fp32
// vectors a, b (8-term fma chains), which costs 10 packed instructions per column instead of 16 for sum (a_k - b_k)^2.  With
// u = 2^-24 and g8 = 8u / (1 - 8u):  | |a|^2_fl - |a|^2 | <= g8 |a|^2,  | a.b_fl - a.b | <= g8 (|a|^2 + |b|^2) / 2,  so the
// exact T = |a - b|^2 satisfies  T >= S (1 - 2u) - E,  E = (2 g8 + u (1 + g8)) (|a|^2 + |b|^2) <= (2 g8 + u (1 + g8)) 2 KD M^2.
// The kernel's other association, S' = fl(|a|^2_fl - 2 fl(-|b|^2_fl / 2 + a.b)) (the chain of 8 fmas starts from the
// halved column norm, an exact scaling), carries g8 (|b|^2 (1 + g8) / 2 + (|a|^2 + |b|^2) / 2) in the chain, twice that after
// the exact factor, plus the two norm roundings: at most g8 (2 |a|^2 + 3 |b|^2)(1 + g8) <= 3 g8 (|a|^2 + |b|^2) (1 + g8)
// beside the final u: the E below (3 g8 in place of 2) covers both associations.
// The stored components are roundings of the exact descriptors, each difference within eta = 3 * 2^-24 * 2M of the exact
// one as above, so s_exact >= T - 2 eta sqrt(KD T) (increasing in T beyond KD eta^2): a pair whose S exceeds the returned
// value certainly has s_exact > limit.
__device__ inline float screen_limit32_dot(float dmaxf, double limit) {
    const double dmax = (dmaxf >= 0.0f && dmaxf < 3.0e38f) ? double(dmaxf) : 3.0e38;
    constexpr double U = 5.9604644775390625e-08, G8 = 8.0 * U / (1.0 - 8.0 * U);
    const double eta = 3.0 * U * 2.0 * dmax;
    const double b = 2.0 * eta * sqrt(double(KD));
    const double y = 0.5 * (b + sqrt(b * b + 4.0 * limit));   // sqrt of the smallest T with T - b sqrt(T) >= limit
    const double E = (3.0 * G8 * (1.0 + G8) + U * (1.0 + G8)) * 2.0 * double(KD) * dmax * dmax;
    const double l32 = (y * y + E) / (1.0 - 2.0 * U) * (1.0 + 1e-6) + 1e-30;
    float f = float(l32);
    if (double(f) < l32) f = __uint_as_float(__float_as_uint(f) + 1u);  // next float up (f > 0)
    return (l32 < 3.0e38) ? f : 3.4e38f;
}

// The descriptor rows of the ns structures staged in LDS (s_x, `pitch` doubles apart; s_q = both bases): T = 256 / S
// consecutive lanes share a structure (atoms sub, sub + T, ...) and reduce with shuffles.  Called by all 256 threads.
__device__ inline void describe_from_lds(const double *s_q, const double *s_x, int pitch, int h, int nf0, int nf1, const double *__restrict__ bias,
                                         int ns, int S, int64_t i0, float *__restrict__ D, double *__restrict__ G, unsigned *__restrict__ dmax_bits) {
    const int T = 256 / S;
    const int sidx = threadIdx.x / T, sub = threadIdx.x - sidx * T;
    const bool mine = sidx < ns;
    double d[DW], g = 0.0;
#pragma unroll
    for (int k = 0; k < DW; ++k) d[k] = 0.0;
    if (mine) {
        const double *x = s_x + sidx * pitch;
        const double *q1 = s_q + KD * nf0;
        for (int a = sub; a < h; a += T) {
            const double n2 = x[3 * a] * x[3 * a] + x[3 * a + 1] * x[3 * a + 1] + x[3 * a + 2] * x[3 * a + 2];
            g += n2;
            if (a < nf0) {
                const double f = sqrt(n2);
#pragma unroll
                for (int k = 0; k < KD; ++k) d[k] = fma(s_q[k * nf0 + a], f, d[k]);
            }
            if (a < nf1) {
                const double f = feature(x, h, 1, a);
#pragma unroll
                for (int k = 0; k < KD; ++k) d[KD + k] = fma(q1[k * nf1 + a], f, d[KD + k]);
            }
        }
    }
    for (int off = T >> 1; off > 0; off >>= 1) {  // T <= 64 consecutive lanes of one wavefront
#pragma unroll
        for (int k = 0; k < DW; ++k) d[k] += __shfl_xor(d[k], off);
        g += __shfl_xor(g, off);
    }
    float mx = 0.0f;
    if (mine && sub == 0) {
        float v[DW];
#pragma unroll
        for (int k = 0; k < DW; ++k) {
            v[(k % KD) * 2 + k / KD] = float(d[k] - bias[k]);
            mx = fmaxf(mx, fabsf(float(d[k] - bias[k])));
        }
        f32x4 *dst = reinterpret_cast<f32x4 *>(D + (i0 + sidx) * DW);
#pragma unroll
        for (int k = 0; k < DW / 4; ++k) dst[k] = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        G[i0 + sidx] = g;
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    // same-address atomics serialise (about 12 ns each): only a wavefront that would raise the maximum sends one
    if ((threadIdx.x & 63) == 0 && mx > __uint_as_float(__hip_atomic_load(dmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        atomicMax(dmax_bits, __float_as_uint(mx));
    // A structure with a NaN or infinite coordinate (its squared norm is not finite) raises the running maximum to the bit pattern of
    // a NaN: the screen's limit then drops nothing (screen_limit32*), the evaluation stages find such a structure similar to nothing
    // (every comparison with a NaN is false, rmsd_pruning.py:75), and the run reports that it saw one (tsc_pass_stats.nonfinite_input;
    // the reference's np.linalg.svd raises LinAlgError on such input, rmsd_pruning.py:19)
    if (__ballot(mine && sub == 0 && !(g < 1.7976931348623157e308)) != 0 && (threadIdx.x & 63) == 0) atomicMax(dmax_bits, NONFINITE_BITS);
}

// D[i][2k + fam] = sum_a Q_fam[k][a] * f_fam,a(x_i) - bias[fam*KD + k]   (fp32, original index space, the two families
// interleaved; the bias is the projection of the mean feature vector and cancels in every difference),
// G[i] = sum_a |x_ia|^2; *dmax_bits = max |D| over everything as the bit pattern of a non-negative float (atomicMax on the
// integer view; zero on entry) -- the pair kernel turns it into the fp32 limit of the screen (screen_limit32).
// A block stages S structures in LDS with coalesced loads (a thread-per-structure walk reads 64 lines per instruction and
// thrashes the L1); T = 256 / S consecutive lanes share a structure (atoms sub, sub + T, ...) and reduce with shuffles.
inline __global__ __launch_bounds__(256) void k_descriptors(const double *__restrict__ heavy, int64_t n, int h, int nf0, int nf1,
                                                      const double *__restrict__ Q, const double *__restrict__ bias, float *__restrict__ D,
                                                      double *__restrict__ G, unsigned *__restrict__ dmax_bits, int S) {
    extern __shared__ __attribute__((aligned(16))) double s_mem[];  // [KD][nf0], [KD][nf1], then S rows of pitch doubles
    const int h3 = h * 3, pitch = h3 | 1;
    double *s_q = s_mem, *s_x = s_mem + KD * (nf0 + nf1);
    for (int e = threadIdx.x; e < KD * (nf0 + nf1); e += 256) s_q[e] = Q[e];
    const int64_t i0 = int64_t(blockIdx.x) * S;
    const int ns = int(min<int64_t>(S, n - i0));
    const double *src = heavy + i0 * h3;
    // eight loads in flight per thread before the first LDS store (a plain copy loop waits for every load in turn)
    for (int base = threadIdx.x; base < ns * h3; base += 256 * 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (base + 256 * j < ns * h3) ? src[base + 256 * j] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = base + 256 * j;
            if (e < ns * h3) {
                const int row = e / h3;
                s_x[row * pitch + (e - row * h3)] = v[j];
            }
        }
    }
    __syncthreads();
    describe_from_lds(s_q, s_x, pitch, h, nf0, nf1, bias, ns, S, i0, D, G, dmax_bits);
}

// K1 with the descriptors of the prune fused in (tsc_pipeline_dev, when the basis is ready before the passing poses are
// embedded): k_transform's workgroup of TR_POSES poses keeps the heavy atoms it has just computed in LDS and takes their
// descriptor rows from there, so the 24 h bytes per structure are not read back by a k_descriptors launch.
// dynamic LDS: transform_lds_bytes(n_mols), then [KD][nf0 + nf1] basis doubles, then TR_POSES rows of (3 h | 1) doubles.
inline __global__ __launch_bounds__(256) void k_transform_describe(const double *__restrict__ frags, FragTable ft, const int32_t *__restrict__ conf_idx,
                                                             const double *__restrict__ rot, const double *__restrict__ pos,
                                                             const int32_t *__restrict__ idx, double *__restrict__ out,
                                                             const int32_t *__restrict__ heavy_slot, int n_heavy, double *__restrict__ heavy_out,
                                                             const int32_t *__restrict__ n_out_dev, int nf0, int nf1, const double *__restrict__ Q,
                                                             const double *__restrict__ bias, float *__restrict__ D, double *__restrict__ G,
                                                             unsigned *__restrict__ dmax_bits, float *__restrict__ heavy32_out = nullptr) {
    // heavy32_out (optional): the float32 copy of the heavy atoms that stage 1 of the pair kernels reads (pair_stage1), zero-padded rows
    extern __shared__ __attribute__((aligned(16))) double s_tr[];
    const int n = ft.n_total, nm = ft.n_mols, tid = threadIdx.x, pitch = (n_heavy * 3) | 1;
    const int pitch32 = heavy32_pitch(n_heavy), pad32 = pitch32 - 3 * n_heavy;
    const int64_t n_out = *n_out_dev;
    int64_t *sP = reinterpret_cast<int64_t *>(s_tr);
    double *sR = s_tr + TR_POSES, *sT = sR + TR_POSES * nm * 9;
    int *sC = reinterpret_cast<int *>(sT + TR_POSES * nm * 3);
    double *s_q = s_tr + (transform_lds_bytes(nm) + 15) / 16 * 2, *s_x = s_q + KD * (nf0 + nf1);
    for (int e = tid; e < KD * (nf0 + nf1); e += 256) s_q[e] = Q[e];
    for (int64_t r0 = int64_t(blockIdx.x) * TR_POSES; r0 < n_out; r0 += int64_t(gridDim.x) * TR_POSES) {
        const int np = int(min<int64_t>(TR_POSES, n_out - r0));
        if (tid < np) sP[tid] = int64_t(idx[r0 + tid]);
        __syncthreads();
        for (int q = tid; q < np * nm * 9; q += 256) {
            const int row = q / (nm * 9), w = q - row * nm * 9;
            sR[q] = rot[sP[row] * nm * 9 + w];
        }
        for (int q = tid; q < np * nm * 3; q += 256) {
            const int row = q / (nm * 3), w = q - row * nm * 3;
            sT[q] = pos[sP[row] * nm * 3 + w];
        }
        for (int q = tid; q < np * nm; q += 256) {
            const int row = q / nm, w = q - row * nm;
            sC[q] = conf_idx[sP[row] * nm + w];
        }
        __syncthreads();
        for (int e = tid; e < np * n; e += 256) {
            const int row = e / n, a = e - row * n;
            const int64_t r = r0 + row;
            double v[3];
            embed_atom_staged(frags, ft, sR, sT, sC, row, a, v);
            if (out) {
                double *o = out + (r * n + a) * 3;
                o[0] = v[0], o[1] = v[1], o[2] = v[2];
            }
            const int hs = heavy_slot[a];
            if (hs >= 0) {
                double *hv = heavy_out + (r * n_heavy + hs) * 3, *x = s_x + row * pitch + hs * 3;
                hv[0] = x[0] = v[0], hv[1] = x[1] = v[1], hv[2] = x[2] = v[2];
                if (heavy32_out) {
                    float *hf = heavy32_out + r * pitch32 + hs * 3;
                    hf[0] = float(v[0]), hf[1] = float(v[1]), hf[2] = float(v[2]);
                }
            }
        }
        if (heavy32_out)
            for (int e = tid; e < np * pad32; e += 256) heavy32_out[(r0 + e / pad32) * pitch32 + 3 * n_heavy + e % pad32] = 0.0f;
        __syncthreads();
        describe_from_lds(s_q, s_x, pitch, n_heavy, nf0, nf1, bias, np, TR_POSES, r0, D, G, dmax_bits);
        __syncthreads();
    }
}

inline size_t transform_describe_lds_bytes(int n_mols, int h) {
    return (transform_lds_bytes(n_mols) + 15) / 16 * 16 + (size_t(KD) * (n_features(h, 0) + n_features(h, 1)) + size_t(TR_POSES) * ((h * 3) | 1)) * sizeof(double);
}

// The trivial basis for small ensembles: component k of a family = its feature k (rows of the identity: |Q x| <= |x|), no
// centring.  Any basis gives the same verdicts; with a few thousand structures the screen has little to do, and estimating
// principal axes (a memset, a moments launch and the one-wavefront basis kernel: about 45 us of latency) costs more than it saves.
inline __global__ void k_identity_basis(int nf0, int nf1, double *__restrict__ Q, double *__restrict__ bias) {
    const int total = KD * (nf0 + nf1);
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int fam = e < KD * nf0 ? 0 : 1, r = fam == 0 ? e : e - KD * nf0, nf = fam == 0 ? nf0 : nf1;
        Q[e] = (r / nf == r % nf) ? 1.0 : 0.0;
    }
    for (int e = threadIdx.x; e < DW; e += blockDim.x) bias[e] = 0.0;
    if (threadIdx.x < NFAM) bias[DW + threadIdx.x] = __builtin_inf();  // (no estimate of the descriptors' spread: k_descriptor_basis)
}

// Device-side descriptor basis: one wavefront per feature family.  Orthonormal rows spanning the leading principal
// axes of the feature covariance by a few steps of block power iteration (the spectrum of these features decays
// fast: one step gives the screening power of three to within 3 % of the pairs that reach H).  Rows are re-orthonormalised by
// CholeskyQR2: lane (i, j) accumulates the Gram entry <W_i, W_j>, lane 0 factors the 8x8 Gram matrix, every lane
// applies L^-1 to its columns -- no cross-lane reduction on the critical path.  Whatever the iteration produced,
// the rows are finally divided by sqrt of a Gershgorin bound of |V V^T|_2, so |V x| <= |x| holds rigorously and
// the screen can never drop a similar pair.
//   M      second moments of the family, (nf+1)^2, upper triangle filled, index nf = the constant 1
//   Qout   [KD][nf] rows of the basis;  bias[k] = V_k . mean
#ifndef TSC_BASIS_ITERS
#define TSC_BASIS_ITERS 1
#endif
constexpr int BASIS_ITERS = TSC_BASIS_ITERS;
constexpr int BASIS_LDS_C = 64;  // covariance staged in LDS up to this many features
inline __global__ __launch_bounds__(64) void k_descriptor_basis(const double *__restrict__ M0, const double *__restrict__ M1, int nf0, int nf1,
                                                          int n_samples, double *__restrict__ Q, double *__restrict__ bias,
                                                          unsigned *__restrict__ zero_word, double *__restrict__ spread_host = nullptr,
                                                          int clear_moments = 0) {
    // clear_moments: the moment matrices are accumulators that k_sample_moments adds into: this kernel, their one reader, leaves them zero
    // spread_host (optional): host-visible copy of the two spread values written at the end (pinned memory; the host pre-sets +inf
    // and looks without waiting: whatever finite values it finds are this kernel's)
    // zero_word (optional): the running max |D| of a descriptor build that follows this kernel, cleared here
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;
    __shared__ double V[KD][DESC_MAX_FEAT], Z[KD][DESC_MAX_FEAT], mu[DESC_MAX_FEAT];
    __shared__ double Cs[BASIS_LDS_C][BASIS_LDS_C + 1];
    __shared__ double Gm[KD][KD];  // Gram matrix of the rows being orthonormalised
    const int fam = blockIdx.x, lane = threadIdx.x;
    const double *M = fam == 0 ? M0 : M1;
    const int nf = fam == 0 ? nf0 : nf1, m = nf + 1;
    double *Qout = Q + (fam == 0 ? 0 : size_t(KD) * nf0);
    double *bout = bias + fam * KD;
    if (nf == 0) {
        if (lane < KD) bout[lane] = 0.0;
        if (lane == 0) {
            bias[DW + fam] = 0.0;  // (a family without features separates nothing)
            if (spread_host) spread_host[fam] = 0.0;
        }
        return;
    }
    const double inv = n_samples > 0 ? 1.0 / n_samples : 0.0;
    for (int a = lane; a < nf; a += 64) mu[a] = M[size_t(a) * m + nf] * inv;
    __builtin_amdgcn_wave_barrier();
    const bool c_in_lds = nf <= BASIS_LDS_C;
    if (c_in_lds)
        for (int e = lane; e < nf * nf; e += 64) {
            const int a = e / nf, b = e - a * nf;
            Cs[a][b] = (a <= b ? M[size_t(a) * m + b] : M[size_t(b) * m + a]) * inv - mu[a] * mu[b];
        }
    for (int k = 0; k < KD; ++k)
        for (int a = lane; a < nf; a += 64) V[k][a] = ((a % KD) == k ? 1.0 : 0.0) + 1e-3 * ((a * 7 + k * 13) % 11 - 5);
    __builtin_amdgcn_wave_barrier();

    // W <- rows of an orthonormal basis of span(W) (zero rows where the span is exhausted)
    auto cholesky_qr = [&](double (*W)[DESC_MAX_FEAT]) {
        const int gi = lane >> 3, gj = lane & 7;  // 64 lanes = the 8 x 8 Gram entries
        double g = 0.0;
#pragma unroll 4
        for (int a = 0; a < nf; ++a) g = fma(W[gi][a], W[gj][a], g);
        Gm[gi][gj] = g;
        __builtin_amdgcn_wave_barrier();
        // every lane factors the 8 x 8 Gram matrix in registers (constant indices only): G = L L^T, then Li = L^-1 -- in float32: this
        // kernel is one wavefront's instruction stream (about 40 us, on the critical path of the pipeline's front half), most of it the
        // float64 divisions and square roots of the factorisation; rows orthonormal to 1e-6 serve as well (the second pass of CholeskyQR2
        // repairs what the first leaves, and the Gershgorin scaling below makes |V x| <= |x| hold for whatever comes out)
        float L[KD][KD], Li[KD][KD];
        float tr = 0.0f;
#pragma unroll
        for (int i = 0; i < KD; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) L[i][j] = float(Gm[i][j]);
            tr += L[i][i];
        }
        bool dead[KD];
        float rinv[KD];  // 1 / L[i][i]
#pragma unroll
        for (int i = 0; i < KD; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                float sacc = L[i][j];
#pragma unroll
                for (int t = 0; t < j; ++t) sacc = fmaf(-L[i][t], L[j][t], sacc);
                if (i == j) {
                    dead[i] = !(sacc > 1e-6f * tr) || !(tr > 0.0f);
                    L[i][i] = dead[i] ? 1.0f : sqrtf(sacc);
                    rinv[i] = 1.0f / L[i][i];
                } else {
                    L[i][j] = dead[j] ? 0.0f : sacc * rinv[j];
                }
            }
            if (dead[i]) {
#pragma unroll
                for (int j = 0; j < i; ++j) L[i][j] = 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < KD; ++j) {
#pragma unroll
            for (int i = j; i < KD; ++i) {
                float sacc = (i == j) ? 1.0f : 0.0f;
#pragma unroll
                for (int t = j; t < i; ++t) sacc = fmaf(-L[i][t], Li[t][j], sacc);
                Li[i][j] = dead[i] ? 0.0f : sacc * rinv[i];
            }
        }
        for (int a = lane; a < nf; a += 64) {
            double w[KD];
#pragma unroll
            for (int k = 0; k < KD; ++k) w[k] = W[k][a];
#pragma unroll
            for (int i = 0; i < KD; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j <= i; ++j) acc = fma(double(Li[i][j]), w[j], acc);
                W[i][a] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // (the start rows are not orthonormalised first: C V spans the same nested subspaces whether V or L^-1 V is fed in -- CholeskyQR mixes
    // row i with rows before it only -- so the rows that come out of the step are the same, and two of the four factorisations of this
    // one-wavefront kernel, which sits on the critical path of the pipeline's front half, are saved)
    for (int it = 0; it < BASIS_ITERS; ++it) {
        // Z_k = C V_k with C[a][b] = M[a][b]/ns - mu_a mu_b
        for (int a = lane; a < nf; a += 64) {
            double acc[KD];
#pragma unroll
            for (int k = 0; k < KD; ++k) acc[k] = 0.0;
#pragma unroll 4
            for (int b = 0; b < nf; ++b) {
                const double cab = c_in_lds ? Cs[a][b] : ((a <= b ? M[size_t(a) * m + b] : M[size_t(b) * m + a]) * inv - mu[a] * mu[b]);
#pragma unroll
                for (int k = 0; k < KD; ++k) acc[k] = fma(cab, V[k][b], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < KD; ++k) Z[k][a] = acc[k];
        }
        __builtin_amdgcn_wave_barrier();
        cholesky_qr(Z);
        cholesky_qr(Z);
        // a direction that C annihilated comes back as a zero row: keep the previous row there if it is still
        // orthogonal to the new ones?  Not needed for validity -- a zero row only weakens the screen.
        for (int k = 0; k < KD; ++k)
            for (int a = lane; a < nf; a += 64) V[k][a] = Z[k][a];
        __builtin_amdgcn_wave_barrier();
    }
    // Gershgorin bound of the largest eigenvalue of V V^T (= |V|_2^2): max_i sum_j |<V_i, V_j>|
    {
        const int gi = lane >> 3, gj = lane & 7;
        double g = 0.0;
        for (int a = 0; a < nf; ++a) g = fma(V[gi][a], V[gj][a], g);
        Gm[gi][gj] = fabs(g);
        __builtin_amdgcn_wave_barrier();
    }
    double gmax = 0.0;
    for (int i = 0; i < KD; ++i) {
        double row = 0.0;
        for (int j = 0; j < KD; ++j) row += Gm[i][j];
        gmax = fmax(gmax, row);
    }
    const double sc = gmax > 0.0 ? 1.0 / sqrt(gmax * (1.0 + 1e-12)) : 0.0;
    for (int a = lane; a < nf; a += 64)
        for (int k = 0; k < KD; ++k) {
            const double v = V[k][a] * sc;
            V[k][a] = v;
            Qout[size_t(k) * nf + a] = v;
        }
    __builtin_amdgcn_wave_barrier();
    if (lane < KD) {
        double b = 0.0;
        for (int a = 0; a < nf; ++a) b = fma(V[lane][a], mu[a], b);
        bout[lane] = b;
    }
    // How far apart two structures of the sample lie in this family's descriptor, on average: E |V (f(p) - f(q))|^2 = 2 sum_k V_k^T C V_k.
    // The host compares it with the screen's limit h thr^2: where both families stay below it the screen can separate (almost)
    // nothing -- every pair would reach H anyway -- and an all-pairs kernel without a screen is the faster route (host.hpp,
    // screen_is_useless).  Only worked out where that kernel exists (up to 32 heavy atoms); +inf otherwise.
    double spread = __builtin_inf();
    if (nf <= 32) {  // (c_in_lds) lane = (row k of V, an eighth of the features a): 64 lanes share the nf^2 KD products
        static_assert(KD == 8 && BASIS_LDS_C >= 32, "lane layout of the spread estimate");
        const int k = lane & 7;
        double lam = 0.0;
        for (int a = lane >> 3; a < nf; a += 8) {
            double z = 0.0;
            for (int b = 0; b < nf; ++b) z = fma(Cs[a][b], V[k][b], z);
            lam = fma(V[k][a], z, lam);
        }
        for (int off = 32; off > 0; off >>= 1) lam += __shfl_xor(lam, off);
        spread = 2.0 * lam;
    }
    if (lane == 0) {
        bias[DW + fam] = spread;
        if (spread_host) {
            spread_host[fam] = spread;
            __threadfence_system();
        }
    }
    if (clear_moments) {
        double *Mw = const_cast<double *>(M);
        for (int e = lane; e < m * m; e += 64) Mw[e] = 0.0;
    }
}

// The first link of the pipeline's basis chain on one device: the sample poses are embedded (heavy atoms, into LDS only) and the second
// moments of their features added up -- k_transform + k_feature_moments (fast form) in one launch, without the round trip of the sample's
// coordinates through memory.  grid (groups of SM_CHUNKS chunks of TR_POSES samples, NFAM); M0 / M1 are zero on entry (k_descriptor_basis clears them again).
// dynamic LDS: sample_moments_lds_bytes
constexpr int SM_CHUNKS = 1;  // chunks of TR_POSES samples per workgroup of k_sample_moments (4: 52 us instead of 26 -- a chunk is a chain of
                              // dependent loads, about 13 us beside the clash kernel, and the chains of a workgroup run one after the other)
__host__ __device__ inline size_t sample_moments_lds_bytes(int n_mols, int h) {
    return (transform_lds_bytes(n_mols) + 15) / 16 * 16 + size_t(TR_POSES) * (size_t(h) * 3 + size_t(h) + 1) * sizeof(double);
}
inline __global__ __launch_bounds__(256) void k_sample_moments(const double *__restrict__ frags, FragTable ft, const int32_t *__restrict__ conf_idx,
                                                        const double *__restrict__ rot, const double *__restrict__ pos,
                                                        const int32_t *__restrict__ sample_idx, int n_samples, const int32_t *__restrict__ heavy_slot,
                                                        int h, int nf0, int nf1, double *__restrict__ M0, double *__restrict__ M1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_sm_raw[];
    const int fam = blockIdx.y, nf = fam == 0 ? nf0 : nf1;
    if (nf == 0) return;
    const int n = ft.n_total, nm = ft.n_mols, tid = threadIdx.x, m = nf + 1;
    double *s_tr = reinterpret_cast<double *>(s_sm_raw);
    int64_t *sP = reinterpret_cast<int64_t *>(s_tr);
    double *sR = s_tr + TR_POSES, *sT = sR + TR_POSES * nm * 9;
    int *sC = reinterpret_cast<int *>(sT + TR_POSES * nm * 3);
    double *s_x = reinterpret_cast<double *>(s_sm_raw + (transform_lds_bytes(nm) + 15) / 16 * 16);   // [TR_POSES][h * 3]
    double *s_n = s_x + size_t(TR_POSES) * h * 3;                                                    // [TR_POSES][m]
    double *__restrict__ Mf = fam == 0 ? M0 : M1;
    const int n_chunks = (n_samples + TR_POSES - 1) / TR_POSES;
    // a workgroup takes SM_CHUNKS consecutive chunks and adds their products up in registers before it touches the accumulators
    constexpr int ACC = 8;
    const bool in_regs = m * m <= 256 * ACC;
    double acc[ACC];
#pragma unroll
    for (int u = 0; u < ACC; ++u) acc[u] = 0.0;
    for (int ch = blockIdx.x * SM_CHUNKS; ch < min(n_chunks, (int(blockIdx.x) + 1) * SM_CHUNKS); ++ch) {
        const int s0 = ch * TR_POSES, ns = min(TR_POSES, n_samples - s0);
        if (tid < ns) sP[tid] = int64_t(sample_idx[s0 + tid]);
        __syncthreads();
        for (int q = tid; q < ns * nm * 9; q += 256) {
            const int row = q / (nm * 9), w = q - row * nm * 9;
            sR[q] = rot[sP[row] * nm * 9 + w];
        }
        for (int q = tid; q < ns * nm * 3; q += 256) {
            const int row = q / (nm * 3), w = q - row * nm * 3;
            sT[q] = pos[sP[row] * nm * 3 + w];
        }
        for (int q = tid; q < ns * nm; q += 256) {
            const int row = q / nm, w = q - row * nm;
            sC[q] = conf_idx[sP[row] * nm + w];
        }
        __syncthreads();
        for (int e = tid; e < ns * n; e += 256) {
            const int row = e / n, a = e - row * n;
            const int hs = heavy_slot[a];
            if (hs < 0) continue;
            double v[3];
            embed_atom_staged(frags, ft, sR, sT, sC, row, a, v);
            double *o = s_x + (size_t(row) * h + hs) * 3;
            o[0] = v[0], o[1] = v[1], o[2] = v[2];
        }
        __syncthreads();
        for (int e = tid; e < ns * m; e += 256) {
            const int sm = e / m, a = e - sm * m;
            s_n[sm * m + a] = (a < nf) ? feature(s_x + size_t(sm) * h * 3, h, fam, a) : 1.0;
        }
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < ACC; ++u) {
                const int e = tid + 256 * u;
                if (e < m * m) {
                    const int a = e / m, b = e - a * m;
                    if (b >= a) {
                        double t = 0.0;
                        for (int sm = 0; sm < ns; ++sm) t += s_n[sm * m + a] * s_n[sm * m + b];
                        acc[u] += t;
                    }
                }
            }
        } else {
            for (int e = tid; e < m * m; e += 256) {
                const int a = e / m, b = e - a * m;
                if (b < a) continue;
                double t = 0.0;
                for (int sm = 0; sm < ns; ++sm) t += s_n[sm * m + a] * s_n[sm * m + b];
                atomicAdd(&Mf[e], t);
            }
        }
        __syncthreads();
    }
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < ACC; ++u) {
            const int e = tid + 256 * u;
            if (e < m * m && e % m >= e / m) atomicAdd(&Mf[e], acc[u]);
        }
    }
}

struct SieveArgs {
    int n;   // upper bound of the active count the grid was sized for (structures of the run)
    int h;
    int tile_begin;
    int tile_stride;
    int seg_cols;
    double thr, maxdev_thr;
    double half_h_thr2;   // h * thr^2 / 2
    double two_thr2;      // 2 thr^2 when the near-duplicate test applies (h >= 4), else -1
    int drain_min;        // queue length that triggers a drain between column tiles (1..64)
    const int32_t *tile_cmax;   // device: per row tile, the largest stop column of its 16 rows (k_open_rows)
    const unsigned *dmax_bits;  // device: largest |descriptor component| of the run (bit pattern of a float), see screen_limit32
    double desc_limit;          // h thr^2: exact squared descriptor distance above which a pair is certainly dissimilar
    const float *heavy32;       // float32 copy of the heavy atoms (heavy32_pitch(h) floats per structure), or null: stage 1 in float64 only
    unsigned long long *dbg;    // -DTSC_DBG_STAMPS builds only: 8 time stamps per wavefront of the launch (tools/stamps.py), else null
};

#ifdef TSC_DBG_STAMPS   // measurement hook: where a wavefront of the pair kernel spends its time (100 MHz wall clock)
#define TSC_STAMP(i)                                                                                                                  \
    do {                                                                                                                              \
        if (a.dbg) {                                                                                                                  \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                               \
            if ((threadIdx.x & 63) == 0)                                                                                              \
                a.dbg[(size_t(blockIdx.y) * gridDim.x + blockIdx.x) * 32 + (threadIdx.x >> 6) * 8 + (i)] = wall_clock64();             \
        }                                                                                                                             \
    } while (0)
#else
#define TSC_STAMP(i) do { } while (0)
#endif

// H = p^T q of one pair read from memory.  `lpp` consecutive lanes (a power of two) share the pair: lane `sub` takes
// atoms sub, sub + lpp, ... and the group sums H with a butterfly, so a batch of few pairs has a short critical path
// (lpp = 64 / batch size) while a full batch runs one pair per lane (lpp = 1).
__device__ inline void pair_H(const double *__restrict__ p, const double *__restrict__ q, int h, int sub, int lpp, double H[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = 0.0;
#pragma unroll 2
    for (int a = sub; a < h; a += lpp) {
        const double px = p[3 * a], py = p[3 * a + 1], pz = p[3 * a + 2];
        const double qx = q[3 * a], qy = q[3 * a + 1], qz = q[3 * a + 2];
        H[0] = fma(px, qx, H[0]), H[1] = fma(px, qy, H[1]), H[2] = fma(px, qz, H[2]);
        H[3] = fma(py, qx, H[3]), H[4] = fma(py, qy, H[4]), H[5] = fma(py, qz, H[5]);
        H[6] = fma(pz, qx, H[6]), H[7] = fma(pz, qy, H[7]), H[8] = fma(pz, qz, H[8]);
    }
    for (int off = lpp >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) H[k] += __shfl_xor(H[k], off);
    }
}

// The float32 copy of the heavy atoms that stage 1 reads first: structure i at heavy32 + i * pitch32, pitch32 = 12 * ceil(h / 4) floats
// (x y z of four atoms per three float4; zero-padded).  The gathers of the pairs that pass the screen are what the pair kernels wait
// for -- with the float64 loads issued twice a C4 step takes 18.8 instead of 12.2 ms, a C3 step 1.00 instead of 0.86 -- and the quartic
// tests on H from this copy, with the rounding bound that goes with it (rmsd.hpp: quartic_gamma32), decide all but the pairs with a
// tiny margin; those are formed again from the float64 coordinates, as is everything the explicit-rotation path needs.

inline __global__ __launch_bounds__(256) void k_heavy32(const double *__restrict__ heavy, int64_t n, int h, float *__restrict__ out) {
    const int h3 = h * 3, pitch = heavy32_pitch(h);
    const int64_t total = n * pitch;
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < total; e += int64_t(gridDim.x) * 256) {
        const int64_t i = e / pitch;
        const int w = int(e - i * pitch);
        out[e] = w < h3 ? float(heavy[i * h3 + w]) : 0.0f;
    }
}

// H = p^T q from the float32 copy: lane `sub` of the pair's `lpp` lanes takes the groups of four atoms sub, sub + lpp, ...; the
// products of two floats are exact in float64, the sums run there
__device__ inline void pair_H32(const float *__restrict__ p, const float *__restrict__ q, int hq, int sub, int lpp, double H[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = 0.0;
    for (int g = sub; g < hq; g += lpp) {
        const f32x4 *pv = reinterpret_cast<const f32x4 *>(p + 12 * g), *qv = reinterpret_cast<const f32x4 *>(q + 12 * g);
        const f32x4 p0 = pv[0], p1 = pv[1], p2 = pv[2], q0 = qv[0], q1 = qv[1], q2 = qv[2];
        const float pf[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
        const float qf[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double px = pf[3 * a], py = pf[3 * a + 1], pz = pf[3 * a + 2];
            const double qx = qf[3 * a], qy = qf[3 * a + 1], qz = qf[3 * a + 2];
            H[0] = fma(px, qx, H[0]), H[1] = fma(px, qy, H[1]), H[2] = fma(px, qz, H[2]);
            H[3] = fma(py, qx, H[3]), H[4] = fma(py, qy, H[4]), H[5] = fma(py, qz, H[5]);
            H[6] = fma(pz, qx, H[6]), H[7] = fma(pz, qy, H[7]), H[8] = fma(pz, qz, H[8]);
        }
    }
    for (int off = lpp >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) H[k] += __shfl_xor(H[k], off);
    }
}

// Stage 1 of a pair that passed the screen: PAIR_DISSIMILAR / PAIR_SIMILAR / PAIR_UNDECIDED (rmsd.hpp).  With the float32 copy the tests
// run on H from that copy alone, with its own rounding bound; what they leave undecided goes to the explicit-rotation stage like any
// other undecided pair (it forms H from the float64 coordinates).  F32 = false: H from the float64 coordinates, as ever (a template
// parameter, not a branch: with both loops in one kernel the float64 one lost a tenth of its speed to the register allocation).
template <bool F32>
__device__ inline int pair_stage1(const double *__restrict__ heavy, const float *__restrict__ heavy32, int64_t i, int64_t j, int h, double half_sum,
                                  double half_h_thr2, double two_thr2, int sub, int lpp) {
    double H[9];
    if (F32) {
        const int pitch = heavy32_pitch(h);
        pair_H32(heavy32 + i * pitch, heavy32 + j * pitch, pitch / 12, sub, lpp, H);
        return pair_verdict_g(H, half_sum, half_h_thr2, two_thr2, quartic_gamma32(h));
    }
    pair_H(heavy + i * h * 3, heavy + j * h * 3, h, sub, lpp, H);
    return pair_verdict(H, half_sum, half_h_thr2, two_thr2, h);
}

// One pair end to end: true iff it is similar in the reference's sense (rmsd < thr and maxdev < 2 thr,
// rmsd_pruning.py:75); exact_taken tells whether the explicit-rotation path ran.
__device__ inline bool pair_is_similar(const double *__restrict__ p, const double *__restrict__ q, int h, double Gp, double Gq,
                                       double half_h_thr2, double thr, double maxdev_thr, bool &exact_taken, int sub, int lpp) {
    double H[9];
    pair_H(p, q, h, sub, lpp, H);
    exact_taken = false;
    if (certainly_dissimilar(H, 0.5 * (Gp + Gq) - half_h_thr2, 0.5 * (Gp + Gq), h)) return false;
    exact_taken = true;
    double rm, md;
    exact_rmsd_maxdev(p, q, h, H, Gp, Gq, rm, md, sub, lpp);
    return rm < thr && md < maxdev_thr;
}

#ifndef TSC_SIEVE_OCC2
#define TSC_SIEVE_OCC2 5
#endif
#ifndef TSC_SIEVE_OCC1
#define TSC_SIEVE_OCC1 6
#endif
// What the pair kernel of a single-rank run does when the LAST work item of a row tile finishes: it applies the tile's
// verdicts (apply_wave_rows, rmsd.hpp) and, if the tile was the last of the pass, closes the pass and opens the next one
// (pass_step_wave) -- no k_apply_pass launch behind the pair kernel.  Work items of a tile = its column segments that start
// below the tile's largest stop column (the others leave at their first test and do not count).
struct FusedApply {
    ApplyArgs ap;
    int32_t *tile_done;  // arrivals per row tile (zero between passes)
    PassTickets *tickets;
    unsigned n_tiles;    // blocks of k_open_rows = row tiles that arrive, here or there
    StepCtx sc;
    StepArgs next;
};

template <int TI, int CPL, bool TRIM, bool F32 = false>
__device__ __forceinline__ void sieve_item(const double *__restrict__ heavy, const int32_t *__restrict__ act,
                                           const double *__restrict__ Gall, const float *__restrict__ D,
                                           const int32_t *__restrict__ cend,
                                           int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                           const PruneState *__restrict__ st, const SieveArgs a, int &A_out, int &bitsel_out) {
    static_assert(TI <= 16, "queue entries keep the row in 4 bits");
    static_assert(DW == 2 * KD, "two families of KD components");
    constexpr int QCAP = TI * 64 * CPL + 64;  // one column tile can add TI * 64 * CPL pairs on top of a remainder below 64
    // CPL columns per lane: a tile is 64 * CPL columns, so one LDS read of a row
    constexpr int TILE_COLS = 64 * CPL;  // descriptor serves 4 x 64 pairs and the loop overhead is paid once per 256
    // an entry packs the row (4 bits) and the column offset inside the segment (12 bits: segments are <= 4096 columns)
    __shared__ unsigned short s_queue[4][QCAP];
    __shared__ unsigned short s_exq[4][128];  // pairs that the sign test could not reject, waiting for the exact path
    __shared__ double s_jacobi[4][32];        // scratch of the Jacobi fallback of the exact path (rmsd.hpp), per wavefront
    constexpr int RS = TRIM ? 20 : DW;  // floats per row record in LDS; TRIM: 16 components, the two squared norms, 2 of padding (80 B)
    __shared__ __attribute__((aligned(16))) float s_rowdesc[4][TI * RS];
    __shared__ f32x2 s_rownorm[4][TRIM ? 1 : TI];  // |row descriptor|^2 per family (TRIM keeps them in the row record)
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * 4 + wid;
    const int tile = a.tile_begin + slot * a.tile_stride;
    const int r0 = tile * TI;
    const int seg_lo = ((r0 + 1) & ~63) + int(blockIdx.y) * a.seg_cols;
    const int seg_hi = seg_lo + a.seg_cols;

    // ---- prologue in one memory round trip: the state, this item's 16 stop columns / best columns (they decide whether
    // it has work at all: most items of a late pass have none) and the descriptors of its rows and first column tile.
    // act[x] (needed for the coordinates of the pairs that reach H) is a valid structure index for every x < n:
    // k_init_run fills it with the identity, every pass rewrites a prefix.
    const int pass_on = st->pass_on, n_active = st->A;
    A_out = n_active, bitsel_out = st->bitsel;  // (for the fused tail: the state block cannot change before this item has arrived there)
    int my_cend = 0, my_best = 0;
    if (lane < TI && r0 + lane < a.n) {
        my_cend = cend[r0 + lane];
        my_best = __hip_atomic_load(&best[r0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static_assert(DW == 16 && TI * DW == 256, "row staging: 4 rows x 16 components per 64 lanes");
    // D holds the descriptors in ACTIVE order (position r = the r-th active structure; k_open_rows copies them there every
    // pass): rows and columns are read by position, nothing is gathered through act[] in front of the screen.  Positions
    // up to n - 1 are readable; those beyond the active count hold stale values that no row's range admits.
    int row_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) row_src[j] = min(r0 + 4 * j + (lane >> 4), a.n - 1);
    int col_src[CPL];
    auto load_cols = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < CPL; ++u) col_src[u] = min(c0 + 64 * u + lane, a.n - 1);
    };
    load_cols(seg_lo);
    if (pass_on == 0 || r0 >= n_active) return;
    const int nrows = min(TI, n_active - r0);
    const bool live0 = lane < nrows && my_cend > max(r0 + lane + 1, seg_lo) && my_best >= seg_lo;
    unsigned alive = unsigned(__ballot(live0));
    if (!alive) return;

    const float limit32 = screen_limit32_dot(__uint_as_float(*a.dmax_bits), a.desc_limit);
    const int limit_bits = __float_as_int(limit32);  // (positive: the integer order of the bit patterns is the float order from zero up)
    float rd_stage[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd_stage[j] = D[int64_t(row_src[j]) * DW + (lane & 15)];
    f32x2 dq[CPL][KD];  // .x = family 0, .y = family 1
    f32x2 cn[CPL];      // their squared norms
    auto load_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(D + int64_t(col_src[u]) * DW);
#pragma unroll
            for (int k = 0; k < KD / 2; ++k) {
                const f32x4 v = src[k];
                dq[u][2 * k] = f32x2{v.x, v.y};
                dq[u][2 * k + 1] = f32x2{v.z, v.w};
            }
            f32x2 nc = {0.0f, 0.0f};  // |column descriptor|^2 per family, once per tile (shared by its 16 rows)
#pragma unroll
            for (int k = 0; k < KD; ++k) nc = __builtin_elementwise_fma(dq[u][k], dq[u][k], nc);
            // TRIM: the dot-product chain of a row starts from -|c|^2 / 2 (exact: a power of two), so that one fma by -2 onto
            // |r|^2 finishes |r|^2 + |c|^2 - 2 r.c without the separate packed add per column
            cn[u] = TRIM ? nc * f32x2{-0.5f, -0.5f} : nc;
        }
    };
    load_tile();

    int cmax = live0 ? my_cend : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
    cmax = min(__builtin_amdgcn_readfirstlane(cmax), seg_hi);

    // the row descriptors of this work item -> LDS (rows beyond nrows are never read back)
    float *rowdesc = s_rowdesc[wid];
#pragma unroll
    for (int j = 0; j < 4; ++j) rowdesc[(4 * j + (lane >> 4)) * RS + (lane & 15)] = rd_stage[j];
    __builtin_amdgcn_wave_barrier();
    if (lane < TI) {  // squared norms of the row descriptors, per family
        const f32x2 *dr = reinterpret_cast<const f32x2 *>(rowdesc + lane * RS);
        f32x2 nr = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < KD; ++k) nr = __builtin_elementwise_fma(dr[k], dr[k], nr);
        if constexpr (TRIM) *reinterpret_cast<f32x2 *>(rowdesc + lane * RS + DW) = nr;
        else s_rownorm[wid][lane] = nr;
    }
    __builtin_amdgcn_wave_barrier();

    TSC_STAMP(1);  // prologue data has arrived
    const int h3 = a.h * 3;
    unsigned short *queue = s_queue[wid];
    int qn = 0;
    unsigned long long n_eval = 0, n_exact = 0;
    int my_screened = 0;  // lanes 0..15: pairs of this item's rows that went through the screen (<= 16 x segment)
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // Queue entries are evaluated in two stages, each over batches of up to 64 pairs with lpp = 64 / pow2ceil(batch) lanes
    // per pair.  Stage 1 forms H and applies the sign test; the few pairs it cannot reject (a few per cent) go to a second
    // queue.  Stage 2 -- the long explicit-rotation path -- runs when 64 of those have gathered (or at the end), so its
    // cost is paid once per 64 candidates instead of once per stage-1 batch with a handful of lanes busy.
    unsigned short *exq = s_exq[wid];
    int qe = 0;
    int64_t si = 0, sj = 0;  // the structures of the pair decoded last
    auto decode = [&](unsigned e, int &t, int &col, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
        t = int(e >> 12);
        col = seg_lo + int(e & 0xfffu);
        const int64_t i = act[r0 + t], j = act[col];
        si = i, sj = j;
        pp = heavy + i * h3, pq = heavy + j * h3;
        Gi = Gall[i], Gj = Gall[j];
    };
    auto sign_stage = [&](int base, int cnt) __attribute__((always_inline)) {
        int lpp = 64;
        while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
        const int g = lane / lpp, sub = lane - g * lpp;
        bool cand = false, sim = false;
        unsigned e = 0;
        int t = 0;
        if (g < cnt) {  // a lane group is either wholly busy or wholly idle: the group shuffles only involve converged lanes
            e = queue[base + g];
            int col;
            const double *pp, *pq;
            double Gi, Gj;
            decode(e, t, col, pp, pq, Gi, Gj);
            const int verdict = pair_stage1<F32>(heavy, a.heavy32, si, sj, a.h, 0.5 * (Gi + Gj), a.half_h_thr2, a.two_thr2, sub, lpp);
            cand = sub == 0 && verdict == PAIR_UNDECIDED;
            sim = sub == 0 && verdict == PAIR_SIMILAR;
            if (sim) atomicMin(&best[r0 + t], col);
        }
        unsigned long long sm = __builtin_amdgcn_ballot_w64(sim);
        while (sm) {  // rows that found a similar column stop being screened (the reference returns there, :75-77)
            const int l = __ffsll((long long)sm) - 1;
            sm &= sm - 1;
            alive &= ~(1u << __builtin_amdgcn_readlane(t, l));
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
        if (m) {
            if (cand) exq[qe + __popcll(m & lt_mask)] = (unsigned short)e;
            qe += __popcll(m);
        }
        n_eval += cnt;
        n_exact += __popcll(m);
        __builtin_amdgcn_wave_barrier();
    };
    auto exact_stage = [&](int base, int cnt) __attribute__((always_inline)) {
        int lpp = 64;
        while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
        const int g = lane / lpp, sub = lane - g * lpp;
        bool sim = false, degenerate = false;
        int t = 0;
        unsigned ent = 0;
        if (g < cnt) {
            int col;
            const double *pp, *pq;
            double Gi, Gj, H[9], e[4];
            ent = exq[base + g];
            decode(ent, t, col, pp, pq, Gi, Gj);
            pair_H(pp, pq, a.h, sub, lpp, H);
            if (rotation_quaternion_fast(H, Gi, Gj, e)) {  // (the lanes of a group hold the same H: they branch together)
                double rm, md;
                residual_rmsd_maxdev(pp, pq, a.h, e, rm, md, sub, lpp);
                sim = sub == 0 && rm < a.thr && md < a.maxdev_thr;  // rmsd_pruning.py:75
                if (sim) atomicMin(&best[r0 + t], col);
            } else {
                degenerate = sub == 0;
            }
        }
        unsigned long long sm = __builtin_amdgcn_ballot_w64(sim);
        while (sm) {  // rows that found a similar column stop being screened (the reference returns there, :75-77)
            const int l = __ffsll((long long)sm) - 1;
            sm &= sm - 1;
            alive &= ~(1u << __builtin_amdgcn_readlane(t, l));
        }
        // Pairs whose top eigenvalue is degenerate (collinear, planar through the origin, mirror-symmetric structures) need
        // the Jacobi eigen-solver.  Its two 4x4 matrices would cost every work item 64 registers, so the whole wavefront
        // takes such a pair on, one at a time: H by all 64 lanes, the solver by lane 0 on a 256-byte LDS area, the residual
        // by all lanes again.
        for (unsigned long long dm = __builtin_amdgcn_ballot_w64(degenerate); dm; dm &= dm - 1) {
            const unsigned e1 = unsigned(__builtin_amdgcn_readlane(int(ent), __ffsll((long long)dm) - 1));
            int t2, col2;
            const double *pp, *pq;
            double Gi, Gj, H[9], e[4], rm, md;
            decode(e1, t2, col2, pp, pq, Gi, Gj);
            pair_H(pp, pq, a.h, lane, 64, H);
            double *jac = s_jacobi[wid];
            if (lane == 0) {
                horn_matrix(H, jac);
                top_eigvec4_mem(jac, jac + 16, e);
                jac[0] = e[0], jac[1] = e[1], jac[2] = e[2], jac[3] = e[3];
            }
            __builtin_amdgcn_wave_barrier();
            e[0] = jac[0], e[1] = jac[1], e[2] = jac[2], e[3] = jac[3];
            __builtin_amdgcn_wave_barrier();
            residual_rmsd_maxdev(pp, pq, a.h, e, rm, md, lane, 64);
            if (rm < a.thr && md < a.maxdev_thr) {  // wave-uniform
                if (lane == 0) atomicMin(&best[r0 + t2], col2);
                alive &= ~(1u << t2);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto drain = [&](int base, int cnt) __attribute__((always_inline)) {
        sign_stage(base, cnt);
#ifdef TSC_DBG_NOEXACT
        qe = 0;
#endif
        if (qe >= 64) {
            exact_stage(qe - 64, 64);
            qe -= 64;
        }
    };

    for (int c0 = seg_lo; c0 < cmax;) {  // (alive != 0 here)
        {   // ---- screen one tile against every live row
            const bool here = lane < nrows && ((alive >> lane) & 1u) && my_cend > c0 && r0 + lane < c0 + TILE_COLS - 1;
            unsigned rows = unsigned(__ballot(here));
            // columns of this tile inside every live row's range (r, cend): counted once per tile by the rows' own lanes (a few
            // vector instructions per tile; as seven scalar ones per (row, tile) it was a quarter of the row loop's instructions)
            my_screened += here ? max(0, min(my_cend, c0 + TILE_COLS) - max(r0 + lane + 1, c0)) : 0;
            if constexpr (TRIM) {
                // rows whose range (r, cend) holds the WHOLE tile need no column mask: one bit per row
                const unsigned full = unsigned(__ballot(here && r0 + lane < c0 && my_cend >= c0 + TILE_COLS));
                // the same screen with fewer vector instructions per (row, tile): one LDS record per row (components and norms
                // behind one address), the norms folded into the dot-product chain, and the two families compared with the limit
                // separately instead of max / max / min / compare (NaN handling unchanged: a NaN component never rejects).
                // Two loops over the same body: first the rows that hold the whole tile (no column mask, no test for it),
                // then the few whose range ends or begins inside it.
                auto screen_row = [&](const int t, auto partial) __attribute__((always_inline)) {
#ifdef TSC_DBG_NOROWLOAD      // (measurement hook: every row of a tile uses row 0's record -- wrong verdicts, the screen without its per-row LDS reads)
                        const f32x2 *rec = reinterpret_cast<const f32x2 *>(rowdesc);
#else
                        const f32x2 *rec = reinterpret_cast<const f32x2 *>(rowdesc + t * RS);
#endif
                        f32x2 rd[KD];
#pragma unroll
                        for (int k = 0; k < KD; ++k) rd[k] = rec[k];
                        const f32x2 nr = rec[KD];
                        // Both family distances against the limit as INTEGER compares of the bit patterns (for a positive limit the
                        // order of non-negative floats; a negative sum -- rounding -- is below it either way; a NaN with a clear
                        // sign bit counts as beyond the limit where a float compare would let it through to H, which rejects it:
                        // :75).  The larger of a column's two families, the smaller of that over the lane's columns, ONE compare
                        // and one branch on it per (row, tile): the scalar side of this loop (27 instructions per trip against 24
                        // vector ones before: each family's compare into a lane mask, the masks combined and tested there) was
                        // what a wavefront spent its trip on.
                        int worst[CPL];
#pragma unroll
                        for (int u = 0; u < CPL; ++u) {
                            f32x2 acc = cn[u];                         // -|c|^2 / 2
#pragma unroll
                            for (int k = 0; k < KD; ++k) acc = __builtin_elementwise_fma(rd[k], dq[u][k], acc);
                            const f32x2 s2 = __builtin_elementwise_fma(acc, f32x2{-2.0f, -2.0f}, nr);
                            worst[u] = max(__float_as_int(s2.x), __float_as_int(s2.y));
                        }
                        if constexpr (decltype(partial)::value) {  // the tile crosses an end of the row's range: only the columns inside count
                            const int r = r0 + t, ce = __builtin_amdgcn_readlane(my_cend, t);
#pragma unroll
                            for (int u = 0; u < CPL; ++u) {
                                const int col = c0 + 64 * u + lane;
                                worst[u] = (col > r && col < ce) ? worst[u] : INT_MAX;
                            }
                        }
                        int nearest = worst[0];
#pragma unroll
                        for (int u = 1; u < CPL; ++u) nearest = min(nearest, worst[u]);
                        if (__builtin_amdgcn_ballot_w64(nearest <= limit_bits)) {   // (rare) some column of the tile is within the limit
#pragma unroll
                            for (int u = 0; u < CPL; ++u) {
                                const unsigned long long m = __builtin_amdgcn_ballot_w64(worst[u] <= limit_bits);
                                if (m) {
                                    if ((m >> lane) & 1ull) queue[qn + __popcll(m & lt_mask)] = (unsigned short)((unsigned(t) << 12) | unsigned(c0 + 64 * u + lane - seg_lo));
                                    qn += __popcll(m);
                                }
                            }
                        }
                };
                auto screen_rows = [&](unsigned todo, auto partial) __attribute__((always_inline)) {
                    while (todo) {
                        const int t = __ffs(todo) - 1;
                        todo &= todo - 1;
                        screen_row(t, partial);
                    }
                };
                if ((rows & full) == 0xffffu) {
                    // all 16 rows hold the whole tile (the usual tile of a large pass): the loop unrolled, every row's record at
                    // a constant LDS offset, no index arithmetic and no loop control between the rows
#pragma unroll
                    for (int t = 0; t < TI; ++t) screen_row(t, std::false_type{});
                } else {
                    screen_rows(rows & full, std::false_type{});
                    screen_rows(rows & ~full, std::true_type{});
                }
            } else
            while (rows) {
                const int t = __ffs(rows) - 1;
                rows &= rows - 1;
                const int r = r0 + t;
                const int ce = __builtin_amdgcn_readlane(my_cend, t);
                const f32x2 *dr = reinterpret_cast<const f32x2 *>(rowdesc + t * DW);
                f32x2 rd[KD];
#pragma unroll
                for (int k = 0; k < KD; ++k) rd[k] = dr[k];
                // larger of the two family distances for the lane's CPL columns, as |r|^2 + |c|^2 - 2 r.c in packed fp32 (one
                // v_pk_fma_f32 per component advances both families; 10 instructions per column, screen_limit32_dot has
                // the error bound)
                const f32x2 nr = s_rownorm[wid][t];
                float mx[CPL];
#pragma unroll
                for (int u = 0; u < CPL; ++u) {
                    f32x2 dot = {0.0f, 0.0f};
#pragma unroll
                    for (int k = 0; k < KD; ++k) dot = __builtin_elementwise_fma(rd[k], dq[u][k], dot);
                    const f32x2 s2 = __builtin_elementwise_fma(dot, f32x2{-2.0f, -2.0f}, nr + cn[u]);
                    mx[u] = fmaxf(s2.x, s2.y);
                }
                if (!(r < c0 && ce >= c0 + TILE_COLS)) {  // the tile crosses an end of the row's range: mask the columns outside
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        const int col = c0 + 64 * u + lane;
                        mx[u] = (col > r && col < ce) ? mx[u] : __builtin_inff();
                    }
                }
                // most rows of a tile have no column within the limit: one test for all CPL * 64 pairs (a NaN distance --
                // NaN coordinates -- is ignored by the minimum; such a pair is not similar for the reference either, :75)
                float mn = mx[0];
#pragma unroll
                for (int u = 1; u < CPL; ++u) mn = fminf(mn, mx[u]);
                if (__builtin_amdgcn_ballot_w64(!(mn > limit32))) {
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        const bool pass = !(mx[u] > limit32);
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                        if (m) {
                            if (pass) queue[qn + __popcll(m & lt_mask)] = (unsigned short)((unsigned(t) << 12) | unsigned(c0 + 64 * u + lane - seg_lo));
                            qn += __popcll(m);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- drain full batches between tiles; a remainder below 64 waits for the next tile
#ifdef TSC_DBG_NODRAIN
        qn = 0;
#endif
        while (qn >= a.drain_min) {
            const int cnt = min(qn, 64);
            drain(qn - cnt, cnt);
            qn -= cnt;
        }
        // the next tile's columns (loaded here, after the drain, so that they do not occupy registers during it)
        c0 += TILE_COLS;
        if (!(c0 < cmax && alive)) break;
#ifndef TSC_DBG_NOTILELOAD   // (measurement hook: keep the first tile's descriptors -- wrong verdicts, the screen without its global loads)
        load_cols(c0);
        load_tile();
#endif
    }
    TSC_STAMP(2);  // screen done
#ifdef TSC_DBG_NODRAIN
    qn = 0;
#endif
    if (qn > 0) drain(0, qn);
#ifdef TSC_DBG_NOEXACT
    qe = 0;
#endif
    if (qe > 0) exact_stage(0, qe);
    TSC_STAMP(3);  // candidates evaluated
    for (int off = 8; off > 0; off >>= 1) my_screened += __shfl_xor(my_screened, off);
    if (lane == 0) {
        count_add(counters, unsigned(slot), CNT_FORMED, n_eval);
        count_add(counters, unsigned(slot), CNT_EXACT, n_exact);
        count_add(counters, unsigned(slot), CNT_SCREENED, (unsigned long long)my_screened);
    }
}

template <int TI, int CPL, bool TRIM = false, bool FUSED = false, bool F32 = false>
inline __global__ __launch_bounds__(256, CPL == 4 ? 4 : (CPL == 2 ? TSC_SIEVE_OCC2 : TSC_SIEVE_OCC1)) void k_rmsd_sieve(const double *__restrict__ heavy, const int32_t *__restrict__ act,
                                                        const double *__restrict__ Gall, const float *__restrict__ D,
                                                        const int32_t *__restrict__ cend,
                                                        int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                                        const PruneState *__restrict__ st, SieveArgs a, FusedApply fa) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = a.tile_begin + slot * a.tile_stride;
    const int r0 = tile * TI;
    const int seg_base = (r0 + 1) & ~63;
    const int seg_lo = seg_base + int(blockIdx.y) * a.seg_cols;
    TSC_STAMP(0);  // the wavefront has started
    if (r0 >= a.n || seg_lo >= a.n) return;  // beyond the upper bound the grid was sized for: nothing to read
    // most items of a pass with long chunks start beyond every stop column of their row tile (0 for a tile without rows or
    // a pass that is gated off: k_open_rows): one scalar load and out
    const int tcm = a.tile_cmax[tile];
    if (tcm <= seg_lo) return;
    int A = 0, bitsel = 0;
    sieve_item<TI, CPL, TRIM, F32>(heavy, act, Gall, D, cend, best, counters, st, a, A, bitsel);
    if constexpr (FUSED) {
        // this item's atomicMin's on best[] are at the L2 before its arrival is
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const int lim = min(a.n, tcm);
        const int n_live = min(int(gridDim.y), (lim - seg_base + a.seg_cols - 1) / a.seg_cols);
        // (the tail of a light pass is a chain of dependent round trips -- arrival, state, best[], act[], the atomics' way out, the pass's
        // counter -- and its length is the pass: the only item of its tile needs no arrival, and the state block was read in the prologue)
        if (n_live > 1) {
            int last = 0;
            if (lane == 0) last = (atomicAdd(&fa.tile_done[tile], 1) == n_live - 1) ? 1 : 0;
            TSC_STAMP(4);  // arrived at the tile's counter
            if (!__builtin_amdgcn_readfirstlane(last)) return;
            if (lane == 0) fa.tile_done[tile] = 0;
        }
        unsigned long long ev_total = 0, rm_total = 0;
        apply_wave_rows(fa.ap, bitsel, r0 + lane, lane < TI && r0 + lane < A, ev_total, rm_total);
        int fin = 0;
        if (lane == 0) {
            count_add(counters, unsigned(slot), CNT_EVALUATED, ev_total);
            count_add(counters, unsigned(slot), CNT_REMOVED, rm_total);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSC_STAMP(5);  // tile applied
        if (lane == 0) fin = tickets_arrive(fa.tickets, unsigned(tile), fa.n_tiles, PT_GROUPS) ? 1 : 0;
        TSC_STAMP(6);  // arrived at the pass's counter
        if (__builtin_amdgcn_readfirstlane(fin)) {
            pass_step_wave(fa.sc, fa.next);
            TSC_STAMP(7);  // pass closed
        }
    }
}

}  // namespace tsc
