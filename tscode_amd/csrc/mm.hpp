// mm.hpp -- the descriptor screen of the pair kernel on the matrix cores (v_mfma_f32_16x16x16_f16), in two levels.
//
// The screen of sieve.hpp asks, per pair and feature family, whether S = |d(p) - d(q)|^2 = |d(p)|^2 + |d(q)|^2 - 2 d(p).d(q) can be below
// h thr^2 (rmsd_pruning.py:75 then needs H; everything else is dropped).  Over a tile of rows x columns that is a matrix product with
// K = 8 components -- and 18 packed-fp32 instructions per (row, 128 columns) on the vector ALU, which is what bounds the large passes
// (bench.py: roofline.bound = "valu_issue").  Here:
//
//   level 1, every pair, one MFMA per 16 rows x 16 columns and family.  Every stored fp32 component d is scaled by a power of two sigma
//   (largest |component| of the run into [32, 64)) and rounded to float16, x = f16(sigma d); |x|^2 of the ROUNDED vector goes along in
//   three float16 pieces n0 + n1 + n2.  The 16 slots of the K dimension:
//        slot   0..7     8  9  10   11 12 13   14 15
//        A      x        n0 n1 n2   1  1  1    0  0        (rows)
//        B     -2 y      1  1  1    m0 m1 m2   0  0        (columns)
//   so that  A.B = |x|^2 + |y|^2 - 2 x.y = |x - y|^2 of the rounded vectors, up to the norms' residuals and the accumulation.
//   A pair whose A.B exceeds screen_limit_mm in EITHER family certainly has h rmsd^2 > h thr^2 (bound below): dropped.
//   level 2 (a build option, TSC_MM_LEVEL2; off: see there), the few pairs level 1 lets through (a few per thousand), where they are decoded
//   for evaluation: the fp32 screen of sieve.hpp itself on the two stored descriptors (screen_limit32_dot), 40 instructions per pair
//   beside the 600 of H.  It drops 2 - 7 % of what level 1 lets through: the float16 form is nearly as tight as the fp32 screen.
//
// Error bound of level 1 (screen_limit_mm_bits; scaled units, M' = sigma * max |component| < 64):
//   * products of two float16 are exact in the fp32 accumulator; the accumulation of the 30-term form was measured on MI355X at
//     2^-22.8 * sum |terms| worst over 10^6 sums with the cancellation of this very use (tools/probe/mfma_probe.hip); the bound takes
//     2^-20 * sum |terms|, sum |terms| <= 2 (|x|^2 + |y|^2) <= 4 KD M'^2 (tests/test_gpu_parity.py::test_matrix_core_screen_values_and_limit
//     holds every value of this form against it);
//   * the norms' residuals 2^-14 each;
//   * x against the exact scaled component: the fp32 rounding of sieve.hpp (3 * 2^-24 * 2 M' per difference) plus the float16 rounding,
//     2^-11 M' (+ 2^-25 below the normal range) per operand: sqrt(S_exact) >= sqrt(S) - sqrt(KD) * eta.
// The rounding to float16 moves the limit by about 2 sqrt(KD) 2^-10 M' sqrt(h) thr sigma -- a per cent or so more pairs than the fp32
// screen lets through reach level 2, which drops exactly those.  As with the choice of basis: what passes where only moves work, never
// a verdict.
//
// Work item = one wavefront = 64 rows (the four row tiles a wavefront of k_open_rows opens) x one column segment; the rows' A operands
// stay in registers (16 VGPRs), a step takes 16 columns: two 8-byte loads per lane, 8 MFMAs, 16 integer maxima (the larger family),
// 16 compares.  Rows that have found their column or passed their stop column get +inf in the n0 slot: they cost nothing and
// never pass.  Nothing else is masked at screen time: what lies left of the diagonal or beyond a stop column is dropped at decode time.
#pragma once
#include "mm_record.hpp"
#include "sieve.hpp"

namespace tsc {

constexpr int MM_ROWS = 64;                 // rows per work item
constexpr int MM_STEP = 16;                 // columns per step
constexpr int MM_QCAP = 2 * MM_ROWS * MM_STEP + 64;   // a step of 32 columns can add 64 x 32 pairs on top of a remainder below 64
static_assert(MM_KD == KD && NFAM == 2 && DW == 2 * MM_KD, "the record layout of mm_record.hpp");

// The limit of level 1 (3.4e38: nothing finite is dropped -- non-finite input or a limit beyond what the scaled distances can
// reach).  limit = h thr^2.  The kernel starts every accumulator at MINUS this value and keeps the pairs that come out negative.
__device__ inline float screen_limit_mm(unsigned dmax_bits, double limit) {
    const float dmaxf = __uint_as_float(dmax_bits);
    if (!(dmaxf >= 0.0f && dmaxf < 3.0e38f)) return 3.4e38f;
    const double sigma = double(mm_scale(dmax_bits)), M = double(dmaxf) * sigma, lim = limit * sigma * sigma;
    constexpr double U = 5.9604644775390625e-08, FL = 6.103515625e-05;    // 2^-24, 2^-14
    const double eta = 6.0 * U * M + 2.0 * (4.8828125e-04 * M + 0.5 * U);   // fp32 storage + the float16 rounding of both operands
    const double b = 2.0 * eta * sqrt(double(KD));
    const double y = 0.5 * (b + sqrt(b * b + 4.0 * lim));   // sqrt of the smallest T with T - b sqrt(T) >= lim
    // (the accumulation error on the term sum with the limit itself among the terms)
    const double E0 = 2.0 * FL + 9.5367431640625e-07 * 4.0 * double(KD) * 64.0 * 64.0;
    const double l = ((y * y + E0) * (1.0 + 9.5367431640625e-07) / (1.0 - 9.5367431640625e-07)) * (1.0 + 1e-6) + 1e-30;
    if (!(l < 1.0e38)) return 3.4e38f;
    float f = float(l);
    if (double(f) <= l) f = __uint_as_float(__float_as_uint(f) + 1u);   // (strictly above: the test is `sum - limit < 0`)
    return f;
}

// records of n structures from their descriptors (runs that get their records outside k_open_rows: the screen's self-test)
inline __global__ __launch_bounds__(256) void k_mm_records(const float *__restrict__ D, int64_t n, const unsigned *__restrict__ dmax_bits, _Float16 *__restrict__ col_rec) {
    const float sigma = mm_scale(*dmax_bits);
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        float d[DW];
#pragma unroll
        for (int q = 0; q < DW / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(D + i * DW + 4 * q);
            d[4 * q] = v.x, d[4 * q + 1] = v.y, d[4 * q + 2] = v.z, d[4 * q + 3] = v.w;
        }
        mm_write_record(d, sigma, col_rec + i * MM_REC_HALVES);
    }
}

// operand registers of a wavefront: lane = (k group g = lane / 16: slots 4 g .. 4 g + 3, row or column lane % 16)
__device__ inline f16x4 mm_load_A(const _Float16 *__restrict__ col_rec, int fam, int g) {   // the row's record
    if (g < 2) return *reinterpret_cast<const f16x4 *>(col_rec + 8 * g + 4 * fam) * _Float16(-0.5f);   // (-2 x) * (-1/2): exact
    if (g == 2) {   // [n0 n1 n2 1] from the column form's [1 1 1 n0 | n1 n2 0 0]
        const f16x4 c2 = *reinterpret_cast<const f16x4 *>(col_rec + 16 + 4 * fam), c3 = *reinterpret_cast<const f16x4 *>(col_rec + 24 + 4 * fam);
        return f16x4{c2[3], c3[0], c3[1], _Float16(1.0f)};
    }
    return f16x4{_Float16(1.0f), _Float16(1.0f), _Float16(0.0f), _Float16(0.0f)};
}
__device__ inline f16x4 mm_load_B(const _Float16 *__restrict__ col_rec, int fam, int g) {   // the column's record
    return *reinterpret_cast<const f16x4 *>(col_rec + 8 * g + 4 * fam);
}
// both families' operands of a column in one 16-byte load
__device__ inline void mm_load_B2(const _Float16 *__restrict__ col_rec, int g, f16x4 (&B)[NFAM]) {
    static_assert(NFAM == 2, "two families side by side");
    const f32x4 v = *reinterpret_cast<const f32x4 *>(col_rec + 8 * g);
    const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    B[0] = __builtin_bit_cast(f16x4, lo), B[1] = __builtin_bit_cast(f16x4, hi);
}

// Level 2: the fp32 screen of sieve.hpp on one pair's stored descriptors (16 floats each, component 2 k + fam): true = the pair may be
// within the limit in both families (S = fl(fl(|a|^2 + |b|^2) - 2 fl(a.b)), the form screen_limit32_dot bounds)
__device__ inline bool mm_pair_within32(const float *__restrict__ da, const float *__restrict__ db, float limit32) {
    f32x2 dot = {0.0f, 0.0f}, na = {0.0f, 0.0f}, nb = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < DW / 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(da + 4 * q), b = *reinterpret_cast<const f32x4 *>(db + 4 * q);
        const f32x2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
        dot = __builtin_elementwise_fma(a0, b0, dot), na = __builtin_elementwise_fma(a0, a0, na), nb = __builtin_elementwise_fma(b0, b0, nb);
        dot = __builtin_elementwise_fma(a1, b1, dot), na = __builtin_elementwise_fma(a1, a1, na), nb = __builtin_elementwise_fma(b1, b1, nb);
    }
    const f32x2 s2 = __builtin_elementwise_fma(dot, f32x2{-2.0f, -2.0f}, na + nb);
    return !(fmaxf(s2.x, s2.y) > limit32);   // (a NaN never rejects: sieve.hpp)
}

// The screen's values themselves, for tests: S[fam][r][c] of rows [0, 64) against columns [0, n_cols) of the records (one wavefront
// per 16 columns), and the limit's bit pattern.
inline __global__ __launch_bounds__(64) void k_mm_screen_dump(const _Float16 *__restrict__ recs, int n_cols,
                                                        const unsigned *__restrict__ dmax_bits, double limit, float *__restrict__ S, int *__restrict__ limit_bits) {
    const int lane = threadIdx.x, g = lane >> 4, rc = lane & 15, c0 = blockIdx.x * MM_STEP;
    if (blockIdx.x == 0 && lane == 0) *limit_bits = __float_as_int(screen_limit_mm(*dmax_bits, limit));
    const int col = min(c0 + rc, n_cols - 1);
#pragma unroll
    for (int fam = 0; fam < NFAM; ++fam) {
        const f16x4 b = mm_load_B(recs + int64_t(col) * MM_REC_HALVES, fam, g);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int row = min(16 * rt + rc, n_cols - 1);
            const f16x4 a = mm_load_A(recs + int64_t(row) * MM_REC_HALVES, fam, g);
            const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c0 + rc < n_cols) S[(size_t(fam) * MM_ROWS + 16 * rt + 4 * g + i) * n_cols + c0 + rc] = acc[i];
        }
    }
}

#ifndef TSC_MM_OCC
#define TSC_MM_OCC 4
#endif
#ifndef TSC_MM_LEVEL2
#define TSC_MM_LEVEL2 0   // 1: level 2 (below) where a pair is decoded in the 64-row kernels.  Measured at C4: 6.95 ms with it, 6.74 without -- the 7 %
                          // more pairs that reach H cost less than its eight loads in front of every batch; the 16-row kernel never had it
#endif
#ifndef TSC_MM_WAVES
#define TSC_MM_WAVES 4
#endif
constexpr int MM_WAVES = TSC_MM_WAVES;   // wavefronts (work items) per workgroup of the walked passes' kernel

// The pair kernel with the screen on the matrix cores.  Grid (groups of 64 rows / 4, column segments); everything around the screen --
// queue, evaluation stages, the fused apply and pass closing -- as in k_rmsd_sieve (sieve.hpp), with 64 rows per work item.
// Dh: the records by POSITION (k_open_rows writes them every pass, like the fp32 copy Dc the other pair kernels read).
template <bool FUSED, bool F32>
inline __global__ __launch_bounds__(64 * MM_WAVES, TSC_MM_OCC) void k_rmsd_sieve_mm(const double *__restrict__ heavy, const int32_t *__restrict__ act,
                                                                    const double *__restrict__ Gall, const float *__restrict__ Dc,
                                                                    const _Float16 *__restrict__ Dh,
                                                                    const int32_t *__restrict__ cend, int32_t *__restrict__ best,
                                                                    PassCounters *__restrict__ counters, const PruneState *__restrict__ st, SieveArgs a,
                                                                    FusedApply fa) {
    __shared__ unsigned short s_queue[MM_WAVES][MM_QCAP];
    __shared__ unsigned short s_exq[MM_WAVES][128];
    __shared__ double s_jacobi[MM_WAVES][32];
    __shared__ __attribute__((aligned(16))) int s_cend[MM_WAVES][MM_ROWS];   // stop column of every row that was looking when the item began (else 0)
    const int lane = threadIdx.x & 63, g = lane >> 4, rc = lane & 15;
    const int wid = MM_WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = a.tile_begin + (int(blockIdx.x) * MM_WAVES + wid) * a.tile_stride;   // (groups of 64 rows dealt round-robin to the ranks; one rank: 0, 1)
    const int R0 = grp * MM_ROWS;
    const int seg_lo = R0 + int(blockIdx.y) * a.seg_cols, seg_hi = seg_lo + a.seg_cols;
    TSC_STAMP(0);  // the wavefront has started
    if (R0 >= a.n || seg_lo >= a.n) return;
    // largest stop columns of the group's four row tiles (0 for tiles beyond the bound the grid was sized for: their entries are stale)
    int tcm_t = 0;
    if (lane < 4 && R0 + 16 * lane < a.n) tcm_t = a.tile_cmax[4 * grp + lane];
    int tcm = tcm_t;
    tcm = max(tcm, __shfl_xor(tcm, 1)), tcm = max(tcm, __shfl_xor(tcm, 2));
    tcm = __builtin_amdgcn_readfirstlane(tcm);
    if (tcm <= seg_lo) return;
    const int pass_on = st->pass_on, A = st->A, bitsel = st->bitsel;
    const int slot = grp;

    do {   // ---- the work item (left by `break`: the fused tail below runs for every item that counts as one of its tiles')
        if (pass_on == 0 || R0 >= A) break;
        const int nrows = min(MM_ROWS, A - R0);
        int my_cend = 0, my_best = 0;
        if (lane < nrows) {
            my_cend = cend[R0 + lane];
            my_best = __hip_atomic_load(&best[R0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool live0 = lane < nrows && my_cend > max(R0 + lane + 1, seg_lo) && my_best >= seg_lo;
        unsigned long long alive = __ballot(live0);
        if (!alive) break;
        const float limit_mm = screen_limit_mm(*a.dmax_bits, a.desc_limit);
        const float limit32 = screen_limit32_dot(__uint_as_float(*a.dmax_bits), a.desc_limit);   // (level 2)
        int *scend = s_cend[wid];
        scend[lane] = live0 ? my_cend : 0;
        int cmax = live0 ? my_cend : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
        cmax = min(__builtin_amdgcn_readfirstlane(cmax), seg_hi);
        f16x4 Ar[4][NFAM];   // the rows' operands
        auto load_A = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int64_t row = min(R0 + 16 * rt + rc, a.n - 1);
#pragma unroll
                for (int fam = 0; fam < NFAM; ++fam) Ar[rt][fam] = mm_load_A(Dh + row * MM_REC_HALVES, fam, g);
            }
        };
        auto load_B = [&](int c0, f16x4 (&B)[NFAM]) __attribute__((always_inline)) {
            mm_load_B2(Dh + int64_t(min(c0 + rc, a.n - 1)) * MM_REC_HALVES, g, B);
        };
        // Rows that are not (or no longer) looking -- beyond the active count, their similar column found, their stop column passed --
        // carry +inf in the n0 slot of family 0 (lanes of k group 2 hold it, element 0): they never come below the limit.  Nothing
        // else is masked at screen time: a pair left of the diagonal or at or beyond its row's stop column (the step in which a range
        // ends; scend[]: 0 for rows that were not looking when the item began) is dropped where it is decoded.  (A row that finds its
        // column keeps its stop column there: pairs queued EARLIER may hold a smaller column still.)
        int next_end = 0;                    // smallest stop column of the rows still looking
        auto mark_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
                if (g == 2 && !((alive >> (16 * rt + rc)) & 1ull)) Ar[rt][0][0] = _Float16(__builtin_inff());
            int e = ((alive >> lane) & 1ull) ? my_cend : INT_MAX;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) e = min(e, __shfl_xor(e, off));
            next_end = __builtin_amdgcn_readfirstlane(e);
            __builtin_amdgcn_wave_barrier();
        };
        TSC_STAMP(1);

        const int h3 = a.h * 3;
        unsigned short *queue = s_queue[wid];
        unsigned short *exq = s_exq[wid];
        int qn = 0, qe = 0;
        unsigned long long n_eval = 0, n_exact = 0;
        int my_screened = 0;
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        int64_t si = 0, sj = 0;
        // queue entry: row (6 bits) | column offset inside the segment (10 bits: segments are <= 1024 columns).
        // Returns whether the pair is still one to look at: right of the diagonal, before the row's stop column and before the similar
        // column the row already has (a later one cannot lower the minimum), and within the limit of the fp32 screen (level 2).
        auto decode = [&](unsigned e, int &t, int &col, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
            t = int(e >> 10);
            col = seg_lo + int(e & 0x3ffu);
            const int64_t i = act[R0 + t], j = act[col];
            si = i, sj = j;
            pp = heavy + i * h3, pq = heavy + j * h3;
            Gi = Gall[i], Gj = Gall[j];
            return col > R0 + t && col < scend[t] && col < __hip_atomic_load(&best[R0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
                   (!TSC_MM_LEVEL2 || mm_pair_within32(Dc + int64_t(R0 + t) * DW, Dc + int64_t(col) * DW, limit32));
        };
        auto sign_stage = [&](int base, int cnt) __attribute__((always_inline)) {
            int lpp = 64;
            while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
            const int gq = lane / lpp, sub = lane - gq * lpp;
            bool cand = false, sim = false, counted = false;
            unsigned e = 0;
            int t = 0;
            if (gq < cnt) {
                e = queue[base + gq];
                int col;
                const double *pp, *pq;
                double Gi, Gj;
                if (decode(e, t, col, pp, pq, Gi, Gj)) {   // (the lanes of a group hold the same pair: they branch together)
                    const int verdict = pair_stage1<F32>(heavy, a.heavy32, si, sj, a.h, 0.5 * (Gi + Gj), a.half_h_thr2, a.two_thr2, sub, lpp);
                    cand = sub == 0 && verdict == PAIR_UNDECIDED;
                    sim = sub == 0 && verdict == PAIR_SIMILAR;
                    counted = sub == 0;
                    if (sim) atomicMin(&best[R0 + t], col);
                }
            }
            unsigned long long sm = __builtin_amdgcn_ballot_w64(sim);
            while (sm) {  // rows that found a similar column stop being screened (the reference returns there, :75-77)
                const int l = __ffsll((long long)sm) - 1;
                sm &= sm - 1;
                alive &= ~(1ull << __builtin_amdgcn_readlane(t, l));
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
            if (m) {
                if (cand) exq[qe + __popcll(m & lt_mask)] = (unsigned short)e;
                qe += __popcll(m);
            }
            n_eval += __popcll(__builtin_amdgcn_ballot_w64(counted));
            n_exact += __popcll(m);
            __builtin_amdgcn_wave_barrier();
        };
        auto exact_stage = [&](int base, int cnt) __attribute__((always_inline)) {
            int lpp = 64;
            while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
            const int gq = lane / lpp, sub = lane - gq * lpp;
            bool sim = false, degenerate = false;
            int t = 0;
            unsigned ent = 0;
            if (gq < cnt) {
                int col;
                const double *pp, *pq;
                double Gi, Gj, H[9], e[4];
                ent = exq[base + gq];
                (void)decode(ent, t, col, pp, pq, Gi, Gj);
                pair_H(pp, pq, a.h, sub, lpp, H);
                if (rotation_quaternion_fast(H, Gi, Gj, e)) {
                    double rm, md;
                    residual_rmsd_maxdev(pp, pq, a.h, e, rm, md, sub, lpp);
                    sim = sub == 0 && rm < a.thr && md < a.maxdev_thr;  // rmsd_pruning.py:75
                    if (sim) atomicMin(&best[R0 + t], col);
                } else {
                    degenerate = sub == 0;
                }
            }
            unsigned long long sm = __builtin_amdgcn_ballot_w64(sim);
            while (sm) {
                const int l = __ffsll((long long)sm) - 1;
                sm &= sm - 1;
                alive &= ~(1ull << __builtin_amdgcn_readlane(t, l));
            }
            // degenerate top eigenvalue: the whole wavefront takes such a pair on, one at a time (sieve.hpp)
            for (unsigned long long dm = __builtin_amdgcn_ballot_w64(degenerate); dm; dm &= dm - 1) {
                const unsigned e1 = unsigned(__builtin_amdgcn_readlane(int(ent), __ffsll((long long)dm) - 1));
                int t2, col2;
                const double *pp, *pq;
                double Gi, Gj, H[9], e[4], rm, md;
                (void)decode(e1, t2, col2, pp, pq, Gi, Gj);
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
                    if (lane == 0) atomicMin(&best[R0 + t2], col2);
                    alive &= ~(1ull << t2);
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        auto drain = [&](int base, int cnt) __attribute__((always_inline)) {
            sign_stage(base, cnt);
            if (qe >= 64) {
                exact_stage(qe - 64, 64);
                qe -= 64;
            }
        };

        load_A();
        mark_rows();
        unsigned long long marked = alive;   // rows whose operand and scend[] entry say what `alive` says
        // a step = 32 columns: two blocks of 16, their operands requested one step ahead (a step's 16 MFMAs and 32 compares stand
        // against one trip to the L2)
        f16x4 Bn[2][NFAM];
        load_B(seg_lo, Bn[0]);
        load_B(seg_lo + MM_STEP, Bn[1]);
        for (int c0 = seg_lo; c0 < cmax;) {   // (alive != 0 here)
            f16x4 Bc[2][NFAM];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int fam = 0; fam < NFAM; ++fam) Bc[u][fam] = Bn[u][fam];
            if (c0 + 2 * MM_STEP < cmax) {
                load_B(c0 + 2 * MM_STEP, Bn[0]);
                load_B(c0 + 3 * MM_STEP, Bn[1]);
            }
            // Every accumulator starts at -limit: a pair is kept iff BOTH families come out negative.  The sign bits are shifted into a
            // mask per family (one v_alignbit_b32 per value; no compare, no branch per value): value j of the step ends at bit 31 - j.
            unsigned m0 = 0, m1 = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const f32x4 z = {-limit_mm, -limit_mm, -limit_mm, -limit_mm};
                    const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x16f16(Ar[rt][0], Bc[u][0], z, 0, 0, 0);
                    const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x16f16(Ar[rt][1], Bc[u][1], z, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(s0[i]), 31);
                        m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(s1[i]), 31);
                    }
                }
            }
            unsigned hits = m0 & m1;
            while (__builtin_amdgcn_ballot_w64(hits != 0u)) {
                // one pair per lane and turn (a lane rarely holds two); kept if it lies right of the diagonal and before its row's stop
                // column -- a batch of candidates is evaluated with 64 / (their number) lanes per pair, and what would be dropped where it
                // is decoded makes the others' chains longer
                bool keep = false;
                unsigned ent = 0;
                if (hits) {
                    const int j = __builtin_clz(hits);                     // the earliest value of the step among this lane's
                    hits &= ~(0x80000000u >> j);
                    const int row = 16 * ((j >> 2) & 3) + 4 * g + (j & 3), col = c0 + MM_STEP * (j >> 4) + rc;
                    keep = col > R0 + row && col < scend[row];
                    ent = (unsigned(row) << 10) | unsigned(col - seg_lo);
                }
                const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
                if (keep) queue[qn + __popcll(km & lt_mask)] = (unsigned short)ent;
                qn += __popcll(km);
            }
            // (statistics: the pairs of this step inside the live rows' ranges)
            my_screened += ((alive >> lane) & 1ull) ? max(0, min(my_cend, c0 + 2 * MM_STEP) - max(R0 + lane + 1, c0)) : 0;
            __builtin_amdgcn_wave_barrier();
            c0 += 2 * MM_STEP;
            while (qn >= 64) {
                drain(qn - 64, 64);
                qn -= 64;
            }
            if (c0 >= next_end) alive &= ~__ballot(((alive >> lane) & 1ull) && my_cend <= c0);   // rows whose range ends here
            if (!alive) break;
            if (alive != marked) {
                mark_rows();
                marked = alive;
            }
        }
        TSC_STAMP(2);
        if (qn > 0) drain(0, qn);
        if (qe > 0) exact_stage(0, qe);
        TSC_STAMP(3);
        for (int off = 32; off > 0; off >>= 1) my_screened += __shfl_xor(my_screened, off);
        if (lane == 0) {
            count_add(counters, unsigned(slot), CNT_FORMED, n_eval);
            count_add(counters, unsigned(slot), CNT_EXACT, n_exact);
            count_add(counters, unsigned(slot), CNT_SCREENED, (unsigned long long)my_screened);
        }
    } while (false);

    if constexpr (FUSED) {
        // (as in k_rmsd_sieve, per row tile: the item that arrives LAST at a tile applies its 16 rows; the last tile closes the pass)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        bool last = false;
        if (lane < 4 && tcm_t > seg_lo) {
            const int lim = min(a.n, tcm_t);
            const int n_live = min(int(gridDim.y), (lim - R0 + a.seg_cols - 1) / a.seg_cols);
            last = n_live <= 1 || atomicAdd(&fa.tile_done[4 * grp + lane], 1) == n_live - 1;
            if (last && n_live > 1) fa.tile_done[4 * grp + lane] = 0;
        }
        TSC_STAMP(4);
        const unsigned long long lm = __ballot(last);   // bits 0..3: the tiles this item applies
        if (!lm) return;
        unsigned long long ev_total = 0, rm_total = 0;
        apply_wave_rows(fa.ap, bitsel, R0 + lane, ((lm >> g) & 1ull) && R0 + lane < A, ev_total, rm_total);
        if (lane == 0) {
            count_add(counters, unsigned(slot), CNT_EVALUATED, ev_total);
            count_add(counters, unsigned(slot), CNT_REMOVED, rm_total);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSC_STAMP(5);
        const bool fin = last && tickets_arrive(fa.tickets, unsigned(4 * grp + lane), fa.n_tiles, PT_GROUPS);
        TSC_STAMP(6);
        if (__ballot(fin)) {
            pass_step_wave(fa.sc, fa.next);
            TSC_STAMP(7);
        }
    }
}

// ---- the same screen for 16-ROW work items: k_rmsd_sieve's items, grid and tail (sieve.hpp) with level 1 in place of the packed-fp32
// screen.  A column tile of 128 (8 KB of records, requested whole) against the item's 16 rows: 16 MFMAs, 64 sign-bit shifts -- against 16 x
// 34 vector instructions.  Four times the column bytes per pair of the 64-row form (each row tile loads its own columns, as the
// packed-fp32 kernel does): the form for runs that are NOT bound by that -- below mm_min_n, where the longer items of the 64-row form lose.
#ifndef TSC_MM16_OCC
#define TSC_MM16_OCC 4
#endif
// work items (wavefronts) per workgroup of the 16-row kernel: 2 where most workgroups of the grid have work (a workgroup keeps its slot until
// its longest item is through: C3's k = 100 .. 2 passes 2 - 4 us shorter than with 4), 4 where most are empty and the launch is its dispatch
// (C3's last pass: 28 us against 44) -- the host picks by the number of column segments (launch_pair_search)
constexpr int MM16_LONG_SEGS = 64;
constexpr int MM16_BLOCKS = 128 / MM_STEP;   // 16-column blocks of a column tile

template <bool F32, int MM16_WAVES>
__device__ __forceinline__ void sieve_item_mm16(const double *__restrict__ heavy, const int32_t *__restrict__ act, const double *__restrict__ Gall,
                                                const _Float16 *__restrict__ Dh,
                                                const int32_t *__restrict__ cend, int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                                const PruneState *__restrict__ st, const SieveArgs &a, int &A_out, int &bitsel_out, const int tile, const int seg) {
    constexpr int TI = 16, TILE_COLS = 128;
    constexpr int QCAP = TI * TILE_COLS + 64;
    __shared__ unsigned short s_queue[MM16_WAVES][QCAP];
    __shared__ unsigned short s_exq[MM16_WAVES][128];
    __shared__ double s_jacobi[MM16_WAVES][32];
    __shared__ int s_cend[MM16_WAVES][TI];   // stop column of every row that was looking when the item began (else 0)
    const int lane = threadIdx.x & 63, g = lane >> 4, rc = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = tile;
    const int r0 = tile * TI;
    const int seg_lo = ((r0 + 1) & ~63) + seg * a.seg_cols;
    const int seg_hi = seg_lo + a.seg_cols;
    const int pass_on = st->pass_on, n_active = st->A;
    A_out = n_active, bitsel_out = st->bitsel;
    int my_cend = 0, my_best = 0;
    if (lane < TI && r0 + lane < a.n) {
        my_cend = cend[r0 + lane];
        my_best = __hip_atomic_load(&best[r0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the rows' operands and the first column tile: requested with the stop columns (one trip for an item of a light pass)
    f16x4 Ar[NFAM];
    {
        const int64_t row = min(r0 + rc, a.n - 1);
#pragma unroll
        for (int fam = 0; fam < NFAM; ++fam) Ar[fam] = mm_load_A(Dh + row * MM_REC_HALVES, fam, g);
    }
    f16x4 B[MM16_BLOCKS][NFAM];
    auto load_tile = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < MM16_BLOCKS; ++b) {
            mm_load_B2(Dh + int64_t(min(c0 + MM_STEP * b + rc, a.n - 1)) * MM_REC_HALVES, g, B[b]);
        }
    };
    load_tile(seg_lo);
    if (pass_on == 0 || r0 >= n_active) return;
    const int nrows = min(TI, n_active - r0);
    const bool live0 = lane < nrows && my_cend > max(r0 + lane + 1, seg_lo) && my_best >= seg_lo;
    unsigned alive = unsigned(__ballot(live0));
    if (!alive) return;
    const float limit_mm = screen_limit_mm(*a.dmax_bits, a.desc_limit);
    int *scend = s_cend[wid];
    if (lane < TI) scend[lane] = live0 ? my_cend : 0;
    int cmax = live0 ? my_cend : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
    cmax = min(__builtin_amdgcn_readfirstlane(cmax), seg_hi);
    // rows that are not (or no longer) looking: +inf in the n0 slot of family 0 (lanes of k group 2, element 0): they never pass
    auto mark_rows = [&]() __attribute__((always_inline)) {
        if (g == 2 && !((alive >> rc) & 1u)) Ar[0][0] = _Float16(__builtin_inff());
    };
    mark_rows();
    __builtin_amdgcn_wave_barrier();
    TSC_STAMP(1);  // prologue data has arrived

    const int h3 = a.h * 3;
    unsigned short *queue = s_queue[wid], *exq = s_exq[wid];
    int qn = 0, qe = 0;
    unsigned long long n_eval = 0, n_exact = 0;
    int my_screened = 0;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int64_t si = 0, sj = 0;
    // queue entry: row (4 bits) | column offset inside the segment (12 bits).  Returns whether the pair is still one to look at: right of
    // the diagonal, before the row's stop column and the similar column it already has, within the limit of the fp32 screen (level 2)
    auto decode = [&](unsigned e, int &t, int &col, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
        t = int(e >> 12);
        col = seg_lo + int(e & 0xfffu);
        const int64_t i = act[r0 + t], j = act[col];
        si = i, sj = j;
        pp = heavy + i * h3, pq = heavy + j * h3;
        Gi = Gall[i], Gj = Gall[j];
        // (no level 2 here: the few per cent it would drop cost an item less as H than as eight more loads in front of every batch)
        return col > r0 + t && col < scend[t] && col < __hip_atomic_load(&best[r0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto sign_stage = [&](int base, int cnt) __attribute__((always_inline)) {
        int lpp = 64;
        while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
        const int gq = lane / lpp, sub = lane - gq * lpp;
        bool cand = false, sim = false, counted = false;
        unsigned e = 0;
        int t = 0;
        if (gq < cnt) {
            e = queue[base + gq];
            int col;
            const double *pp, *pq;
            double Gi, Gj;
            if (decode(e, t, col, pp, pq, Gi, Gj)) {   // (the lanes of a group hold the same pair: they branch together)
                const int verdict = pair_stage1<F32>(heavy, a.heavy32, si, sj, a.h, 0.5 * (Gi + Gj), a.half_h_thr2, a.two_thr2, sub, lpp);
                cand = sub == 0 && verdict == PAIR_UNDECIDED;
                sim = sub == 0 && verdict == PAIR_SIMILAR;
                counted = sub == 0;
                if (sim) atomicMin(&best[r0 + t], col);
            }
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
        n_eval += __popcll(__builtin_amdgcn_ballot_w64(counted));
        n_exact += __popcll(m);
        __builtin_amdgcn_wave_barrier();
    };
    auto exact_stage = [&](int base, int cnt) __attribute__((always_inline)) {
        int lpp = 64;
        while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
        const int gq = lane / lpp, sub = lane - gq * lpp;
        bool sim = false, degenerate = false;
        int t = 0;
        unsigned ent = 0;
        if (gq < cnt) {
            int col;
            const double *pp, *pq;
            double Gi, Gj, H[9], e[4];
            ent = exq[base + gq];
            (void)decode(ent, t, col, pp, pq, Gi, Gj);
            pair_H(pp, pq, a.h, sub, lpp, H);
            if (rotation_quaternion_fast(H, Gi, Gj, e)) {
                double rm, md;
                residual_rmsd_maxdev(pp, pq, a.h, e, rm, md, sub, lpp);
                sim = sub == 0 && rm < a.thr && md < a.maxdev_thr;  // rmsd_pruning.py:75
                if (sim) atomicMin(&best[r0 + t], col);
            } else {
                degenerate = sub == 0;
            }
        }
        unsigned long long sm = __builtin_amdgcn_ballot_w64(sim);
        while (sm) {
            const int l = __ffsll((long long)sm) - 1;
            sm &= sm - 1;
            alive &= ~(1u << __builtin_amdgcn_readlane(t, l));
        }
        for (unsigned long long dm = __builtin_amdgcn_ballot_w64(degenerate); dm; dm &= dm - 1) {   // (sieve.hpp: the Jacobi fallback)
            const unsigned e1 = unsigned(__builtin_amdgcn_readlane(int(ent), __ffsll((long long)dm) - 1));
            int t2, col2;
            const double *pp, *pq;
            double Gi, Gj, H[9], e[4], rm, md;
            (void)decode(e1, t2, col2, pp, pq, Gi, Gj);
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
        if (qe >= 64) {
            exact_stage(qe - 64, 64);
            qe -= 64;
        }
    };

    unsigned marked = alive;
    for (int c0 = seg_lo; c0 < cmax;) {   // (alive != 0 here)
        // every accumulator starts at -limit: a pair is kept iff both families come out negative; sign bits into a mask per family, value
        // j = 4 b + i of the tile -- row 4 g + i, column 16 b + rc -- at bit 31 - j
        unsigned m0 = 0, m1 = 0;
#pragma unroll
        for (int b = 0; b < MM16_BLOCKS; ++b) {
            const f32x4 z = {-limit_mm, -limit_mm, -limit_mm, -limit_mm};
            const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x16f16(Ar[0], B[b][0], z, 0, 0, 0);
            const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x16f16(Ar[1], B[b][1], z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(s0[i]), 31);
                m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(s1[i]), 31);
            }
        }
        // (statistics: the pairs of this tile inside the live rows' ranges, as k_rmsd_sieve counts them)
        my_screened += (lane < TI && ((alive >> lane) & 1u)) ? max(0, min(my_cend, c0 + TILE_COLS) - max(r0 + lane + 1, c0)) : 0;
        unsigned hits = m0 & m1;
        while (__builtin_amdgcn_ballot_w64(hits != 0u)) {
            // one pair per lane and turn; kept if it lies right of the diagonal and before its row's stop column (k_rmsd_sieve_mm)
            bool keep = false;
            unsigned ent = 0;
            if (hits) {
                const int j = __builtin_clz(hits);
                hits &= ~(0x80000000u >> j);
                const int row = 4 * g + (j & 3), col = c0 + MM_STEP * (j >> 2) + rc;
                keep = col > r0 + row && col < scend[row];
                ent = (unsigned(row) << 12) | unsigned(col - seg_lo);
            }
            const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
            if (keep) queue[qn + __popcll(km & lt_mask)] = (unsigned short)ent;
            qn += __popcll(km);
        }
        __builtin_amdgcn_wave_barrier();
        while (qn >= a.drain_min) {
            const int cnt = min(qn, 64);
            drain(qn - cnt, cnt);
            qn -= cnt;
        }
        c0 += TILE_COLS;
        alive &= ~unsigned(__ballot(lane < TI && ((alive >> lane) & 1u) && my_cend <= c0));   // rows whose range ends here
        if (!(c0 < cmax && alive)) break;
        load_tile(c0);   // (here, after the batches, so that the tile does not occupy registers during them)
        if (alive != marked) {
            mark_rows();
            marked = alive;
        }
    }
    TSC_STAMP(2);  // screen done
    if (qn > 0) drain(0, qn);
    if (qe > 0) exact_stage(0, qe);
    TSC_STAMP(3);  // candidates evaluated
    for (int off = 8; off > 0; off >>= 1) my_screened += __shfl_xor(my_screened, off);
    if (lane == 0) {
        count_add(counters, unsigned(slot), CNT_FORMED, n_eval);
        count_add(counters, unsigned(slot), CNT_EXACT, n_exact);
        count_add(counters, unsigned(slot), CNT_SCREENED, (unsigned long long)my_screened);
    }
}

template <bool FUSED, bool F32, int MM16_WAVES>
inline __global__ __launch_bounds__(64 * MM16_WAVES, TSC_MM16_OCC) void k_rmsd_sieve_mm16(const double *__restrict__ heavy, const int32_t *__restrict__ act,
                                                                        const double *__restrict__ Gall,                                                                         const _Float16 *__restrict__ Dh,
                                                                        const int32_t *__restrict__ cend, int32_t *__restrict__ best,
                                                                        PassCounters *__restrict__ counters, const PruneState *__restrict__ st, SieveArgs a,
                                                                        FusedApply fa) {
    constexpr int TI = 16;
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * MM16_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = a.tile_begin + slot * a.tile_stride;
    const int r0 = tile * TI;
    const int seg_base = (r0 + 1) & ~63;
    const int seg_lo = seg_base + int(blockIdx.y) * a.seg_cols;
    TSC_STAMP(0);  // the wavefront has started
    if (r0 >= a.n || seg_lo >= a.n) return;
    const int tcm = a.tile_cmax[tile];
    if (tcm <= seg_lo) return;
    int A = 0, bitsel = 0;
    sieve_item_mm16<F32, MM16_WAVES>(heavy, act, Gall, Dh, cend, best, counters, st, a, A, bitsel, tile, int(blockIdx.y));
    if constexpr (FUSED) {   // (k_rmsd_sieve's tail, sieve.hpp)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const int lim = min(a.n, tcm);
        const int n_live = min(int(gridDim.y), (lim - seg_base + a.seg_cols - 1) / a.seg_cols);
        if (n_live > 1) {
            int last = 0;
            if (lane == 0) last = (atomicAdd(&fa.tile_done[tile], 1) == n_live - 1) ? 1 : 0;
            TSC_STAMP(4);
            if (!__builtin_amdgcn_readfirstlane(last)) return;
            if (lane == 0) fa.tile_done[tile] = 0;
        }
        unsigned long long ev_total = 0, rm_total = 0;
        apply_wave_rows(fa.ap, bitsel, r0 + lane, lane < TI && r0 + lane < A, ev_total, rm_total);
        int fin = 0;
        if (lane == 0) {
            count_add(counters, unsigned(tile), CNT_EVALUATED, ev_total);
            count_add(counters, unsigned(tile), CNT_REMOVED, rm_total);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSC_STAMP(5);
        if (lane == 0) fin = tickets_arrive(fa.tickets, unsigned(tile), fa.n_tiles, PT_GROUPS) ? 1 : 0;
        TSC_STAMP(6);
        if (__builtin_amdgcn_readfirstlane(fin)) {
            pass_step_wave(fa.sc, fa.next);
            TSC_STAMP(7);
        }
    }
}

}  // namespace tsc
