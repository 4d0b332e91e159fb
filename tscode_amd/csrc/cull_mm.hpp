// cull_mm.hpp -- the pair kernel of a culled pass (cull.hpp) with the descriptor screen on the matrix cores (mm.hpp).
//
// cull.hpp's culling (row tiles of 16 and column tiles of 128 positions of the sorted layout, each with its bounding box; a tile pair
// whose boxes lie beyond the limit is never multiplied) in front of level 1 of mm.hpp; level 2 and everything cull.hpp's decode tests
// (visited once, inside the row's range, before the similar column the row already has) where a pair is decoded.
#pragma once
#include "cull.hpp"
#include "mm.hpp"

namespace tsc {

constexpr int CMM_SEG = 1024;             // columns per work item (queue entries keep the column offset in 10 bits)
constexpr int CMM_TILES = CMM_SEG / CULL_COLS;

struct CullMmArgs {
    const _Float16 *Dhs;         // the float16 records (mm_record.hpp) by sorted position, like CullArgs::Ds
    const int32_t *cstruct;      // the structure at every sorted position (act[crank[pos]])
};

#ifndef TSC_CMM_OCC
#define TSC_CMM_OCC 4
#endif

// One wavefront = 64 consecutive positions of the sorted layout (four row tiles of 16, each with its bounding box) x one segment of
// 1024 columns (eight column tiles of 128, each with its box) at or behind them, to the end of the rows' chunk.  A (row tile, column
// tile) pair whose boxes lie further apart than the screen's limit is never multiplied; a column tile that no row tile needs is never
// loaded.  (16-row items with 4096-column segments and a whole column tile requested ahead -- cull.hpp's shape -- were measured too:
// C4's k = 2 pass 2.57 ms against 1.98 in this form; the commit before this one.)
template <bool F32>
__device__ __forceinline__ void sieve_item_sorted_mm(const double *__restrict__ heavy, const int32_t *__restrict__ act, const double *__restrict__ Gall,
                                                     const int32_t *__restrict__ cend, int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                                     const PruneState *__restrict__ st, const SieveArgs a, const CullArgs ca, const CullMmArgs cm, const int grp,
                                                     const int seg) {
    static_assert(CULL_COLS == 128 && CMM_TILES == 8 && DW == 16, "tile shapes");
    __shared__ unsigned short s_queue[1][MM_QCAP];
    __shared__ unsigned short s_exq[1][128];
    __shared__ double s_jacobi[1][32];
    const int lane = threadIdx.x & 63, g = lane >> 4, rc = lane & 15;
    constexpr int wid = 0;   // (one wavefront per workgroup: below)
    const int p0 = grp * MM_ROWS;
    const int pass_on = st->pass_on, A = st->A;
    TSC_STAMP(0);  // started
    if (pass_on == 0 || p0 >= A) return;
    const int nrows = min(MM_ROWS, A - p0);
    // the chunk of the group's LAST row ends the column range (a group may straddle a chunk boundary; a pair across it fails the
    // range test at decode time: the higher rank lies at or beyond the lower one's stop column)
    int col_end;
    {
        const int cb = lane <= ca.k ? ca.cbase[lane] : INT_MAX;            // (k <= 63 chunks + the end)
        const unsigned long long le = __ballot(cb <= p0 + nrows - 1);
        const int c_last = __popcll(le) - 1;
        col_end = __builtin_amdgcn_readlane(cb, c_last + 1);
    }
    const int seg_lo = (p0 & ~(CULL_COLS - 1)) + seg * CMM_SEG;
    const int seg_hi = min(seg_lo + CMM_SEG, col_end);
    if (seg_lo >= seg_hi) return;
    // which (row tile, column tile) pairs lie within the limit: lane = (column tile, row tile), one round trip
    const float limit32 = screen_limit32_dot(__uint_as_float(*a.dmax_bits), a.desc_limit);
    unsigned need;   // bit 4 ct + rt
    {
        const int n_ct = (seg_hi - seg_lo + CULL_COLS - 1) / CULL_COLS;    // <= 8
        const int ct = lane >> 2, rt = lane & 3;
        bool near = false;
        if (ct < n_ct && p0 + 16 * rt < A) {
            const f32x4 *rb = reinterpret_cast<const f32x4 *>(ca.rbox + int64_t(p0 / 16 + rt) * CULL_BOX);
            const f32x4 *cb = reinterpret_cast<const f32x4 *>(ca.cbox + int64_t(seg_lo / CULL_COLS + ct) * CULL_BOX);
            float g0 = 0.0f, g1 = 0.0f;
#pragma unroll
            for (int q = 0; q < DW / 4; ++q) {
                const f32x4 rl = rb[q], rh = rb[DW / 4 + q], cl = cb[q], ch = cb[DW / 4 + q];
                const float gx = fmaxf(0.0f, fmaxf(cl.x - rh.x, rl.x - ch.x)), gy = fmaxf(0.0f, fmaxf(cl.y - rh.y, rl.y - ch.y));
                const float gz = fmaxf(0.0f, fmaxf(cl.z - rh.z, rl.z - ch.z)), gw = fmaxf(0.0f, fmaxf(cl.w - rh.w, rl.w - ch.w));
                g0 = fmaf(gx, gx, fmaf(gz, gz, g0));  // components 4q, 4q + 2: family 0
                g1 = fmaf(gy, gy, fmaf(gw, gw, g1));  // components 4q + 1, 4q + 3: family 1
            }
            near = fmaxf(g0, g1) <= limit32 * 1.001f;   // (cull.hpp: every pair of the two tiles is at least sqrt(g) apart in that family)
        }
        need = unsigned(__ballot(near));   // (lanes 32..63 hold no tile pair)
    }
    TSC_STAMP(1);  // boxes tested
    if (!need) return;
    const float limit_mm = screen_limit_mm(*a.dmax_bits, a.desc_limit);

    // the rows' operands; rows beyond the active count carry +inf in the n0 slot of family 0 (mm.hpp): they never pass
    f16x4 Ar[4][NFAM];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t row = min(p0 + 16 * rt + rc, A - 1);
#pragma unroll
        for (int fam = 0; fam < NFAM; ++fam) Ar[rt][fam] = mm_load_A(cm.Dhs + row * MM_REC_HALVES, fam, g);
        if (g == 2 && 16 * rt + rc >= nrows) Ar[rt][0][0] = _Float16(__builtin_inff());
    }
    auto load_B = [&](int c0, f16x4 (&B)[NFAM]) __attribute__((always_inline)) {
        mm_load_B2(cm.Dhs + int64_t(min(c0 + rc, A - 1)) * MM_REC_HALVES, g, B);
    };

    const int h3 = a.h * 3;
    unsigned short *queue = s_queue[wid], *exq = s_exq[wid];
    int qn = 0, qe = 0;
    unsigned long long n_eval = 0, n_exact = 0, n_screened = 0;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // an entry = (row of the group, 6 bits | column position inside the segment, 10 bits); the pair = the two structures at those
    // positions, the one with the lower active rank playing the reference's `ref` (rmsd_pruning.py:92: the row), the other its column
    int64_t si = 0, sj = 0;
    auto decode = [&](unsigned e, int &lo, int &hi, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
        const int prow = p0 + int(e >> 10), pcol = seg_lo + int(e & 0x3ffu);
        const int r1 = ca.crank[prow], r2 = ca.crank[pcol];
        const int s1 = cm.cstruct[prow], s2 = cm.cstruct[pcol];   // (with the ranks, not behind them: act[] would be a round trip more)
        lo = min(r1, r2), hi = max(r1, r2);
        const int64_t i = r1 < r2 ? s1 : s2, j = r1 < r2 ? s2 : s1;
        si = i, sj = j;
        pp = heavy + i * h3, pq = heavy + j * h3;
        Gi = Gall[i], Gj = Gall[j];
        // visited once (from the earlier position), inside the segment; the column inside the row's range (rows of another chunk, or
        // behind a cache hit, are not) and before the similar column the row already has; within the fp32 screen's limit (level 2)
        return pcol > prow && pcol < seg_hi && hi < cend[lo] && hi < __hip_atomic_load(&best[lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
               (!TSC_MM_LEVEL2 || mm_pair_within32(ca.Ds + int64_t(prow) * DW, ca.Ds + int64_t(pcol) * DW, limit32));
    };
    auto exact_stage = [&](int base, int cnt) __attribute__((always_inline)) {
        int lpp = 64;
        while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
        const int gq = lane / lpp, sub = lane - gq * lpp;
        bool degenerate = false;
        unsigned ent = 0;
        if (gq < cnt) {
            int lo, hi;
            const double *pp, *pq;
            double Gi, Gj, H[9], e[4];
            ent = exq[base + gq];
            (void)decode(ent, lo, hi, pp, pq, Gi, Gj);
            pair_H(pp, pq, a.h, sub, lpp, H);
            if (rotation_quaternion_fast(H, Gi, Gj, e)) {
                double rm, md;
                residual_rmsd_maxdev(pp, pq, a.h, e, rm, md, sub, lpp);
                if (sub == 0 && rm < a.thr && md < a.maxdev_thr) atomicMin(&best[lo], hi);  // rmsd_pruning.py:75
            } else {
                degenerate = sub == 0;
            }
        }
        for (unsigned long long dm = __builtin_amdgcn_ballot_w64(degenerate); dm; dm &= dm - 1) {  // (sieve.hpp: the Jacobi fallback, one pair per wavefront)
            const unsigned e1 = unsigned(__builtin_amdgcn_readlane(int(ent), __ffsll((long long)dm) - 1));
            int lo, hi;
            const double *pp, *pq;
            double Gi, Gj, H[9], e[4], rm, md;
            (void)decode(e1, lo, hi, pp, pq, Gi, Gj);
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
            if (rm < a.thr && md < a.maxdev_thr && lane == 0) atomicMin(&best[lo], hi);
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto sign_stage = [&](int base, int cnt) __attribute__((always_inline)) {
        int lpp = 64;
        while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
        const int gq = lane / lpp, sub = lane - gq * lpp;
        bool cand = false, counted = false;
        unsigned e = 0;
        if (gq < cnt) {
            e = queue[base + gq];
            int lo, hi;
            const double *pp, *pq;
            double Gi, Gj;
            if (decode(e, lo, hi, pp, pq, Gi, Gj)) {  // (the lanes of a group hold the same pair: they branch together)
                const int verdict = pair_stage1<F32>(heavy, a.heavy32, si, sj, a.h, 0.5 * (Gi + Gj), a.half_h_thr2, a.two_thr2, sub, lpp);
                cand = sub == 0 && verdict == PAIR_UNDECIDED;
                counted = sub == 0;
                if (sub == 0 && verdict == PAIR_SIMILAR) atomicMin(&best[lo], hi);
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
        if (m) {
            if (cand) exq[qe + __popcll(m & lt_mask)] = (unsigned short)e;
            qe += __popcll(m);
        }
        n_eval += __popcll(__builtin_amdgcn_ballot_w64(counted));
        n_exact += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (qe >= 64) {
            exact_stage(qe - 64, 64);
            qe -= 64;
        }
    };
    // the column tiles some row tile needs, one after the other; a step = 32 columns (two blocks of 16), its operands requested a step
    // ahead (the first step of the next needed tile behind the last one of this)
    unsigned tiles = 0;   // bit ct: some row tile needs column tile ct
#pragma unroll
    for (int ct = 0; ct < CMM_TILES; ++ct) tiles |= ((need >> (4 * ct)) & 0xfu) ? (1u << ct) : 0u;
    TSC_STAMP(2);  // the rows' operands
#ifdef TSC_DBG_STAMPS
    if (a.dbg && lane == 0) a.dbg[(size_t(blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wid * 8 + 7] = (unsigned long long)(__popc(tiles) * 64 + __popc(need));
#endif
    f16x4 Bn[2][NFAM];
    {
        const int c_first = seg_lo + CULL_COLS * (__ffs(tiles) - 1);
        load_B(c_first, Bn[0]);
        load_B(c_first + MM_STEP, Bn[1]);
    }
    for (; tiles; tiles &= tiles - 1) {
        const int ct = __ffs(tiles) - 1;
        const unsigned rts = (need >> (4 * ct)) & 0xfu;   // the row tiles that need this column tile
        const unsigned rest = tiles & (tiles - 1);
        const int c_tile = seg_lo + CULL_COLS * ct;
        const int c_next_tile = rest ? seg_lo + CULL_COLS * (__ffs(rest) - 1) : -1;
        n_screened += (unsigned long long)(16 * __popc(rts)) * (unsigned long long)max(0, min(CULL_COLS, seg_hi - c_tile));
        for (int sub = 0; sub < CULL_COLS / (2 * MM_STEP); ++sub) {
            const int c0 = c_tile + 2 * MM_STEP * sub;
            f16x4 Bc[2][NFAM];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int fam = 0; fam < NFAM; ++fam) Bc[u][fam] = Bn[u][fam];
            const int c_pre = sub + 1 < CULL_COLS / (2 * MM_STEP) ? c0 + 2 * MM_STEP : c_next_tile;
            if (c_pre >= 0) {
                load_B(c_pre, Bn[0]);
                load_B(c_pre + MM_STEP, Bn[1]);
            }
            // every accumulator starts at -limit: a pair is kept iff both families come out negative; sign bits into a mask per family,
            // value j = 16 u + 4 rt + i of the step at bit 31 - j (mm.hpp)
            unsigned m0 = 0, m1 = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    if ((rts >> rt) & 1u) {
                        const f32x4 z = {-limit_mm, -limit_mm, -limit_mm, -limit_mm};
                        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x16f16(Ar[rt][0], Bc[u][0], z, 0, 0, 0);
                        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x16f16(Ar[rt][1], Bc[u][1], z, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(s0[i]), 31);
                            m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(s1[i]), 31);
                        }
                    } else {
                        m0 <<= 4, m1 <<= 4;
                    }
                }
            }
            unsigned hits = m0 & m1;
            for (unsigned long long hm = __builtin_amdgcn_ballot_w64(hits != 0u); hm; hm = __builtin_amdgcn_ballot_w64(hits != 0u)) {
                if (hits) {   // one pair per lane and turn
                    const int j = __builtin_clz(hits);
                    hits &= ~(0x80000000u >> j);
                    const int u = j >> 4, rt = (j >> 2) & 3, i = j & 3;
                    queue[qn + __popcll(hm & lt_mask)] = (unsigned short)((unsigned(16 * rt + 4 * g + i) << 10) | unsigned(c0 + MM_STEP * u + rc - seg_lo));
                }
                qn += __popcll(hm);
            }
            __builtin_amdgcn_wave_barrier();
            while (qn >= 64) {
                sign_stage(qn - 64, 64);
                qn -= 64;
            }
        }
    }
    TSC_STAMP(3);  // screened (evaluation batches of 64 included)
    if (qn > 0) sign_stage(0, qn);
    if (qe > 0) exact_stage(0, qe);
    TSC_STAMP(4);  // the rest evaluated
    if (lane == 0) {
        count_add(counters, unsigned(grp), CNT_FORMED, n_eval);
        count_add(counters, unsigned(grp), CNT_EXACT, n_exact);
        count_add(counters, unsigned(grp), CNT_SCREENED, n_screened);
    }
}

// grid: ONE WAVEFRONT per workgroup = (a group of 64 rows, a column segment).  (Four per workgroup, as in the other pair kernels, kept a
// workgroup's slot until its longest item was through -- in a sorted layout the near-duplicates sit together and one item of four may
// carry fifty evaluation batches: measured, 53 % of the wavefront slots in use.)  With xcd_seg the groups of a segment are dealt so that
// runs of CULL_XCD_RUN consecutive groups stay on one XCD (cull.hpp: k_rmsd_sieve_sorted).
template <bool F32>
inline __global__ __launch_bounds__(64, TSC_CMM_OCC) void k_rmsd_sieve_sorted_mm(const double *__restrict__ heavy, const int32_t *__restrict__ act,
                                                                           const double *__restrict__ Gall, const int32_t *__restrict__ cend,
                                                                           int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                                                           const PruneState *__restrict__ st, SieveArgs a, CullArgs ca, CullMmArgs cm, int n_groups,
                                                                           int n_seg) {
    const int groups = n_groups;   // workgroups per segment
    int seg, grp;
    if (ca.xcd_seg) {
        const int x = int(blockIdx.x & 7u);
        const long long j = (long long)(blockIdx.x >> 3);
        const int runs_per_xcd = (((groups + CULL_XCD_RUN - 1) / CULL_XCD_RUN) + 7) / 8;
        const long long per_seg = (long long)runs_per_xcd * CULL_XCD_RUN;
        seg = int(j / per_seg);
        const int rem = int(j - (long long)seg * per_seg);
        grp = ((rem / CULL_XCD_RUN) * 8 + x) * CULL_XCD_RUN + rem % CULL_XCD_RUN;
    } else {
        seg = int(blockIdx.x / unsigned(groups)), grp = int(blockIdx.x - unsigned(seg) * unsigned(groups));
    }
    if (seg >= n_seg || grp >= groups) return;
    // (several ranks: runs of tile_block / 4 consecutive groups of the sorted layout stay on one rank, as cull.hpp deals its tiles)
    const int tbg = max(1, ca.tile_block / 4);
    const int g64 = a.tile_stride <= 1 ? grp : ((grp / tbg) * a.tile_stride + a.tile_begin) * tbg + grp % tbg;
    sieve_item_sorted_mm<F32>(heavy, act, Gall, cend, best, counters, st, a, ca, cm, g64, seg);
}

}  // namespace tsc
