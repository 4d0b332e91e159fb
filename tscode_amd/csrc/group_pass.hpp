// group_pass.hpp -- the FIRST passes of a prune run in one launch, where their chunks are short.
//
// A pass of prune_conformers_rmsd only compares structures of the same chunk (tscode/rmsd_pruning.py:136-147), and the chunks of the first
// passes of a run are a few dozen to a few hundred structures long.  Each of those passes is, on its own, a chain of dependent round trips
// behind a kernel boundary (at 57 000 structures the five passes k = 2000 ... 100 take 40 - 50 us each for 1 - 6 M pairs).
// Here a workgroup owns one chunk of the LAST pass of the group and runs every pass of the group on what that chunk depends on:
//
//     region of pass G-1  = the own chunk
//     region of pass q    = the chunks of pass q that intersect the region of pass q + 1        (whole chunks: a chunk is the unit a pass decides on)
//
// so the region grows by less than a chunk of pass q on either side per step back, and the structures outside the own chunk (the halo) are
// decided redundantly by the neighbouring workgroups as well -- the verdicts are a pure function of the mask and the cache a pass starts
// from, so every workgroup that decides a structure decides it alike.  A workgroup keeps the mask of its region and the cache views of the
// later passes of the group in LDS, and COMMITS (mask bytes, bit copy, scan counts, cache keys for the passes behind the group, statistics)
// only the rows of its own chunk.  No grid barrier, no kernel boundary between the passes.
//
// What cannot be known inside: the gate of the reference (`k == 1 or 20 k < count_nonzero(mask)`, rmsd_pruning.py:192) for the second and
// later passes of the group needs the GLOBAL number of survivors of the pass before.  The group runs them as open; the per-pass records it
// leaves hold the true counts, and the host -- which reads them at the end of the run anyway -- repeats the run pass by pass should a gate
// turn out to have been closed (more than half of an ensemble would have to go in its first passes; prune.hip, prune_run).
// The group is only ever the first passes of a run: the mask is all ones, the cache empty, the first gate known on the host.
//
// The region's descriptors are staged in LDS once (one contiguous, coalesced read: a region is a range of structure indices), component pairs
// side by side per structure position -- a workgroup has two wavefronts per SIMD, far too few to hide a gather from memory per column tile (a
// first version that read them from memory took 480 us for C3's five passes; the separate passes take 218).
#pragma once
#include "local_pass.hpp"

namespace tsc {

constexpr int LPG_MAX = 6;                  // passes in one launch at most
constexpr int LPG_ROWS = 1792;              // structures of a workgroup's region at most (own chunk + halos; the host checks): their descriptors
                                            // (64 B each) live in LDS beside 46 KB of queues and rank tables
constexpr int LPG_WORDS = LPG_ROWS / 64;
#ifndef TSC_LPG_WAVES
#define TSC_LPG_WAVES 8
#endif
constexpr int LPG_WAVES = TSC_LPG_WAVES;    // a workgroup's row tiles are dealt to its wavefronts: the longest chunk's tiles bound a pass
constexpr int LPG_THREADS = LPG_WAVES * 64;

struct GroupArgs {
    int n, G;
    int k[LPG_MAX], cs[LPG_MAX];  // chunks and chunk size (n // k, rmsd_pruning.py:136) of every pass of the group, in schedule order
    int slot[LPG_MAX];            // their schedule slots (records)
    int h, use_cache, algo_tag;
    double thr, maxdev_thr, half_h_thr2, two_thr2, desc_limit;
    const unsigned *dmax_bits;
    unsigned long long *gstat;    // [LPG_MAX][8] statistics of the passes, summed over the workgroups (zero on entry, zeroed again on the way out)
};

__host__ __device__ inline void group_chunk_bounds(int n, int k, int cs, int i, int &first, int &last) {
    int c = i / cs;
    if (c >= k) c = k - 1;
    first = c * cs;                              // :140
    last = (c == k - 1) ? n : first + cs;        // :141-144
}
// [lo, hi): the structures pass q of the group has to decide so that own chunk c of the last pass comes out right
__host__ __device__ inline void group_region(const GroupArgs &a, int c, int q, int &lo, int &hi) {
    const int L = a.G - 1;
    lo = c * a.cs[L];
    hi = (c == a.k[L] - 1) ? a.n : lo + a.cs[L];
    for (int p = L - 1; p >= q; --p) {
        int f, l, f2, l2;
        group_chunk_bounds(a.n, a.k[p], a.cs[p], lo, f, l);
        group_chunk_bounds(a.n, a.k[p], a.cs[p], hi - 1, f2, l2);
        lo = f, hi = l2;
    }
}

// Closes every pass of the group and opens the pass behind it: what pass_step_wave (rmsd.hpp) does for one pass.  ONE wavefront.
__device__ inline void group_step_wave(const GroupArgs &a, const StepCtx &sc, const StepArgs &sa) {
    PruneState *st = sc.st;
    const int lane = threadIdx.x & 63;
    unsigned long long v[5] = {0, 0, 0, 0, 0};
    if (lane < a.G) {
#pragma unroll
        for (int w = 0; w < 5; ++w) v[w] = __hip_atomic_load(&a.gstat[lane * 8 + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int n0 = __hip_atomic_load(&st->n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int bitsel = __hip_atomic_load(&st->bitsel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // survivors entering pass q = n0 - removed by the passes before it
    long long before = 0;
    {
        long long incl = (long long)v[CNT_REMOVED];
        for (int off = 1; off < 8; off <<= 1) {
            const long long t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        before = incl - (long long)v[CNT_REMOVED];
    }
    const long long total_removed = __shfl(before + (long long)v[CNT_REMOVED], a.G - 1);
    if (lane < a.G) {
        PassRecord &r = sc.rec[a.slot[lane]];
        r.k = a.k[lane], r.n_before = (long long)n0 - before, r.n_after = (long long)n0 - before - (long long)v[CNT_REMOVED];
        r.formed = (long long)v[CNT_FORMED], r.exact = (long long)v[CNT_EXACT], r.screened = (long long)v[CNT_SCREENED];
        r.evaluated = (long long)v[CNT_EVALUATED], r.removed = (long long)v[CNT_REMOVED];
        r.on = 1, r.algo = a.algo_tag;
    }
    // the scan blocks' prefix from their counts (the workgroups kept the counts current)
    int run = 0;
    for (int b0 = 0; b0 < sc.n_blocks; b0 += 64) {
        const int b = b0 + lane;
        const int c = b < sc.n_blocks ? __hip_atomic_load(&sc.bsum[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (b < sc.n_blocks) sc.boff[b] = run + incl - c;
        run += __shfl(incl, 63);
    }
    if (lane == 0) {
        sc.boff[sc.n_blocks] = run;
        const int n_active = n0 - int(total_removed);
        st->n_active = n_active;
        st->bitsel = bitsel ^ 1;  // the copy the group cleared its removed rows in is the current one now
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
    for (int e = lane; e < CNT_BUCKETS * 8; e += 64) sc.cnt->w[e >> 3][e & 7] = 0;
    for (int e = lane; e < sc.ticket_lines; e += 64) sc.tickets[32 * e] = 0;
    for (int e = lane; e < LPG_MAX * 8; e += 64) a.gstat[e] = 0;
}

inline __global__ __launch_bounds__(LPG_THREADS) void k_pass_group(GroupArgs a, PruneState *__restrict__ st, uint8_t *__restrict__ mask,
                                                                    unsigned long long *__restrict__ bits, int bit_words, const double *__restrict__ heavy,
                                                                    const double *__restrict__ Gall, const float *__restrict__ D, CacheViews cv,
                                                                    int32_t *__restrict__ bsum, int block_items, StepCtx sc, StepArgs next,
                                                                    LocalTickets *__restrict__ tickets, int cap) {
    __shared__ unsigned long long s_mb[LPG_WORDS + 2], s_db[LPG_MAX][LPG_WORDS + 2];
    __shared__ unsigned short s_wpre[LPG_WORDS + 2];
    __shared__ unsigned short s_act[LPG_ROWS], s_cend[LPG_ROWS];
    __shared__ int s_best[LPG_ROWS];
    __shared__ unsigned short s_queue[LPG_WAVES][LP_QCAP], s_exq[LPG_WAVES][128];
    __shared__ __attribute__((aligned(16))) float s_rowdesc[LPG_WAVES][LP_TI * DW];
    extern __shared__ __attribute__((aligned(16))) f32x2 s_desc[];  // [KD][cap]: component pair u of the structure at region offset t is s_desc[u * cap + t]
    __shared__ unsigned long long s_stat[LPG_MAX][8];
    __shared__ int s_A, s_last;
    static_assert(DW == 16 && LP_TI * DW == 256, "row staging: 4 rows x 16 components per 64 lanes");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long *mbit = bits + size_t(st->bitsel) * bit_words;
    unsigned long long *mbit_next = bits + size_t(st->bitsel ^ 1) * bit_words;
    const int c = blockIdx.x;  // own chunk of the last pass of the group
    int own_lo, own_hi, reg_lo, reg_hi;
    group_region(a, c, a.G - 1, own_lo, own_hi);
    group_region(a, c, 0, reg_lo, reg_hi);
    const int reg_len = reg_hi - reg_lo;  // <= LPG_ROWS (checked by the host for every chunk)
    const int nw = (reg_len + 63) >> 6;

    // the region's mask as bits; empty views of the group's later passes (the run has made no key yet)
    for (int w = tid; w < nw + 2; w += LPG_THREADS) {
        unsigned long long m = 0;
        if (w < nw) {
            m = extract64(mbit, int64_t(reg_lo) + 64 * w);
            const int rem = reg_len - 64 * w;
            if (rem < 64) m &= (1ull << rem) - 1ull;
        }
        s_mb[w] = m;
        for (int q = 0; q < LPG_MAX; ++q) s_db[q][w] = 0;
    }
    if (tid < LPG_MAX * 8) (&s_stat[0][0])[tid] = 0;
    {   // the region's descriptors: float4 e = components 4 (e & 3) .. + 3 of structure reg_lo + (e >> 2)
        const f32x4 *src = reinterpret_cast<const f32x4 *>(D + int64_t(reg_lo) * DW);
        for (int e = tid; e < reg_len * 4; e += LPG_THREADS) {
            const f32x4 v = src[e];
            const int t = e >> 2, u = 2 * (e & 3);
            s_desc[u * cap + t] = f32x2{v.x, v.y};
            s_desc[(u + 1) * cap + t] = f32x2{v.z, v.w};
        }
    }
    // the other bit copy may lag behind this one (a superset): the own chunk's words brought up to date (commutes with the neighbours' and-ing
    // of the words they share, and with the bits cleared below)
    for (int w = (own_lo >> 6) + tid; w <= ((own_hi - 1) >> 6); w += LPG_THREADS) atomicAnd(&mbit_next[w], mbit[w]);
    __syncthreads();

    const int h3 = a.h * 3;
    const float limit32 = screen_limit32(__uint_as_float(*a.dmax_bits), a.desc_limit);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned short *queue = s_queue[wid], *exq = s_exq[wid];
    float *rowdesc = s_rowdesc[wid];
    auto is_own = [&](int t) { return reg_lo + t >= own_lo && reg_lo + t < own_hi; };

    for (int q = 0; q < a.G; ++q) {
        const int kq = a.k[q], csq = a.cs[q];
        int p_lo, p_hi;
        group_region(a, c, q, p_lo, p_hi);
        // ---- 1. ranks of the region's active structures
        if (tid == 0) {
            int run = 0;
            for (int w = 0; w < nw; ++w) {
                s_wpre[w] = (unsigned short)run;
                run += __popcll(s_mb[w]);
            }
            s_wpre[nw] = s_wpre[nw + 1] = (unsigned short)run;
            s_A = run;
        }
        __syncthreads();
        const int A = s_A;
        auto rank_of = [&](int t) { return int(s_wpre[t >> 6]) + __popcll(s_mb[t >> 6] & ((t & 63) ? (~0ull >> (64 - (t & 63))) : 0ull)); };
        for (int t = tid; t < reg_len; t += LPG_THREADS)
            if ((s_mb[t >> 6] >> (t & 63)) & 1ull) s_act[rank_of(t)] = (unsigned short)t;
        __syncthreads();
        const int ra = rank_of(p_lo - reg_lo), rb = rank_of(p_hi - reg_lo);  // the rows of this pass: ranks [ra, rb)
        // ---- 2. stop column of every row (first active column whose key is cached, :65-67; else the end of its chunk), best = none
        for (int r = ra + tid; r < rb; r += LPG_THREADS) {
            const int t = s_act[r];
            int f, l;
            group_chunk_bounds(a.n, kq, csq, reg_lo + t, f, l);
            f -= reg_lo, l -= reg_lo;
            int found = l;
            if (a.use_cache && q > 0) {
                const int len = l - t - 1;  // candidate deltas d = 1 .. len
                for (int d0 = 1; d0 <= len; d0 += 64) {
                    unsigned long long w = lds_extract64(s_mb, t + d0) & lds_extract64(s_db[q], f + d0);
                    const int rem = len - d0 + 1;
                    if (rem < 64) w &= (1ull << rem) - 1ull;
                    if (w) {
                        found = t + d0 + __ffsll((long long)w) - 1;
                        break;
                    }
                }
            }
            s_cend[r] = (unsigned short)rank_of(found);
            s_best[r] = INT_MAX;
        }
        __syncthreads();

        // ---- 3. pairs: one wavefront per (16 rows) x (all their columns), 64 columns at a time (local_pass.hpp)
        unsigned long long n_eval = 0, n_exact = 0, n_evaluated = 0, n_removed = 0;
        int my_screened = 0;
        const int n_tiles = (rb - ra + LP_TI - 1) / LP_TI;
        for (int rt = wid; rt < n_tiles; rt += LPG_WAVES) {
            const int r0 = ra + rt * LP_TI;
            const int nrows = min(LP_TI, rb - r0);
            const int my_cend = lane < nrows ? int(s_cend[r0 + lane]) : 0;
            const bool own_row = lane < nrows && is_own(s_act[r0 + lane]);
            const bool live0 = lane < nrows && my_cend > r0 + lane + 1;
            unsigned alive = unsigned(__builtin_amdgcn_ballot_w64(live0));
            if (!alive) continue;
            const unsigned own_rows = unsigned(__builtin_amdgcn_ballot_w64(own_row));
            int cmax = live0 ? my_cend : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
            cmax = __builtin_amdgcn_readfirstlane(cmax);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = min(r0 + 4 * u + (lane >> 4), A - 1);
                const float *comp = reinterpret_cast<const float *>(s_desc + ((lane & 15) >> 1) * cap + s_act[rr]);
                rowdesc[64 * u + lane] = comp[lane & 1];
            }
            __builtin_amdgcn_wave_barrier();
            int qn = 0, qe = 0;

            auto decode = [&](unsigned e, int &t, int &col, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
                t = int(e >> 12);
                col = int(e & 0xfffu);
                const int64_t i = reg_lo + s_act[r0 + t], jj = reg_lo + s_act[col];
                pp = heavy + i * h3, pq = heavy + jj * h3;
                Gi = Gall[i], Gj = Gall[jj];
            };
            auto note_similar = [&](bool sim, int t, int col) __attribute__((always_inline)) {
                if (sim) atomicMin(&s_best[r0 + t], col);
                unsigned long long sm = __builtin_amdgcn_ballot_w64(sim);
                while (sm) {  // rows that found a similar column stop being screened (the reference returns there, :75-77)
                    const int l = __ffsll((long long)sm) - 1;
                    sm &= sm - 1;
                    alive &= ~(1u << __builtin_amdgcn_readlane(t, l));
                }
            };
            auto exact_stage = [&](int base, int cnt) __attribute__((always_inline)) {
                int lpp = 64;
                while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
                const int grp = lane / lpp, sub = lane - grp * lpp;
                bool sim = false;
                int t = 0, col = 0;
                if (grp < cnt) {
                    const double *pp, *pq;
                    double Gi, Gj, H[9], rm, md;
                    decode(exq[base + grp], t, col, pp, pq, Gi, Gj);
                    pair_H(pp, pq, a.h, sub, lpp, H);
                    exact_rmsd_maxdev(pp, pq, a.h, H, Gi, Gj, rm, md, sub, lpp);
                    sim = sub == 0 && rm < a.thr && md < a.maxdev_thr;  // rmsd_pruning.py:75
                }
                note_similar(sim, t, col);
                __builtin_amdgcn_wave_barrier();
            };
            auto sign_stage = [&](int base, int cnt) __attribute__((always_inline)) {
                int lpp = 64;
                while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
                const int grp = lane / lpp, sub = lane - grp * lpp;
                bool cand = false, sim = false, counted = false;
                unsigned e = 0;
                int t = 0, col = 0;
                if (grp < cnt) {
                    e = queue[base + grp];
                    const double *pp, *pq;
                    double Gi, Gj, H[9];
                    decode(e, t, col, pp, pq, Gi, Gj);
                    pair_H(pp, pq, a.h, sub, lpp, H);
                    const int verdict = pair_verdict(H, 0.5 * (Gi + Gj), a.half_h_thr2, a.two_thr2, a.h);
                    cand = sub == 0 && verdict == PAIR_UNDECIDED;
                    sim = sub == 0 && verdict == PAIR_SIMILAR;
                    counted = sub == 0 && ((own_rows >> t) & 1u);  // (statistics: the own chunk's rows only -- the halo's belong to the neighbours)
                }
                note_similar(sim, t, col);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
                if (m) {
                    if (cand) exq[qe + __popcll(m & lt_mask)] = (unsigned short)e;
                    qe += __popcll(m);
                }
                n_eval += __popcll(__builtin_amdgcn_ballot_w64(counted));
                n_exact += __popcll(__builtin_amdgcn_ballot_w64(cand && counted));
                __builtin_amdgcn_wave_barrier();
                if (qe >= 64) {
                    exact_stage(qe - 64, 64);
                    qe -= 64;
                }
            };

            auto load_cols = [&](int c0, f32x2 (&dst)[KD]) __attribute__((always_inline)) {
                const int t = s_act[min(c0 + lane, A - 1)];
#pragma unroll
                for (int u = 0; u < KD; ++u) dst[u] = s_desc[u * cap + t];
            };
            f32x2 dq[KD], dq_next[KD];
            load_cols((r0 + 1) & ~63, dq_next);
            for (int c0 = (r0 + 1) & ~63; c0 < cmax && alive; c0 += 64) {
                {
                    const int col = c0 + lane;
#pragma unroll
                    for (int u = 0; u < KD; ++u) dq[u] = dq_next[u];
                    if (c0 + 64 < cmax) load_cols(c0 + 64, dq_next);
                    const bool here = lane < nrows && ((alive >> lane) & 1u) && my_cend > c0 && r0 + lane < c0 + 63;
                    unsigned rows = unsigned(__builtin_amdgcn_ballot_w64(here));
                    my_screened += (here && own_row) ? max(0, min(my_cend, c0 + 64) - max(r0 + lane + 1, c0)) : 0;
                    while (rows) {
                        const int t = __ffs(rows) - 1;
                        rows &= rows - 1;
                        const int r = r0 + t;
                        const int ce = __builtin_amdgcn_readlane(my_cend, t);
                        const f32x2 *dr = reinterpret_cast<const f32x2 *>(rowdesc + t * DW);
                        f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
                        for (int u = 0; u < KD; ++u) {
                            const f32x2 d = dr[u] - dq[u];
                            s2 = __builtin_elementwise_fma(d, d, s2);
                        }
                        const bool pass = col > r && col < ce && !(fmaxf(s2.x, s2.y) > limit32);
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                        if (m) {
                            if (pass) queue[qn + __popcll(m & lt_mask)] = (unsigned short)((unsigned(t) << 12) | unsigned(col));
                            qn += __popcll(m);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
#ifdef TSC_DBG_GROUP_NODRAIN
                qn = 0;
#endif
                while (qn >= 64) {
                    sign_stage(qn - 64, 64);
                    qn -= 64;
                }
            }
            if (qn > 0) sign_stage(0, qn);
            if (qe > 0) exact_stage(0, qe);
        }
        for (int off = 8; off > 0; off >>= 1) my_screened += __shfl_xor(my_screened, off);
        __syncthreads();

        // ---- 4. apply: every row of the pass in the region's own mask and views; the own chunk's rows in memory as well
        for (int rt = wid; rt < n_tiles; rt += LPG_WAVES) {
            const int r = ra + rt * LP_TI + lane;
            bool removed = false, own = false;
            int my_block = -1, delta = 0, f_abs = 0;
            unsigned long long ev = 0;
            if (lane < LP_TI && r < rb) {
                const int b = s_best[r];
                const int t_r = s_act[r];
                own = is_own(t_r);
                if (b != INT_MAX) {
                    const int t_b = s_act[b];
                    int l_abs;
                    group_chunk_bounds(a.n, kq, csq, reg_lo + t_r, f_abs, l_abs);
                    delta = t_b - t_r;
                    removed = true;
                    atomicAnd(&s_mb[t_r >> 6], ~(1ull << (t_r & 63)));
                    // the key (first, first + (j - i)) of the removed row (:69-73) where a later pass of the group will look for it
                    if (a.use_cache) {
                        for (int q2 = q + 1; q2 < a.G; ++q2) {
                            const int c2 = f_abs / a.cs[q2];
                            if (c2 * a.cs[q2] != f_abs || c2 >= a.k[q2]) continue;  // not the start of one of its chunks
                            const int end2 = (c2 == a.k[q2] - 1) ? a.n : f_abs + a.cs[q2];
                            const int bkey = f_abs + delta;
                            if (bkey < end2 && bkey < reg_hi) atomicOr(&s_db[q2][(bkey - reg_lo) >> 6], 1ull << ((bkey - reg_lo) & 63));
                        }
                    }
                    if (own) {
                        const int i = reg_lo + t_r;
                        mask[i] = 0;
                        atomicAnd(&mbit_next[i >> 6], ~(1ull << (i & 63)));
                        my_block = i / block_items;
                        ev = (unsigned long long)(b - r);  // columns r+1 .. b were evaluated
                    }
                } else if (own) {
                    ev = (unsigned long long)(int(s_cend[r]) - r - 1);  // every active column before the stop column
                }
            }
            const bool commit = removed && own;
            for (unsigned long long left = __builtin_amdgcn_ballot_w64(commit); left;) {
                const int l = __ffsll((long long)left) - 1;
                const int blk = __shfl(my_block, l);
                const unsigned long long same = __builtin_amdgcn_ballot_w64(commit && my_block == blk);
                if (lane == l) atomicSub(&bsum[blk], __popcll(same));
                left &= ~same;
            }
            n_removed += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(commit));
            views_insert_wave(cv, commit, f_abs, f_abs + delta);  // ... and where the passes BEHIND the group will
            for (int off = 32; off > 0; off >>= 1) ev += __shfl_down(ev, off);
            n_evaluated += ev;
        }
        if (lane == 0) {
            if (n_eval) atomicAdd(&s_stat[q][CNT_FORMED], n_eval);
            if (n_exact) atomicAdd(&s_stat[q][CNT_EXACT], n_exact);
            if (my_screened) atomicAdd(&s_stat[q][CNT_SCREENED], (unsigned long long)my_screened);
            if (n_evaluated) atomicAdd(&s_stat[q][CNT_EVALUATED], n_evaluated);
            if (n_removed) atomicAdd(&s_stat[q][CNT_REMOVED], n_removed);
        }
        __syncthreads();  // the region's mask and views as the next pass finds them
    }

    // ---- statistics of the own chunk, per pass; the last workgroup closes the group and opens the pass behind it
    if (tid < a.G * 5) {
        const unsigned long long v = s_stat[tid / 5][tid % 5];
        if (v) atomicAdd(&a.gstat[(tid / 5) * 8 + tid % 5], v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int last = 0;
        const unsigned grp = blockIdx.x % LP_TICKET_GROUPS;
        const unsigned in_group = (gridDim.x - grp + LP_TICKET_GROUPS - 1) / LP_TICKET_GROUPS;
        if (atomicAdd(&tickets->group[grp][0], 1u) == in_group - 1) {
            const unsigned groups = min(unsigned(LP_TICKET_GROUPS), gridDim.x);
            last = (atomicAdd(&tickets->top, 1u) == groups - 1) ? 1 : 0;
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last && tid < 64) group_step_wave(a, sc, next);  // (zeroes the tickets as well)
}

}  // namespace tsc
