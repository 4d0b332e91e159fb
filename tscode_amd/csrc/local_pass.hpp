// local_pass.hpp -- a whole pass of prune_conformers_rmsd in ONE launch, for passes whose chunks are short.
//
// A pass only ever compares structures of the same chunk (tscode/rmsd_pruning.py:136-147), so when the longest chunk of
// a pass holds at most LP_MAX_ROWS structures a workgroup can own a chunk end to end: it ranks the chunk's active
// structures (the mask bits, by ballot), finds every row's stop column in the cache view, screens and evaluates the
// pairs, and applies the verdicts (mask, cache keys, scan counts, statistics) -- the work of k_open_rows, k_rmsd_sieve and the
// tile-wise apply without the kernel boundary in between.  The host takes this path for passes
// whose chunks are a few row tiles long and for small ensembles (see tsc_prune_pass_local), where a pass is bound by
// launch latency, not by work.
//
// Long chunks are shared: block j of the NB blocks of a chunk takes the row tiles j, j + NB, ... (rows of a pass are
// independent, :92,101-113); every block of a chunk repeats the cheap ranking, only its own rows go further.
// Same verdict functions as the big kernel (sieve.hpp): fp32 descriptor screen -> H -> quartic tests -> explicit rotation.
#pragma once
#include "rmsd.hpp"
#include "sieve.hpp"

namespace tsc {

constexpr int LP_MAX_ROWS = 2048;  // longest chunk (structures) the chunk-local kernel takes
constexpr int LP_WAVES = 4;
constexpr int LP_THREADS = LP_WAVES * 64;
constexpr int LP_TI = 16;                       // rows per work item
#ifndef TSC_LP_TPB
#define TSC_LP_TPB 4
#endif
// row tiles one block is sized for: one per wavefront (measured against two: a wavefront walks its tiles one after the other, and
// a pass of this kernel is as long as its slowest wavefront -- C3's k = 500 pass 44 -> 39 us, a 20 000-structure call 0.71 -> 0.64 ms)
constexpr int LP_TILES_PER_BLOCK = TSC_LP_TPB;
constexpr int LP_QCAP = LP_TI * 64 + 64;
constexpr int LP_WORDS = LP_MAX_ROWS / 64;
constexpr int LP_TICKET_GROUPS = 16;  // two-level ticket: same-address atomics serialise (~12 ns each)
#ifndef TSC_LP_OCC
#define TSC_LP_OCC 4
#endif
constexpr int LP_OCCUPANCY = TSC_LP_OCC;  // workgroups per CU the register allocation aims at (unbounded, the kernel takes 255 VGPRs: one)

struct LocalPassArgs {
    int h;
    int use_cache;
    int nb_regular;  // blocks per chunk for chunks 0 .. k-2
    int nb_last;     // blocks of the last chunk (it takes the remainder, :141-142)
    int c_lo;        // first chunk of this launch and how many regular chunks (not the last one of the pass) follow it: 0 and k - 1,
    int n_reg;       // unless the pass is partitioned over ranks (rmsd.hpp, k_pass_merge): then this rank's chunks; blocks beyond
                     // n_reg * nb_regular belong to the last chunk of the pass
    unsigned long long *exch;  // rank-partitioned pass (else null): removed rows are noted here and applied by k_pass_merge
    double thr, maxdev_thr, half_h_thr2, two_thr2, desc_limit;
    const unsigned *dmax_bits;
};

struct LocalTickets {  // zeroed by k_init_run and again by the wavefront that closes a pass
    unsigned group[LP_TICKET_GROUPS][32];  // one counter per 128-byte line
    unsigned top;
    unsigned pad[31];
};

__device__ inline unsigned long long lds_extract64(const unsigned long long *bits, int start) {
    const int w = start >> 6, sh = start & 63;
    const unsigned long long lo = bits[w] >> sh;
    const unsigned long long hi = sh ? (bits[w + 1] << (64 - sh)) : 0ull;
    return lo | hi;
}

inline __global__ __launch_bounds__(LP_THREADS, LP_OCCUPANCY) void k_pass_chunks(PassGeom g, LocalPassArgs a, PruneState *__restrict__ st, uint8_t *__restrict__ mask,
                                                             unsigned long long *__restrict__ bits, int bit_words,
                                                             const unsigned long long *__restrict__ dbit, const double *__restrict__ heavy,
                                                             const double *__restrict__ Gall, const float *__restrict__ D,
                                                             CacheViews cv, PassCounters *__restrict__ cnt, int32_t *__restrict__ bsum, int block_items,
                                                             StepCtx sc, StepArgs next, LocalTickets *__restrict__ tickets) {
    __shared__ unsigned long long s_mb[LP_WORDS + 2], s_db[LP_WORDS + 2];
    __shared__ unsigned short s_wpre[LP_WORDS + 2];
    __shared__ unsigned short s_act[LP_MAX_ROWS], s_cend[LP_MAX_ROWS];
    __shared__ int s_best[LP_MAX_ROWS];
    __shared__ unsigned short s_queue[LP_WAVES][LP_QCAP], s_exq[LP_WAVES][128];
    __shared__ __attribute__((aligned(16))) float s_rowdesc[LP_WAVES][LP_TI * DW];
    __shared__ unsigned long long s_stat[8];  // block totals of the five statistics
    __shared__ int s_A, s_anydb, s_last;
    static_assert(DW == 16 && LP_TI * DW == 256, "row staging: 4 rows x 16 components per 64 lanes");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool pass_on = st->pass_on != 0;
    // the pass reads one bit copy of the mask and clears the rows it removes in the other (every row of a pass sees the mask
    // as it was when the pass began, rmsd_pruning.py:151-157, whatever order the workgroups run in)
    const unsigned long long *mbit = bits + size_t(st->bitsel) * bit_words;
    unsigned long long *mbit_next = bits + size_t(st->bitsel ^ 1) * bit_words;
    // which chunk, and which share of its row tiles
    int c, j, nb;
    {
        const int reg_blocks = a.n_reg * a.nb_regular;
        if (int(blockIdx.x) < reg_blocks) {
            c = int(blockIdx.x) / a.nb_regular, j = int(blockIdx.x) - c * a.nb_regular, nb = a.nb_regular;
            c += a.c_lo;
        } else {
            c = g.k - 1, j = int(blockIdx.x) - reg_blocks, nb = a.nb_last;
        }
    }
    const int first = c * g.cs;                                  // :140
    const int L = ((c == g.k - 1) ? g.n : first + g.cs) - first;  // :141-144  (<= LP_MAX_ROWS, checked by the host)
    const int nw = (L + 63) >> 6;
    if (tid < 8) s_stat[tid] = 0;
    unsigned long long n_screened = 0, n_eval = 0, n_exact = 0, n_evaluated = 0, n_removed = 0;
    int my_screened = 0;  // lanes 0..15 of a wavefront: pairs of its tiles' rows that went through the screen

    if (pass_on) {
        // ---- 1. the chunk's mask and cache view as bits; ranks of the active structures.  The other bit copy may lag one pass
        // behind (a superset of this one): and-ing this chunk's words into it brings it up to date, and commutes with the
        // bits other workgroups -- of this very chunk when it is shared -- clear there in step 4
        if (j == 0 && !a.exch)
            for (int w = (first >> 6) + tid; w <= ((first + L - 1) >> 6); w += LP_THREADS) atomicAnd(&mbit_next[w], mbit[w]);
        for (int w = tid; w < nw; w += LP_THREADS) {
            unsigned long long m = extract64(mbit, int64_t(first) + 64 * w);
            unsigned long long v = a.use_cache ? extract64(dbit, int64_t(first) + 64 * w) : 0ull;
            const int rem = L - 64 * w;
            if (rem < 64) m &= (1ull << rem) - 1ull, v &= (1ull << rem) - 1ull;
            s_mb[w] = m;
            s_db[w] = v;
        }
        if (tid < 2) s_mb[nw + tid] = 0, s_db[nw + tid] = 0;
        __syncthreads();
        if (tid == 0) {
            int run = 0, any = 0;
            for (int w = 0; w < nw; ++w) {
                s_wpre[w] = (unsigned short)run;
                run += __popcll(s_mb[w]);
                any |= (s_db[w] != 0ull) ? 1 : 0;
            }
            s_wpre[nw] = (unsigned short)run;
            s_A = run, s_anydb = any;
        }
        __syncthreads();
        const int A = s_A;
        auto rank_of = [&](int t) { return int(s_wpre[t >> 6]) + __popcll(s_mb[t >> 6] & ((t & 63) ? (~0ull >> (64 - (t & 63))) : 0ull)); };
        for (int t = tid; t < L; t += LP_THREADS)
            if ((s_mb[t >> 6] >> (t & 63)) & 1ull) s_act[rank_of(t)] = (unsigned short)t;
        __syncthreads();
        // ---- 2. stop column of this block's rows (first active column whose key is cached, :65-67), best = none
        const int n_tiles = (A + LP_TI - 1) / LP_TI;
        const int my_tiles = n_tiles > j ? (n_tiles - j + nb - 1) / nb : 0;
        for (int idx = tid; idx < my_tiles * LP_TI; idx += LP_THREADS) {
            const int r = (j + (idx / LP_TI) * nb) * LP_TI + (idx % LP_TI);
            if (r >= A) continue;
            const int t = s_act[r];
            int found = L;
            if (s_anydb) {
                const int len = L - t - 1;  // candidate deltas d = 1 .. len
                for (int d0 = 1; d0 <= len; d0 += 64) {
                    unsigned long long w = lds_extract64(s_mb, t + d0) & lds_extract64(s_db, d0);
                    const int rem = len - d0 + 1;
                    if (rem < 64) w &= (1ull << rem) - 1ull;
                    if (w) {
                        found = t + d0 + __ffsll((long long)w) - 1;
                        break;
                    }
                }
            }
            s_cend[r] = (unsigned short)(found >= L ? A : rank_of(found));
            s_best[r] = INT_MAX;
        }
        __syncthreads();

        // ---- 3. pairs: one wavefront per (16 rows) x (all their columns), 64 columns at a time
        const int h3 = a.h * 3;
        const float limit32 = screen_limit32(__uint_as_float(*a.dmax_bits), a.desc_limit);
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        unsigned short *queue = s_queue[wid], *exq = s_exq[wid];
        float *rowdesc = s_rowdesc[wid];
        int tile_no = 0;
        for (int rt = j; rt < n_tiles; rt += nb, ++tile_no) {
            if ((tile_no % LP_WAVES) != wid) continue;  // this block's tiles are dealt round-robin to its wavefronts
            const int r0 = rt * LP_TI;
            const int nrows = min(LP_TI, A - r0);
            const int my_cend = lane < nrows ? int(s_cend[r0 + lane]) : 0;
            const bool live0 = lane < nrows && my_cend > r0 + lane + 1;
            unsigned alive = unsigned(__builtin_amdgcn_ballot_w64(live0));
            if (!alive) continue;
            int cmax = live0 ? my_cend : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off));
            cmax = __builtin_amdgcn_readfirstlane(cmax);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = min(r0 + 4 * q + (lane >> 4), A - 1);
                rowdesc[64 * q + lane] = D[int64_t(first + s_act[rr]) * DW + (lane & 15)];
            }
            __builtin_amdgcn_wave_barrier();
            int qn = 0, qe = 0;

            auto decode = [&](unsigned e, int &t, int &col, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
                t = int(e >> 12);
                col = int(e & 0xfffu);
                const int64_t i = first + s_act[r0 + t], jj = first + s_act[col];
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
                bool cand = false, sim = false;
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
                }
                note_similar(sim, t, col);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(cand);
                if (m) {
                    if (cand) exq[qe + __popcll(m & lt_mask)] = (unsigned short)e;
                    qe += __popcll(m);
                }
                n_eval += cnt;
                n_exact += __popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (qe >= 64) {
                    exact_stage(qe - 64, 64);
                    qe -= 64;
                }
            };

            // column tiles are double-buffered: the gather of the next tile is in flight while this one is screened
            auto load_cols = [&](int c0, f32x2 (&dst)[KD]) __attribute__((always_inline)) {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(D + int64_t(first + s_act[min(c0 + lane, A - 1)]) * DW);
#pragma unroll
                for (int k = 0; k < KD / 2; ++k) {
                    const f32x4 v = src[k];
                    dst[2 * k] = f32x2{v.x, v.y};
                    dst[2 * k + 1] = f32x2{v.z, v.w};
                }
            };
            f32x2 dq[KD], dq_next[KD];
            load_cols((r0 + 1) & ~63, dq_next);
            for (int c0 = (r0 + 1) & ~63; c0 < cmax && alive; c0 += 64) {
                {
                    const int col = c0 + lane;
#pragma unroll
                    for (int k = 0; k < KD; ++k) dq[k] = dq_next[k];
                    if (c0 + 64 < cmax) load_cols(c0 + 64, dq_next);
                    const bool here = lane < nrows && ((alive >> lane) & 1u) && my_cend > c0 && r0 + lane < c0 + 63;
                    unsigned rows = unsigned(__builtin_amdgcn_ballot_w64(here));
                    // (columns of this tile inside every live row's range: counted once per tile by the rows' own lanes, not by
                    // seven scalar instructions per row -- sieve.hpp)
                    my_screened += here ? max(0, min(my_cend, c0 + 64) - max(r0 + lane + 1, c0)) : 0;
                    while (rows) {
                        const int t = __ffs(rows) - 1;
                        rows &= rows - 1;
                        const int r = r0 + t;
                        const int ce = __builtin_amdgcn_readlane(my_cend, t);
                        const f32x2 *dr = reinterpret_cast<const f32x2 *>(rowdesc + t * DW);
                        f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
                        for (int k = 0; k < KD; ++k) {
                            const f32x2 d = dr[k] - dq[k];
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
                while (qn >= 64) {
                    sign_stage(qn - 64, 64);
                    qn -= 64;
                }
            }
            if (qn > 0) sign_stage(0, qn);
            if (qe > 0) exact_stage(0, qe);
        }
        for (int off = 8; off > 0; off >>= 1) my_screened += __shfl_xor(my_screened, off);
        n_screened = (unsigned long long)my_screened;
        __syncthreads();

        // ---- 4. apply this block's rows: mask, one cache key per removed row (:69-73), scan counts, evaluation count
        for (int rt = j; rt < n_tiles; rt += nb) {
            const int r = rt * LP_TI + lane;  // one wavefront per tile keeps the ballots simple; tiles dealt as above
            if (((rt - j) / nb) % LP_WAVES != wid) continue;
            bool removed = false;
            int my_block = -1, delta = 0;
            unsigned long long ev = 0;
            if (lane < LP_TI && r < A) {
                const int b = s_best[r];
                if (b != INT_MAX) {
                    const int t_r = s_act[r], t_b = s_act[b];
                    if (a.exch) {
                        atomicOr(&a.exch[(first + t_r) >> 6], 1ull << ((first + t_r) & 63));
                    } else {
                        mask[first + t_r] = 0;
                        atomicAnd(&mbit_next[(first + t_r) >> 6], ~(1ull << ((first + t_r) & 63)));
                        my_block = (first + t_r) / block_items;
                    }
                    delta = t_b - t_r;
                    removed = true;
                    ev = (unsigned long long)(b - r);  // columns r+1 .. b were evaluated
                } else {
                    ev = (unsigned long long)(int(s_cend[r]) - r - 1);  // every active column before the stop column
                }
            }
            for (unsigned long long left = __builtin_amdgcn_ballot_w64(removed && my_block >= 0); left;) {
                const int l = __ffsll((long long)left) - 1;
                const int blk = __shfl(my_block, l);
                const unsigned long long same = __builtin_amdgcn_ballot_w64(removed && my_block == blk);
                if (lane == l) atomicSub(&bsum[blk], __popcll(same));
                left &= ~same;
            }
            const int n_rm = __popcll(__builtin_amdgcn_ballot_w64(removed));
            views_insert_wave(cv, removed, first, first + delta);  // the cache keys (:69-73), where later passes will look for them
            for (int off = 32; off > 0; off >>= 1) ev += __shfl_down(ev, off);
            n_evaluated += ev, n_removed += (unsigned long long)n_rm;
        }
        // ---- 5. statistics: one set of atomics per block
        if (lane == 0) {
            if (n_eval) atomicAdd(&s_stat[CNT_FORMED], n_eval);
            if (n_exact) atomicAdd(&s_stat[CNT_EXACT], n_exact);
            if (n_screened) atomicAdd(&s_stat[CNT_SCREENED], n_screened);
            if (n_evaluated) atomicAdd(&s_stat[CNT_EVALUATED], n_evaluated);
            if (n_removed) atomicAdd(&s_stat[CNT_REMOVED], n_removed);
        }
        __syncthreads();
        if (tid < 5) count_add(cnt, blockIdx.x, tid, s_stat[tid]);
    }

    // ---- 6. the last block closes this pass and opens the next (as k_apply_pass does); two-level ticket
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
    if (s_last && tid < 64) pass_step_wave(sc, next);  // (zeroes the tickets as well)
}

}  // namespace tsc
