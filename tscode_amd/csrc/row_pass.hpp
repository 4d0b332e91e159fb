// row_pass.hpp -- a whole pass of prune_conformers_rmsd in ONE launch, for passes whose chunks are short.
//
// A pass only ever compares structures of the same chunk (tscode/rmsd_pruning.py:136-147).  On the two-launch flow a pass with
// short chunks is a chain of fixed costs -- k_open_rows (three dependent round trips + a launch), the pair kernel's prologue (two
// more), its tile and pass counters: 45 - 55 us for 1.3 - 5.7 M pairs at 57 046 structures.  Here the wavefront that OPENS a row
// tile (k_open_rows' work: which structures its 16 rows are, their stop columns -- the same code, open_tile_rows) keeps going: it
// stages the bits of the mask behind its first row, finds its columns by select on those words, screens, evaluates the candidates
// and applies its rows' verdicts.  The rows of a tile are its own, so nothing is exchanged with other wavefronts but the arrival
// at the pass's counter, and the last arrival closes the pass and opens the next (pass_step_wave).
//
// The host takes this path when no chunk of the pass is longer than "local_max_chunk" structures (rows only look inside their
// chunk, so no tile then walks more columns than that).  What the kernel costs was measured with its own time stamps
// (tools/stamps.py, C3 pass k = 1000, 3 553 tiles, median / slowest wavefront): prefix to LDS 0.3, search + select of the rows
// 1.9, descriptor + cache view + rank of the stop column 2.1, mask words + first 64 columns' descriptors 2.0, screen 3.4 (16 rows
// x 64 columns), candidates 1.9, apply 1.1 = 12.8 / 22 us -- seven dependent phases of about 2 us each, half memory round trip,
// half dependent integer code, with 3.5 wavefronts per SIMD all in the same phase.  It replaced the chunk-local kernel of rounds
// 1 - 3 (k_pass_chunks, a workgroup per chunk: 38 / 40 / 51 us for C3's k = 1000 / 500 / 200 against 36 / 39 / 47 here) and takes
// partitioned passes the same way; as the OPENER of passes with long chunks (finishing the short tiles, leaving the long ones to the
// pair kernel behind it) it lost to k_open_rows + k_rmsd_sieve on every pass of C3 and was taken out again.
//
// Same verdict functions as the other pair kernels: fp32 descriptor screen (dot-product form, screen_limit32_dot) -> H -> quartic
// tests -> explicit rotation (rmsd.hpp, sieve.hpp).
#pragma once
#include "rmsd.hpp"
#include "sieve.hpp"

namespace tsc {

constexpr int RP_WORDS = 64;         // words of the mask's bit copy a wavefront stages: 4096 positions from the word of its first row on
constexpr int RP_MAX_COLS = 1536;    // longest chunk (structures) of a pass this kernel takes: compile-time cap of "local_max_chunk"
constexpr int RP_LDS_BLOCKS = 1024;  // scan blocks whose prefix is staged in LDS (2 M structures); beyond: read from memory
constexpr int RP_QCAP = 16 * 64 + 64;
constexpr int RP_RS = 20;            // floats per row record in LDS: 16 components, the two squared norms, 2 of padding
#ifndef TSC_RP_OCC
#define TSC_RP_OCC 4
#endif

struct RowPassArgs {
    int h;
    double thr, maxdev_thr, half_h_thr2, two_thr2, desc_limit;
    const unsigned *dmax_bits;
    const double *heavy, *Gall;
    ApplyArgs ap;  // (its act / cend / best are not read: the verdicts are applied from registers)
};

__global__ __launch_bounds__(256, TSC_RP_OCC) void k_pass_rows(PassGeom g, OpenArgs oa, RowPassArgs ra, StepCtx sc, StepArgs next, const float *__restrict__ D) {
    static_assert(SCAN_BLOCK_WORDS == 32 && 64 / OPEN_LPR == 16 && DESC_WORDS == 4 * OPEN_LPR && DW == 16, "one wavefront = one row tile; a float4 of the descriptor per lane");
    static_assert(RP_MAX_COLS + 64 <= 64 * RP_WORDS && RP_MAX_COLS < 4096, "queue entries keep the column in 12 bits");
    __shared__ int s_boff[RP_LDS_BLOCKS + 1];
    __shared__ int s_state[5];
    __shared__ unsigned long long s_words[4][RP_WORDS];
    __shared__ unsigned short s_wpre[4][RP_WORDS];                // set bits of the staged words before word w
    __shared__ unsigned short s_colpos[4][RP_MAX_COLS + 64];      // position of column c of the rows being walked, relative to the first staged word
    __shared__ unsigned short s_queue[4][RP_QCAP], s_exq[4][128];
    __shared__ double s_jacobi[4][32];
    __shared__ __attribute__((aligned(16))) float s_rowdesc[4][16 * RP_RS];
    __shared__ int s_best[4][16];
    __shared__ int s_rowpos[4][16];
    const PruneState *st = sc.st;
    const int lane = threadIdx.x & 63, sub = lane / OPEN_LPR, sl = lane % OPEN_LPR;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool in_lds = oa.n_blocks <= oa.lds_cap;
    const int boff_mine = (in_lds && int(threadIdx.x) <= oa.n_blocks) ? oa.boff[threadIdx.x] : 0;
    // (one read of the state per workgroup: k_open_rows)
    if (threadIdx.x == 0) s_state[0] = st->pass_on, s_state[1] = st->A, s_state[2] = st->bitsel, s_state[3] = st->row_lo, s_state[4] = st->n_active;
    __syncthreads();
    const int pass_on = s_state[0], A = s_state[1], sel = s_state[2], row_lo = s_state[3], n_all = s_state[4];
    const unsigned long long *X = oa.bits + size_t(sel) * oa.bit_words;
    const unsigned tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = int(tile) * 16;
    TSC_OPEN_STAMP(0);  // started, state read
    if (pass_on && !ra.ap.exch) {
        // the other bit copy lags one pass behind (a superset of this one); rows removed by THIS launch are cleared there while other
        // workgroups may not have brought their words up to date yet: an AND commutes with those clears
        unsigned long long *Xo = oa.bits + size_t(sel ^ 1) * oa.bit_words;
        for (int w = blockIdx.x * 256 + threadIdx.x; w < oa.bit_words; w += gridDim.x * 256) {
            const unsigned long long x = X[w];
            if (x != ~0ull) atomicAnd(&Xo[w], x);
        }
    }
    if (pass_on && in_lds && int(blockIdx.x) * 64 < A) {  // (block-uniform)
        if (int(threadIdx.x) <= oa.n_blocks) s_boff[threadIdx.x] = boff_mine;
        for (int e = threadIdx.x + 256; e <= oa.n_blocks; e += 256) s_boff[e] = oa.boff[e];
        __syncthreads();
    }
    if (tile >= oa.n_tiles) return;  // (padding of the last block)
    TSC_OPEN_STAMP(1);  // other bit copy brought up to date, prefix staged
    if (pass_on && r0 < A) {
        unsigned long long n_eval = 0, n_exact = 0, ev_total = 0, rm_total = 0;
        int n_screened = 0;
        // The wavefront walks its rows' columns on 4096 bits of the mask staged from the word of its first row on.  That holds the stop
        // positions of all 16 rows -- unless the mask has become so sparse that the rows of a tile lie further apart than that (an ensemble
        // that collapsed: a chunk holds a handful of survivors).  Then the leading rows that do fit are finished and the tile is opened
        // again from the first row that did not, as often as it takes: nothing but the statistics lives across a round.
        for (int r_base = r0, rows_left = min(16, A - r0); rows_left > 0;) {
            const OpenedRow o = open_tile_rows(g, oa, s_boff, in_lds, X, A, row_lo, n_all, r_base, D);
            // ---- the rows, one per lane 0 .. 15
            const int src = (lane & 15) * OPEN_LPR;
            const int row_c = __shfl(o.my_c, src);
            const int64_t row_i = __shfl((long long)o.i, src), row_first = __shfl((long long)o.first, src), row_found = __shfl((long long)o.found, src);
            const int64_t i0 = __shfl((long long)row_i, 0);
            const int64_t w0 = i0 >> 6;
            // leading rows whose stop position lies inside the staged words (the first one always: its range is a chunk at most)
            const unsigned long long fits = __builtin_amdgcn_ballot_w64(lane < rows_left && row_found - 64 * w0 <= 64 * RP_WORDS);
            const int nrows = min(rows_left, int(__ffsll((long long)~fits)) - 1);
            const int my_cend = lane < nrows ? row_c : 0;
            const bool live0 = lane < nrows && my_cend > r_base + lane + 1;
            unsigned alive = unsigned(__builtin_amdgcn_ballot_w64(live0));
            __builtin_amdgcn_wave_barrier();  // (a second round: the first has read these arrays)
            if (lane < 16) s_best[wid][lane] = INT_MAX, s_rowpos[wid][lane] = int(row_i);
            int cmax = my_cend;
            long long pmax = lane < nrows ? (long long)row_found : 0ll;
            for (int off = 8; off > 0; off >>= 1) {
                cmax = max(cmax, __shfl_xor(cmax, off));
                pmax = max(pmax, __shfl_xor(pmax, off));
            }
            cmax = __shfl(cmax, 0), pmax = __shfl(pmax, 0);
            const int ncols = cmax - r_base - 1;  // columns r_base + 1 .. cmax - 1 are all any of the rows may look at: column c has rank r_base + 1 + c
            if (alive) {
                // ---- the mask behind the first row, as bits: RP_WORDS words, one per lane; column c = the c-th set bit
                unsigned long long word = (w0 + lane < oa.bit_words && 64 * (w0 + lane) < pmax) ? X[w0 + lane] : 0ull;
                if (lane == 0) word &= (i0 & 63) == 63 ? 0ull : (~0ull << ((i0 & 63) + 1));
                const int pc = __popcll(word);
                int incl = pc;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if (lane >= off) incl += t;
                }
                s_words[wid][lane] = word;
                s_wpre[wid][lane] = (unsigned short)(incl - pc);
                *reinterpret_cast<f32x4 *>(&s_rowdesc[wid][sub * RP_RS + 4 * sl]) = o.dval;
                __builtin_amdgcn_wave_barrier();
                if (lane < 16) {  // squared norms of the row descriptors, per family, behind the components (one LDS record per row: sieve.hpp)
                    const f32x2 *dr = reinterpret_cast<const f32x2 *>(&s_rowdesc[wid][lane * RP_RS]);
                    f32x2 nr = {0.0f, 0.0f};
#pragma unroll
                    for (int k = 0; k < KD; ++k) nr = __builtin_elementwise_fma(dr[k], dr[k], nr);
                    *reinterpret_cast<f32x2 *>(&s_rowdesc[wid][lane * RP_RS + DW]) = nr;
                }
                __builtin_amdgcn_wave_barrier();

                const int h3 = ra.h * 3;
                const int limit_bits = __float_as_int(screen_limit32_dot(__uint_as_float(*ra.dmax_bits), ra.desc_limit));  // (positive: integer order = float order)
                const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                unsigned short *queue = s_queue[wid], *exq = s_exq[wid];
                const float *rowdesc = s_rowdesc[wid];
                int qn = 0, qe = 0;

                auto decode = [&](unsigned e, int &t, int &c, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
                    t = int(e >> 12);
                    c = int(e & 0xfffu);
                    const int64_t i = s_rowpos[wid][t], j = 64 * w0 + s_colpos[wid][c];
                    pp = ra.heavy + i * h3, pq = ra.heavy + j * h3;
                    Gi = ra.Gall[i], Gj = ra.Gall[j];
                };
                auto note_similar = [&](bool sim, int t, int c) __attribute__((always_inline)) {
                    if (sim) atomicMin(&s_best[wid][t], c);
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
                    const int grp = lane / lpp, sb = lane - grp * lpp;
                    bool sim = false, degenerate = false;
                    int t = 0, c = 0;
                    unsigned ent = 0;
                    if (grp < cnt) {
                        const double *pp, *pq;
                        double Gi, Gj, H[9], e[4];
                        ent = exq[base + grp];
                        decode(ent, t, c, pp, pq, Gi, Gj);
                        pair_H(pp, pq, ra.h, sb, lpp, H);
                        if (rotation_quaternion_fast(H, Gi, Gj, e)) {  // (the lanes of a group hold the same H: they branch together)
                            double rm, md;
                            residual_rmsd_maxdev(pp, pq, ra.h, e, rm, md, sb, lpp);
                            sim = sb == 0 && rm < ra.thr && md < ra.maxdev_thr;  // rmsd_pruning.py:75
                        } else {
                            degenerate = sb == 0;
                        }
                    }
                    note_similar(sim, t, c);
                    // degenerate top eigenvalue: the Jacobi solver, one pair at a time by the whole wavefront (sieve.hpp)
                    for (unsigned long long dm = __builtin_amdgcn_ballot_w64(degenerate); dm; dm &= dm - 1) {
                        const unsigned e1 = unsigned(__builtin_amdgcn_readlane(int(ent), __ffsll((long long)dm) - 1));
                        int t2, c2;
                        const double *pp, *pq;
                        double Gi, Gj, H[9], e[4], rm, md;
                        decode(e1, t2, c2, pp, pq, Gi, Gj);
                        pair_H(pp, pq, ra.h, lane, 64, H);
                        double *jac = s_jacobi[wid];
                        if (lane == 0) {
                            horn_matrix(H, jac);
                            top_eigvec4_mem(jac, jac + 16, e);
                            jac[0] = e[0], jac[1] = e[1], jac[2] = e[2], jac[3] = e[3];
                        }
                        __builtin_amdgcn_wave_barrier();
                        e[0] = jac[0], e[1] = jac[1], e[2] = jac[2], e[3] = jac[3];
                        __builtin_amdgcn_wave_barrier();
                        residual_rmsd_maxdev(pp, pq, ra.h, e, rm, md, lane, 64);
                        if (rm < ra.thr && md < ra.maxdev_thr) {  // wave-uniform
                            if (lane == 0) atomicMin(&s_best[wid][t2], c2);
                            alive &= ~(1u << t2);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                };
                auto sign_stage = [&](int base, int cnt) __attribute__((always_inline)) {
                    int lpp = 64;
                    while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
                    const int grp = lane / lpp, sb = lane - grp * lpp;
                    bool cand = false, sim = false;
                    unsigned e = 0;
                    int t = 0, c = 0;
                    if (grp < cnt) {
                        e = queue[base + grp];
                        const double *pp, *pq;
                        double Gi, Gj, H[9];
                        decode(e, t, c, pp, pq, Gi, Gj);
                        pair_H(pp, pq, ra.h, sb, lpp, H);
                        const int verdict = pair_verdict(H, 0.5 * (Gi + Gj), ra.half_h_thr2, ra.two_thr2, ra.h);
                        cand = sb == 0 && verdict == PAIR_UNDECIDED;
                        sim = sb == 0 && verdict == PAIR_SIMILAR;
                    }
                    note_similar(sim, t, c);
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

                // column c0 + lane: its position (select on the staged words) and its descriptor; the gather of the next 64 columns is
                // in flight while these are screened
                auto load_cols = [&](int c0, f32x2 (&dst)[KD]) __attribute__((always_inline)) {
                    const int c = min(c0 + lane, ncols - 1);
                    int w = 0;
#pragma unroll
                    for (int step = RP_WORDS / 2; step > 0; step >>= 1)
                        if (int(s_wpre[wid][w + step]) <= c) w += step;
                    const int pos = 64 * w + select64(s_words[wid][w], c - int(s_wpre[wid][w]));
                    if (c0 + lane < ncols) s_colpos[wid][c] = (unsigned short)pos;
                    const f32x4 *srcp = reinterpret_cast<const f32x4 *>(D + (64 * w0 + pos) * DW);
#pragma unroll
                    for (int k = 0; k < KD / 2; ++k) {
                        const f32x4 v = srcp[k];
                        dst[2 * k] = f32x2{v.x, v.y};
                        dst[2 * k + 1] = f32x2{v.z, v.w};
                    }
                };
                f32x2 dq[KD], dq_next[KD], cn;
                load_cols(0, dq_next);
                TSC_OPEN_STAMP(4);  // words staged, first 64 columns found and their descriptors arrived
                int my_screened = 0;
                for (int c0 = 0; c0 < ncols && alive; c0 += 64) {
                    {
                        const int c = c0 + lane;
#pragma unroll
                        for (int k = 0; k < KD; ++k) dq[k] = dq_next[k];
                        if (c0 + 64 < ncols) load_cols(c0 + 64, dq_next);
                        cn = f32x2{0.0f, 0.0f};  // -|column descriptor|^2 / 2 per family: where the dot-product chain of a row starts (sieve.hpp, TRIM)
#pragma unroll
                        for (int k = 0; k < KD; ++k) cn = __builtin_elementwise_fma(dq[k], dq[k], cn);
                        cn = cn * f32x2{-0.5f, -0.5f};
                        // row t (rank r_base + t) looks at the columns t <= c < cend - r_base - 1
                        const int c_end = my_cend - r_base - 1;
                        const bool here = lane < nrows && ((alive >> lane) & 1u) && c_end > c0 && lane < c0 + 64;
                        unsigned rows = unsigned(__builtin_amdgcn_ballot_w64(here));
                        my_screened += here ? max(0, min(c_end, c0 + 64) - max(lane, c0)) : 0;
                        // |r|^2 + |c|^2 - 2 r.c in packed fp32 for both families (screen_limit32_dot has the bound), compared as bit patterns
                        auto row_worst = [&](const int t) __attribute__((always_inline)) {
                            const f32x2 *rec = reinterpret_cast<const f32x2 *>(rowdesc + t * RP_RS);
                            f32x2 acc = cn;
#pragma unroll
                            for (int k = 0; k < KD; ++k) acc = __builtin_elementwise_fma(rec[k], dq[k], acc);
                            const f32x2 s2 = __builtin_elementwise_fma(acc, f32x2{-2.0f, -2.0f}, rec[KD]);
                            const int ce = __builtin_amdgcn_readlane(c_end, t);
                            return (c >= t && c < ce) ? max(__float_as_int(s2.x), __float_as_int(s2.y)) : INT_MAX;
                        };
                        auto push = [&](const int t, const int worst) __attribute__((always_inline)) {
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(worst <= limit_bits);
                            if (m) {
                                if ((m >> lane) & 1ull) queue[qn + __popcll(m & lt_mask)] = (unsigned short)((unsigned(t) << 12) | unsigned(c));
                                qn += __popcll(m);
                            }
                        };
                        while (rows) {  // two rows per trip: their chains are independent, so one hides the other's latencies
                            const int t0 = __ffs(rows) - 1;
                            rows &= rows - 1;
                            const int t1 = rows ? __ffs(rows) - 1 : t0;
                            rows &= rows - 1;
                            const int w0s = row_worst(t0), w1s = row_worst(t1);
                            if (__builtin_amdgcn_ballot_w64(min(w0s, w1s) <= limit_bits)) {  // (rare) some column is within the limit
                                push(t0, w0s);
                                if (t1 != t0) push(t1, w1s);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    while (qn >= 64) {
                        sign_stage(qn - 64, 64);
                        qn -= 64;
                    }
                }
                TSC_OPEN_STAMP(5);  // screened (full batches of candidates evaluated on the way)
                if (qn > 0) sign_stage(0, qn);
                if (qe > 0) exact_stage(0, qe);
                TSC_OPEN_STAMP(6);  // candidates evaluated
                for (int off = 8; off > 0; off >>= 1) my_screened += __shfl_xor(my_screened, off);
                n_screened += my_screened;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- the verdicts of these rows (lanes 0 .. nrows - 1), applied from registers
            bool removed = false;
            int64_t delta = 0;
            unsigned long long ev = 0;
            if (lane < nrows) {
                const int b = s_best[wid][lane];
                if (b != INT_MAX) {
                    removed = true;
                    delta = 64 * w0 + int64_t(s_colpos[wid][b]) - row_i;
                    ev = (unsigned long long)(b + 1 - lane);  // columns r + 1 .. r_base + 1 + b were evaluated
                } else {
                    ev = (unsigned long long)(my_cend - (r_base + lane) - 1);  // every active column before the stop column
                }
            }
            apply_rows_core(ra.ap, sel, removed, row_i, row_first, delta, ev, ev_total, rm_total);
            r_base += nrows, rows_left -= nrows;
        }
        if (lane == 0) {
            count_add(sc.cnt, tile, CNT_FORMED, n_eval);
            count_add(sc.cnt, tile, CNT_EXACT, n_exact);
            count_add(sc.cnt, tile, CNT_SCREENED, (unsigned long long)n_screened);
            count_add(sc.cnt, tile, CNT_EVALUATED, ev_total);
            count_add(sc.cnt, tile, CNT_REMOVED, rm_total);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSC_OPEN_STAMP(7);  // verdicts applied
    int fin = 0;
    if (lane == 0) fin = tickets_arrive(oa.tickets, tile, oa.n_tiles, PT_GROUPS) ? 1 : 0;
    if (__builtin_amdgcn_readfirstlane(fin)) pass_step_wave(sc, next);
}

}  // namespace tsc
