// cull.hpp -- the large passes of prune_conformers_rmsd with SPATIAL CULLING in front of the pair kernel's screen.
//
// The screen of k_rmsd_sieve (sieve.hpp) drops a pair when its descriptors lie further apart than h thr^2 in either feature
// family; in the big passes of a large ensemble it still LOOKS at every pair of a row's range -- 99.95 % of them to no avail
// (profiles/r03_culling_study.json: at one million conformers 0.05 % of the pairs pass).  Here the active structures of a pass
// are laid out along a Morton curve through the leading descriptor components, chunk by chunk; rows are taken 16 and columns
// 128 consecutive structures OF THAT ORDER at a time, every such tile has a bounding box in descriptor space, and a (row tile,
// column tile) pair whose boxes lie further apart than the screen's limit is never loaded.  On C4 (1M x 50) that leaves 8-23 %
// of the tile pairs the left-to-right walk visits.
//
// What changes and what does not (tscode/rmsd_pruning.py:43-79):
//   * which rows a pass removes depends only on the SET of a row's similar columns inside its range (i, jc(i)): the row is removed
//     iff the set is not empty, and its cache key needs the smallest member js(i).  Both come out of an atomicMin over the
//     columns' active ranks whatever order the pairs are visited in -- best[] is exactly what the ordered walk produces;
//   * the early exit does not survive: the walk in index order stops a row at its first similar column, the culled walk has to
//     find every similar column of the row (there are few) because a smaller index may turn up later;
//   * a pair is visited ONCE, from the tile pair (R, C) with C at or behind R in the sorted order, and credited to whichever of
//     the two structures has the lower index: rows also collect verdicts as columns of other tiles, so the verdicts of a pass are
//     applied by k_apply_pass behind the pair kernel (not tile by tile inside it);
//   * the number of pair evaluations the REFERENCE would have made follows from best[] and cend[] as before.
#pragma once
#include "rmsd.hpp"
#include "sieve.hpp"

namespace tsc {

constexpr int CULL_MORTON_BITS = 5;                         // bits per dimension of the (coarse) Morton key
constexpr int CULL_MORTON_DIMS = 3;                         // leading components of family 0
constexpr int CULL_BUCKETS = 1 << (CULL_MORTON_BITS * CULL_MORTON_DIMS);   // 32 768 cells: a counting sort, no comparison sort
constexpr int CULL_MAX_CHUNKS = 64;                         // passes with more chunks are not culled (their chunks are short anyway)
constexpr int CULL_COLS = 128;                              // columns per tile (the pair kernel's CPL = 2)
constexpr int CULL_BOX = 2 * DW;                            // floats per bounding box: lo[16], hi[16]
#ifndef TSC_CULL_XCD_RUN
#define TSC_CULL_XCD_RUN 32
#endif
constexpr int CULL_XCD_RUN = TSC_CULL_XCD_RUN;              // option "cull_xcd": consecutive row groups (of 4 tiles) that stay on one XCD

// ---- once per run: the structures in (coarse) Morton order of their descriptors ---------------------------------------------
__device__ inline unsigned morton_cell(const float *__restrict__ d, float inv_dmax) {
    unsigned code = 0;
#pragma unroll
    for (int k = 0; k < CULL_MORTON_DIMS; ++k) {
        const float x = d[2 * k] * inv_dmax * 0.5f + 0.5f;  // component k of family 0 (families interleaved), mapped to [0, 1]
        int q = int(x * float(1 << CULL_MORTON_BITS));
        q = q < 0 ? 0 : (q >= (1 << CULL_MORTON_BITS) ? (1 << CULL_MORTON_BITS) - 1 : q);   // (a NaN component lands in cell 0)
#pragma unroll
        for (int b = 0; b < CULL_MORTON_BITS; ++b) code |= ((unsigned(q) >> b) & 1u) << (b * CULL_MORTON_DIMS + k);
    }
    return code;
}
// The order has to come out THE SAME on every rank of a sharded run (the ranks deal the row tiles of the sorted layout among
// themselves: with layouts of their own they would miss pairs), so it is a STABLE sort by cell, ties in index order: two passes of
// a least-significant-digit radix sort over 8 bits, each an ordered partition into 256 buckets -- per block of 2048 entries the
// number of entries per bucket; their exclusive prefix over blocks and buckets; a scatter that ranks the entries of a wavefront by
// ballots (entry order kept).
constexpr int RADIX_BUCKETS = 256;
__device__ inline int radix_digit(const float *__restrict__ D, const int32_t *__restrict__ in, int64_t m, int64_t n, float inv, int shift, int &id) {
    id = 0;
    if (m >= n) return -1;
    id = in ? in[m] : int(m);
    return int((morton_cell(D + int64_t(id) * DW, inv) >> shift) & (RADIX_BUCKETS - 1));
}
inline __global__ __launch_bounds__(256) void k_radix_count(const float *__restrict__ D, const int32_t *__restrict__ in, int64_t n, const unsigned *__restrict__ dmax_bits,
                                                      int shift, int32_t *__restrict__ blk_cnt) {
    __shared__ int s_cnt[RADIX_BUCKETS];
    const float dmax = __uint_as_float(*dmax_bits);
    const float inv = (dmax > 0.0f && dmax < 3.0e38f) ? 1.0f / dmax : 0.0f;
    const int tid = threadIdx.x;
    s_cnt[tid] = 0;
    __syncthreads();
    const int64_t m0 = int64_t(blockIdx.x) * 2048;
    for (int u = 0; u < 8; ++u) {
        int id;
        const int d = radix_digit(D, in, m0 + u * 256 + tid, n, inv, shift, id);
        if (d >= 0) atomicAdd(&s_cnt[d], 1);
    }
    __syncthreads();
    blk_cnt[int64_t(blockIdx.x) * RADIX_BUCKETS + tid] = s_cnt[tid];
}
// blk_cnt[b][d] -> entries with digit d in the blocks before b (one wavefront per digit), tot[d] = entries with digit d
inline __global__ __launch_bounds__(64) void k_radix_scan(int n_blocks, int32_t *__restrict__ blk_cnt, int32_t *__restrict__ tot) {
    const int d = blockIdx.x, lane = threadIdx.x & 63;
    int run = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < n_blocks ? blk_cnt[int64_t(b) * RADIX_BUCKETS + d] : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (b < n_blocks) blk_cnt[int64_t(b) * RADIX_BUCKETS + d] = run + incl - v;
        run += __shfl(incl, 63);
    }
    if (lane == 0) tot[d] = run;
}
// tot[d] -> first output slot of digit d (exclusive prefix over the 256 digits; one block)
inline __global__ __launch_bounds__(256) void k_radix_base(int32_t *__restrict__ tot) {
    __shared__ int s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int v = tot[tid];
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wv; ++q) base += s_part[q];
    tot[tid] = base + incl - v;
}
inline __global__ __launch_bounds__(256) void k_radix_scatter(const float *__restrict__ D, const int32_t *__restrict__ in, int64_t n, const unsigned *__restrict__ dmax_bits,
                                                        int shift, const int32_t *__restrict__ blk_base, const int32_t *__restrict__ digit_base,
                                                        int32_t *__restrict__ out) {
    __shared__ int s_cnt[32][RADIX_BUCKETS];
    const float dmax = __uint_as_float(*dmax_bits);
    const float inv = (dmax > 0.0f && dmax < 3.0e38f) ? 1.0f / dmax : 0.0f;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    for (int e = tid; e < 32 * RADIX_BUCKETS; e += 256) (&s_cnt[0][0])[e] = 0;
    __syncthreads();
    const int64_t m0 = int64_t(blockIdx.x) * 2048;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int my_d[8], my_id[8], my_rank[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        my_d[u] = radix_digit(D, in, m0 + u * 256 + tid, n, inv, shift, my_id[u]);
        my_rank[u] = 0;
        for (unsigned long long left = __ballot(my_d[u] >= 0); left;) {   // the lanes of a wavefront that share a digit, in lane order
            const int dd = __shfl(my_d[u], __ffsll((long long)left) - 1);
            const unsigned long long same = __ballot(my_d[u] == dd);
            if (my_d[u] == dd) my_rank[u] = __popcll(same & lt);
            if (lane == 0) s_cnt[u * 4 + wv][dd] = __popcll(same);
            left &= ~same;
        }
    }
    __syncthreads();
    {   // exclusive prefix over the block's 32 (round, wavefront) slots per digit, on top of the digit's base and the block's offset in it
        int run = digit_base[tid] + blk_base[int64_t(blockIdx.x) * RADIX_BUCKETS + tid];
        for (int sl = 0; sl < 32; ++sl) {
            const int v = s_cnt[sl][tid];
            s_cnt[sl][tid] = run;
            run += v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (my_d[u] >= 0) out[s_cnt[u * 4 + wv][my_d[u]] + my_rank[u]] = my_id[u];
}

// ---- once per culled pass ----------------------------------------------------------------------------------------------
// cbase[c] = active structures before chunk c of the open pass (c = 0 .. k; cbase[k] = A): chunk c takes the positions
// [cbase[c], cbase[c + 1]) of the sorted layout -- the rank range of the chunk, in another order.  Also clears the fill counters.
inline __global__ __launch_bounds__(64) void k_chunk_bases(PassGeom g, const PruneState *__restrict__ st, const int32_t *__restrict__ boff,
                                                     const unsigned long long *__restrict__ bits, int bit_words, int n_blocks, int32_t *__restrict__ cbase,
                                                     int32_t *__restrict__ cfill) {
    const unsigned long long *X = bits + size_t(st->bitsel) * bit_words;
    const int n_all = st->n_active;
    const int c = blockIdx.x;  // one wavefront per chunk boundary (grid = k + 1)
    const int64_t pos = c < g.k ? int64_t(c) * g.cs : int64_t(g.n);
    // (a pass partitioned over ranks runs on the LOCAL rows 0 .. A - 1 of this rank's chunks: ranks relative to row_lo, clamped to them)
    const int r = rank_below_wave(boff, X, n_blocks, n_all, pos) - st->row_lo;
    if ((threadIdx.x & 63) == 0) cbase[c] = r < 0 ? 0 : (r > st->A ? st->A : r), cfill[c] = 0;
}

// Culled or walked?  The ordered walk looks at the pairs inside the rows' ranges (r, cend[r]) -- W of them, summed by k_open_rows;
// the culled kernel at the tile pairs whose boxes lie within the limit, whatever the ranges are: a share f of ALL pairs of the
// chunks that falls with the chunk length (measured at 1M x 50: 0.27 at 22 000 active structures per chunk, 0.19 at 62 000,
// 0.13 at 206 000: about 0.27 (22 000 / len)^(1/3)) at two thirds of the walk's rate per pair (what the boxes leave are the near
// tiles, dense in candidates).  Where the reference's cache ends most rows early (W a few per cent of all pairs: C4's k = 5 and
// k = 1 passes) the walk has little to do and keeps the pass.  One wavefront; the verdict goes to pinned host memory, the host
// waits for it (a pass that large takes a millisecond or more; enqueueing both kernels and letting the loser's 10^5 workgroups
// start and leave cost 0.1-0.2 ms a pass) and launches one flow or the other.
inline __global__ __launch_bounds__(64) void k_cull_decide(PruneState *__restrict__ st, const PassCounters *__restrict__ cnt, const int32_t *__restrict__ cbase, int k,
                                                     int force, int *__restrict__ flag_host) {
    const int lane = threadIdx.x & 63;
    unsigned long long w = __hip_atomic_load(&cnt->w[lane][CNT_WALK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off);
    double all = 0.0, longest = 0.0;
    if (lane < k) {
        const double len = double(cbase[lane + 1] - cbase[lane]);
        all = 0.5 * len * len, longest = len;
    }
    for (int off = 32; off > 0; off >>= 1) all += __shfl_xor(all, off), longest = fmax(longest, __shfl_xor(longest, off));
    if (lane == 0) {
        const double f = 0.27 * cbrt(22000.0 / fmax(longest, 1.0));
        const int on = (w > 0 && (force || double(w) > 1.5 * f * all)) ? 1 : 0;  // (force: option "cull" = 2, tests; w = 0: a pass that is gated off, or whose rows have no columns left)
        st->cull_on = on;
        flag_host[1] = st->A;   // (the rows of the pass: a pass that is WALKED is launched for these, not for the ensemble the run began with)
        *flag_host = on;
        __threadfence_system();
    }
}

// The active structures of the pass, chunk by chunk, each chunk in the run's Morton order -- an ORDERED partition of that order by
// chunk, so that 128 consecutive positions of a chunk are 128 neighbours on the curve (a layout that only kept a block's structures
// together gave column tiles that mixed two distant places of the curve, and the boxes culled a quarter of the tiles instead of
// three quarters).  Three launches: per block of 2048 entries the number of active structures per chunk; their exclusive prefix
// over the blocks (one wavefront per chunk); the scatter, which ranks every structure inside its block by ballots -- entry order
// is kept inside a block as well -- and moves what the pair kernel reads BY POSITION: the descriptor and the structure's active
// rank (crank: what the verdicts are expressed in).
constexpr int CULL_LAYOUT_ITEMS = 2048;
constexpr int CULL_LAYOUT_SLOTS = CULL_LAYOUT_ITEMS / 64;   // (round, wavefront) slots of a block: 64 consecutive entries each

// chunk of entry m of the Morton order if that structure is active, else -1
struct LayoutRange {
    int s_lo, s_hi;   // structures [s_lo, s_hi) take part (a pass partitioned over ranks: this rank's chunks; else all n)
};
__device__ inline int layout_chunk(const PassGeom &g, const LayoutRange lr, const int32_t *__restrict__ order, const unsigned long long *__restrict__ X, int64_t m, int &i) {
    i = 0;
    if (m >= g.n) return -1;
    i = order[m];
    if (i < lr.s_lo || i >= lr.s_hi || !((X[i >> 6] >> (i & 63)) & 1ull)) return -1;
    const int c = i / g.cs;
    return c >= g.k ? g.k - 1 : c;
}
// s_cnt[slot][chunk] = active structures of chunk `chunk` among the 64 entries of slot `slot`; mine = this lane's chunk (-1: none);
// returns the lane's rank among the lanes of its wavefront with the same chunk
__device__ inline int layout_wave_rank(int mine, int *s_cnt_slot) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank = 0;
    for (unsigned long long left = __ballot(mine >= 0); left;) {
        const int c = __shfl(mine, __ffsll((long long)left) - 1);
        const unsigned long long same = __ballot(mine == c);
        if (mine == c) rank = __popcll(same & lt);
        if (lane == 0 && s_cnt_slot) s_cnt_slot[c] = __popcll(same);
        left &= ~same;
    }
    return rank;
}
inline __global__ __launch_bounds__(256) void k_layout_count(PassGeom g, LayoutRange lr, const PruneState *__restrict__ st, const int32_t *__restrict__ order,
                                                      const unsigned long long *__restrict__ bits, int bit_words, int32_t *__restrict__ blk_cnt) {
    __shared__ int s_cnt[CULL_MAX_CHUNKS];
    if (st->pass_on == 0) return;
    const unsigned long long *X = bits + size_t(st->bitsel) * bit_words;
    const int tid = threadIdx.x;
    if (tid < CULL_MAX_CHUNKS) s_cnt[tid] = 0;
    __syncthreads();
    const int64_t m0 = int64_t(blockIdx.x) * CULL_LAYOUT_ITEMS;
    for (int u = 0; u < CULL_LAYOUT_ITEMS / 256; ++u) {
        int i;
        const int c = layout_chunk(g, lr, order, X, m0 + u * 256 + tid, i);
        for (unsigned long long left = __ballot(c >= 0); left;) {   // one LDS atomic per (wavefront, chunk present)
            const int cc = __shfl(c, __ffsll((long long)left) - 1);
            const unsigned long long same = __ballot(c == cc);
            if ((tid & 63) == 0) atomicAdd(&s_cnt[cc], __popcll(same));
            left &= ~same;
        }
    }
    __syncthreads();
    if (tid < g.k) blk_cnt[int64_t(blockIdx.x) * CULL_MAX_CHUNKS + tid] = s_cnt[tid];
}
// blk_cnt[b][c] -> first position of block b's structures of chunk c: cbase[c] + structures of chunk c in the blocks before b
inline __global__ __launch_bounds__(64) void k_layout_scan(const PruneState *__restrict__ st, int n_layout_blocks, const int32_t *__restrict__ cbase,
                                                     int32_t *__restrict__ blk_cnt) {
    if (st->pass_on == 0) return;
    const int c = blockIdx.x, lane = threadIdx.x & 63;
    int run = cbase[c];
    for (int b0 = 0; b0 < n_layout_blocks; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < n_layout_blocks ? blk_cnt[int64_t(b) * CULL_MAX_CHUNKS + c] : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (b < n_layout_blocks) blk_cnt[int64_t(b) * CULL_MAX_CHUNKS + c] = run + incl - v;
        run += __shfl(incl, 63);
    }
}
inline __global__ __launch_bounds__(256) void k_layout_scatter(PassGeom g, LayoutRange lr, const PruneState *__restrict__ st, const int32_t *__restrict__ order,
                                                        const unsigned long long *__restrict__ bits, int bit_words, const int32_t *__restrict__ rank_of,
                                                        const float *__restrict__ Dc, const int32_t *__restrict__ blk_base, float *__restrict__ Ds,
                                                        int32_t *__restrict__ crank, const _Float16 *__restrict__ Dh = nullptr,
                                                        _Float16 *__restrict__ Dhs = nullptr, int32_t *__restrict__ cstruct = nullptr) {
    // cstruct (optional): the structure at every sorted position (= act[crank[pos]]: cull_mm.hpp reads a candidate's coordinates one
    // round trip earlier with it)
    // Dh -> Dhs (optional): the float16 records of the matrix-core screen (mm_record.hpp) move along with the descriptors
    __shared__ int s_cnt[CULL_LAYOUT_SLOTS][CULL_MAX_CHUNKS];
    if (st->pass_on == 0) return;
    const unsigned long long *X = bits + size_t(st->bitsel) * bit_words;
    const int tid = threadIdx.x, wv = tid >> 6;
    for (int e = tid; e < CULL_LAYOUT_SLOTS * CULL_MAX_CHUNKS; e += 256) (&s_cnt[0][0])[e] = 0;
    __syncthreads();
    const int64_t m0 = int64_t(blockIdx.x) * CULL_LAYOUT_ITEMS;
    int my_c[CULL_LAYOUT_ITEMS / 256], my_i[CULL_LAYOUT_ITEMS / 256], my_rank[CULL_LAYOUT_ITEMS / 256];
#pragma unroll
    for (int u = 0; u < CULL_LAYOUT_ITEMS / 256; ++u) {
        my_c[u] = layout_chunk(g, lr, order, X, m0 + u * 256 + tid, my_i[u]);
        my_rank[u] = layout_wave_rank(my_c[u], s_cnt[u * 4 + wv]);
    }
    __syncthreads();
    // exclusive prefix over the block's 32 slots, per chunk, on top of the block's base
    if (tid < g.k) {
        int run = blk_base[int64_t(blockIdx.x) * CULL_MAX_CHUNKS + tid];
        for (int sl = 0; sl < CULL_LAYOUT_SLOTS; ++sl) {
            const int v = s_cnt[sl][tid];
            s_cnt[sl][tid] = run;
            run += v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CULL_LAYOUT_ITEMS / 256; ++u) {
        if (my_c[u] < 0) continue;
        const int pos = s_cnt[u * 4 + wv][my_c[u]] + my_rank[u];
        const int r = rank_of[my_i[u]];
        crank[pos] = r;
        if (cstruct) cstruct[pos] = my_i[u];
        const f32x4 *src = reinterpret_cast<const f32x4 *>(Dc + int64_t(r) * DW);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Ds + int64_t(pos) * DW);
#pragma unroll
        for (int q = 0; q < DW / 4; ++q) dst[q] = src[q];
        if (Dhs) {
            const f32x4 *hs = reinterpret_cast<const f32x4 *>(Dh + int64_t(r) * MM_REC_HALVES);
            f32x4 *hd = reinterpret_cast<f32x4 *>(Dhs + int64_t(pos) * MM_REC_HALVES);
#pragma unroll
            for (int q = 0; q < MM_REC_HALVES / 8; ++q) hd[q] = hs[q];
        }
    }
}

// Bounding boxes of the sorted layout: per 128 positions one column box and eight row boxes (16 positions each), lo[16] then
// hi[16].  Positions beyond the active count do not exist: an empty box (lo = +inf, hi = -inf) is infinitely far from everything.
inline __global__ __launch_bounds__(128) void k_tile_boxes(const PruneState *__restrict__ st, const float *__restrict__ Ds, float *__restrict__ cbox,
                                                     float *__restrict__ rbox) {
    __shared__ float s_d[CULL_COLS][DW + 1];
    const int A = st->pass_on ? st->A : 0;
    const int64_t p0 = int64_t(blockIdx.x) * CULL_COLS;
    const int tid = threadIdx.x;
    for (int e = tid; e < CULL_COLS * DW; e += 128) {
        const int p = e / DW, k = e - p * DW;
        s_d[p][k] = (p0 + p < A) ? Ds[(p0 + p) * DW + k] : __builtin_nanf("");
    }
    __syncthreads();
    const int t = tid / DW, k = tid - t * DW;  // row tile t of this block, component k
    float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float v = s_d[16 * t + q][k];
        lo = fminf(lo, v), hi = fmaxf(hi, v);  // (fminf / fmaxf ignore the NaN of an absent position)
    }
    float *rb = rbox + (int64_t(blockIdx.x) * 8 + t) * CULL_BOX;
    rb[k] = lo, rb[DW + k] = hi;
    __syncthreads();
    s_d[t][k] = lo, s_d[8 + t][k] = hi;
    __syncthreads();
    if (tid < DW) {
        float clo = __builtin_inff(), chi = -__builtin_inff();
#pragma unroll
        for (int q = 0; q < 8; ++q) clo = fminf(clo, s_d[q][tid]), chi = fmaxf(chi, s_d[8 + q][tid]);
        float *cb = cbox + int64_t(blockIdx.x) * CULL_BOX;
        cb[tid] = clo, cb[DW + tid] = chi;
    }
}

struct CullArgs {
    const float *Ds;        // descriptors by sorted position
    const int32_t *crank;   // sorted position -> active rank
    const int32_t *cbase;   // [k + 1] first position of every chunk
    const float *cbox, *rbox;
    int k;                  // chunks of the pass
    int tile_block;         // row tiles are dealt to the ranks in runs of this many consecutive tiles of the sorted layout (1: one by one)
    int xcd_seg;            // 1: runs of row groups keyed to XCDs inside every column segment (k_rmsd_sieve_sorted), one work item per workgroup
};

// The pair kernel of a culled pass: one wavefront = (16 consecutive positions of the sorted layout) x (one segment of the
// columns at or behind them, to the end of the rows' chunk).  Per column tile of 128: skipped unless its box lies within the
// screen's limit of the row tile's box in both families; else the same packed-fp32 screen as k_rmsd_sieve, the same queue,
// the same two evaluation stages.  A pair that passes the screen is credited to its lower-ranked structure; whether the higher
// one lies inside that row's range (beyond the row, before its stop column, same chunk) is checked where the pair is decoded.
template <int TI, bool F32>
__device__ __forceinline__ void sieve_item_sorted(const double *__restrict__ heavy, const int32_t *__restrict__ act, const double *__restrict__ Gall,
                                                  const int32_t *__restrict__ cend, int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                                  const PruneState *__restrict__ st, const SieveArgs a, const CullArgs ca, const int slot, const int seg) {
    constexpr int CPL = 2, TILE_COLS = 64 * CPL, RS = 20;
    constexpr int QCAP = TI * TILE_COLS + 64;
    static_assert(TI == 16 && TILE_COLS == CULL_COLS && DW == 16, "tile shape");
    __shared__ unsigned short s_queue[4][QCAP];
    __shared__ unsigned short s_exq[4][128];
    __shared__ double s_jacobi[4][32];
    __shared__ __attribute__((aligned(16))) float s_rowdesc[4][TI * RS];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (runs of consecutive tiles stay on one rank: neighbours on the curve look at the same columns, and what a row has already found --
    // best[] -- is known where its other candidates are evaluated)
    const int tile = ca.tile_block <= 1 ? a.tile_begin + slot * a.tile_stride
                                        : ((slot / ca.tile_block) * a.tile_stride + a.tile_begin) * ca.tile_block + slot % ca.tile_block;
    const int p0 = tile * TI;
    const int pass_on = st->pass_on, A = st->A;
    if (pass_on == 0 || p0 >= A) return;
    const int nrows = min(TI, A - p0);
    // the chunk of the tile's LAST row ends the column range (a tile may straddle a chunk boundary; a pair across it fails the
    // range test at decode time: the higher rank lies at or beyond the lower one's stop column)
    int col_end;
    {
        const int cb = lane <= ca.k ? ca.cbase[lane] : INT_MAX;            // (k <= 63 chunks + the end)
        const unsigned long long le = __ballot(cb <= p0 + nrows - 1);        // cbase is ascending: the chunks that begin at or before the last row
        const int c_last = __popcll(le) - 1;
        col_end = __builtin_amdgcn_readlane(cb, c_last + 1);
    }
    const int seg_lo = (p0 & ~(TILE_COLS - 1)) + seg * a.seg_cols;
    const int seg_hi = min(seg_lo + a.seg_cols, col_end);
    if (seg_lo >= seg_hi) return;
    // which column tiles of the segment lie within the limit of this row tile: one lane per column tile, one round trip
    const float limit32 = screen_limit32_dot(__uint_as_float(*a.dmax_bits), a.desc_limit);
    const int limit_bits = __float_as_int(limit32);
    unsigned long long need;
    {
        const int n_ct = (seg_hi - seg_lo + TILE_COLS - 1) / TILE_COLS;    // <= 32 (segments of at most 4096 columns)
        bool near = false;
        if (lane < n_ct) {
            const f32x4 *rb = reinterpret_cast<const f32x4 *>(ca.rbox + int64_t(tile) * CULL_BOX);
            const f32x4 *cb = reinterpret_cast<const f32x4 *>(ca.cbox + int64_t(seg_lo / TILE_COLS + lane) * CULL_BOX);
            float g0 = 0.0f, g1 = 0.0f;
#pragma unroll
            for (int q = 0; q < DW / 4; ++q) {
                const f32x4 rl = rb[q], rh = rb[DW / 4 + q], cl = cb[q], ch = cb[DW / 4 + q];
                const float gx = fmaxf(0.0f, fmaxf(cl.x - rh.x, rl.x - ch.x)), gy = fmaxf(0.0f, fmaxf(cl.y - rh.y, rl.y - ch.y));
                const float gz = fmaxf(0.0f, fmaxf(cl.z - rh.z, rl.z - ch.z)), gw = fmaxf(0.0f, fmaxf(cl.w - rh.w, rl.w - ch.w));
                g0 = fmaf(gx, gx, fmaf(gz, gz, g0));  // components 4q, 4q + 2: family 0
                g1 = fmaf(gy, gy, fmaf(gw, gw, g1));  // components 4q + 1, 4q + 3: family 1
            }
            // every pair of the two tiles is at least sqrt(g) apart in that family; the screen keeps a pair only below limit32 in
            // both (a little slack for the rounding of g itself; a NaN gap -- an empty box -- compares false: skipped)
            near = fmaxf(g0, g1) <= limit32 * 1.001f;
        }
        need = __ballot(near);
    }
    if (!need) return;

    // the row descriptors of this work item -> LDS (record: 16 components, the two squared norms)
    float *rowdesc = s_rowdesc[wid];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int src = min(p0 + 4 * j + (lane >> 4), A - 1);
        rowdesc[(4 * j + (lane >> 4)) * RS + (lane & 15)] = ca.Ds[int64_t(src) * DW + (lane & 15)];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < TI) {
        const f32x2 *dr = reinterpret_cast<const f32x2 *>(rowdesc + lane * RS);
        f32x2 nr = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < KD; ++k) nr = __builtin_elementwise_fma(dr[k], dr[k], nr);
        *reinterpret_cast<f32x2 *>(rowdesc + lane * RS + DW) = nr;
    }
    __builtin_amdgcn_wave_barrier();

    const int h3 = a.h * 3;
    unsigned short *queue = s_queue[wid], *exq = s_exq[wid];
    int qn = 0, qe = 0;
    unsigned long long n_eval = 0, n_exact = 0, n_screened = 0;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // an entry = (row of the tile, column position inside the segment); the pair = the two structures at those positions, the one
    // with the lower active rank playing the reference's `ref` (rmsd_pruning.py:92: the row), the other its column
    int64_t si = 0, sj = 0;  // the structures of the pair decoded last
    auto decode = [&](unsigned e, int &lo, int &hi, const double *&pp, const double *&pq, double &Gi, double &Gj) __attribute__((always_inline)) {
        const int r1 = ca.crank[p0 + int(e >> 12)], r2 = ca.crank[seg_lo + int(e & 0xfffu)];
        lo = min(r1, r2), hi = max(r1, r2);
        const int64_t i = act[lo], j = act[hi];
        si = i, sj = j;
        pp = heavy + i * h3, pq = heavy + j * h3;
        Gi = Gall[i], Gj = Gall[j];
        // the column lies inside the row's range (rows of another chunk, or behind a cache hit, do not) -- and before the similar column
        // the row already has: what is left of the ordered walk's early exit (a later column cannot lower the minimum)
        return hi < cend[lo] && hi < __hip_atomic_load(&best[lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto exact_stage = [&](int base, int cnt) __attribute__((always_inline)) {
        int lpp = 64;
        while (lpp > 1 && 64 / lpp < cnt) lpp >>= 1;
        const int g = lane / lpp, sub = lane - g * lpp;
        bool degenerate = false;
        unsigned ent = 0;
        if (g < cnt) {
            int lo, hi;
            const double *pp, *pq;
            double Gi, Gj, H[9], e[4];
            ent = exq[base + g];
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
        const int g = lane / lpp, sub = lane - g * lpp;
        bool cand = false, counted = false;
        unsigned e = 0;
        if (g < cnt) {
            e = queue[base + g];
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

    for (; need; need &= need - 1) {
        const int c0 = seg_lo + TILE_COLS * (__ffsll((long long)need) - 1);
        f32x2 dq[CPL][KD], cn[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(ca.Ds + int64_t(min(c0 + 64 * u + lane, A - 1)) * DW);
#pragma unroll
            for (int k = 0; k < KD / 2; ++k) {
                const f32x4 v = src[k];
                dq[u][2 * k] = f32x2{v.x, v.y};
                dq[u][2 * k + 1] = f32x2{v.z, v.w};
            }
            f32x2 nc = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < KD; ++k) nc = __builtin_elementwise_fma(dq[u][k], dq[u][k], nc);
            cn[u] = nc * f32x2{-0.5f, -0.5f};
        }
        n_screened += (unsigned long long)nrows * (unsigned long long)max(0, min(TILE_COLS, seg_hi - c0));
#pragma unroll
        for (int t = 0; t < TI; ++t) {  // (rows beyond nrows repeat the last row's record: their pairs fail the position test below)
            const f32x2 *rec = reinterpret_cast<const f32x2 *>(rowdesc + t * RS);
            f32x2 rd[KD];
#pragma unroll
            for (int k = 0; k < KD; ++k) rd[k] = rec[k];
            const f32x2 nr = rec[KD];
            int worst[CPL];
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                f32x2 acc = cn[u];
#pragma unroll
                for (int k = 0; k < KD; ++k) acc = __builtin_elementwise_fma(rd[k], dq[u][k], acc);
                const f32x2 s2 = __builtin_elementwise_fma(acc, f32x2{-2.0f, -2.0f}, nr);
                worst[u] = max(__float_as_int(s2.x), __float_as_int(s2.y));
            }
            if (__builtin_amdgcn_ballot_w64(min(worst[0], worst[1]) <= limit_bits)) {  // (rare) some column of the tile is within the limit
#pragma unroll
                for (int u = 0; u < CPL; ++u) {
                    const int col = c0 + 64 * u + lane;
                    // a pair is visited once: from the row tile at or before the column's, and inside the tile pair that holds both,
                    // by the earlier position
                    const bool pass = worst[u] <= limit_bits && t < nrows && col > p0 + t && col < seg_hi;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                    if (m) {
                        if (pass) queue[qn + __popcll(m & lt_mask)] = (unsigned short)((unsigned(t) << 12) | unsigned(col - seg_lo));
                        qn += __popcll(m);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        while (qn >= a.drain_min) {
            const int cnt = min(qn, 64);
            sign_stage(qn - cnt, cnt);
            qn -= cnt;
        }
    }
    if (qn > 0) {
        // (what is left is below drain_min <= 64)
        sign_stage(0, qn);
    }
    if (qe > 0) exact_stage(0, qe);
    if (lane == 0) {
        count_add(counters, unsigned(slot), CNT_FORMED, n_eval);
        count_add(counters, unsigned(slot), CNT_EXACT, n_exact);
        count_add(counters, unsigned(slot), CNT_SCREENED, n_screened);
    }
}

#ifndef TSC_SORTED_OCC
#define TSC_SORTED_OCC TSC_SIEVE_OCC2
#endif
template <bool F32>
inline __global__ __launch_bounds__(256, TSC_SORTED_OCC) void k_rmsd_sieve_sorted(const double *__restrict__ heavy, const int32_t *__restrict__ act,
                                                                            const double *__restrict__ Gall, const int32_t *__restrict__ cend,
                                                                            int32_t *__restrict__ best, PassCounters *__restrict__ counters,
                                                                            const PruneState *__restrict__ st, SieveArgs a, CullArgs ca, int my_tiles, int n_seg) {
    // a fixed grid walks the (row tile, column segment) items: half of them lie beyond their chunk's end and leave at once, and a
    // workgroup per item would be 10^5 launches of nothing.  The four wavefronts of a workgroup take four consecutive row tiles of
    // one segment (the same column tiles: one trip through the caches), segment after segment
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int groups = (my_tiles + 3) / 4;
    if (ca.xcd_seg) {
        // Row-group RUNS keyed to XCDs.  Workgroups go to the XCDs round-robin (workgroup b runs on XCD b % 8), each XCD has an L2 of its own,
        // and a work item's columns are the 4096-position window behind its rows: with the items in plain order (below) the ~1 300
        // workgroups in flight are consecutive row groups of ONE segment spread over all eight XCDs -- every L2 sees the whole
        // 80 000-position window of descriptors (5 MB against its 4 MB; measured L2 hit rate 64 %).  Here, inside a segment, runs of
        // CULL_XCD_RUN consecutive row groups go to the XCDs in turn: the workgroups an XCD has in flight come from a few runs and share
        // their windows.  (Whole SEGMENTS keyed to XCDs -- round 5's first form -- took the L2 hit rate to 92 % and the fabric reads
        // down 4.6x, and the pass 2.3 - 5x up: in a culled pass the work sits in the segments next to the diagonal, which two XCDs then
        // did alone; profiles/r05_xcd_study.)
        const int x = int(blockIdx.x & 7u);
        const long long j = (long long)(blockIdx.x >> 3);
        const int runs_per_xcd = (((groups + CULL_XCD_RUN - 1) / CULL_XCD_RUN) + 7) / 8;
        const long long per_seg = (long long)runs_per_xcd * CULL_XCD_RUN;
        const int seg = int(j / per_seg);
        const int rem = int(j - (long long)seg * per_seg);
        const int grp = ((rem / CULL_XCD_RUN) * 8 + x) * CULL_XCD_RUN + rem % CULL_XCD_RUN;
        if (seg < n_seg && grp < groups) sieve_item_sorted<16, F32>(heavy, act, Gall, cend, best, counters, st, a, ca, grp * 4 + wid, seg);
        return;
    }
    if ((long long)gridDim.x >= (long long)groups * n_seg) {  // (the usual launch: a workgroup per item -- no loop to carry state across)
        const int seg = int(blockIdx.x / unsigned(groups)), grp = int(blockIdx.x - unsigned(seg) * unsigned(groups));
        if (seg < n_seg) sieve_item_sorted<16, F32>(heavy, act, Gall, cend, best, counters, st, a, ca, grp * 4 + wid, seg);
        return;
    }
    for (long long item = blockIdx.x; item < (long long)groups * n_seg; item += gridDim.x) {
        const int seg = int(item / groups), slot = int(item - (long long)seg * groups) * 4 + wid;
        sieve_item_sorted<16, F32>(heavy, act, Gall, cend, best, counters, st, a, ca, slot, seg);
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace tsc
