// scan.hpp -- ordered compaction primitives: exclusive scan of a byte mask, bit packing, row gathers.
// NumPy's structures[mask] keeps order, and the pruning result depends on that order, so every
// compaction here is a stable one driven by an exclusive prefix sum (no atomics).
#pragma once
#include "common.hpp"

namespace tsc {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;                             // bytes per thread: one 64-bit load
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;     // 2048 mask bytes per block

__device__ inline uint64_t load_mask8(const uint8_t *__restrict__ mask, int64_t base, int64_t n) {
    // 8 mask bytes starting at base (base % 8 == 0), zero beyond n; mask is hipMalloc-aligned.
    if (base + 8 <= n) return *reinterpret_cast<const uint64_t *>(mask + base);
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b)
        if (base + b < n) v |= uint64_t(mask[base + b]) << (8 * b);
    return v;
}

__device__ inline int count_nonzero_bytes(uint64_t v) {
    // number of non-zero bytes in v
    uint64_t t = v | (v >> 4);
    t |= t >> 2;
    t |= t >> 1;
    t &= 0x0101010101010101ull;
    return __popcll(t);
}

// pass 1: per-block count of non-zero mask bytes
// gate (optional): the kernels of a speculatively enqueued pass return at once when *gate == 0.
inline __global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const uint8_t *__restrict__ mask, int64_t n,
                                                                   int32_t *__restrict__ bsum, const int *__restrict__ gate) {
    __shared__ int s_w[SCAN_THREADS / WAVE];
    if (gate && *gate == 0) return;
    int64_t base = (int64_t(blockIdx.x) * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
    int c = (base < n) ? count_nonzero_bytes(load_mask8(mask, base, n)) : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// pass 2: pos[i] = number of non-zero mask bytes before i (pos[n] = total); optionally the list of
// kept indices (act_idx[pos[i]] = i) and the mask as a bit array (bit i of mbit = mask[i] != 0).
// (the work of one block; s_w, s_o: SCAN_THREADS / WAVE ints of LDS each)
__device__ inline void scan_write_block(const uint8_t *__restrict__ mask, int64_t n, const int32_t *__restrict__ bsum,
                                        int32_t *__restrict__ pos, int32_t *__restrict__ act_idx, uint8_t *__restrict__ mbit_bytes,
                                        int32_t *__restrict__ total_out, int *s_w, int *s_o, volatile int32_t *total_host = nullptr) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // exclusive offset of this block = sum of the counts of the blocks before it (a few hundred values at most)
    int part = 0;
    for (int i = threadIdx.x; i < int(blockIdx.x); i += SCAN_THREADS) part += bsum[i];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) s_o[wid] = part;
    int64_t base = (int64_t(blockIdx.x) * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
    uint64_t v = (base < n) ? load_mask8(mask, base, n) : 0;
    int c = count_nonzero_bytes(v);
    int incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    const int block_off = s_o[0] + s_o[1] + s_o[2] + s_o[3];
    int run = block_off + woff + incl - c;
    uint8_t bits = 0;
    for (int b = 0; b < SCAN_ITEMS; ++b) {
        int64_t i = base + b;
        if (i >= n) break;
        bool on = ((v >> (8 * b)) & 0xff) != 0;
        if (pos) pos[i] = run;
        if (on) {
            if (act_idx) act_idx[run] = int32_t(i);
            bits |= uint8_t(1u << b);
            ++run;
        }
    }
    if (mbit_bytes && base < n) mbit_bytes[base >> 3] = bits;
    if (base <= n && n < base + SCAN_ITEMS) {  // the thread owning the tail knows the total
        if (pos) pos[n] = run;
        if (total_out) *total_out = run;
        if (total_host) {  // (pinned host memory the host is looking at: no copy command, no event in the stream)
            *total_host = run;
            __threadfence_system();
        }
    }
}

inline __global__ __launch_bounds__(SCAN_THREADS) void k_scan_write(const uint8_t *__restrict__ mask, int64_t n,
                                                              const int32_t *__restrict__ bsum, int32_t *__restrict__ pos,
                                                              int32_t *__restrict__ act_idx, uint8_t *__restrict__ mbit_bytes,
                                                              int32_t *__restrict__ total_out, const int *__restrict__ gate,
                                                              int32_t *total_host = nullptr) {
    __shared__ int s_w[SCAN_THREADS / WAVE];
    __shared__ int s_o[SCAN_THREADS / WAVE];
    if (gate && *gate == 0) return;
    scan_write_block(mask, n, bsum, pos, act_idx, mbit_bytes, total_out, s_w, s_o, total_host);
}

// Enqueue the two passes.  bsum must hold ceil(n / SCAN_TILE) + 1 ints.  pos may be null (then only
// act_idx / mbit are produced); total_dev (optional) receives count_nonzero(mask).
// bsum_current: the caller keeps bsum up to date itself (the prune run does, through k_apply_pass), so the
// counting pass is skipped.
inline int scan_grid_blocks(int64_t n) { return int(ceil_div<int64_t>(n + 1, SCAN_TILE)); }

inline int scan_mask(hipStream_t st, const uint8_t *mask, int64_t n, int32_t *bsum, int32_t *pos, int32_t *act_idx,
                     uint8_t *mbit_bytes, int32_t *total_dev, const int *gate = nullptr, bool bsum_current = false, int32_t *total_host = nullptr) {
    int nb = int(ceil_div<int64_t>(n + 1, SCAN_TILE));  // n + 1: some thread always owns index n (writes pos[n])
    if (!bsum_current) hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, st, mask, n, bsum, gate);
    hipLaunchKernelGGL(k_scan_write, dim3(nb), dim3(SCAN_THREADS), 0, st, mask, n, (const int32_t *)bsum, pos, act_idx, mbit_bytes, total_dev, gate, total_host);
    TSC_HIP(hipGetLastError());
    return 0;
}

inline size_t scan_bsum_count(int64_t n) { return size_t(ceil_div<int64_t>(n + 1, SCAN_TILE)) + 2; }

// dst[r] = src[idx[r]] for rows of row_words 8-byte words; idx == nullptr means identity.
// sel (optional) picks words inside the row: dst[r][w] = src[idx[r]][sel[w]] for w < out_words
// (used for the heavy-atom gather, where a "word" is one coordinate triple = 3 doubles handled as 3 words).
inline __global__ __launch_bounds__(256) void k_gather_rows(const uint64_t *__restrict__ src, const int32_t *__restrict__ idx,
                                                      int64_t n_out, int row_words, const int32_t *__restrict__ sel,
                                                      int out_words, uint64_t *__restrict__ dst) {
    int64_t total = n_out * out_words;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        int64_t r = e / out_words;
        int w = int(e - r * out_words);
        int64_t s = idx ? idx[r] : r;
        int sw = sel ? sel[w] : w;
        dst[e] = src[s * row_words + sw];
    }
}

inline int launch_gather_rows(hipStream_t st, const void *src, const int32_t *idx, int64_t n_out, int row_words,
                              const int32_t *sel, int out_words, void *dst) {
    if (n_out <= 0) return 0;
    int64_t total = n_out * out_words;
    int blocks = int(std::min<int64_t>(ceil_div<int64_t>(total, 256), 256 * 16));
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, st, static_cast<const uint64_t *>(src), idx, n_out,
                       row_words, sel, out_words, static_cast<uint64_t *>(dst));
    TSC_HIP(hipGetLastError());
    return 0;
}

}  // namespace tsc
