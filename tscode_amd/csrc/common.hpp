// common.hpp -- context, error plumbing and the stream-ordered scratch cache of libtscode_hip.
#pragma once

#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/tscode_hip.h"

namespace tsc {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define TSC_HIP(call)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return ::tsc::fail(e_ == hipErrorOutOfMemory ? TSC_ERR_NOMEM : TSC_ERR_HIP, "%s failed: %s (%s:%d)", \
                               #call, hipGetErrorString(e_), __FILE__, __LINE__);                          \
    } while (0)

#define TSC_TRY(call)          \
    do {                       \
        int rc_ = (call);      \
        if (rc_ != 0) return rc_; \
    } while (0)

#define TSC_REQUIRE(cond, ...)                                        \
    do {                                                              \
        if (!(cond)) return ::tsc::fail(TSC_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// No C++ exception crosses the C ABI: every `extern "C"` entry point that returns a status runs its body between these two.
// The containers of tsc_ctx (std::map / std::vector) and `new` are what can throw in this library: std::bad_alloc -> TSC_ERR_NOMEM.
#define TSC_API_GUARD_BEGIN try {
#define TSC_API_GUARD_END                                                                                  \
    }                                                                                                      \
    catch (const std::bad_alloc &) {                                                                       \
        return ::tsc::fail(TSC_ERR_NOMEM, "%s: out of host memory", __func__);                             \
    }                                                                                                      \
    catch (const std::exception &e_) {                                                                     \
        return ::tsc::fail(TSC_ERR_INVALID, "%s: unexpected C++ exception: %s", __func__, e_.what());     \
    }                                                                                                      \
    catch (...) {                                                                                          \
        return ::tsc::fail(TSC_ERR_INVALID, "%s: unexpected C++ exception", __func__);                    \
    }

constexpr int WAVE = 64;

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) {
    return (a + b - 1) / b;
}

}  // namespace tsc

struct tsc_ctx;
static inline bool want_heavy32(const tsc_ctx *c, double heavy_bytes);

// One per (process, device).  All work is enqueued on `stream`; scratch blocks are recycled in
// stream order, so a block handed back by one call can be reused by the next without a sync.
struct tsc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;     // scalar read-backs that should not wait for work enqueued after their producer
    bool own_stream = true;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_sync = nullptr;         // orders the auxiliary stream behind the main one (no timing)
    hipStream_t basis_stream = nullptr;   // the descriptor basis of the pipeline, built beside the clash kernel
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::multimap<size_t, void *> cache;  // free scratch blocks by size
    std::map<void *, size_t> live;        // blocks handed out
    void *pinned = nullptr;               // small pinned host buffer for scalar read-backs
    size_t pinned_bytes = 0;
    int prune_algo = 0;                   // tsc_ctx_set_option("prune_algo"): 0 auto, 1 register-tiled, 2 sieve
    int seg_cols = 0;                     // columns per pair-kernel work item (0 = chosen from the problem size)
    int drain_min = 32;                   // sieve: queued pairs that trigger an evaluation batch between column tiles (swept 16..64 after the row
                                          // loop was trimmed: 32 is 1.3 % ahead of 64 at 1M structures, level elsewhere)
    int sieve_trim = 1;                   // pair kernel: the screen with fewer vector instructions per (row, tile) (norms folded into the fma chain, per-family compares)
    int sieve_mm = 1;                     // pair kernels of a one-rank run with the screen on the matrix cores, 64 rows per work item (mm.hpp, cull_mm.hpp): 0 never (the
                                          // packed-fp32 screen of sieve.hpp / cull.hpp), 1 for runs of at least mm_min_n structures, 2 always
    int64_t mm_min_n = 100000;            // (measured: at 57 000 structures a pass is a few thousand work items and bound by their chains of memory round trips, which
                                          // the longer 64-row items lengthen -- C3 0.80 - 0.89 ms against 0.79; at 483 000 the passes are bound by issue: C4 9.05 -> 7.8 ms)
    int sieve_mm16 = 1;                   // runs below mm_min_n: the walked passes' pair kernel with the matrix-core screen on 16-row items (mm.hpp: k_rmsd_sieve_mm16); 0: the packed-fp32 kernel
    int mm_seg_cols = 0;                  // ... columns per work item of the walked passes' kernel (0: 1024 where rows' ranges reach 2048 columns, else 512)
    int sieve_cpl = 2;                    // columns per lane of the pair kernel's screen: 2 = 128-column tiles at 5 waves/SIMD (default), 4 = 256-column tiles at 4, 1 = 64-column tiles at 6
    int64_t pca_min_n = 6000;             // below this many structures the descriptors use the identity basis (no principal-axis estimate)
    int fuse_descriptors = 1;             // ... and the descriptors by the kernel that embeds the passing poses (needs early_basis)
    int early_basis = 1;                  // tsc_pipeline_dev: descriptor basis from a sample of unfiltered poses, on its own stream
    int clash_first = 0;                  // ... whose chain is enqueued in front of the clash launch (0) or behind it (1: rounds 1 - 3)
    int cull_tile_block = 256;             // culled passes dealt by row tiles: consecutive tiles of the sorted layout per rank and turn
    int stage1_f32 = 1;                   // stage 1 of the pair kernels reads a float32 copy of the coordinates first (sieve.hpp: pair_stage1): 0 never, 2 always,
                                          // 1: from 128 MB of heavy atoms on -- and from 8 MB on where the matrix-core kernels run (want_heavy32 below)
    int local_max_chunk = 384;            // longest chunk (structures) of a pass that the chunk-local kernel takes
    int local_pass = 1;                   // passes with short chunks run in one launch (local_pass.hpp)
    void *dbg_buf = nullptr;              // -DTSC_DBG_STAMPS builds: time stamps of the pair kernel's wavefronts
    size_t dbg_bytes = 0;
    int64_t dbg_waves = 0, dbg_stamp_k = -1;
    int fused_apply = 1;                  // single-rank sieve passes: the pair kernel applies the verdicts tile by tile and closes the pass (sieve.hpp)
    int open_lds_blocks = 1 << 30;        // k_open_rows stages the scan-block prefix in LDS up to this many blocks (tests lower it to take the other path)
    int clash_fp32 = 1;                   // clash verdicts (max_clashes = 0, no counts): packed-fp32 minimum with fp64 fallback
    int clash_lanes = 1;                  // ... of two fragments, the smaller of at most 32 atoms, fused with the embed: one pose per lane (k_clash_lanes)
    int deterministic_basis = 0;          // the descriptor basis from fixed-order sums (sieve.hpp, k_feature_moments): a sharded run sets it -- its ranks
                                          // must derive bit-identical descriptors (the culled passes deal the tiles of a layout sorted by them)
    int cull = 1;                         // large passes of the sieve lay their structures out along a Morton curve and skip tile pairs by bounding box (cull.hpp)
    double cull_min_pairs = 2.0e9;        // ... passes of at least this many pairs (n * (n / k) / 2)
    int64_t cull_grid = 1 << 30;          // workgroups of the culled pair kernel at most (each walks work items with that stride)
    int cull_xcd = 1;                     // 1 (default): the culled pair kernel keys runs of 32 row groups to XCDs (workgroup b runs on XCD b % 8): the workgroups an
                                          // XCD has in flight share their column windows in its L2 (cull.hpp; an experiment of round 5)
    int pass_timing = 0;                  // HIP events per pass: 0 none, 1 on the pair kernel's dispatch, 2 also around the whole pass
    // basis estimated by tsc_embed_clash_compact_dev beside its clash kernel, for the tsc_prune_create that follows (consumed once)
    double *eb_block = nullptr;           // [sample coordinates | moment accumulators | basis]
    int eb_h = 0, eb_samples = 0;
    bool eb_valid = false;
    // descriptors written by tsc_embed_masked_dev with the poses it embeds, for the tsc_prune_create that follows on the same
    // structures (consumed once; the run borrows the buffers until it is destroyed)
    float *xd_D = nullptr;
    double *xd_G = nullptr;
    unsigned *xd_dmax = nullptr;
    float *xd_heavy32 = nullptr;          // ... and the float32 copy of the heavy atoms written beside them (large runs)
    int64_t xd_h32_cap = 0;               // floats
    bool xd_h32_valid = false;
    int64_t xd_cap = 0;                   // structures the buffers hold
    int xd_h = 0;
    const double *xd_heavy = nullptr;     // the heavy-atom array they describe
    bool xd_valid = false;
    int xd_borrowers = 0;                 // live prune runs that read the xd_* buffers (tsc_prune_create borrowed them): no release / regrow meanwhile
    // Bookkeeping that runs of one context driven from DIFFERENT host threads touch (tsc_prune_create / tsc_prune_destroy): the pinned flag
    // words handed to runs (host.hpp: PINNED_FLAG_OFFSET; bit s set = word s is some live run's) and the list of live runs, which
    // tsc_ctx_destroy destroys with the context -- a host whose finalisers run in no particular order cannot leak a run's events.
    std::mutex runs_mutex;
    unsigned long long flag_slots_used = 0;
    std::vector<tsc_prune *> live_runs;
    std::vector<int32_t> sample_host;     // pose indices of the basis sample of the last tsc_pipeline_dev call and their device copy
    int32_t *sample_dev = nullptr;
    double *mom_acc = nullptr;            // moment accumulators of the pipeline's basis chain (k_sample_moments adds, k_descriptor_basis clears)
    size_t mom_cap = 0;                   // doubles
    bool mom_clean = false;               // all zero (as far as the host can tell: every chain enqueued so far ended with the clearing kernel)
    std::vector<int32_t> slot_host;       // heavy-atom slot table of the last tsc_pipeline_dev call and its device copy
    int32_t *slot_dev = nullptr;
    std::vector<hipEvent_t> event_pool;   // recycled timing events of prune runs

    int alloc(size_t bytes, void **out) {
        if (bytes == 0) bytes = 8;
        bytes = (bytes + 255) & ~size_t(255);
        auto it = cache.lower_bound(bytes);
        if (it != cache.end() && it->first <= bytes * 2 + (1u << 20)) {
            *out = it->second;
            live[*out] = it->first;
            cache.erase(it);
            return 0;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            // drop the cache and retry once
            (void)hipGetLastError();
            trim();
            e = hipMalloc(&p, bytes);
            if (e != hipSuccess) return tsc::fail(TSC_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        }
        live[p] = bytes;
        *out = p;
        return 0;
    }
    void release(void *p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;
        cache.emplace(it->second, p);
        live.erase(it);
    }
    void trim() {
        (void)hipStreamSynchronize(stream);
        for (auto &kv : cache) (void)hipFree(kv.second);
        cache.clear();
    }
};

// Does a run over `heavy_bytes` of heavy atoms (n * h * 24) keep the float32 copy that stage 1 of the pair kernels reads?  It pays where the
// candidates' gathers come from HBM (128 MB and more: C4 12.1 -> 10.6 ms in round 4) and where the pair kernel is a chain of round trips rather
// than VALU issue -- the matrix-core kernels: half the bytes and half the trips per evaluation batch (C3: 0.745 -> 0.704 ms); the packed-fp32
// kernel at C3's size lost 2 % to the conversions.
static inline bool want_heavy32(const tsc_ctx *c, double heavy_bytes) {
    if (c->stage1_f32 != 1) return c->stage1_f32 == 2;
    return heavy_bytes >= 128e6 || (heavy_bytes >= 8e6 && (c->sieve_mm != 0 || c->sieve_mm16 != 0));
}


namespace tsc {

// RAII bundle of scratch blocks of one call: everything goes back to the cache on scope exit.
struct Scratch {
    tsc_ctx *ctx;
    std::vector<void *> blocks;
    explicit Scratch(tsc_ctx *c) : ctx(c) {}
    ~Scratch() {
        for (void *p : blocks) ctx->release(p);
    }
    template <typename T>
    int get(size_t count, T **out) {
        void *p = nullptr;
        int rc = ctx->alloc(count * sizeof(T), &p);
        if (rc) return rc;
        blocks.push_back(p);
        *out = static_cast<T *>(p);
        return 0;
    }
};

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace tsc
