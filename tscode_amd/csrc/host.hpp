// host.hpp -- host-side helpers shared by the translation units of libtscode_hip (one .hip per kernel family: ctx, embed, prune,
// pairs_*, adjacent, pipeline).  Functions declared here without a body are defined in exactly one of them; the library is built
// with -fvisibility=hidden, so none of this is exported.
#pragma once

#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>

#include "common.hpp"
#include "embed_clash.hpp"
#include "sieve.hpp"

using namespace tsc;

constexpr int TILE_ROWS = 16;
constexpr int MAX_HP = 32;  // register-tiled kernel only; the sieve kernel takes any h

enum { ALGO_AUTO = 0, ALGO_TILE = 1, ALGO_SIEVE = 2, ALGO_LOCAL = 3 /* reported only: a pass run by the chunk-local kernel */ };

static inline int make_frag_table(const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf, int n_mols, FragTable *ft) {
    TSC_REQUIRE(frag_off && n_atoms && n_conf, "null fragment table");
    TSC_REQUIRE(n_mols >= 1 && n_mols <= MAX_MOLS, "n_mols = %d not in 1..%d", n_mols, MAX_MOLS);
    memset(ft, 0, sizeof(*ft));
    ft->n_mols = n_mols;
    int off = 0;
    for (int m = 0; m < n_mols; ++m) {
        TSC_REQUIRE(n_atoms[m] > 0 && n_conf[m] > 0 && frag_off[m] >= 0, "bad fragment %d", m);
        ft->frag_off[m] = frag_off[m];
        ft->n_atoms[m] = n_atoms[m];
        ft->n_conf[m] = n_conf[m];
        ft->atom_off[m] = off;
        off += n_atoms[m];
    }
    for (int m = n_mols; m <= MAX_MOLS; ++m) ft->atom_off[m] = off;
    ft->n_total = off;
    return 0;
}

static inline int grid_for(int64_t work_items, int per_block, int cap = 256 * 16) {
    return int(std::max<int64_t>(1, std::min<int64_t>(ceil_div<int64_t>(work_items, per_block), cap)));
}

template <typename T>
static inline int upload(tsc_ctx *c, Scratch &s, const T *host, size_t count, T **dev) {
    TSC_TRY(s.get(count ? count : 1, dev));
    if (count) TSC_HIP(hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}

static inline int64_t frags_total_doubles(const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf, int n_mols) {
    int64_t end = 0;
    for (int m = 0; m < n_mols; ++m) end = std::max<int64_t>(end, frag_off[m] + int64_t(n_conf[m]) * n_atoms[m] * 3);
    return end;
}


// Fetch a device scalar that the work enqueued so far has produced WITHOUT waiting for what is enqueued after this call:
// the copy goes to the auxiliary stream behind an event; read_i32_finish waits for that stream only.
static inline int read_i32_begin(tsc_ctx *c, const int32_t *dev) {
    TSC_HIP(hipEventRecord(c->ev_sync, c->stream));
    TSC_HIP(hipStreamWaitEvent(c->aux_stream, c->ev_sync, 0));
    TSC_HIP(hipMemcpyAsync(c->pinned, dev, sizeof(int32_t), hipMemcpyDeviceToHost, c->aux_stream));
    return 0;
}
static inline int read_i32_finish(tsc_ctx *c, int32_t *host_out) {
    TSC_HIP(hipStreamSynchronize(c->aux_stream));
    *host_out = *static_cast<int32_t *>(c->pinned);
    return 0;
}

static inline int read_i32(tsc_ctx *c, const int32_t *dev, int32_t *host_out) {
    TSC_HIP(hipMemcpyAsync(c->pinned, dev, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    *host_out = *static_cast<int32_t *>(c->pinned);
    return 0;
}

// Doubles of a descriptor basis: KD rows per feature family, then the DW projections of the mean feature vector (+ 1 spare)
// ... and, behind the DW projections, the two families' mean squared descriptor distance over the sample (k_descriptor_basis)
static inline size_t basis_doubles(int h) { return size_t(KD) * (n_features(h, 0) + n_features(h, 1)) + DW + NFAM + 1; }
static inline size_t basis_spread_offset(int h) { return size_t(KD) * (n_features(h, 0) + n_features(h, 1)) + DW; }
constexpr size_t PINNED_SPREAD_OFFSET = 8192;  // where a basis' two spread values land in the context's pinned buffer
constexpr size_t PINNED_FLAG_OFFSET = PINNED_SPREAD_OFFSET + 128;  // ... and the culled-or-walked verdicts of the context's runs, one 64-byte line each
constexpr int PINNED_FLAG_SLOTS = 64;
constexpr size_t PINNED_COUNT_OFFSET = PINNED_FLAG_OFFSET + 64 * size_t(PINNED_FLAG_SLOTS);  // ... and the pipeline's count of passing poses (scan.hpp: total_host)
constexpr int64_t AUTO_TILE_MIN_N = 30000;     // a prune of its own (no pipeline around it) spends a synchronisation on the question from here on

// "Would the screen let (almost) every pair through?"  Two structures of the sample lie 2 sum_k lambda_k apart, on average, in a
// family's squared descriptor distance; the screen drops a pair only where a family's distance exceeds h thr^2.  Where BOTH families'
// averages stay below that limit most pairs reach H = p^T q whatever the screen does, and the register-tiled all-pairs kernel, which
// forms H for every pair at 2.5e10 pairs/s, beats the sieve's evaluation stage at 5e9 (profiles/r03_hard_workloads.json: 8 ms
// against 46 on 100 000 structures whose descriptors coincide, cache-free mode).  Either kernel gives the same verdicts.
// Asked in the cache-free mode only: in the reference-exact mode the cache's stop columns end nearly every row early on such an ensemble
// (13 M pair evaluations where the cache-free mode makes 191 M), and with so few pairs the sieve's cheaper passes win (2.6 against 3.6 ms).
static inline bool screen_is_useless(const double *spread, int h, double thr) {
    const double limit = double(h) * thr * thr;
    return spread[0] < limit && spread[1] < limit;  // (false for NaN / +inf: no estimate)
}

// Basis of the descriptors: leading principal axes of the two feature families (sieve.hpp) over `n_samples` structures
// heavy[stride * i], into d_Q (basis_doubles(h)).  Enqueued on `st`; the scratch it takes from `s` must outlive the kernels.
static inline size_t moment_doubles(int h) {  // (MOM_BLOCKS partial matrices per family: k_feature_moments)
    const size_t a = size_t(n_features(h, 0) + 1), b = size_t(n_features(h, 1) + 1);
    return size_t(MOM_BLOCKS) * (a * a + b * b) + 1;  // (+ the two arrival counters of k_feature_moments, in the last double)
}

// (prune.hip) d_moments (optional): moment_doubles(h) doubles already zeroed on `st` by the caller; otherwise taken from `s` and cleared there
int build_basis(tsc_ctx *c, hipStream_t st, Scratch &s, const double *heavy, int h, int n_samples, int64_t stride, double *d_Q,
                unsigned *zero_word = nullptr, double *d_moments = nullptr, double *spread_host = nullptr);

// Descriptors built outside the prune (tsc_pipeline_dev: by the kernel that embeds the passing poses)
struct ExternalDescriptors {
    float *D = nullptr;
    double *G = nullptr;
    unsigned *dmax_bits = nullptr;
    float *heavy32 = nullptr;  // (optional) the float32 copy of the heavy atoms, written by the kernel that embedded them
};


static inline int get_event(tsc_ctx *c, hipEvent_t *e) {
    if (!c->event_pool.empty()) {
        *e = c->event_pool.back();
        c->event_pool.pop_back();
        return 0;
    }
    TSC_HIP(hipEventCreate(e));
    return 0;
}

// (pipeline.hip) the basis that tsc_embed_clash_compact_dev left for the tsc_prune_create that follows, or null
const double *pending_basis(const tsc_ctx *c, int h);

// (prune.hip) one whole run on device data; mask_host (optional) also receives the verdicts, copied before the run's single
// synchronisation (the statistics read-back)
int prune_run(tsc_ctx *c, const double *heavy, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask, uint8_t *mask_host,
              tsc_pass_stats *stats, int *n_passes, const double *basis = nullptr, const ExternalDescriptors *ext = nullptr, int force_algo = -1);
