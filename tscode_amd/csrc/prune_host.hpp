// prune_host.hpp -- the host-side state of one prune run (tsc_prune) and the launchers of its pair kernels, which live in translation
// units of their own (pairs_tile.hip, pairs_sieve.hip, pairs_sorted.hip: the template instantiations are most of the library's build time).
#pragma once

#include "host.hpp"
#include "rmsd.hpp"
#include "scan.hpp"
#include "local_pass.hpp"
#include "cull.hpp"
#include "mm.hpp"
#include "cull_mm.hpp"

// --------------------------------------------------------------------------------------------------
// K3: prune_conformers_rmsd
//
// A run is a sequence of passes over the schedule of rmsd_pruning.py:186-188.  Nothing in a pass waits for the
// host: the gate of :192 is evaluated on the device (k_pass_step) and every pass that COULD run (20 k < N) is
// enqueued with grids sized for N structures; kernels of a pass that is gated off, and blocks beyond the number
// of still-active structures, return at once.  The host reads the per-pass records once, at the end.

static const double KS[TSC_MAX_PASSES] = {5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1};  // :186-188

static_assert(MAX_SLOTS == TSC_MAX_PASSES, "one cache view per schedule slot");

struct tsc_prune {
    tsc_ctx *ctx = nullptr;
    const double *heavy = nullptr;
    int64_t n = 0, npad = 0;
    int h = 0, hp = 0;
    double thr = 0;
    int mode = 0;
    int algo = ALGO_SIEVE;  // pair kernel of this run
    // device state
    uint8_t *mask = nullptr;
    int32_t *act = nullptr, *cend = nullptr, *best = nullptr;
    int32_t *bsum = nullptr, *boff = nullptr, *tile_cmax = nullptr, *tile_done = nullptr;
    int n_blocks = 0;                      // scan blocks (SCAN_TILE structures each) the mask is ranked by
    unsigned long long *bits = nullptr;    // two bit copies of the mask (the pass in flight reads one, clears removed rows in the other)
    unsigned long long *views = nullptr;   // cache view of every pass of the schedule, each followed by its summary (rmsd.hpp, CacheViews)
    ViewPass *view_pass = nullptr;         // device: the pass of each view (chunk count, chunk size, its division constants)
    int view_of_slot[TSC_MAX_PASSES];      // schedule slot -> view index (-1: the pass can never run)
    int n_views = 0;
    size_t bit_words = 0, dsum_words = 0;
    bool cur_fused = false;                // the open pass is applied by the pair kernel itself
    double *Xr = nullptr, *Xc = nullptr, *G = nullptr;                       // register-tiled kernel
    float *Dall = nullptr;   // sieve kernel: fp32 descriptors of every structure, [n][DW]
    float *Dc = nullptr;     // ... in active order, rewritten by k_open_rows every pass: what the pair kernel reads
    bool mm64 = false;                      // this run takes the 64-row matrix-core kernels (mm.hpp, cull_mm.hpp); else, with records, the 16-row form
    _Float16 *Dh = nullptr;                 // ... and as the float16 records of the matrix-core screen (mm.hpp), in active order too ("sieve_mm")
    double *Gall = nullptr;
    struct Tickets {
        PassTickets pass;
        LocalTickets local;
    } *tickets = nullptr;  // arrival counters of the fused pair kernel and of the chunk-local pass kernel
    bool cur_local = false;           // the open pass ran (whole) in tsc_prune_pass_local
    // culled passes (cull.hpp): allocated when the first one comes up
    int32_t *morton_order = nullptr, *rank_of = nullptr, *crank = nullptr, *cbase = nullptr, *cfill = nullptr, *blk_cnt = nullptr;
    float *Ds = nullptr, *cbox = nullptr, *rbox = nullptr;
    int32_t *cstruct = nullptr;                // the structure at every sorted position (cull_mm.hpp)
    _Float16 *Dhs = nullptr;                   // the float16 records of the matrix-core screen by sorted position (cull_mm.hpp)
    float *heavy32 = nullptr;            // float32 copy of the heavy atoms for stage 1 of the pair kernels (sieve.hpp: pair_stage1)
    bool morton_sorted = false;          // the run's Morton order exists (made when the first pass is really culled)
    // rank-partitioned passes (rmsd.hpp, k_pass_merge): set by tsc_prune_set_partition
    int part_rank = 0, part_world = 1, part_min_chunks = 0;
    unsigned long long *exch = nullptr;  // caller-owned exchange buffer: bit_words words of removed rows + 8 of statistics
    bool cur_range = false;              // the open pass is run by tsc_prune_pass_range / tsc_prune_pass_merge
    int range_ready_slot = -1;           // slot whose row range the device already holds (set by the k_pass_merge before it)
    bool views_split = false;            // partitioned passes have run: the cache views of the remaining passes hold this rank's keys only
    uint8_t *export_mask_host = nullptr;  // set by prune_run: pinned host buffer that receives the mask with the statistics
    unsigned *dmax_bits = nullptr;  // device scalar: largest |descriptor component| as float bits (zeroed by k_init_run)
    PassCounters *counters = nullptr;
    PruneState *state = nullptr;
    PassRecord *records = nullptr;  // [TSC_MAX_PASSES]
    std::vector<void *> blocks;
    // host state
    int next_ks = 0;       // next index into KS to consider
    int cur_slot = -1;     // schedule slot of the pass in flight (-1 = none)
    int last_slot = -1;    // slot of the last pass that was enqueued and not yet closed on the device
    int opened_slot = -1;  // slot that the device has already opened (done by the apply kernel of the pass before it)
    int64_t cur_k = 0;
    bool local_done = false;
    bool slot_used[TSC_MAX_PASSES] = {false};
    hipEvent_t ev[TSC_MAX_PASSES][4] = {{nullptr}};  // per slot: pass begin, pair kernel begin, pair kernel end, pass end
    tsc_pass_stats stats[TSC_MAX_PASSES];
    int n_passes = 0;
    bool collected = false;
    bool borrows_xd = false;   // the descriptors (and the float32 copy) are the context's xd_* buffers, written by tsc_embed_masked_dev: the context
                               // must not release them while this run lives (tsc_ctx::xd_borrowers)
    bool det_desc = false;     // the descriptors were built with fixed-order sums ("deterministic_basis"): every rank of a sharded run that fed
                               // its run the same sample holds the same bits -- what row tiles of a SORTED layout dealt among ranks rely on
    bool auto_tile = false;    // ALGO_TILE was this run's own choice (screen_is_useless on its own basis estimate), not the caller's
    int flag_slot = -1;        // this run's word in the context's pinned buffer (the culled-or-walked verdict of a candidate pass)
};

template <typename T>
static int palloc(tsc_prune *p, size_t count, T **out) {
    void *q = nullptr;
    TSC_TRY(p->ctx->alloc(count * sizeof(T), &q));
    p->blocks.push_back(q);
    *out = static_cast<T *>(q);
    return 0;
}

// The pass that follows the open one (what tsc_prune_next_pass will hand out next; not consumed here): the kernel that
// finishes a pass also closes it and opens this one on the device.
static inline StepArgs next_step_args(const tsc_prune *p, int *next_slot) {
    int nxt = -1;
    for (int s = p->next_ks; s < TSC_MAX_PASSES; ++s) {
        const int64_t k = int64_t(KS[s]);
        if (k == 1 || 20 * k < p->n) {
            nxt = s;
            break;
        }
    }
    *next_slot = nxt;
    return StepArgs{p->cur_slot, nxt, nxt >= 0 ? (long long)KS[nxt] : 0ll, p->algo, p->cur_local ? ALGO_LOCAL : -1};
}

// range_close: the context of the kernels of a rank-partitioned pass -- their last unit leaves the statistics in the exchange buffer
// instead of closing the pass (pass_step_wave)
static inline StepCtx step_ctx(const tsc_prune *p, bool range_close = false) {
    return StepCtx{p->state, p->counters, p->records, p->bsum, p->boff, p->n_blocks, reinterpret_cast<unsigned *>(p->tickets),
                   int(sizeof(*p->tickets) / 128), range_close ? p->exch + p->bit_words : nullptr};
}

// Chunks [c_lo, c_hi) of a pass of k chunks that START inside rank's block [n rank / world, n (rank + 1) / world) of the
// structure axis, and the structures [s_lo, s_hi) they cover (the last chunk of the pass runs to n, rmsd_pruning.py:141-144).
static inline void partition_bounds(int64_t n, int64_t k, int rank, int world, int64_t *c_lo, int64_t *c_hi, int64_t *s_lo, int64_t *s_hi) {
    const int64_t cs = n / k;
    auto first_chunk = [&](int r) { return r <= 0 ? int64_t(0) : (r >= world ? k : std::min<int64_t>(ceil_div<int64_t>(n * r / world, cs), k)); };
    *c_lo = first_chunk(rank), *c_hi = first_chunk(rank + 1);
    *s_lo = *c_lo < k ? *c_lo * cs : n, *s_hi = *c_hi < k ? *c_hi * cs : n;
}
static inline bool pass_is_partitioned(const tsc_prune *p, int64_t k) {
    return p->part_world > 1 && p->exch && p->algo == ALGO_SIEVE && k >= int64_t(p->part_min_chunks) * p->part_world;
}

// Views of the passes AFTER the open one (where the rows it removes leave their cache keys); none in cache-free mode.
static inline CacheViews later_views(const tsc_prune *p) {
    CacheViews cv;
    cv.views = p->views, cv.stride = (long long)(p->bit_words + p->dsum_words), cv.bit_words = int(p->bit_words), cv.n = int(p->n);
    cv.first = p->mode == 0 ? p->view_of_slot[p->cur_slot] + 1 : 0;
    cv.count = p->mode == 0 ? p->n_views : 0;
    cv.pass = p->view_pass;
    return cv;
}

static inline const unsigned long long *view_of_open_pass(const tsc_prune *p) {
    return p->mode == 0 ? p->views + size_t(p->view_of_slot[p->cur_slot]) * (p->bit_words + p->dsum_words) : p->views;
}

static inline ApplyArgs apply_args(const tsc_prune *p) {
    ApplyArgs a;
    a.g = PassGeom{int(p->n), int(p->cur_k), int(p->n / p->cur_k)};
    a.act = p->act, a.cend = p->cend, a.best = p->best, a.mask = p->mask, a.bits = p->bits, a.bit_words = int(p->bit_words);
    a.bsum = p->bsum, a.block_items = SCAN_TILE, a.cv = later_views(p);
    a.exch = p->cur_range ? p->exch : nullptr;
    return a;
}

// ---- the pair kernels' launchers (plain arguments: the callers fill the argument blocks) ----
// k_rmsd_tile<hp, 16> (pairs_tile.hip); hp = 4, 8, ... 32
int launch_rmsd_tile(int hp, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *Xr, const double *Xc, const double *G, const int32_t *cend,
                     int32_t *best, PassCounters *counters, const PruneState *state, const TileArgs &a);
// k_rmsd_sieve<16, cpl, trim, fused, f32> (pairs_sieve.hip: fused, pairs_sieve_plain.hip: not); the shapes that exist: cpl 1 / 2 / 4 untrimmed, cpl 2
// trimmed, and cpl 2 trimmed with f32
int launch_rmsd_sieve_fused(int cpl, bool trim, bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act,
                            const double *Gall, const float *Dc, const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state,
                            const SieveArgs &a, const FusedApply &fa);
int launch_rmsd_sieve_plain(int cpl, bool trim, bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act,
                            const double *Gall, const float *Dc, const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state,
                            const SieveArgs &a, const FusedApply &fa);
// k_rmsd_sieve_mm<fused, f32> (pairs_mm.hip): the screen on the matrix cores, 64 rows per work item
int launch_rmsd_sieve_mm(bool fused, bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                         const float *Dc, const _Float16 *Dh, const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state, const SieveArgs &a,
                         const FusedApply &fa);
// k_rmsd_sieve_sorted<f32> and k_pass_chunks (pairs_sorted.hip)
int launch_rmsd_sieve_sorted(bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                             const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state, const SieveArgs &a, const CullArgs &ca,
                             int my_tiles, int n_seg);
// k_rmsd_sieve_mm16<fused, f32> (pairs_mm.hip): the matrix-core screen for 16-row items (k_rmsd_sieve's grid)
int launch_rmsd_sieve_mm16(bool fused, bool f32, int waves, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                           const _Float16 *Dh, const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state,
                           const SieveArgs &a, const FusedApply &fa);
// k_rmsd_sieve_sorted_mm<f32> (pairs_mm.hip): the culled pass with the screen on the matrix cores
int launch_rmsd_sieve_sorted_mm(bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                                const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state, const SieveArgs &a, const CullArgs &ca,
                                const CullMmArgs &cm, int n_groups, int n_seg);
int launch_pass_chunks(hipStream_t st, unsigned blocks, hipEvent_t e0, hipEvent_t e1, const PassGeom &g, const LocalPassArgs &a, PruneState *state, uint8_t *mask,
                       unsigned long long *bits, int bit_words, const unsigned long long *view, const double *heavy, const double *Gall, const float *Dall,
                       const CacheViews &cv, PassCounters *counters, int32_t *bsum, int block_items, const StepCtx &sc, const StepArgs &sa, LocalTickets *tickets);
