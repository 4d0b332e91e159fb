// pipeline.hip -- the fused device pipeline: embed -> clash mask -> ordered compaction -> RMSD prune on resident buffers
// gfx950 only.  There is deliberately no CPU implementation behind these entry points.
#include "host.hpp"
#include "scan.hpp"

// --------------------------------------------------------------------------------------------------
// pipeline

// heavy_slot[a] = rank of atom a among the heavy atoms, -1 for the others; kept on the device between calls
static int heavy_slot_table(tsc_ctx *c, const FragTable &ft, const int32_t *heavy_idx, int n_heavy, int32_t **d_slot) {
    std::vector<int32_t> slot(size_t(ft.n_total), -1);
    for (int a = 0; a < n_heavy; ++a) {
        TSC_REQUIRE(heavy_idx[a] >= 0 && heavy_idx[a] < ft.n_total && (a == 0 || heavy_idx[a] > heavy_idx[a - 1]),
                    "heavy_idx must be strictly increasing atom indices");
        slot[size_t(heavy_idx[a])] = a;
    }
    if (!(c->slot_dev && c->slot_host == slot)) {  // (same heavy-atom pattern as the last call: no upload)
        if (c->slot_dev) c->release(c->slot_dev);
        c->slot_dev = nullptr;
        void *q = nullptr;
        TSC_TRY(c->alloc(slot.size() * sizeof(int32_t), &q));
        c->slot_dev = static_cast<int32_t *>(q);
        c->slot_host = slot;
        TSC_HIP(hipMemcpyAsync(c->slot_dev, c->slot_host.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    *d_slot = c->slot_dev;
    return 0;
}

// pose indices 0, stride, 2 stride, ... of the basis sample, cached on the device between calls
static int basis_sample_table(tsc_ctx *c, int64_t n_poses, int *n_samples_out) {
    const int n_samples = int(std::min<int64_t>(n_poses, DESC_SAMPLE));
    const int64_t stride = std::max<int64_t>(1, n_poses / n_samples);
    if (!(c->sample_dev && int(c->sample_host.size()) == n_samples && c->sample_host.back() == int32_t(stride * (n_samples - 1)))) {
        if (c->sample_dev) c->release(c->sample_dev);
        c->sample_dev = nullptr;
        c->sample_host.resize(size_t(n_samples));
        for (int i = 0; i < n_samples; ++i) c->sample_host[size_t(i)] = int32_t(stride * i);
        void *q = nullptr;
        TSC_TRY(c->alloc(size_t(n_samples) * sizeof(int32_t), &q));
        c->sample_dev = static_cast<int32_t *>(q);
        TSC_HIP(hipMemcpyAsync(c->sample_dev, c->sample_host.data(), size_t(n_samples) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    *n_samples_out = n_samples;
    return 0;
}

// The descriptor basis of a prune from a sample of unfiltered poses, on the side stream: begin() records the fork point on the main
// stream (the inputs are ordered there) and makes room in the context's persistent block; launch() enqueues the chain -- sample
// embed, moments, basis: about 50 us -- on the side stream and marks the basis valid for the next consumer (tsc_prune_create or
// tsc_embed_masked_dev on this context, which wait for ev_join).  Work enqueued on the main stream between the two hides the chain.
struct BasisFork {
    int n_samples = 0;
    double *sample = nullptr, *moments = nullptr, *basis = nullptr;
};
static int basis_fork_begin(tsc_ctx *c, int64_t n_poses, int n_heavy, BasisFork *bf) {
    *bf = BasisFork();
    if (!(c->early_basis && c->prune_algo != ALGO_TILE)) return 0;
    TSC_TRY(basis_sample_table(c, n_poses, &bf->n_samples));
    const size_t need = size_t(bf->n_samples) * n_heavy * 3 + moment_doubles(n_heavy) + basis_doubles(n_heavy);
    if (!(c->eb_block && c->eb_h == n_heavy && c->eb_samples == bf->n_samples)) {
        TSC_HIP(hipStreamSynchronize(c->basis_stream));
        if (c->eb_block) c->release(c->eb_block);
        c->eb_block = nullptr;
        void *q = nullptr;
        TSC_TRY(c->alloc(need * sizeof(double), &q));
        c->eb_block = static_cast<double *>(q), c->eb_h = n_heavy, c->eb_samples = bf->n_samples;
    }
    bf->sample = c->eb_block, bf->moments = bf->sample + size_t(bf->n_samples) * n_heavy * 3, bf->basis = bf->moments + moment_doubles(n_heavy);
    TSC_HIP(hipEventRecord(c->ev_fork, c->stream));
    return 0;
}
static int basis_fork_launch(tsc_ctx *c, const BasisFork &bf, Scratch &s, const double *frags, const FragTable &ft, const int32_t *conf_idx, const double *rot,
                             const double *pos, const int32_t *d_slot, int n_heavy) {
    if (!bf.basis) return 0;
    TSC_HIP(hipStreamWaitEvent(c->basis_stream, c->ev_fork, 0));
    hipLaunchKernelGGL(k_transform, dim3(grid_for(bf.n_samples, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), c->basis_stream, frags, ft, conf_idx,
                       rot, pos, (const int32_t *)c->sample_dev, int64_t(bf.n_samples), (double *)nullptr, (const int32_t *)d_slot, n_heavy, bf.sample,
                       (const int32_t *)nullptr, bf.moments, int(moment_doubles(n_heavy)));
    TSC_TRY(build_basis(c, c->basis_stream, s, bf.sample, n_heavy, bf.n_samples, 1, bf.basis, nullptr, bf.moments));
    TSC_HIP(hipEventRecord(c->ev_join, c->basis_stream));
    c->eb_valid = true;
    return 0;
}
const double *pending_basis(const tsc_ctx *c, int h) {
    return (c->eb_valid && c->eb_h == h && c->prune_algo != ALGO_TILE) ? c->eb_block + size_t(c->eb_samples) * h * 3 + moment_doubles(h) : nullptr;
}

// Fork the descriptor basis of the prune that will follow from a sample of these poses (all device pointers, as
// tsc_transform_batch_dev) -- a call of its own for hosts that run the clash verdicts and the embedding as separate steps
// (the multi-rank front half of tscode_amd/pipeline.py): enqueue it first, and the chain runs beside whatever follows.
extern "C" __attribute__((visibility("default"))) int tsc_basis_from_poses_dev(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                                                               const int32_t *n_conf, int n_mols, const int32_t *conf_idx, const double *rot,
                                                                               const double *pos, int64_t n_poses, const int32_t *heavy_idx, int n_heavy) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && conf_idx && rot && pos && heavy_idx, "tsc_basis_from_poses_dev: null argument");
    TSC_REQUIRE(n_poses > 0 && n_poses < INT32_MAX, "bad n_poses");
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    TSC_REQUIRE(n_heavy > 0 && n_heavy <= ft.n_total, "bad n_heavy");
    DeviceGuard guard(c->device);
    Scratch s(c);
    int32_t *d_slot;
    TSC_TRY(heavy_slot_table(c, ft, heavy_idx, n_heavy, &d_slot));
    BasisFork bf;
    TSC_TRY(basis_fork_begin(c, n_poses, n_heavy, &bf));
    return basis_fork_launch(c, bf, s, frags, ft, conf_idx, rot, pos, d_slot, n_heavy);
    TSC_API_GUARD_END
}

// The poses selected by a mask that is ALREADY on the device (clash verdicts gathered from every rank, say), embedded in order:
// structures f64[n_sel, n_atoms, 3] and / or heavy f64[n_sel, n_heavy, 3] (either may be NULL, not both).  With `heavy`, and a
// basis pending on this context (tsc_basis_from_poses_dev), the kernel also writes the descriptors of the prune that follows; the
// next tsc_prune_create on this context over the same `heavy` array takes them instead of reading the coordinates back.
// n_sel_host (optional): the number of selected poses; the call synchronises for it while the embed runs.
extern "C" __attribute__((visibility("default"))) int tsc_embed_masked_dev(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                                                           const int32_t *n_conf, int n_mols, const int32_t *conf_idx, const double *rot,
                                                                           const double *pos, int64_t n_poses, const uint8_t *mask, const int32_t *heavy_idx,
                                                                           int n_heavy, double *structures, double *heavy, int64_t *n_sel_host) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && conf_idx && rot && pos && heavy_idx && mask && (structures || heavy), "tsc_embed_masked_dev: null argument");
    TSC_REQUIRE(n_poses >= 0 && n_poses < INT32_MAX, "bad n_poses");
    if (n_sel_host) *n_sel_host = 0;
    if (n_poses == 0) return 0;
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    TSC_REQUIRE(n_heavy > 0 && n_heavy <= ft.n_total, "bad n_heavy");
    DeviceGuard guard(c->device);
    hipStream_t st = c->stream;
    Scratch s(c);
    int32_t *d_slot, *bsum, *act, *total;
    TSC_TRY(heavy_slot_table(c, ft, heavy_idx, n_heavy, &d_slot));
    TSC_TRY(s.get(scan_bsum_count(n_poses), &bsum));
    TSC_TRY(s.get(size_t(n_poses), &act));
    TSC_TRY(s.get(1, &total));
    TSC_TRY(scan_mask(st, mask, n_poses, bsum, nullptr, act, nullptr, total));
    if (n_sel_host) TSC_TRY(read_i32_begin(c, total));
    const double *basis = heavy ? pending_basis(c, n_heavy) : nullptr;
    c->xd_valid = false;
    if (basis && c->fuse_descriptors && transform_describe_lds_bytes(ft.n_mols, n_heavy) <= 64 * 1024) {
        if (!(c->xd_D && c->xd_cap >= n_poses)) {
            if (c->xd_borrowers > 0)
                return fail(TSC_ERR_STATE, "tsc_embed_masked_dev: %d live prune run(s) still read the descriptor buffers of an earlier call, which %lld poses "
                                           "would outgrow: destroy them first (tsc_prune_destroy)", c->xd_borrowers, (long long)n_poses);
            for (void *q : {static_cast<void *>(c->xd_D), static_cast<void *>(c->xd_G), static_cast<void *>(c->xd_dmax)})
                if (q) c->release(q);
            c->xd_D = nullptr, c->xd_G = nullptr, c->xd_dmax = nullptr, c->xd_cap = 0;
            void *q = nullptr;
            TSC_TRY(c->alloc(size_t(n_poses) * DW * sizeof(float), &q));
            c->xd_D = static_cast<float *>(q);
            TSC_TRY(c->alloc(size_t(n_poses) * sizeof(double), &q));
            c->xd_G = static_cast<double *>(q);
            TSC_TRY(c->alloc(4 * sizeof(unsigned), &q));
            c->xd_dmax = static_cast<unsigned *>(q);
            c->xd_cap = n_poses;
        }
        const int nf0 = n_features(n_heavy, 0), nf1 = n_features(n_heavy, 1);
        // the float32 copy for stage 1 of the pair kernels, where the run can be large enough for it (the count is not known yet)
        float *h32 = nullptr;
        if (want_heavy32(c, double(n_poses) * n_heavy * 24.0)) {
            const int64_t need = n_poses * heavy32_pitch(n_heavy);
            if (c->xd_h32_cap < need) {
                if (c->xd_borrowers > 0)
                    return fail(TSC_ERR_STATE, "tsc_embed_masked_dev: %d live prune run(s) still read the float32 copy of an earlier call: destroy them first",
                                c->xd_borrowers);
                if (c->xd_heavy32) c->release(c->xd_heavy32);
                c->xd_heavy32 = nullptr, c->xd_h32_cap = 0;
                void *q = nullptr;
                TSC_TRY(c->alloc(size_t(need) * sizeof(float), &q));
                c->xd_heavy32 = static_cast<float *>(q), c->xd_h32_cap = need;
            }
            h32 = c->xd_heavy32;
        }
        TSC_HIP(hipMemsetAsync(c->xd_dmax, 0, sizeof(unsigned), st));
        TSC_HIP(hipStreamWaitEvent(st, c->ev_join, 0));  // the basis from the side stream
        hipLaunchKernelGGL(k_transform_describe, dim3(grid_for(n_poses, TR_POSES, 256 * 64)), dim3(256), transform_describe_lds_bytes(ft.n_mols, n_heavy), st,
                           frags, ft, conf_idx, rot, pos, (const int32_t *)act, structures, (const int32_t *)d_slot, n_heavy, heavy, (const int32_t *)total,
                           nf0, nf1, basis, (const double *)(basis + size_t(KD) * (nf0 + nf1)), c->xd_D, c->xd_G, c->xd_dmax, h32);
        c->xd_valid = true, c->xd_h = n_heavy, c->xd_heavy = heavy, c->xd_h32_valid = h32 != nullptr;
        c->eb_valid = false;  // (the basis went into the descriptors)
    } else {
        hipLaunchKernelGGL(k_transform, dim3(grid_for(n_poses, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), st, frags, ft, conf_idx, rot, pos,
                           (const int32_t *)act, int64_t(0), structures, (const int32_t *)d_slot, heavy ? n_heavy : 0, heavy, (const int32_t *)total);
    }
    TSC_HIP(hipGetLastError());
    if (n_sel_host) {
        int32_t n_sel = 0;
        TSC_TRY(read_i32_finish(c, &n_sel));
        *n_sel_host = n_sel;
    }
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_embed_clash_compact_dev(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                                                                  const int32_t *n_conf, int n_mols, const int32_t *conf_idx, const double *rot,
                                                                                  const double *pos, int64_t n_poses, const int32_t *heavy_idx, int n_heavy,
                                                                                  double clash_thresh, int64_t max_clashes, uint8_t *clash_mask,
                                                                                  double *structures, double *heavy, int64_t *n_pass_host) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && conf_idx && rot && pos && heavy_idx && clash_mask && heavy && n_pass_host, "tsc_embed_clash_compact_dev: null argument");
    TSC_REQUIRE(n_poses >= 0 && n_poses < INT32_MAX, "bad n_poses");
    *n_pass_host = 0;
    if (n_poses == 0) return 0;
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    TSC_REQUIRE(n_heavy > 0 && n_heavy <= ft.n_total, "bad n_heavy");
    DeviceGuard guard(c->device);
    hipStream_t st = c->stream;
    Scratch s(c);
    int32_t *d_slot, *bsum, *act, *total;
    TSC_TRY(heavy_slot_table(c, ft, heavy_idx, n_heavy, &d_slot));
    TSC_TRY(s.get(scan_bsum_count(n_poses), &bsum));
    TSC_TRY(s.get(size_t(n_poses), &act));
    TSC_TRY(s.get(1, &total));
    // The descriptor basis of the prune that follows (tsc_prune_create on the gathered survivors), from a sample of THIS block's
    // unfiltered poses, on the side stream beside the clash kernel -- as tsc_pipeline_dev does.  Every rank of a sharded run
    // ends up with a basis of its own; any basis gives the same verdicts.
    BasisFork bf;
    TSC_TRY(basis_fork_begin(c, n_poses, n_heavy, &bf));
    TSC_TRY(tsc_embed_clash_mask_dev(c, frags, frag_off, n_atoms, n_conf, n_mols, conf_idx, rot, pos, n_poses, clash_thresh, max_clashes, clash_mask, nullptr));
    TSC_TRY(basis_fork_launch(c, bf, s, frags, ft, conf_idx, rot, pos, d_slot, n_heavy));
    TSC_TRY(scan_mask(st, clash_mask, n_poses, bsum, nullptr, act, nullptr, total));
    TSC_TRY(read_i32_begin(c, total));
    hipLaunchKernelGGL(k_transform, dim3(grid_for(n_poses, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), st, frags, ft, conf_idx, rot, pos, (const int32_t *)act,
                       int64_t(0), structures, (const int32_t *)d_slot, n_heavy, heavy, (const int32_t *)total);
    TSC_HIP(hipGetLastError());
    int32_t n_pass = 0;
    TSC_TRY(read_i32_finish(c, &n_pass));
    *n_pass_host = n_pass;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_pipeline_dev(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf,
                                int n_mols, const int32_t *conf_idx, const double *rot, const double *pos, int64_t n_poses,
                                const int32_t *heavy_idx, int n_heavy, double clash_thresh, int64_t max_clashes, double rmsd_thr, int mode,
                                uint8_t *clash_mask, double *structures, uint8_t *keep_mask, uint8_t *keep_mask_host, int64_t *n_pass_host,
                                int64_t *n_keep_host, tsc_pass_stats *stats, int *n_passes, float *timings_ms) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && conf_idx && rot && pos && heavy_idx && clash_mask && structures && keep_mask, "tsc_pipeline_dev: null argument");
    TSC_REQUIRE(n_poses > 0 && n_poses < INT32_MAX, "bad n_poses");
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    TSC_REQUIRE(n_heavy > 0 && n_heavy <= ft.n_total, "bad n_heavy");
    DeviceGuard guard(c->device);
    hipStream_t st = c->stream;
    Scratch s(c);
    // stage timings only on request ("pass_timing" = 2): four events in the stream cost about 4 us each
    const bool timed = timings_ms && c->pass_timing >= 2;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    struct EvGuard {
        tsc_ctx *c;
        hipEvent_t *e;
        ~EvGuard() {
            for (int i = 0; i < 4; ++i)
                if (e[i]) c->event_pool.push_back(e[i]);
        }
    } evg{c, ev};
    if (timed)
        for (auto &e : ev) TSC_TRY(get_event(c, &e));
    int32_t *d_slot, *bsum, *act, *total;
    double *d_heavy;
    TSC_TRY(heavy_slot_table(c, ft, heavy_idx, n_heavy, &d_slot));
    TSC_TRY(s.get(scan_bsum_count(n_poses), &bsum));
    TSC_TRY(s.get(size_t(n_poses), &act));
    TSC_TRY(s.get(1, &total));
    if (timed) TSC_HIP(hipEventRecord(ev[0], st));
    // The descriptor basis of the prune (sieve.hpp) from a sample of the UNFILTERED poses, on its own stream beside the clash
    // kernel: 3 small launches and a one-wavefront kernel (about 45 us of latency at C3) leave the critical path.  Any
    // orthonormal basis gives the same verdicts; poses that fail the clash check are as good a sample of the geometry.
    double *d_basis = nullptr;
    ExternalDescriptors ext;  // set when the embedding of the passing poses also writes their descriptors
    struct BasisJoin {  // declared after the scratch: whatever path leaves this function, the side stream is idle before its blocks go back
        tsc_ctx *c;
        bool pending;
        ~BasisJoin() {
            if (pending) (void)hipStreamSynchronize(c->basis_stream);
        }
    } basis_join{c, false};
    int n_samples = 0;
    double *d_sample = nullptr, *d_moments = nullptr;
    bool fused_sample = false;
    if (c->early_basis && c->prune_algo != ALGO_TILE) {
        TSC_TRY(basis_sample_table(c, n_poses, &n_samples));
        TSC_TRY(s.get(size_t(n_samples) * n_heavy * 3, &d_sample));
        TSC_TRY(s.get(basis_doubles(n_heavy), &d_basis));
        if (c->fuse_descriptors && transform_describe_lds_bytes(ft.n_mols, n_heavy) <= 64 * 1024) {
            TSC_TRY(s.get(size_t(n_poses) * DW, &ext.D));
            TSC_TRY(s.get(size_t(n_poses), &ext.G));
            TSC_TRY(s.get(4, &ext.dmax_bits));
            // (the float32 copy for stage 1 of the pair kernels, where the run can be large enough for it: the count is not known yet)
            if (want_heavy32(c, double(n_poses) * n_heavy * 24.0))
                TSC_TRY(s.get(size_t(n_poses) * heavy32_pitch(n_heavy), &ext.heavy32));
        }
        // one device: the sample is embedded and reduced by ONE kernel into accumulators the context keeps zero between runs (sieve.hpp,
        // k_sample_moments); a sharded run takes the fixed-order sums instead (k_transform + k_feature_moments, "deterministic_basis")
        fused_sample = !c->deterministic_basis && sample_moments_lds_bytes(ft.n_mols, n_heavy) <= 64 * 1024;
        if (fused_sample) {
            const size_t a = size_t(n_features(n_heavy, 0) + 1), b = size_t(n_features(n_heavy, 1) + 1), need = a * a + b * b;
            if (c->mom_cap < need) {
                if (c->mom_acc) c->release(c->mom_acc);
                c->mom_acc = nullptr, c->mom_cap = 0;
                void *q = nullptr;
                TSC_TRY(c->alloc(need * sizeof(double), &q));
                c->mom_acc = static_cast<double *>(q), c->mom_cap = need, c->mom_clean = false;
            }
            if (!c->mom_clean) TSC_HIP(hipMemsetAsync(c->mom_acc, 0, c->mom_cap * sizeof(double), st));
        } else {
            TSC_TRY(s.get(moment_doubles(n_heavy), &d_moments));
        }
        TSC_HIP(hipEventRecord(c->ev_fork, st));  // the inputs (and the tables above) are ordered on the main stream
    }
    auto enqueue_basis_chain = [&]() -> int {
        // The side stream's chain (sample embed + moments, basis: 37 us + its event's way back across queues) is enqueued IN FRONT of the
        // clash launch since round 4: with the clash kernel at 24 us (k_clash_lanes; 60 before) the chain is what the embed of the passing
        // poses waits for, and every microsecond of host time in front of it is on the critical path ("clash_first" 1: the old order).
        TSC_HIP(hipStreamWaitEvent(c->basis_stream, c->ev_fork, 0));
        basis_join.pending = true;
        // (the two families' descriptor spread is written to pinned host memory by the basis kernel itself: no copy, no wait -- the host
        // pre-sets "no estimate" and looks after it has fetched the count below)
        double *spread_host = reinterpret_cast<double *>(static_cast<char *>(c->pinned) + PINNED_SPREAD_OFFSET);
        spread_host[0] = spread_host[1] = __builtin_inf();
        if (fused_sample) {
            const int nf0 = n_features(n_heavy, 0), nf1 = n_features(n_heavy, 1);
            c->mom_clean = false;
            hipLaunchKernelGGL(k_sample_moments, dim3(ceil_div(ceil_div(n_samples, TR_POSES), SM_CHUNKS), NFAM), dim3(256), sample_moments_lds_bytes(ft.n_mols, n_heavy), c->basis_stream,
                               frags, ft, conf_idx, rot, pos, (const int32_t *)c->sample_dev, n_samples, (const int32_t *)d_slot, n_heavy, nf0, nf1, c->mom_acc,
                               c->mom_acc + size_t(nf0 + 1) * (nf0 + 1));
            hipLaunchKernelGGL(k_descriptor_basis, dim3(NFAM), dim3(64), 0, c->basis_stream, (const double *)c->mom_acc,
                               (const double *)(c->mom_acc + size_t(nf0 + 1) * (nf0 + 1)), nf0, nf1, n_samples, d_basis, d_basis + size_t(KD) * (nf0 + nf1),
                               ext.dmax_bits, spread_host, 1);
            TSC_HIP(hipGetLastError());
            c->mom_clean = true;
        } else {
            hipLaunchKernelGGL(k_transform, dim3(grid_for(n_samples, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), c->basis_stream, frags, ft,
                               conf_idx, rot, pos, (const int32_t *)c->sample_dev, int64_t(n_samples), (double *)nullptr, (const int32_t *)d_slot, n_heavy, d_sample,
                               (const int32_t *)nullptr, d_moments, int(moment_doubles(n_heavy)));
            TSC_TRY(build_basis(c, c->basis_stream, s, d_sample, n_heavy, n_samples, 1, d_basis, ext.dmax_bits, d_moments, spread_host));
        }
        TSC_HIP(hipEventRecord(c->ev_join, c->basis_stream));
        return 0;
    };
    if (d_basis && !c->clash_first) TSC_TRY(enqueue_basis_chain());
    // K1+K2 fused verdicts
    TSC_TRY(tsc_embed_clash_mask_dev(c, frags, frag_off, n_atoms, n_conf, n_mols, conf_idx, rot, pos, n_poses, clash_thresh, max_clashes,
                                     clash_mask, nullptr));
    if (timed) TSC_HIP(hipEventRecord(ev[1], st));
    if (d_basis && c->clash_first) TSC_TRY(enqueue_basis_chain());
    // ordered compaction: embed only the passing poses, all atoms + heavy atoms
    // (the count goes to the host through pinned memory the scan kernel writes itself: an event record + a copy on a second stream in front
    // of the next launch cost the stream two packets, about 6 us each)
    volatile int32_t *count_host = reinterpret_cast<volatile int32_t *>(static_cast<char *>(c->pinned) + PINNED_COUNT_OFFSET);
    *count_host = -1;
    TSC_TRY(scan_mask(st, clash_mask, n_poses, bsum, nullptr, act, nullptr, total, nullptr, false, const_cast<int32_t *>(count_host)));
    // the passing poses are embedded (all atoms + heavy atoms) by a launch sized for every pose that reads the count on the
    // device: it runs while the host fetches the count it needs to set up the prune (the schedule depends on it)
    TSC_TRY(s.get(size_t(n_poses) * n_heavy * 3, &d_heavy));
    if (ext.D) {
        const int nf0 = n_features(n_heavy, 0), nf1 = n_features(n_heavy, 1);
        TSC_HIP(hipStreamWaitEvent(st, c->ev_join, 0));  // the basis (and the cleared maximum) from the side stream
        hipLaunchKernelGGL(k_transform_describe, dim3(grid_for(n_poses, TR_POSES, 256 * 64)), dim3(256), transform_describe_lds_bytes(ft.n_mols, n_heavy), st,
                           frags, ft, conf_idx, rot, pos, (const int32_t *)act, structures, (const int32_t *)d_slot, n_heavy, d_heavy, (const int32_t *)total,
                           nf0, nf1, (const double *)d_basis, (const double *)(d_basis + size_t(KD) * (nf0 + nf1)), ext.D, ext.G, ext.dmax_bits, ext.heavy32);
    } else {
        hipLaunchKernelGGL(k_transform, dim3(grid_for(n_poses, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), st, frags, ft, conf_idx, rot, pos,
                           (const int32_t *)act, int64_t(0), structures, (const int32_t *)d_slot, n_heavy, d_heavy, (const int32_t *)total);
    }
    TSC_HIP(hipGetLastError());
    int32_t n_pass = -1;
    {   // the host looks until the scan kernel has written (a device that never does -- a fault -- is left to the copy path's error after 5 s)
        const auto t_look = std::chrono::steady_clock::now();
        for (unsigned spins = 0; (n_pass = *count_host) < 0; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 0xffffu) == 0xffffu && std::chrono::steady_clock::now() - t_look > std::chrono::seconds(5)) {
                TSC_TRY(read_i32(c, total, &n_pass));
                break;
            }
        }
    }
    if (n_pass_host) *n_pass_host = n_pass;
    int64_t n_keep = 0;
    int np = 0;
    if (d_basis) TSC_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
    // automatic kernel choice: where the sample's descriptors hardly differ the screen separates nothing and the all-pairs kernel is the
    // faster route (screen_is_useless).  The basis kernel writes the two spreads to pinned memory; the HOST has to wait for that kernel
    // itself -- the stream-side wait above orders only the stream, and since round 4 the chain (about 37 us + its event's way across
    // queues) is no shorter than the clash kernel + scan it runs beside: the count can be there before the spreads are.  A read of the
    // pre-set +inf would not change a verdict, only the kernel choice (sieve where the all-pairs kernel is 5x faster) -- from run to run.
    int force_algo = -1;
    if (d_basis && c->prune_algo == ALGO_AUTO && mode == 1 && n_heavy <= MAX_HP) {
        TSC_HIP(hipEventSynchronize(c->ev_join));
        const volatile double *sh = reinterpret_cast<const volatile double *>(static_cast<const char *>(c->pinned) + PINNED_SPREAD_OFFSET);
        const double spread[NFAM] = {sh[0], sh[1]};
        if (screen_is_useless(spread, n_heavy, rmsd_thr)) force_algo = ALGO_TILE;
    }
    const bool sieve_run = force_algo != ALGO_TILE;
    if (n_pass > 0) {
        if (timed) TSC_HIP(hipEventRecord(ev[2], st));
        TSC_TRY(prune_run(c, d_heavy, n_pass, n_heavy, rmsd_thr, mode, keep_mask, keep_mask_host, stats, &np, sieve_run ? d_basis : nullptr,
                          (sieve_run && ext.D) ? &ext : nullptr, force_algo));
        for (int i = 0; i < np; ++i) n_keep = stats ? stats[i].n_active_after : 0;
        if (!stats) {  // count survivors without the stats array
            TSC_TRY(scan_mask(st, keep_mask, n_pass, bsum, nullptr, nullptr, nullptr, total));
            int32_t t = 0;
            TSC_TRY(read_i32(c, total, &t));
            n_keep = t;
        }
    } else if (timed) {
        TSC_HIP(hipEventRecord(ev[2], st));
    }
    if (timed) {
        TSC_HIP(hipEventRecord(ev[3], st));
        TSC_HIP(hipEventSynchronize(ev[3]));
    } else {
        TSC_HIP(hipStreamSynchronize(st));
    }
    if (n_passes) *n_passes = np;
    if (n_keep_host) *n_keep_host = n_keep;
    if (timings_ms && !timed) timings_ms[0] = timings_ms[1] = timings_ms[2] = timings_ms[3] = 0.0f;
    if (timed) {
        TSC_HIP(hipEventElapsedTime(&timings_ms[0], ev[0], ev[1]));
        TSC_HIP(hipEventElapsedTime(&timings_ms[1], ev[1], ev[2]));
        TSC_HIP(hipEventElapsedTime(&timings_ms[2], ev[2], ev[3]));
        TSC_HIP(hipEventElapsedTime(&timings_ms[3], ev[0], ev[3]));
    }
    return 0;
    TSC_API_GUARD_END
}
