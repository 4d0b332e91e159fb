// prune.hip -- K3: prune_conformers_rmsd -- the run, its passes, the sharded protocol
// gfx950 only.  There is deliberately no CPU implementation behind these entry points.
#include "prune_host.hpp"

// --------------------------------------------------------------------------------------------------
// K3: pairs

extern "C" __attribute__((visibility("default"))) int tsc_rmsd_pairs_dev(tsc_ctx *c, const double *heavy, int64_t n_structs, int h, const int32_t *pairs, int64_t n_pairs,
                                  double *rmsd, double *maxdev) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && heavy && pairs && rmsd && maxdev, "tsc_rmsd_pairs_dev: null argument");
    TSC_REQUIRE(n_structs >= 0 && h > 0 && n_pairs >= 0, "bad sizes");
    if (n_pairs == 0) return 0;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_rmsd_pairs, dim3(grid_for(n_pairs, 256)), dim3(256), 0, c->stream, heavy, h, pairs, n_pairs, rmsd, maxdev);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_rmsd_pairs(tsc_ctx *c, const double *heavy, int64_t n_structs, int h, const int32_t *pairs, int64_t n_pairs,
                              double *rmsd, double *maxdev) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && heavy && pairs && rmsd && maxdev, "tsc_rmsd_pairs: null argument");
    TSC_REQUIRE(n_structs >= 0 && h > 0 && n_pairs >= 0, "bad sizes");
    for (int64_t k = 0; k < 2 * n_pairs; ++k)
        TSC_REQUIRE(pairs[k] >= 0 && pairs[k] < n_structs, "pairs[%lld] = %d out of range", (long long)k, pairs[k]);
    if (n_pairs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_heavy, *d_r, *d_m;
    int32_t *d_pairs;
    TSC_TRY(upload(c, s, heavy, size_t(n_structs) * h * 3, &d_heavy));
    TSC_TRY(upload(c, s, pairs, size_t(n_pairs) * 2, &d_pairs));
    TSC_TRY(s.get(size_t(n_pairs), &d_r));
    TSC_TRY(s.get(size_t(n_pairs), &d_m));
    TSC_TRY(tsc_rmsd_pairs_dev(c, d_heavy, n_structs, h, d_pairs, n_pairs, d_r, d_m));
    TSC_HIP(hipMemcpyAsync(rmsd, d_r, size_t(n_pairs) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(maxdev, d_m, size_t(n_pairs) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_screen_mm_values(tsc_ctx *c, const float *D, int64_t n, double limit, float *S, int32_t *limit_bits,
                                                                           float *out_scale) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && D && S && limit_bits && out_scale, "tsc_screen_mm_values: null argument");
    TSC_REQUIRE(n >= 1 && n <= (1 << 20), "n = %lld not supported", (long long)n);
    DeviceGuard guard(c->device);
    Scratch s(c);
    float dmax = 0.0f;
    for (int64_t e = 0; e < n * DW; ++e) dmax = std::fabs(D[e]) > dmax || std::isnan(D[e]) ? (std::isnan(D[e]) ? NAN : std::fabs(D[e])) : dmax;
    unsigned bits;
    memcpy(&bits, &dmax, sizeof(bits));
    if (std::isnan(dmax)) bits = NONFINITE_BITS;
    float *d_D, *d_S;
    unsigned *d_bits;
    int *d_lim;
    _Float16 *d_rec;
    TSC_TRY(upload(c, s, D, size_t(n) * DW, &d_D));
    TSC_TRY(upload(c, s, &bits, 1, &d_bits));
    TSC_TRY(s.get(size_t(n) * MM_REC_HALVES, &d_rec));
    TSC_TRY(s.get(size_t(NFAM) * MM_ROWS * n, &d_S));
    TSC_TRY(s.get(1, &d_lim));
    hipLaunchKernelGGL(k_mm_records, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (const float *)d_D, n, (const unsigned *)d_bits, d_rec);
    hipLaunchKernelGGL(k_mm_screen_dump, dim3(unsigned(ceil_div<int64_t>(n, MM_STEP))), dim3(64), 0, c->stream, (const _Float16 *)d_rec, int(n), (const unsigned *)d_bits,
                       limit, d_S, d_lim);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(S, d_S, size_t(NFAM) * MM_ROWS * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(limit_bits, d_lim, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    *out_scale = mm_scale(bits);
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_destroy(tsc_prune *p) {
    TSC_API_GUARD_BEGIN
    if (!p) return 0;
    DeviceGuard guard(p->ctx->device);
    {
        std::lock_guard<std::mutex> lock(p->ctx->runs_mutex);
        if (p->borrows_xd && p->ctx->xd_borrowers > 0) --p->ctx->xd_borrowers;
        if (p->flag_slot >= 0) p->ctx->flag_slots_used &= ~(1ull << p->flag_slot);
        auto &lr = p->ctx->live_runs;
        lr.erase(std::remove(lr.begin(), lr.end(), p), lr.end());
    }
    for (void *q : p->blocks) p->ctx->release(q);
    for (auto &slot : p->ev)
        for (hipEvent_t e : slot)
            if (e) p->ctx->event_pool.push_back(e);
    delete p;
    return 0;
    TSC_API_GUARD_END
}


// d_moments (optional): moment_doubles(h) doubles already zeroed on `st` by the caller; otherwise taken from `s` and cleared here
int build_basis(tsc_ctx *c, hipStream_t st, Scratch &s, const double *heavy, int h, int n_samples, int64_t stride, double *d_Q,
                       unsigned *zero_word, double *d_moments, double *spread_host) {
    const int nf[NFAM] = {n_features(h, 0), n_features(h, 1)};
    const size_t q_doubles = size_t(KD) * (nf[0] + nf[1]);
    double *d_M[NFAM], *d_zero;
    // the moment matrices of both families in one block (one memset)
    const size_t m0 = size_t(MOM_BLOCKS) * (nf[0] + 1) * (nf[0] + 1), m1 = size_t(MOM_BLOCKS) * (nf[1] + 1) * (nf[1] + 1);
    if (d_moments) {
        d_zero = d_moments;
    } else {
        TSC_TRY(s.get(m0 + m1 + 1, &d_zero));
        // (the arrival counters of the deterministic form -- its partial matrices are written whole --, or the matrices the fast form adds into)
        if (c->deterministic_basis) TSC_HIP(hipMemsetAsync(d_zero + m0 + m1, 0, sizeof(double), st));
        else TSC_HIP(hipMemsetAsync(d_zero, 0, (m0 + m1 + 1) * sizeof(double), st));
    }
    d_M[0] = d_zero, d_M[1] = d_zero + m0;
    {
        const size_t lds = size_t(32) * (std::max(nf[0], nf[1]) + 1) * sizeof(double);
        if (lds > 64 * 1024)
            TSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_feature_moments), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        if (c->deterministic_basis)
            hipLaunchKernelGGL(k_feature_moments, dim3(MOM_BLOCKS, NFAM), dim3(256), lds, st, heavy, h, nf[0], nf[1], stride, n_samples, d_M[0], d_M[1],
                               reinterpret_cast<unsigned *>(d_zero + m0 + m1));
        else
            hipLaunchKernelGGL(k_feature_moments, dim3(ceil_div(n_samples, 32), NFAM), dim3(256), lds, st, heavy, h, nf[0], nf[1], stride, n_samples, d_M[0],
                               d_M[1], (unsigned *)nullptr);
    }
    hipLaunchKernelGGL(k_descriptor_basis, dim3(NFAM), dim3(64), 0, st, (const double *)d_M[0], (const double *)d_M[1], nf[0], nf[1], n_samples, d_Q,
                       d_Q + q_doubles, zero_word, spread_host);
    TSC_HIP(hipGetLastError());
    return 0;
}


// rows are valid: the choice only moves how many pairs the screen drops); otherwise it is estimated from the structures.
// Everything is enqueued; nothing waits for the host.
static int build_descriptors(tsc_prune *p, const double *basis) {
    tsc_ctx *c = p->ctx;
    hipStream_t st = c->stream;
    const int h = p->h;
    const int nf[NFAM] = {n_features(h, 0), n_features(h, 1)};
    const size_t q_doubles = size_t(KD) * (nf[0] + nf[1]);
    Scratch s(c);
    const double *d_Q = basis;
    if (!basis) {
        const int n_samples = int(std::min<int64_t>(p->n, DESC_SAMPLE));
        const int64_t stride = std::max<int64_t>(1, p->n / n_samples);
        double *q;
        TSC_TRY(s.get(basis_doubles(h), &q));
        if (p->n < c->pca_min_n) {  // small ensemble: the identity basis, one tiny launch (sieve.hpp)
            hipLaunchKernelGGL(k_identity_basis, dim3(1), dim3(256), 0, st, nf[0], nf[1], q, q + q_doubles);
            TSC_HIP(hipGetLastError());
        } else {
            TSC_TRY(build_basis(c, st, s, p->heavy, h, n_samples, stride, q));
        }
        d_Q = q;
    }
    // structures per block of k_descriptors: as many as fit 48 KB of LDS next to the basis (a power of two, 4..64: the
    // 256 / S lanes that share a structure must be one wavefront at most)
    const size_t pitch = size_t(h * 3) | 1;
    int S = 64;
    while (S > 4 && (q_doubles + size_t(S) * pitch) * sizeof(double) > 48 * 1024) S >>= 1;
    const size_t lds_desc = (q_doubles + size_t(S) * pitch) * sizeof(double);
    TSC_REQUIRE(lds_desc <= 150 * 1024, "%d heavy atoms per structure exceed what the descriptor kernel can stage in LDS", h);
    if (lds_desc > 64 * 1024)
        TSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_descriptors), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_desc)));
    hipLaunchKernelGGL(k_descriptors, dim3(ceil_div<int64_t>(p->n, S)), dim3(256), lds_desc, st, p->heavy, p->n, h, nf[0], nf[1], d_Q,
                       (const double *)(d_Q + q_doubles), p->Dall, p->Gall, p->dmax_bits, S);
    TSC_HIP(hipGetLastError());
    return 0;  // the scratch blocks go back to the stream-ordered cache: later users run after these kernels
}

static int prune_create_impl(tsc_ctx *c, const double *heavy_dev, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask_buffer, tsc_prune **out,
                             const double *basis = nullptr, const ExternalDescriptors *ext = nullptr, int force_algo = -1) {
    TSC_REQUIRE(c && heavy_dev && out, "tsc_prune_create: null argument");
    TSC_REQUIRE(n > 0 && n < INT32_MAX - 4096, "n = %lld not supported", (long long)n);
    TSC_REQUIRE(h > 0, "no heavy atoms: the reference divides by zero here (rmsd_pruning.py:35)");
    TSC_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (reference-exact) or 1 (cache-free)");
    TSC_REQUIRE(rmsd_thr > 0, "rmsd_thr must be positive");
    TSC_REQUIRE(c->prune_algo != ALGO_TILE || h <= MAX_HP, "prune_algo=1 (register-tiled kernel) supports at most %d heavy atoms, got %d", MAX_HP, h);
    *out = nullptr;
    DeviceGuard guard(c->device);
    tsc_prune *p = new (std::nothrow) tsc_prune();
    if (!p) return fail(TSC_ERR_NOMEM, "out of host memory");
    p->ctx = c;
    p->heavy = heavy_dev;
    p->n = n;
    p->npad = (n + 63) / 64 * 64 + 320;  // the last column tile of a segment reads up to 255 columns past the active count
    p->h = h;
    p->hp = (h + 3) / 4 * 4;
    p->thr = rmsd_thr;
    p->mode = mode;
    p->algo = (c->prune_algo == ALGO_TILE) ? ALGO_TILE : ALGO_SIEVE;
    p->det_desc = c->deterministic_basis != 0;
    {   // this run's word of pinned memory (the culled-or-walked verdict of a candidate pass): its own for as long as it lives
        std::lock_guard<std::mutex> lock(c->runs_mutex);
        static_assert(PINNED_FLAG_SLOTS == 64, "one bit per flag word");
        const int slot = ~c->flag_slots_used ? __builtin_ctzll(~c->flag_slots_used) : -1;
        if (slot < 0) {
            delete p;
            return fail(TSC_ERR_STATE, "tsc_prune_create: %d prune runs are alive on this context, the most it serves at a time (destroy some, or use a "
                                       "context per thread)", PINNED_FLAG_SLOTS);
        }
        c->flag_slots_used |= 1ull << slot;
        p->flag_slot = slot;
        c->live_runs.push_back(p);
    }
    Scratch s_basis(c);
    if (force_algo >= 0) {
        p->algo = force_algo;
    } else if (c->prune_algo == ALGO_AUTO && mode == 1 && h <= MAX_HP && n >= AUTO_TILE_MIN_N && !basis && !(ext && ext->D)) {
        // automatic choice, no basis from a pipeline around this run: estimate it now and ask whether the screen can separate
        // anything (one synchronisation, some 20 us, on a run of at least 30 000 structures)
        const int n_samples = int(std::min<int64_t>(n, DESC_SAMPLE));
        double *q = nullptr;
        int rc0 = s_basis.get(basis_doubles(h), &q);
        if (!rc0) rc0 = build_basis(c, c->stream, s_basis, heavy_dev, h, n_samples, std::max<int64_t>(1, n / n_samples), q);
        if (!rc0) {
            double *host = reinterpret_cast<double *>(static_cast<char *>(c->pinned) + PINNED_SPREAD_OFFSET);
            hipError_t e = hipMemcpyAsync(host, q + basis_spread_offset(h), NFAM * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc0 = fail(TSC_ERR_HIP, "descriptor spread read-back failed: %s", hipGetErrorString(e));
            else if (screen_is_useless(host, h, rmsd_thr)) p->algo = ALGO_TILE, p->auto_tile = true;
            else basis = q;  // (the sieve's descriptors are built from it further down)
        }
        if (rc0) {
            tsc_prune_destroy(p);
            return rc0;
        }
    }
    p->bit_words = size_t(n / 64 + 40);  // (k_open_rows reads the 32 words of a whole scan block, also of the last, partial one)
    p->n_blocks = int(scan_bsum_count(n));
    int rc = 0;
    if (mask_buffer)
        p->mask = mask_buffer;  // the caller's verdict buffer serves as the working mask (8-byte aligned, n bytes)
    else if (!rc)
        rc = palloc(p, size_t(n), &p->mask);
    if (!rc) rc = palloc(p, size_t(n), &p->act);
    if (!rc) rc = palloc(p, size_t(n), &p->cend);
    if (!rc) rc = palloc(p, size_t(n), &p->best);
    if (!rc) rc = palloc(p, size_t(p->n_blocks) + 1, &p->bsum);
    if (!rc) rc = palloc(p, size_t(p->n_blocks) + 1, &p->boff);
    if (!rc) rc = palloc(p, size_t(n) / 16 + 8, &p->tile_cmax);
    if (!rc) rc = palloc(p, size_t(n) / 16 + 8, &p->tile_done);
    if (!rc) rc = palloc(p, 2 * p->bit_words, &p->bits);
    p->dsum_words = p->bit_words / 1024 + 4;  // one summary bit per 1024 cache-view bits, kept right behind the view
    p->n_views = 0;
    for (int slot = 0; slot < TSC_MAX_PASSES; ++slot) {
        const bool can_run = int64_t(KS[slot]) == 1 || 20 * int64_t(KS[slot]) < n;  // (tsc_prune_next_pass)
        p->view_of_slot[slot] = (can_run && mode == 0) ? p->n_views++ : -1;
    }
    if (!rc) rc = palloc(p, std::max<size_t>(1, size_t(p->n_views) * (p->bit_words + p->dsum_words)), &p->views);
    if (!rc) rc = palloc(p, TSC_MAX_PASSES, &p->view_pass);
    if (!rc) rc = palloc(p, 1, &p->counters);
    if (!rc) rc = palloc(p, 1, &p->state);
    if (!rc) rc = palloc(p, TSC_MAX_PASSES, &p->records);
    const bool own_desc = !(ext && ext->D);  // (external descriptors: already enqueued on this stream, the caller owns the buffers)
    if (!rc && p->algo == ALGO_SIEVE) {
        if (own_desc) {
            rc = palloc(p, size_t(n) * DW, &p->Dall);
            if (!rc) rc = palloc(p, size_t(n), &p->Gall);
            if (!rc) rc = palloc(p, 4, &p->dmax_bits);
        } else {
            p->Dall = ext->D, p->Gall = ext->G, p->dmax_bits = ext->dmax_bits;
        }
        if (!rc) rc = palloc(p, size_t(n) * DW, &p->Dc);
        // (mm.hpp; decided per run: the 64-row kernels for large runs, the 16-row form of the walked kernel -- "sieve_mm16" -- below that)
        p->mm64 = c->sieve_mm == 2 || (c->sieve_mm == 1 && n >= c->mm_min_n);
        const bool want_mm = p->mm64 || c->sieve_mm16 != 0;
        if (!rc && want_mm) rc = palloc(p, size_t(n) * MM_REC_HALVES, &p->Dh);
        // the float32 copy stage 1 reads (sieve.hpp, pair_stage1): from the embedding kernel where there was one, else converted here
        // (it pays where the gathers come from HBM: 41 MB of heavy atoms at C3 sit in the 256 MB infinity cache and the conversions cost the
        // VALU-bound kernel 2 %; at C4's 348 MB a step goes from 12.1 to 10.6 ms.  "stage1_f32": 0 never, 1 from 128 MB on, 2 always)
        if (!rc && want_heavy32(c, double(n) * h * 24.0)) {
            if (ext && ext->heavy32) {  // written by the kernel that embedded the structures
                p->heavy32 = ext->heavy32;
            } else {
                rc = palloc(p, size_t(n) * heavy32_pitch(h), &p->heavy32);
                if (!rc) hipLaunchKernelGGL(k_heavy32, dim3(unsigned(std::min<int64_t>(ceil_div<int64_t>(n * heavy32_pitch(h), 256), 65536))), dim3(256), 0, c->stream,
                                            heavy_dev, n, h, p->heavy32);
            }
        }
    }
    if (!rc) rc = palloc(p, 1, &p->tickets);
    if (!rc && p->algo == ALGO_TILE) {
        const size_t hp3 = size_t(p->hp) * 3;
        rc = palloc(p, size_t(p->npad) * hp3, &p->Xr);
        if (!rc) rc = palloc(p, size_t(p->npad) * hp3, &p->Xc);
        if (!rc) rc = palloc(p, size_t(p->npad), &p->G);
    }
    if (!rc) {
        hipStream_t st = c->stream;
        int first_slot = -1;  // the pass tsc_prune_next_pass will hand out first: opened by k_init_run itself
        for (int slot = 0; slot < TSC_MAX_PASSES && first_slot < 0; ++slot)
            if (int64_t(KS[slot]) == 1 || 20 * int64_t(KS[slot]) < n) first_slot = slot;
        p->opened_slot = first_slot;
        InitArgs ia;
        memset(&ia, 0, sizeof(ia));
        ia.n = n, ia.mask = p->mask, ia.bits = p->bits, ia.bit_words = int(p->bit_words);
        ia.views = p->views, ia.view_words = int64_t(p->n_views) * int64_t(p->bit_words + p->dsum_words), ia.view_pass = p->view_pass, ia.n_views = p->n_views;
        for (int slot = 0; slot < TSC_MAX_PASSES; ++slot)
            if (p->view_of_slot[slot] >= 0) ia.sched[p->view_of_slot[slot]] = view_pass(int(n), int(KS[slot]));
        ia.st = p->state, ia.rec = p->records, ia.n_rec = TSC_MAX_PASSES, ia.cnt = p->counters;
        ia.bsum = p->bsum, ia.boff = p->boff, ia.n_blocks = p->n_blocks, ia.block_items = SCAN_TILE;
        ia.dmax_bits = own_desc ? p->dmax_bits : nullptr;
        // the arrival counters and the per-tile ones lie in two blocks: the tickets are zeroed here, tile_done by its own loop below
        ia.zero_words = reinterpret_cast<unsigned *>(p->tickets), ia.n_zero_words = int64_t(sizeof(*p->tickets) / sizeof(unsigned));
        ia.act = p->act, ia.first_slot = first_slot, ia.first_k = first_slot >= 0 ? (long long)KS[first_slot] : 0ll, ia.first_algo = p->algo;
        ia.tile_done = p->tile_done, ia.n_tile_done = n / 16 + 8;
        hipLaunchKernelGGL(k_init_run, dim3(grid_for(n / 8 + 1, 256, 512)), dim3(256), 0, st, ia);
        hipError_t e = hipGetLastError();
        // padded columns of the compacted layouts are read by the last column tile of a segment but never used; the
        // register-tiled kernel's buffers are zeroed once so that those reads see finite numbers
        if (e == hipSuccess && p->Xc) e = hipMemsetAsync(p->Xc, 0, size_t(p->npad) * p->hp * 3 * sizeof(double), st);
        if (e == hipSuccess && p->Xr) e = hipMemsetAsync(p->Xr, 0, size_t(p->npad) * p->hp * 3 * sizeof(double), st);
        if (e == hipSuccess && p->G) e = hipMemsetAsync(p->G, 0, size_t(p->npad) * sizeof(double), st);
        if (e != hipSuccess) rc = fail(TSC_ERR_HIP, "prune state setup failed: %s", hipGetErrorString(e));
    }
    if (!rc && p->Dall && own_desc) rc = build_descriptors(p, basis);
    if (rc) {
        tsc_prune_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}


extern "C" __attribute__((visibility("default"))) int tsc_prune_create(tsc_ctx *c, const double *heavy_dev, int64_t n, int h, double rmsd_thr, int mode, tsc_prune **out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c != nullptr, "tsc_prune_create: null argument");
    // descriptors that tsc_embed_masked_dev wrote with this very array are used once, by the run created next ...
    if (c->xd_valid && c->xd_h == h && c->xd_heavy == heavy_dev && n <= c->xd_cap && c->prune_algo != ALGO_TILE) {
        ExternalDescriptors ext;
        ext.D = c->xd_D, ext.G = c->xd_G, ext.dmax_bits = c->xd_dmax;
        ext.heavy32 = c->xd_h32_valid ? c->xd_heavy32 : nullptr;
        c->xd_valid = false;
        TSC_TRY(prune_create_impl(c, heavy_dev, n, h, rmsd_thr, mode, nullptr, out, nullptr, &ext));
        std::lock_guard<std::mutex> lock(c->runs_mutex);
        (*out)->borrows_xd = true;   // (tsc_embed_masked_dev will not release or regrow the buffers under this run)
        ++c->xd_borrowers;
        return 0;
    }
    c->xd_valid = false;
    // ... and so is a basis that tsc_embed_clash_compact_dev / tsc_basis_from_poses_dev estimated on the side stream
    const double *basis = pending_basis(c, h);
    if (basis) {
        DeviceGuard guard(c->device);
        TSC_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    c->eb_valid = false;
    return prune_create_impl(c, heavy_dev, n, h, rmsd_thr, mode, nullptr, out, basis);
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_next_pass(tsc_prune *p, int64_t *k_out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && k_out, "null argument");
    if (p->cur_k != 0) return fail(TSC_ERR_STATE, "tsc_prune_next_pass: previous pass not finished");
    *k_out = 0;
    while (p->next_ks < TSC_MAX_PASSES) {
        const int slot = p->next_ks++;
        const int64_t k = int64_t(KS[slot]);  // int(k): the reference itself fails for float k
        // count_nonzero(mask) <= n, so a pass with 20 k >= n can never pass the gate of :192; the others are enqueued
        // and gated on the device
        if (k == 1 || 20 * k < p->n) {
            p->cur_k = k;
            p->cur_slot = slot;
            p->local_done = false;
            *k_out = k;
            return 0;
        }
    }
    return 0;
    TSC_API_GUARD_END
}

// Rough size of the open pass in pairs (n structures, every row against half of an average chunk); it depends only
// on n and k, so every rank of a sharded run computes the same number.
extern "C" __attribute__((visibility("default"))) int tsc_prune_pass_estimate(tsc_prune *p, int64_t *pairs) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && pairs, "null argument");
    if (p->cur_k == 0) return fail(TSC_ERR_STATE, "tsc_prune_pass_estimate: no pass open");
    *pairs = p->n * (p->n / p->cur_k) / 2;
    return 0;
    TSC_API_GUARD_END
}

// The pair search of one rank's row tiles of the open pass (step 3 of a pass; steps 1-2 have run).
// rows_ub: upper bound of the rows of the pass on this device (n; in a rank-partitioned pass the structures of this rank's chunks)
// rows_now (optional, <= rows_ub): the rows the pass really has, where the host has learnt it (a pass that waited for k_cull_decide): the grid is
// sized for them; everything the kernels COUNT arrivals by stays with rows_ub, which k_open_rows was launched with
static int launch_pair_search(tsc_prune *p, int rank, int world, int64_t rows_ub, int64_t rows_now = -1) {
    tsc_ctx *c = p->ctx;
    hipStream_t st = c->stream;
    const int64_t n = p->n, k = p->cur_k;
    const int slot = p->cur_slot;
    const int A = int(std::max<int64_t>(rows_ub, 1));
    PassGeom g{int(n), int(k), int(n / k)};
    const int64_t longest_chunk = n - (k - 1) * g.cs;
    // 3. pairs: rows dealt round-robin over ranks in tiles of 16, columns cut into segments for load balance
    const int n_tiles = ceil_div(A, TILE_ROWS);
    const int max_range = int(std::min<int64_t>(A, longest_chunk));
    // a wavefront walks its segment tile by tile: short segments keep the critical path short when a pass has little
    // work (many small chunks), long ones amortise the per-item setup when it has a lot
    // (measured on MI355X, tools/sweep.py: 512 columns at 57k structures, 1024 at 126k, 4096 at 483k; "seg_cols" overrides)
    int seg_cols = c->seg_cols > 0 ? c->seg_cols : (n <= 100000 ? 512 : (n <= 400000 ? 1024 : 4096));
    while (seg_cols > 256 && max_range < seg_cols * 4) seg_cols /= 2;
    // the screen on the matrix cores (mm.hpp): one rank, 64 rows per work item and segments of their own length
    const bool mm = p->algo == ALGO_SIEVE && p->Dh && p->mm64;
    const bool mm16 = p->algo == ALGO_SIEVE && p->Dh && !p->mm64 && c->sieve_cpl == 2 && c->sieve_trim != 0;
    if (mm) seg_cols = c->mm_seg_cols > 0 ? c->mm_seg_cols : (max_range >= 2048 ? 1024 : 512);
    const int n_seg = ceil_div(max_range + 64, seg_cols);  // + 64: a segment starts at the 64-aligned column below r0 + 1
    const int my_tiles = (n_tiles - rank + world - 1) / world;
    // (mm: groups of 64 rows dealt round-robin to the ranks; the 16-row matrix-core kernel: two items per workgroup, four where most workgroups are empty)
    const int mm16_waves = mm16 ? (n_seg <= MM16_LONG_SEGS ? 2 : 4) : 4;
    const int A_grid = rows_now >= 0 ? int(std::min<int64_t>(std::max<int64_t>(rows_now, 1), A)) : A;
    const int grid_tiles = (ceil_div(A_grid, TILE_ROWS) - rank + world - 1) / world;
    dim3 grid(std::max(1, mm ? ceil_div((ceil_div(A_grid, MM_ROWS) - rank + world - 1) / world, MM_WAVES) : ceil_div(std::min(my_tiles, grid_tiles), mm16_waves)), n_seg);
    // the pair kernel's own start / stop events ride on its dispatch packet (no extra packets in the stream; a
    // hipEventRecord before and after it costs about 4 us each on MI355X)
    hipEvent_t e0 = c->pass_timing >= 1 ? p->ev[slot][1] : nullptr, e1 = c->pass_timing >= 1 ? p->ev[slot][2] : nullptr;
    if (p->algo == ALGO_TILE) {
        TileArgs a;
        a.ld = p->npad, a.h = p->h;
        a.tile_begin = rank, a.tile_stride = world, a.seg_cols = seg_cols;
        a.thr = p->thr, a.maxdev_thr = 2 * p->thr;  // :95
        a.half_h_thr2 = 0.5 * double(p->h) * p->thr * p->thr;
        TSC_TRY(launch_rmsd_tile(p->hp, st, grid, e0, e1, (const double *)p->Xr, (const double *)p->Xc, (const double *)p->G, (const int32_t *)p->cend, p->best,
                                 p->counters, (const PruneState *)p->state, a));
    } else {
        SieveArgs a;
        // (the bound the grid was sized for: tiles of the last block beyond it must leave before they read a stale tile_cmax[] entry
        // and arrive at the pass's counter as a tile that does not exist)
        a.n = A, a.h = p->h;
        a.tile_begin = rank, a.tile_stride = world, a.seg_cols = seg_cols;
        a.thr = p->thr, a.maxdev_thr = 2 * p->thr;  // :95
        a.half_h_thr2 = 0.5 * double(p->h) * p->thr * p->thr;
        a.two_thr2 = p->h >= 4 ? 2.0 * p->thr * p->thr : -1.0;
        a.dmax_bits = p->dmax_bits, a.desc_limit = double(p->h) * p->thr * p->thr;
        a.heavy32 = p->heavy32;
        a.tile_cmax = p->tile_cmax;
        a.drain_min = c->drain_min;
        a.dbg = nullptr;
#ifdef TSC_DBG_STAMPS
        if (c->dbg_stamp_k == k) {
            const size_t bytes = size_t(grid.x) * grid.y * 32 * sizeof(unsigned long long);
            if (c->dbg_bytes < bytes) {
                if (c->dbg_buf) (void)hipFree(c->dbg_buf);
                TSC_HIP(hipMalloc(&c->dbg_buf, bytes));
                c->dbg_bytes = bytes;
            }
            TSC_HIP(hipMemsetAsync(c->dbg_buf, 0, bytes, st));
            c->dbg_waves = int64_t(grid.x) * grid.y * 4;
            a.dbg = static_cast<unsigned long long *>(c->dbg_buf);
        }
#endif
        FusedApply fa;
        memset(&fa, 0, sizeof(fa));
        if (p->cur_fused) {
            fa.ap = apply_args(p);
            fa.tile_done = p->tile_done, fa.tickets = &p->tickets->pass, fa.n_tiles = unsigned(ceil_div(A, TILE_ROWS));
            fa.sc = step_ctx(p, p->cur_range);
            int nxt = -1;
            fa.next = next_step_args(p, &nxt);
            if (!p->cur_range) {
                p->opened_slot = nxt;
                p->last_slot = -1;  // closed on the device, by the pair kernel's last tile
            }
        }
        const bool trim = c->sieve_cpl == 2 && c->sieve_trim;
        if (mm16) {
            TSC_TRY(launch_rmsd_sieve_mm16(p->cur_fused, a.heavy32 != nullptr, mm16_waves, st, grid, e0, e1, p->heavy, (const int32_t *)p->act, (const double *)p->Gall,
                                           (const _Float16 *)p->Dh, (const int32_t *)p->cend, p->best, p->counters,
                                           (const PruneState *)p->state, a, fa));
            return 0;
        }
        if (mm) {
            TSC_TRY(launch_rmsd_sieve_mm(p->cur_fused, a.heavy32 != nullptr, st, grid, e0, e1, p->heavy, (const int32_t *)p->act, (const double *)p->Gall,
                                         (const float *)p->Dc, (const _Float16 *)p->Dh, (const int32_t *)p->cend, p->best, p->counters, (const PruneState *)p->state, a, fa));
            return 0;
        }
        // (stage 1 on the float32 copy exists for the default shape of the kernel only)
        TSC_TRY((p->cur_fused ? launch_rmsd_sieve_fused : launch_rmsd_sieve_plain)(
            c->sieve_cpl, trim, trim && a.heavy32 != nullptr, st, grid, e0, e1, p->heavy, (const int32_t *)p->act, (const double *)p->Gall, (const float *)p->Dc,
            (const int32_t *)p->cend, p->best, p->counters, (const PruneState *)p->state, a, fa));
    }
    TSC_HIP(hipGetLastError());
    return 0;
}

// The launches of a pass on this device.  range = false: the rows dealt to (rank, world) by tiles, of all chunks (tsc_prune_pass_local).
// range = true: every row of the chunks that belong to this rank (tsc_prune_pass_range); rank / world are then 0 / 1 for the
// kernels -- they see an ensemble made of this rank's rows.
static int pass_launch(tsc_prune *p, int rank, int world, bool range) {
    tsc_ctx *c = p->ctx;
    DeviceGuard guard(c->device);
    hipStream_t st = c->stream;
    const int64_t n = p->n, k = p->cur_k;
    const int slot = p->cur_slot;
    int64_t c_lo = 0, c_hi = k, s_lo = 0, s_hi = n;
    if (range) partition_bounds(n, k, p->part_rank, p->part_world, &c_lo, &c_hi, &s_lo, &s_hi);
    const int A = int(std::max<int64_t>(s_hi - s_lo, 1));  // grids are sized for the upper bound; kernels read the true count from the state block
    PassGeom g{int(n), int(k), int(n / k)};
    p->cur_range = range;
    for (int i = 0; i < 4; ++i)
        if (!p->ev[slot][i]) TSC_TRY(get_event(c, &p->ev[slot][i]));
    if (c->pass_timing >= 2) TSC_HIP(hipEventRecord(p->ev[slot][0], st));
    // 0. open this pass: gate (:192), counters, cache-view bitmap -- already done by the apply kernel of the pass before
    //    it (its last block), by a one-block launch for the first pass of a run
    if (p->opened_slot != slot) {
        StepArgs sa{p->last_slot, slot, (long long)k, p->algo, -1};
        hipLaunchKernelGGL(k_pass_step, dim3(1), dim3(64), 0, st, step_ctx(p), sa);
    }
    if (range && p->range_ready_slot != slot)  // no k_pass_merge in front of this pass (the first of a run): which rows are this rank's
        hipLaunchKernelGGL(k_range_open, dim3(1), dim3(64), 0, st, p->state, (const int32_t *)p->boff, (const unsigned long long *)p->bits, int(p->bit_words),
                           p->n_blocks, int(s_lo), int(s_hi));
    p->last_slot = slot;
    p->slot_used[slot] = true;
    const int use_cache = (p->mode == 0);
    // the last chunk takes the remainder (:141-142); of this rank's chunks, in a partitioned pass
    const int64_t longest_chunk = c_hi == k ? n - (k - 1) * g.cs : g.cs;
    // Short chunks: the whole pass in one launch, a workgroup (or a few) per chunk (local_pass.hpp)
    // (measured on MI355X: a block of the chunk-local kernel is a chain of dependent memory round trips, so it wins where
    // chunks are a few row tiles long -- at 57k structures the passes k = 1000, 500 and 200 take 37, 39 and 50 us instead of
    // 52-58 -- and loses beyond: k = 100 takes 58 us there against 53 on the two-launch path; "local_max_chunk" moves the limit)
    // (the longest chunk counts, i.e. the last one with its remainder: at 57 046 structures in 2 000 chunks -- 28 each, 1 074 in the last --
    // the chunk-local kernel was tried with the long chunk on workgroups of its own: 97 us against 37 for the two launches)
    p->cur_local = p->algo == ALGO_SIEVE && world == 1 && c->local_pass != 0 && std::max<int64_t>(longest_chunk, g.cs) <= std::min(LP_MAX_ROWS, c->local_max_chunk) &&
                   c_hi > c_lo;
    p->cur_fused = false;
    if (p->cur_local) {
        LocalPassArgs a;
        a.h = p->h, a.use_cache = use_cache;
        a.nb_regular = std::max(1, ceil_div(ceil_div(g.cs, LP_TI), LP_TILES_PER_BLOCK));
        a.nb_last = c_hi == k ? std::max(1, ceil_div(ceil_div(int(n - (k - 1) * g.cs), LP_TI), LP_TILES_PER_BLOCK)) : 0;
        a.c_lo = int(c_lo), a.n_reg = int(std::min<int64_t>(c_hi, k - 1) - c_lo);
        a.exch = range ? p->exch : nullptr;
        a.thr = p->thr, a.maxdev_thr = 2 * p->thr;  // :95
        a.half_h_thr2 = 0.5 * double(p->h) * p->thr * p->thr;
        a.two_thr2 = p->h >= 4 ? 2.0 * p->thr * p->thr : -1.0;
        a.desc_limit = double(p->h) * p->thr * p->thr;
        a.dmax_bits = p->dmax_bits;
        int nxt = -1;
        const StepArgs sa = next_step_args(p, &nxt);
        const int64_t blocks = int64_t(a.n_reg) * a.nb_regular + a.nb_last;
        // (its own events only at pass_timing 2: level 1 is what a timed region carries for the PAIR kernel's durations, and a pair of
        // events costs a small pass about 6 us)
        hipEvent_t e0 = c->pass_timing >= 2 ? p->ev[slot][1] : nullptr, e1 = c->pass_timing >= 2 ? p->ev[slot][2] : nullptr;
        TSC_TRY(launch_pass_chunks(st, unsigned(blocks), e0, e1, g, a, p->state, p->mask, p->bits, int(p->bit_words), view_of_open_pass(p), p->heavy,
                                   (const double *)p->Gall, (const float *)p->Dall, later_views(p), p->counters, p->bsum, SCAN_TILE, step_ctx(p, range), sa,
                                   &p->tickets->local));
        if (!range) {
            p->opened_slot = nxt;
            p->last_slot = -1;  // closed on the device
        }
        p->local_done = true;
        return 0;
    }
    // Large passes: the structures laid out along a Morton curve, tile pairs skipped by bounding box (cull.hpp); the verdicts are
    // applied by k_apply_pass behind the pair kernel (tsc_prune_pass_finish), on one rank or several
    // (the pairs a rank gets to look at: its chunks in a partitioned pass, its row tiles in a pass dealt by tiles -- the layout and
    // the boxes are made by every rank for itself and have to pay for themselves on that share)
    const double my_pairs = range ? double(s_hi - s_lo) * double(n / k) * 0.5 : double(n) * double(n / k) * 0.5 / double(world);
    // (row tiles dealt to several ranks: twice the threshold -- every rank lays the whole pass out for an eighth, say, of its tiles;
    // measured at 1M x 50 and eight ranks the culled k = 2 pass costs a rank 0.82 ms against 0.77 for the walk)
    // Row tiles of a pass dealt to several ranks (tsc_prune_pass_local / _rows with world > 1): the ranks deal the tiles of ONE sorted layout,
    // so every rank must hold bit-identical descriptors -- only runs created under "deterministic_basis" may be culled that way; the others
    // walk the pass in index order, every rank alike.  (Inside a pass partitioned by chunks a rank culls its own chunks with a layout of
    // its own: no such condition.)
    const bool shared_layout_ok = world == 1 || range || p->det_desc;
    if (world > 1 && p->auto_tile && !p->det_desc)
        return fail(TSC_ERR_STATE, "tsc_prune_pass_local: this run chose the all-pairs kernel from its own basis estimate; ranks of a sharded run could "
                                   "choose differently -- create the runs under deterministic_basis = 1, or force prune_algo 1 or 2 on every rank");
    const bool culled = p->algo == ALGO_SIEVE && c->cull != 0 && c->sieve_cpl == 2 && k < CULL_MAX_CHUNKS && shared_layout_ok &&
                        my_pairs >= c->cull_min_pairs * ((world > 1 && !range) ? 2.0 : 1.0);
    if (culled && !p->morton_order) {
        int rc = palloc(p, size_t(n), &p->morton_order);
        if (!rc) rc = palloc(p, size_t(n), &p->rank_of);
        if (!rc) rc = palloc(p, size_t(n) + 256, &p->crank);
        if (!rc) rc = palloc(p, size_t(CULL_MAX_CHUNKS) + 1, &p->cbase);
        if (!rc) rc = palloc(p, size_t(CULL_MAX_CHUNKS) + 1, &p->cfill);
        if (!rc) rc = palloc(p, (size_t(n) / CULL_LAYOUT_ITEMS + 2) * CULL_MAX_CHUNKS, &p->blk_cnt);
        if (!rc) rc = palloc(p, (size_t(n) + 256) * DW, &p->Ds);
        if (!rc && p->Dh && p->mm64) rc = palloc(p, (size_t(n) + 256) * MM_REC_HALVES, &p->Dhs);
        if (!rc && p->Dh && p->mm64) rc = palloc(p, size_t(n) + 256, &p->cstruct);
        if (!rc) rc = palloc(p, (size_t(n) / CULL_COLS + 2) * CULL_BOX, &p->cbox);
        if (!rc) rc = palloc(p, (size_t(n) / CULL_COLS + 2) * 8 * CULL_BOX, &p->rbox);
        if (rc) return rc;
    }
    // 1. per row: which structure it is, its stop column, best[] = none, its descriptor by position (k_open_rows, rmsd.hpp)
    p->cur_fused = p->algo == ALGO_SIEVE && world == 1 && (c->fused_apply != 0 || range);
    {
        // (the fp32 rows by position: read by the packed-fp32 kernels, by level 2 of the 64-row matrix-core kernels where it is built in, and by a
        // culled pass's layout)
        const bool need_dc = !(p->algo == ALGO_SIEVE && p->Dh && (p->mm64 ? !TSC_MM_LEVEL2 : (c->sieve_cpl == 2 && c->sieve_trim != 0)) && !culled);
        OpenArgs oa;
        oa.use_cache = use_cache, oa.fused = p->cur_fused ? 1 : 0, oa.lds_cap = std::min(c->open_lds_blocks, OPEN_LDS_BLOCKS);
        oa.view = view_of_open_pass(p), oa.bits = p->bits, oa.bit_words = int(p->bit_words);
        oa.boff = p->boff, oa.n_blocks = p->n_blocks, oa.block_items = SCAN_TILE;
        oa.n_tiles = unsigned(ceil_div(A, 16)), oa.tickets = &p->tickets->pass;
        oa.rank_of = culled ? p->rank_of : nullptr;
        oa.Dh = p->Dh, oa.dmax_bits = p->dmax_bits;
        oa.dbg = nullptr;
#ifdef TSC_DBG_STAMPS
        if (c->dbg_stamp_k == -k) {  // (a negative k selects k_open_rows of pass k)
            const size_t bytes = size_t(ceil_div(ceil_div(A, 16), 4)) * 32 * sizeof(unsigned long long);
            if (c->dbg_bytes < bytes) {
                if (c->dbg_buf) (void)hipFree(c->dbg_buf);
                TSC_HIP(hipMalloc(&c->dbg_buf, bytes));
                c->dbg_bytes = bytes;
            }
            TSC_HIP(hipMemsetAsync(c->dbg_buf, 0, bytes, st));
            c->dbg_waves = int64_t(ceil_div(ceil_div(A, 16), 4)) * 4;
            oa.dbg = static_cast<unsigned long long *>(c->dbg_buf);
        }
#endif
        int nxt = -1;
        const StepArgs sa = p->cur_fused ? next_step_args(p, &nxt) : StepArgs{-1, -1, 0ll, 0, -1};
        static_assert(SCAN_TILE == 64 * SCAN_BLOCK_WORDS && DW == DESC_WORDS, "k_open_rows");
        hipLaunchKernelGGL(k_open_rows, dim3(ceil_div(ceil_div(A, 16), 16)), dim3(256), 0, st, g, oa, step_ctx(p, range), sa, p->act, p->cend, p->best, p->tile_cmax,
                           (const float *)p->Dall, need_dc ? p->Dc : nullptr);
    }
    if (p->algo == ALGO_TILE) {
        const int hp3 = p->hp * 3;
        size_t lds = size_t(64) * (hp3 + 1) * sizeof(double);
        hipLaunchKernelGGL(k_compact_coords, dim3(ceil_div(A, 64)), dim3(256), lds, st, p->heavy, p->h, hp3, p->act, (const PruneState *)p->state,
                           p->Xr, p->Xc, p->npad, p->G);
    }
    bool run_culled = false;
    int64_t rows_now = -1;   // (the pass's rows, where the host has waited for the device anyway)
    if (culled) {
        // culled, or walked in index order?  The rows' ranges decide (k_cull_decide); the host waits for the verdict -- a pass this
        // large takes a millisecond or more, the round trip some 20 us
        volatile int *flag = reinterpret_cast<volatile int *>(static_cast<char *>(c->pinned) + PINNED_FLAG_OFFSET + 64 * size_t(p->flag_slot));
        *flag = 0;
        hipLaunchKernelGGL(k_chunk_bases, dim3(unsigned(k + 1)), dim3(64), 0, st, g, (const PruneState *)p->state, (const int32_t *)p->boff,
                           (const unsigned long long *)p->bits, int(p->bit_words), p->n_blocks, p->cbase, p->cfill);
        hipLaunchKernelGGL(k_cull_decide, dim3(1), dim3(64), 0, st, p->state, (const PassCounters *)p->counters, (const int32_t *)p->cbase, int(k),
                           c->cull == 2 ? 1 : 0, const_cast<int *>(flag));
        TSC_HIP(hipStreamSynchronize(st));
        run_culled = *flag != 0;
        rows_now = flag[1];
    }
    if (run_culled && !p->morton_sorted) {
        // once per run: the structures in coarse Morton order of their descriptors -- a stable two-digit radix sort by cell, so that
        // every rank of a sharded run comes to the same order (cull.hpp)
        Scratch s(c);
        int32_t *tmp, *blk, *tot;
        const int n_rb = int(ceil_div<int64_t>(n, 2048));
        TSC_TRY(s.get(size_t(n), &tmp));
        TSC_TRY(s.get(size_t(n_rb) * RADIX_BUCKETS, &blk));
        TSC_TRY(s.get(size_t(RADIX_BUCKETS), &tot));
        static_assert(CULL_MORTON_BITS * CULL_MORTON_DIMS <= 16, "two 8-bit digits");
        for (int pass = 0; pass < 2; ++pass) {
            const int32_t *in = pass == 0 ? nullptr : tmp;
            int32_t *out = pass == 0 ? tmp : p->morton_order;
            hipLaunchKernelGGL(k_radix_count, dim3(unsigned(n_rb)), dim3(256), 0, st, (const float *)p->Dall, in, n, (const unsigned *)p->dmax_bits, 8 * pass, blk);
            hipLaunchKernelGGL(k_radix_scan, dim3(RADIX_BUCKETS), dim3(64), 0, st, n_rb, blk, tot);
            hipLaunchKernelGGL(k_radix_base, dim3(1), dim3(256), 0, st, tot);
            hipLaunchKernelGGL(k_radix_scatter, dim3(unsigned(n_rb)), dim3(256), 0, st, (const float *)p->Dall, in, n, (const unsigned *)p->dmax_bits, 8 * pass,
                               (const int32_t *)blk, (const int32_t *)tot, out);
        }
        TSC_HIP(hipGetLastError());
        p->morton_sorted = true;
    }
    // (the culled pass with the screen on the matrix cores: one rank's own pass -- row tiles of a layout dealt to several ranks keep the
    // kernel of cull.hpp, whose items are single row tiles)
    const bool cull_mm = run_culled && p->Dhs && p->mm64;
    if (run_culled) {
        p->cur_fused = false;  // rows collect verdicts as columns of other tiles too: the pass is applied behind the pair kernel (k_apply_pass)
        const int n_lb = int(ceil_div<int64_t>(n, CULL_LAYOUT_ITEMS));
        const LayoutRange lr{int(s_lo), int(s_hi)};
        hipLaunchKernelGGL(k_layout_count, dim3(unsigned(n_lb)), dim3(256), 0, st, g, lr, (const PruneState *)p->state, (const int32_t *)p->morton_order,
                           (const unsigned long long *)p->bits, int(p->bit_words), p->blk_cnt);
        hipLaunchKernelGGL(k_layout_scan, dim3(unsigned(k)), dim3(64), 0, st, (const PruneState *)p->state, n_lb, (const int32_t *)p->cbase, p->blk_cnt);
        hipLaunchKernelGGL(k_layout_scatter, dim3(unsigned(n_lb)), dim3(256), 0, st, g, lr, (const PruneState *)p->state, (const int32_t *)p->morton_order,
                           (const unsigned long long *)p->bits, int(p->bit_words), (const int32_t *)p->rank_of, (const float *)p->Dc,
                           (const int32_t *)p->blk_cnt, p->Ds, p->crank, (const _Float16 *)(cull_mm ? p->Dh : nullptr), cull_mm ? p->Dhs : nullptr, cull_mm ? p->cstruct : nullptr);
        hipLaunchKernelGGL(k_tile_boxes, dim3(unsigned(ceil_div<int64_t>(n, CULL_COLS))), dim3(128), 0, st, (const PruneState *)p->state, (const float *)p->Ds,
                           p->cbox, p->rbox);
        SieveArgs a;
        memset(&a, 0, sizeof(a));
        a.n = A, a.h = p->h;
        a.tile_begin = rank, a.tile_stride = world, a.seg_cols = 4096;
        a.thr = p->thr, a.maxdev_thr = 2 * p->thr;  // :95
        a.half_h_thr2 = 0.5 * double(p->h) * p->thr * p->thr;
        a.two_thr2 = p->h >= 4 ? 2.0 * p->thr * p->thr : -1.0;
        a.dmax_bits = p->dmax_bits, a.desc_limit = double(p->h) * p->thr * p->thr;
        a.heavy32 = p->heavy32;
        a.drain_min = c->drain_min;
        const int tb = world > 1 ? std::max(1, c->cull_tile_block) : 1;
        CullArgs ca{p->Ds, p->crank, p->cbase, p->cbox, p->rbox, int(k), tb, c->cull_xcd};
        const int n_tiles = ceil_div(A, TILE_ROWS);
        // (slots of this rank: one by one, or whole runs of tb tiles -- an upper bound; slots beyond the last tile leave at once)
        const int my_tiles = tb <= 1 ? (n_tiles - rank + world - 1) / world : (n_tiles / (tb * world) + 1) * tb;
        // columns of a row tile: from its own 128-aligned position to the end of its (last row's) chunk -- a chunk and a tile more at most
        const int n_seg = ceil_div(int(std::min<int64_t>(A, longest_chunk)) + 2 * CULL_COLS, a.seg_cols);
        hipEvent_t e0 = c->pass_timing >= 1 ? p->ev[slot][1] : nullptr, e1 = c->pass_timing >= 1 ? p->ev[slot][2] : nullptr;
        const int64_t items = int64_t(ceil_div(my_tiles, 4)) * n_seg;
        // ("cull_xcd": one work item per workgroup, runs of row groups keyed to XCDs -- 8 XCDs x segments x the runs an XCD holds of a segment)
        const int64_t xitems = int64_t(8) * n_seg * ceil_div(ceil_div(ceil_div(my_tiles, 4), CULL_XCD_RUN), 8) * CULL_XCD_RUN;
        const dim3 sgrid(unsigned(std::max<int64_t>(1, c->cull_xcd ? xitems : std::min<int64_t>(items, c->cull_grid))));
        if (cull_mm) {
            // (one wavefront per workgroup; several ranks: this rank's share of the groups, in runs of tile_block / 4 -- an upper bound)
            const int all_groups = ceil_div(A, MM_ROWS), tbg = std::max(1, tb / 4);
            const int n_groups = world <= 1 ? all_groups : (all_groups / (tbg * world) + 1) * tbg, wgs = n_groups;
            const int n_seg_mm = ceil_div(int(std::min<int64_t>(A, longest_chunk)) + 2 * CULL_COLS, CMM_SEG);
            const int64_t grid_mm = c->cull_xcd ? int64_t(8) * n_seg_mm * ceil_div(ceil_div(wgs, CULL_XCD_RUN), 8) * CULL_XCD_RUN : int64_t(wgs) * n_seg_mm;
#ifdef TSC_DBG_STAMPS
            if (c->dbg_stamp_k == k) {
                const size_t bytes = size_t(grid_mm) * 32 * sizeof(unsigned long long);
                if (c->dbg_bytes < bytes) {
                    if (c->dbg_buf) (void)hipFree(c->dbg_buf);
                    TSC_HIP(hipMalloc(&c->dbg_buf, bytes));
                    c->dbg_bytes = bytes;
                }
                TSC_HIP(hipMemsetAsync(c->dbg_buf, 0, bytes, st));
                c->dbg_waves = grid_mm * 4;   // (8 of every 32 words used: one wavefront per workgroup)
                a.dbg = static_cast<unsigned long long *>(c->dbg_buf);
            }
#endif
            CullMmArgs cm{p->Dhs, p->cstruct};
            TSC_TRY(launch_rmsd_sieve_sorted_mm(a.heavy32 != nullptr, st, dim3(unsigned(std::max<int64_t>(1, grid_mm))), e0, e1, p->heavy, (const int32_t *)p->act,
                                                (const double *)p->Gall, (const int32_t *)p->cend, p->best, p->counters, (const PruneState *)p->state, a, ca, cm,
                                                n_groups, n_seg_mm));
        } else
        TSC_TRY(launch_rmsd_sieve_sorted(a.heavy32 != nullptr, st, sgrid, e0, e1, p->heavy, (const int32_t *)p->act, (const double *)p->Gall, (const int32_t *)p->cend,
                                         p->best, p->counters, (const PruneState *)p->state, a, ca, my_tiles, n_seg));
        if (range) {
            // a partitioned pass is closed by tsc_prune_pass_merge after the exchange: this rank's verdicts go into the exchange buffer
            // now (k_apply_pass in its noting form), its last block leaves the statistics there
            int nxt = -1;
            const StepArgs sa2 = next_step_args(p, &nxt);
            const int blocks = int(std::min<int64_t>(ceil_div<int64_t>(A, 256), 512));
            hipLaunchKernelGGL(k_apply_pass, dim3(blocks), dim3(256), 0, st, apply_args(p), step_ctx(p, true), sa2);
            TSC_HIP(hipGetLastError());
        }
        p->local_done = true;
        return 0;
    }
    TSC_TRY(launch_pair_search(p, rank, world, A, rows_now));
    p->local_done = true;
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_pass_local(tsc_prune *p, int rank, int world) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    TSC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
    if (p->cur_k == 0 || p->local_done) return fail(TSC_ERR_STATE, "tsc_prune_pass_local: no pass open (call tsc_prune_next_pass)");
    if (p->views_split && p->mode == 0)
        return fail(TSC_ERR_STATE, "tsc_prune_pass_local: rank-partitioned passes have run; sum the cache views over the ranks first "
                                   "(tsc_prune_views_ptr, tsc_prune_views_merged)");
    return pass_launch(p, rank, world, false);
    TSC_API_GUARD_END
}

// ---- rank-partitioned passes (rmsd.hpp, k_pass_merge) ----
// words of the exchange buffer of a run over n structures: the removed-row bits of a pass (bit_words of prune_create_impl), eight
// words of statistics, then -- reference-exact mode -- the storage of every cache view of the run (so that the host can sum the
// views of the remaining passes over the ranks in place, as part of a buffer it owns)
static int64_t views_words_of(int64_t n, int mode) {
    if (mode != 0) return 0;
    int n_views = 0;
    for (int slot = 0; slot < TSC_MAX_PASSES; ++slot) n_views += (int64_t(KS[slot]) == 1 || 20 * int64_t(KS[slot]) < n) ? 1 : 0;
    const int64_t bit_words = n / 64 + 40;
    return int64_t(n_views) * (bit_words + bit_words / 1024 + 4);
}
extern "C" __attribute__((visibility("default"))) int tsc_prune_exchange_words(int64_t n, int mode, int64_t *words) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(words && n > 0, "bad argument");
    *words = n / 64 + 40 + 8 + views_words_of(n, mode);
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_set_partition(tsc_prune *p, int rank, int world, int min_chunks_per_rank, void *exch_dev,
                                                                              int64_t exch_words) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && exch_dev, "null argument");
    TSC_REQUIRE(world >= 1 && rank >= 0 && rank < world && min_chunks_per_rank >= 1, "bad rank %d / world %d / min_chunks_per_rank %d", rank, world,
                min_chunks_per_rank);
    const int64_t need = int64_t(p->bit_words) + 8 + views_words_of(p->n, p->mode);
    TSC_REQUIRE(exch_words >= need, "exchange buffer of %lld words, %lld needed (tsc_prune_exchange_words)", (long long)exch_words, (long long)need);
    if (p->cur_k != 0 || p->next_ks != 0) return fail(TSC_ERR_STATE, "tsc_prune_set_partition: call it right after tsc_prune_create");
    DeviceGuard guard(p->ctx->device);
    p->part_rank = rank, p->part_world = world, p->part_min_chunks = min_chunks_per_rank;
    p->exch = static_cast<unsigned long long *>(exch_dev);
    TSC_HIP(hipMemsetAsync(p->exch, 0, size_t(need) * sizeof(unsigned long long), p->ctx->stream));
    if (p->mode == 0) p->views = p->exch + p->bit_words + 8;  // the cache views live in the caller's buffer from here on (still empty)
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_pass_partitioned(tsc_prune *p, int *flag) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && flag, "null argument");
    if (p->cur_k == 0) return fail(TSC_ERR_STATE, "tsc_prune_pass_partitioned: no pass open");
    *flag = pass_is_partitioned(p, p->cur_k) ? 1 : 0;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_pass_range(tsc_prune *p) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    if (p->cur_k == 0 || p->local_done) return fail(TSC_ERR_STATE, "tsc_prune_pass_range: no pass open (call tsc_prune_next_pass)");
    if (!pass_is_partitioned(p, p->cur_k)) return fail(TSC_ERR_STATE, "tsc_prune_pass_range: the open pass (k = %lld) is not rank-partitioned", (long long)p->cur_k);
    TSC_TRY(pass_launch(p, 0, 1, true));
    p->views_split = true;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_pass_merge(tsc_prune *p) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    if (p->cur_k == 0 || !p->local_done || !p->cur_range) return fail(TSC_ERR_STATE, "tsc_prune_pass_merge: tsc_prune_pass_range has not run");
    tsc_ctx *c = p->ctx;
    DeviceGuard guard(c->device);
    int nxt = -1;
    const StepArgs sa = next_step_args(p, &nxt);
    MergeArgs ma;
    ma.n = int(p->n), ma.bit_words = int(p->bit_words), ma.n_blocks = p->n_blocks, ma.bits = p->bits, ma.exch = p->exch, ma.mask = p->mask;
    ma.next_s_lo = ma.next_s_hi = -1;
    if (nxt >= 0 && pass_is_partitioned(p, int64_t(KS[nxt]))) {
        int64_t c_lo, c_hi, s_lo, s_hi;
        partition_bounds(p->n, int64_t(KS[nxt]), p->part_rank, p->part_world, &c_lo, &c_hi, &s_lo, &s_hi);
        ma.next_s_lo = int(s_lo), ma.next_s_hi = int(s_hi);
        p->range_ready_slot = nxt;
    }
    hipLaunchKernelGGL(k_pass_merge, dim3(1), dim3(1024), 0, c->stream, ma, step_ctx(p), sa);
    TSC_HIP(hipGetLastError());
    p->opened_slot = nxt;
    p->last_slot = -1;  // closed on the device
    if (c->pass_timing >= 2) TSC_HIP(hipEventRecord(p->ev[p->cur_slot][3], c->stream));
    p->cur_k = 0;
    p->cur_slot = -1;
    p->cur_range = false;
    p->collected = false;
    return 0;
    TSC_API_GUARD_END
}

// The cache views of the passes that have not run yet (the open one included), as one block of 64-bit words: after partitioned
// passes they hold the keys of this rank's removed rows only.  *words = 0: nothing to exchange (cache-free mode, or no partitioned
// pass has run).  Otherwise: sum the block over the ranks (the ranks' bits are disjoint), then tsc_prune_views_merged.
extern "C" __attribute__((visibility("default"))) int tsc_prune_views_ptr(tsc_prune *p, void **views_dev, int64_t *offset_words, int64_t *words) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && views_dev && offset_words && words, "null argument");
    if (p->cur_k == 0) return fail(TSC_ERR_STATE, "tsc_prune_views_ptr: no pass open");
    *views_dev = nullptr, *offset_words = 0, *words = 0;
    if (!p->views_split || p->mode != 0) return 0;
    const int v = p->view_of_slot[p->cur_slot];
    *views_dev = p->views + size_t(v) * (p->bit_words + p->dsum_words);
    *offset_words = int64_t(p->views - p->exch) + int64_t(v) * int64_t(p->bit_words + p->dsum_words);
    *words = int64_t(p->n_views - v) * int64_t(p->bit_words + p->dsum_words);
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_views_merged(tsc_prune *p) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    if (p->cur_k == 0) return fail(TSC_ERR_STATE, "tsc_prune_views_merged: no pass open");
    if (p->views_split && p->mode == 0) {
        DeviceGuard guard(p->ctx->device);
        const int v = p->view_of_slot[p->cur_slot], count = p->n_views - v;
        hipLaunchKernelGGL(k_views_summaries, dim3(unsigned(std::min<int64_t>(ceil_div<int64_t>(int64_t(count) * p->dsum_words * 64, 256), 2048))), dim3(256), 0, p->ctx->stream,
                           p->views + size_t(v) * (p->bit_words + p->dsum_words), (long long)(p->bit_words + p->dsum_words), int(p->bit_words),
                           int(p->dsum_words), count);
        TSC_HIP(hipGetLastError());
    }
    p->views_split = false;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_pass_rows(tsc_prune *p, int rank, int world) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    TSC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
    if (p->cur_k == 0 || !p->local_done || p->cur_local || p->cur_fused)
        return fail(TSC_ERR_STATE, "tsc_prune_pass_rows: needs an open pass whose tsc_prune_pass_local ran with world_size > 1");
    DeviceGuard guard(p->ctx->device);
    return launch_pair_search(p, rank, world, p->n);
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_best_ptr(tsc_prune *p, void **best_dev, int64_t *n_entries) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && best_dev && n_entries, "null argument");
    if (p->cur_k == 0) return fail(TSC_ERR_STATE, "tsc_prune_best_ptr: no pass open");
    *best_dev = p->best;
    *n_entries = p->n;  // entries beyond the (device-side) active count are not touched by the pass
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_use_best_buffer(tsc_prune *p, void *best_dev) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && best_dev, "null argument");
    if (p->cur_k != 0 || p->next_ks != 0) return fail(TSC_ERR_STATE, "tsc_prune_use_best_buffer: call it right after tsc_prune_create");
    p->best = static_cast<int32_t *>(best_dev);
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_pass_finish(tsc_prune *p) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    if (p->cur_k == 0 || !p->local_done) return fail(TSC_ERR_STATE, "tsc_prune_pass_finish: tsc_prune_pass_local has not run");
    if (p->cur_range) return fail(TSC_ERR_STATE, "tsc_prune_pass_finish: a rank-partitioned pass is closed by tsc_prune_pass_merge");
    tsc_ctx *c = p->ctx;
    DeviceGuard guard(c->device);
    if (!p->cur_local && !p->cur_fused) {  // (a chunk-local pass, and the pair kernel of a fused one, have applied the verdicts already)
        int nxt = -1;
        const StepArgs sa = next_step_args(p, &nxt);
        const int blocks = int(std::min<int64_t>(ceil_div<int64_t>(p->n, 256), 512));
        hipLaunchKernelGGL(k_apply_pass, dim3(blocks), dim3(256), 0, c->stream, apply_args(p), step_ctx(p), sa);
        p->opened_slot = nxt;
        p->last_slot = -1;  // closed on the device
        TSC_HIP(hipGetLastError());
    }
    if (c->pass_timing >= 2) TSC_HIP(hipEventRecord(p->ev[p->cur_slot][3], c->stream));
    p->cur_k = 0;
    p->cur_slot = -1;
    p->collected = false;
    return 0;
    TSC_API_GUARD_END
}

// Runs every pass that needs no exchange between ranks (all of them for world == 1; for world > 1 those whose estimate is
// below min_pairs: every rank computes them whole and reaches the same verdicts) and returns with the first pass that does
// open (*k_out = its k; the caller runs tsc_prune_pass_local(rank, world), merges best[], tsc_prune_pass_finish) or with
// *k_out = 0 when the schedule is exhausted.  One host call instead of three per small pass.
extern "C" __attribute__((visibility("default"))) int tsc_prune_run_replicated(tsc_prune *p, int world, int64_t min_pairs, int64_t *k_out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && k_out && world >= 1, "null argument");
    for (;;) {
        int64_t k = 0;
        TSC_TRY(tsc_prune_next_pass(p, &k));
        *k_out = k;
        if (k == 0) return 0;
        // (a partitioned pass, or the first pass after partitioned ones -- the cache views must be summed over the ranks first --
        // goes back to the caller as well)
        if (world > 1 && (pass_is_partitioned(p, k) || (p->views_split && p->mode == 0) || p->n * (p->n / k) / 2 >= min_pairs)) return 0;
        TSC_TRY(tsc_prune_pass_local(p, 0, 1));
        TSC_TRY(tsc_prune_pass_finish(p));
    }
    TSC_API_GUARD_END
}

// The whole pass loop of a SHARDED run behind one call (the multi-rank variant of the prune; 8e: the protocol).  The library
// walks the schedule exactly as tscode_amd/pipeline.py::sharded_step does -- passes below `min_pairs` whole on every rank, passes with at
// least `min_chunks_per_rank` chunks per rank partitioned by chunks (removed-row bits summed), the cache views summed once before the
// first pass of the other kind, the remaining large passes dealt by row tiles (best[] min-merged) -- and hands the host nothing but the
// collectives: `exchange(user, kind, buf, count)` must reduce the `count` elements at device address `buf` over the ranks IN PLACE, in
// stream order with the context's stream (enqueue it there, or synchronise on both sides), and return 0.  One process per GPU owns the
// communicator (RCCL through torch.distributed, or ncclAllReduce on the context's stream from a C host); the library opens none.
extern "C" __attribute__((visibility("default"))) int tsc_prune_run_sharded(tsc_prune *p, int rank, int world, int min_chunks_per_rank, int64_t min_pairs,
                                                                            void *exch_dev, int64_t exch_words, tsc_exchange_fn exchange, void *user,
                                                                            tsc_exchange_record *log, int log_cap, int *n_log) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    TSC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
    TSC_REQUIRE(world == 1 || exchange != nullptr, "tsc_prune_run_sharded: %d ranks and no exchange function", world);
    TSC_REQUIRE(min_chunks_per_rank >= 0 && log_cap >= 0 && (log || log_cap == 0), "bad argument");
    int logged = 0;
    if (n_log) *n_log = 0;
    auto xchg = [&](int kind, void *buf, int64_t count, int64_t k) -> int {
        if (log && logged < log_cap) log[logged] = tsc_exchange_record{k, kind, count};
        ++logged;
        if (n_log) *n_log = std::min(logged, log_cap);
        if (world == 1 && !exchange) return 0;
        const int rc = exchange(user, kind, buf, count);
        if (rc != 0) return fail(TSC_ERR_STATE, "tsc_prune_run_sharded: the exchange function returned %d (pass k = %lld, kind %d, %lld elements)", rc,
                                 (long long)k, kind, (long long)count);
        return 0;
    };
    const bool can_partition = world > 1 && min_chunks_per_rank > 0 && exch_dev != nullptr && p->algo == ALGO_SIEVE;
    if (can_partition) TSC_TRY(tsc_prune_set_partition(p, rank, world, min_chunks_per_rank, exch_dev, exch_words));
    for (;;) {
        int64_t k = 0;
        TSC_TRY(tsc_prune_run_replicated(p, world, min_pairs, &k));  // (every pass that needs no exchange; returns with the first that does, open)
        if (k == 0) break;
        if (pass_is_partitioned(p, k)) {
            // the whole pass on this rank's chunks; what the ranks tell each other is which rows they removed (+ the statistics)
            TSC_TRY(tsc_prune_pass_range(p));
            TSC_TRY(xchg(TSC_XCHG_SUM_I64, p->exch, int64_t(p->bit_words) + 8, k));
            TSC_TRY(tsc_prune_pass_merge(p));
            continue;
        }
        if (p->views_split && p->mode == 0) {
            // first pass after the partitioned ones: every rank needs every rank's cache keys from here on
            void *views = nullptr;
            int64_t off = 0, words = 0;
            TSC_TRY(tsc_prune_views_ptr(p, &views, &off, &words));
            if (words > 0) TSC_TRY(xchg(TSC_XCHG_SUM_I64, views, words, -k));
            TSC_TRY(tsc_prune_views_merged(p));
        }
        if (world > 1 && p->n * (p->n / k) / 2 >= min_pairs) {
            TSC_TRY(tsc_prune_pass_local(p, rank, world));  // this rank's row tiles only ...
            TSC_TRY(xchg(TSC_XCHG_MIN_I32, p->best, p->n, k));  // ... merged
        } else {
            TSC_TRY(tsc_prune_pass_local(p, 0, 1));  // (a small pass that only came back for the views' exchange)
        }
        TSC_TRY(tsc_prune_pass_finish(p));
    }
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_mask_dev(tsc_prune *p, const uint8_t **mask_dev) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && mask_dev, "null argument");
    *mask_dev = p->mask;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_copy_mask_dev(tsc_prune *p, uint8_t *dst) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p && dst, "null argument");
    DeviceGuard guard(p->ctx->device);
    TSC_HIP(hipMemcpyAsync(dst, p->mask, size_t(p->n), hipMemcpyDeviceToDevice, p->ctx->stream));
    return 0;
    TSC_API_GUARD_END
}

// Close the last pass on the device, read the records back (the one synchronisation of a run) and build the
// per-pass statistics of the passes whose gate was open.
extern "C" __attribute__((visibility("default"))) int tsc_prune_stats(tsc_prune *p, tsc_pass_stats *stats, int *n_passes) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(p != nullptr, "null argument");
    if (p->cur_k != 0) return fail(TSC_ERR_STATE, "tsc_prune_stats: a pass is still open");
    tsc_ctx *c = p->ctx;
    DeviceGuard guard(c->device);
    if (!p->collected) {
        hipStream_t st = c->stream;
        if (p->last_slot >= 0) {
            StepArgs sa{p->last_slot, -1, 0ll, 0, -1};
            hipLaunchKernelGGL(k_pass_step, dim3(1), dim3(64), 0, st, step_ctx(p), sa);
            p->last_slot = -1;
        }
        static_assert(sizeof(PassRecord) * TSC_MAX_PASSES <= 4096 && sizeof(PassRecord) % 8 == 0, "records fit the pinned staging buffer");
        {
            const int rec_words = int(sizeof(PassRecord) * TSC_MAX_PASSES / 8);
            const int64_t mask_words = p->export_mask_host ? p->n / 8 : 0;
            hipLaunchKernelGGL(k_export_run, dim3(grid_for(std::max<int64_t>(mask_words, rec_words), 256, 64)), dim3(256), 0, st,
                               reinterpret_cast<const unsigned long long *>(p->records), rec_words, static_cast<unsigned long long *>(c->pinned),
                               reinterpret_cast<const unsigned long long *>(p->mask), mask_words, (const uint8_t *)p->mask, p->n,
                               reinterpret_cast<unsigned long long *>(p->export_mask_host), (const unsigned *)p->dmax_bits);
            TSC_HIP(hipGetLastError());
            p->export_mask_host = nullptr;
        }
        TSC_HIP(hipStreamSynchronize(st));
        const PassRecord *rec = static_cast<const PassRecord *>(c->pinned);
        const bool nonfinite = unsigned(static_cast<const unsigned long long *>(c->pinned)[sizeof(PassRecord) * TSC_MAX_PASSES / 8]) >= 0x7f800000u;
        p->n_passes = 0;
        for (int slot = 0; slot < TSC_MAX_PASSES; ++slot) {
            if (!p->slot_used[slot] || !rec[slot].on) continue;
            tsc_pass_stats &s = p->stats[p->n_passes++];
            memset(&s, 0, sizeof(s));
            s.k = rec[slot].k, s.n_active_before = rec[slot].n_before, s.n_active_after = rec[slot].n_after;
            s.pairs_evaluated = rec[slot].evaluated, s.pairs_computed = rec[slot].formed, s.candidates = rec[slot].exact;
            s.pairs_screened = rec[slot].screened, s.new_keys = rec[slot].removed, s.algo = rec[slot].algo;
            s.nonfinite_input = nonfinite ? 1 : 0;
            float ms = 0;
            if (c->pass_timing >= 2 && hipEventElapsedTime(&ms, p->ev[slot][0], p->ev[slot][3]) == hipSuccess) s.gpu_ms = ms;
            if (c->pass_timing >= (rec[slot].algo == ALGO_LOCAL ? 2 : 1) && hipEventElapsedTime(&ms, p->ev[slot][1], p->ev[slot][2]) == hipSuccess) s.tile_ms = ms;
        }
        p->collected = true;
    }
    if (stats) memcpy(stats, p->stats, sizeof(tsc_pass_stats) * size_t(p->n_passes));
    if (n_passes) *n_passes = p->n_passes;
    return 0;
    TSC_API_GUARD_END
}

// One whole run on device data; mask_host (optional) also receives the verdicts, copied before the run's single
// synchronisation (the statistics read-back).
int prune_run(tsc_ctx *c, const double *heavy, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask, uint8_t *mask_host,
                     tsc_pass_stats *stats, int *n_passes, const double *basis, const ExternalDescriptors *ext, int force_algo) {
    tsc_prune *p = nullptr;
    const bool in_place = (reinterpret_cast<uintptr_t>(mask) & 7u) == 0;  // run on the caller's buffer: no copy at the end
    TSC_TRY(prune_create_impl(c, heavy, n, h, rmsd_thr, mode, in_place ? mask : nullptr, &p, basis, ext, force_algo));
    int rc = 0;
    for (;;) {
        int64_t k = 0;
        if ((rc = tsc_prune_next_pass(p, &k)) != 0 || k == 0) break;
        if ((rc = tsc_prune_pass_local(p, 0, 1)) != 0) break;
        if ((rc = tsc_prune_pass_finish(p)) != 0) break;
    }
    if (!rc) {
        DeviceGuard guard(c->device);
        hipError_t e = in_place ? hipSuccess : hipMemcpyAsync(mask, p->mask, size_t(n), hipMemcpyDeviceToDevice, c->stream);
        // the verdicts go to the host with the statistics (one launch, k_export_run) when the buffer is pinned host memory the
        // device can write; any other pointer takes a copy command
        p->export_mask_host = nullptr;
        if (e == hipSuccess && mask_host) {
            hipPointerAttribute_t at;
            const bool mapped = (reinterpret_cast<uintptr_t>(mask_host) & 7u) == 0 && (reinterpret_cast<uintptr_t>(p->mask) & 7u) == 0 &&
                                hipPointerGetAttributes(&at, mask_host) == hipSuccess && at.type == hipMemoryTypeHost;
            // the address the DEVICE sees: for hipHostRegister'ed or non-mapped pinned memory it need not be the host address,
            // and may not exist at all
            uint8_t *dev_view = mapped ? static_cast<uint8_t *>(at.devicePointer) : nullptr;
            if (dev_view && (reinterpret_cast<uintptr_t>(dev_view) & 7u) == 0) {
                p->export_mask_host = dev_view;
            } else {
                (void)hipGetLastError();
                e = hipMemcpyAsync(mask_host, p->mask, size_t(n), hipMemcpyDeviceToHost, c->stream);
            }
        }
        if (e != hipSuccess) rc = fail(TSC_ERR_HIP, "mask copy failed: %s", hipGetErrorString(e));
    }
    if (!rc) rc = tsc_prune_stats(p, stats, n_passes);
    tsc_prune_destroy(p);
    return rc;
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_rmsd_dev(tsc_ctx *c, const double *heavy, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask,
                                  tsc_pass_stats *stats, int *n_passes) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && heavy && mask, "tsc_prune_rmsd_dev: null argument");
    if (n == 0) {
        if (n_passes) *n_passes = 0;
        return 0;
    }
    return prune_run(c, heavy, n, h, rmsd_thr, mode, mask, nullptr, stats, n_passes);
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_rmsd(tsc_ctx *c, const double *heavy, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask,
                              tsc_pass_stats *stats, int *n_passes) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && heavy && mask, "tsc_prune_rmsd: null argument");
    TSC_REQUIRE(n >= 0 && h > 0, "bad sizes");
    if (n == 0) {
        if (n_passes) *n_passes = 0;
        return 0;
    }
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_heavy;
    uint8_t *d_mask;
    TSC_TRY(upload(c, s, heavy, size_t(n) * h * 3, &d_heavy));
    TSC_TRY(s.get(size_t(n), &d_mask));
    TSC_TRY(tsc_prune_rmsd_dev(c, d_heavy, n, h, rmsd_thr, mode, d_mask, stats, n_passes));
    TSC_HIP(hipMemcpyAsync(mask, d_mask, size_t(n), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// prune_conformers_rmsd as the reference calls it (rmsd_pruning.py:164-206): ALL atoms of every structure in host memory plus
// the indices of the heavy ones.  The heavy-atom gather `structures[:, atomnos != 1]` (:178-179) runs on the device: on the
// host it is a strided 40 MB copy that costs ten times the prune at 57k structures.
extern "C" __attribute__((visibility("default"))) int tsc_prune_structures(tsc_ctx *c, const double *structures, int64_t n, int n_atoms, const int32_t *heavy_idx,
                                                                           int n_heavy, double rmsd_thr, int mode, uint8_t *mask, tsc_pass_stats *stats,
                                                                           int *n_passes) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && structures && heavy_idx && mask, "tsc_prune_structures: null argument");
    TSC_REQUIRE(n >= 0 && n_atoms > 0 && n_heavy > 0 && n_heavy <= n_atoms, "bad sizes");
    if (n == 0) {
        if (n_passes) *n_passes = 0;
        return 0;
    }
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_all, *d_heavy;
    uint8_t *d_mask;
    TSC_TRY(upload(c, s, structures, size_t(n) * n_atoms * 3, &d_all));
    TSC_TRY(s.get(size_t(n) * n_heavy * 3, &d_heavy));
    TSC_TRY(s.get(size_t(n), &d_mask));
    TSC_TRY(tsc_gather_heavy_dev(c, d_all, nullptr, n, n_atoms, heavy_idx, n_heavy, d_heavy, nullptr));
    TSC_TRY(tsc_prune_rmsd_dev(c, d_heavy, n, n_heavy, rmsd_thr, mode, d_mask, stats, n_passes));
    TSC_HIP(hipMemcpyAsync(mask, d_mask, size_t(n), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

