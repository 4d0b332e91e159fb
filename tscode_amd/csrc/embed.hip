// embed.hip -- K1 / K2: batched embed, clash masks, distances, ordered compaction
// gfx950 only.  There is deliberately no CPU implementation behind these entry points.
#include "host.hpp"
#include "scan.hpp"

// --------------------------------------------------------------------------------------------------
// K1

extern "C" __attribute__((visibility("default"))) int tsc_transform_batch_dev(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                       const int32_t *n_conf, int n_mols, const int32_t *conf_idx, const double *rot,
                                       const double *pos, int64_t n_poses, double *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && conf_idx && rot && pos && out, "tsc_transform_batch_dev: null argument");
    TSC_REQUIRE(n_poses >= 0, "negative n_poses");
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    if (n_poses == 0) return 0;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_transform, dim3(grid_for(n_poses, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), c->stream, frags, ft, conf_idx, rot,
                       pos, (const int32_t *)nullptr, n_poses, out, (const int32_t *)nullptr, 0, (double *)nullptr, (const int32_t *)nullptr);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}


extern "C" __attribute__((visibility("default"))) int tsc_transform_batch(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                   const int32_t *n_conf, int n_mols, const int32_t *conf_idx, const double *rot,
                                   const double *pos, int64_t n_poses, double *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && conf_idx && rot && pos && out, "tsc_transform_batch: null argument");
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    TSC_REQUIRE(n_poses >= 0, "negative n_poses");
    for (int64_t i = 0; i < n_poses * n_mols; ++i)
        TSC_REQUIRE(conf_idx[i] >= 0 && conf_idx[i] < n_conf[i % n_mols], "conf_idx[%lld] out of range", (long long)i);
    if (n_poses == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_frags, *d_rot, *d_pos, *d_out;
    int32_t *d_ci;
    TSC_TRY(upload(c, s, frags, size_t(frags_total_doubles(frag_off, n_atoms, n_conf, n_mols)), &d_frags));
    TSC_TRY(upload(c, s, conf_idx, size_t(n_poses) * n_mols, &d_ci));
    TSC_TRY(upload(c, s, rot, size_t(n_poses) * n_mols * 9, &d_rot));
    TSC_TRY(upload(c, s, pos, size_t(n_poses) * n_mols * 3, &d_pos));
    TSC_TRY(s.get(size_t(n_poses) * ft.n_total * 3, &d_out));
    TSC_TRY(tsc_transform_batch_dev(c, d_frags, frag_off, n_atoms, n_conf, n_mols, d_ci, d_rot, d_pos, n_poses, d_out));
    TSC_HIP(hipMemcpyAsync(out, d_out, size_t(n_poses) * ft.n_total * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// K2

static int make_clash_args(int64_t n_poses, int n_atoms, const int32_t *ids, int n_ids, double thresh, int64_t max_clashes,
                           ClashArgs *a) {
    TSC_REQUIRE(n_poses >= 0 && n_atoms > 0, "bad sizes");
    TSC_REQUIRE(n_ids == 0 || n_ids == 2 || n_ids == 3, "ids must have 0 (None), 2 or 3 entries, got %d", n_ids);
    memset(a, 0, sizeof(*a));
    a->n_poses = n_poses;
    a->n = n_atoms;
    a->max_clashes = max_clashes;
    if (n_ids == 0) {
        a->self_mode = 1;
        a->first_row = 0;
        a->n_mols = 1;
        a->sq_bound = clash_sq_bound(0.5);  // numba_functions.py:54
        for (int m = 1; m <= MAX_MOLS; ++m) a->atom_off[m] = n_atoms;
    } else {
        TSC_REQUIRE(ids != nullptr, "ids is null");
        int off = 0;
        // like the reference (numba_functions.py:77-78, 88-90) the LAST fragment takes whatever is left
        for (int m = 0; m < n_ids; ++m) {
            a->atom_off[m] = off;
            TSC_REQUIRE(ids[m] >= 0, "negative fragment length");
            off += ids[m];
        }
        TSC_REQUIRE(off - ids[n_ids - 1] <= n_atoms, "fragment lengths exceed the atom count");
        for (int m = n_ids; m <= MAX_MOLS; ++m) a->atom_off[m] = n_atoms;
        a->n_mols = n_ids;
        a->first_row = a->atom_off[1];
        a->sq_bound = clash_sq_bound(thresh);
    }
    int rows = std::max(1, n_atoms - a->first_row);
    a->lp = std::min(64, std::max(4, pow2_ceil(rows)));
    return 0;
}

template <bool FUSED, bool SELF, bool MINMODE>
static int launch_clash_impl(tsc_ctx *c, const ClashArgs &a, const double *coords, const double *frags, const FragTable &ft,
                             const int32_t *conf_idx, const double *rot, const double *pos, uint8_t *mask, int32_t *counts) {
    const int ppw = 64 / a.lp;
    size_t lds = size_t(4) * clash_lds_per_wave(a.n, a.lp, MINMODE);
    TSC_REQUIRE(lds <= 160 * 1024, "pose too large for the LDS staging of the clash kernel (%d atoms)", a.n);
    int64_t waves = ceil_div<int64_t>(a.n_poses, ppw);
    int blocks = grid_for(waves, 4, 256 * 32);  // (measured: 8192 workgroups beat 2048 by 15 % at 500k x 200 -- a wavefront that loops over poses is a chain of load latencies)
    if (lds > 64 * 1024)
        TSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_clash<FUSED, SELF, MINMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL((k_clash<FUSED, SELF, MINMODE>), dim3(blocks), dim3(256), lds, c->stream, a, coords, frags, ft, conf_idx, rot, pos, mask, counts);
    TSC_HIP(hipGetLastError());
    return 0;
}

template <bool FUSED>
static int launch_clash(tsc_ctx *c, const ClashArgs &a, const double *coords, const double *frags, const FragTable &ft,
                        const int32_t *conf_idx, const double *rot, const double *pos, uint8_t *mask, int32_t *counts) {
    if (a.n_poses == 0) return 0;
    if (a.self_mode) return launch_clash_impl<FUSED, true, false>(c, a, coords, frags, ft, conf_idx, rot, pos, mask, counts);
    // verdict only and no clash allowed: the packed-fp32 minimum with its fp64 fallback (embed_clash.hpp)
    const bool minmode = !counts && a.max_clashes == 0 && c->clash_fp32 != 0 && 4 * clash_lds_per_wave(a.n, a.lp, true) <= 160 * 1024;
    if (minmode && FUSED && c->clash_lanes != 0 && a.n_mols == 2 && ft.n_mols == 2 && a.atom_off[1] == ft.atom_off[1] && a.n == ft.n_total &&
        std::min(ft.n_atoms[0], ft.n_atoms[1]) >= 1 && std::min(ft.n_atoms[0], ft.n_atoms[1]) <= 32) {
        // one pose per lane, the smaller fragment in registers (embed_clash.hpp, k_clash_lanes)
        const int mA = ft.n_atoms[0] <= ft.n_atoms[1] ? 0 : 1, mB = 1 - mA, na2 = (ft.n_atoms[mA] + 1) / 2;
        const dim3 grid(unsigned(grid_for(ceil_div<int64_t>(a.n_poses, 64), 4, 256 * 32)));
#define TSC_LAUNCH_CLASH_LANES(N)                                                                                                          \
    hipLaunchKernelGGL(k_clash_lanes<N>, grid, dim3(256), 0, c->stream, a.n_poses, frags, ft, mA, mB, conf_idx, rot, pos, a.sq_bound, mask)
        if (na2 <= 4) TSC_LAUNCH_CLASH_LANES(4);
        else if (na2 <= 8) TSC_LAUNCH_CLASH_LANES(8);
        else if (na2 <= 12) TSC_LAUNCH_CLASH_LANES(12);
        else if (na2 <= 13) TSC_LAUNCH_CLASH_LANES(13);
        else TSC_LAUNCH_CLASH_LANES(16);
#undef TSC_LAUNCH_CLASH_LANES
        TSC_HIP(hipGetLastError());
        return 0;
    }
    if (minmode && FUSED && c->clash_lanes != 0 && (a.n_mols == 2 || a.n_mols == 3) && ft.n_mols == a.n_mols && a.n == ft.n_total &&
        a.atom_off[1] == ft.atom_off[1] && (a.n_mols == 2 || a.atom_off[2] == ft.atom_off[2])) {
        // one pose per lane for fragments of any size and for three of them: the "A" fragment of every pair in register tiles, the poses
        // packed again between the fragment pairs (embed_clash.hpp, k_clash_lanes_multi).  The tile size that wastes the fewest padded
        // distances over the pairs (numba_functions.py:87-105: (m2, m1), (m3, m2), (m1, m3)), the B atoms embedded once per tile counted in
        auto cost = [&](int na2) {
            int64_t w = 0;
            const int n_pairs = a.n_mols == 2 ? 1 : 3;
            for (int pr = 0; pr < n_pairs; ++pr) {
                const int mA = a.n_mols == 2 ? 1 : (pr == 0 ? 1 : (pr == 1 ? 2 : 0)), mB = a.n_mols == 2 ? 0 : (pr == 0 ? 0 : (pr == 1 ? 1 : 2));
                const int64_t tiles = ceil_div(ft.n_atoms[mA], 2 * na2);
                w += tiles * ft.n_atoms[mB] * (2 * na2 + 4);
            }
            return w;
        };
        int best = 12;
        for (int na2 : {16, 18})
            if (cost(na2) < cost(best)) best = na2;
        const dim3 grid(unsigned(grid_for(ceil_div<int64_t>(a.n_poses, CLM_POSES), 1, 256 * 32)));
#define TSC_LAUNCH_CLASH_MULTI(N)                                                                                                          \
    hipLaunchKernelGGL(k_clash_lanes_multi<N>, grid, dim3(256), 0, c->stream, a.n_poses, frags, ft, conf_idx, rot, pos, a.sq_bound, mask)
        if (best == 12) TSC_LAUNCH_CLASH_MULTI(12);
        else if (best == 16) TSC_LAUNCH_CLASH_MULTI(16);
        else TSC_LAUNCH_CLASH_MULTI(18);
#undef TSC_LAUNCH_CLASH_MULTI
        TSC_HIP(hipGetLastError());
        return 0;
    }
    if (minmode) return launch_clash_impl<FUSED, false, true>(c, a, coords, frags, ft, conf_idx, rot, pos, mask, counts);
    return launch_clash_impl<FUSED, false, false>(c, a, coords, frags, ft, conf_idx, rot, pos, mask, counts);
}

extern "C" __attribute__((visibility("default"))) int tsc_clash_mask_dev(tsc_ctx *c, const double *coords, int64_t n_poses, int n_atoms, const int32_t *ids, int n_ids,
                                  double thresh, int64_t max_clashes, uint8_t *mask, int32_t *counts) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && mask, "tsc_clash_mask_dev: null argument");
    ClashArgs a;
    TSC_TRY(make_clash_args(n_poses, n_atoms, ids, n_ids, thresh, max_clashes, &a));
    DeviceGuard guard(c->device);
    FragTable ft;
    memset(&ft, 0, sizeof(ft));
    return launch_clash<false>(c, a, coords, nullptr, ft, nullptr, nullptr, nullptr, mask, counts);
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_clash_mask(tsc_ctx *c, const double *coords, int64_t n_poses, int n_atoms, const int32_t *ids, int n_ids,
                              double thresh, int64_t max_clashes, uint8_t *mask, int32_t *counts) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && mask, "tsc_clash_mask: null argument");
    ClashArgs a;
    TSC_TRY(make_clash_args(n_poses, n_atoms, ids, n_ids, thresh, max_clashes, &a));
    if (n_poses == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_coords;
    uint8_t *d_mask;
    int32_t *d_counts = nullptr;
    TSC_TRY(upload(c, s, coords, size_t(n_poses) * n_atoms * 3, &d_coords));
    TSC_TRY(s.get(size_t(n_poses), &d_mask));
    if (counts) TSC_TRY(s.get(size_t(n_poses), &d_counts));
    TSC_TRY(tsc_clash_mask_dev(c, d_coords, n_poses, n_atoms, ids, n_ids, thresh, max_clashes, d_mask, d_counts));
    TSC_HIP(hipMemcpyAsync(mask, d_mask, size_t(n_poses), hipMemcpyDeviceToHost, c->stream));
    if (counts) TSC_HIP(hipMemcpyAsync(counts, d_counts, size_t(n_poses) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_embed_clash_mask_dev(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                        const int32_t *n_conf, int n_mols, const int32_t *conf_idx, const double *rot,
                                        const double *pos, int64_t n_poses, double thresh, int64_t max_clashes, uint8_t *mask,
                                        int32_t *counts) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && conf_idx && rot && pos && mask, "tsc_embed_clash_mask_dev: null argument");
    TSC_REQUIRE(n_mols == 2 || n_mols == 3, "the fused embed+clash path needs 2 or 3 fragments, got %d", n_mols);
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    ClashArgs a;
    TSC_TRY(make_clash_args(n_poses, ft.n_total, n_atoms, n_mols, thresh, max_clashes, &a));
    DeviceGuard guard(c->device);
    return launch_clash<true>(c, a, nullptr, frags, ft, conf_idx, rot, pos, mask, counts);
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_all_dists(tsc_ctx *c, const double *A, int na, const double *B, int nb, double *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && A && B && out && na >= 0 && nb >= 0, "tsc_all_dists: bad argument");
    if (na == 0 || nb == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *dA, *dB, *dO;
    TSC_TRY(upload(c, s, A, size_t(na) * 3, &dA));
    TSC_TRY(upload(c, s, B, size_t(nb) * 3, &dB));
    TSC_TRY(s.get(size_t(na) * nb, &dO));
    hipLaunchKernelGGL(k_all_dists, dim3(grid_for(int64_t(na) * nb, 256)), dim3(256), 0, c->stream, dA, na, dB, nb, dO);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(out, dO, size_t(na) * nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}


// --------------------------------------------------------------------------------------------------
// ordered compaction

extern "C" __attribute__((visibility("default"))) int tsc_compact_rows_dev(tsc_ctx *c, const void *src, const uint8_t *mask, int64_t n_rows, int64_t row_bytes, void *dst,
                                    int64_t *n_kept_host) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && src && mask && dst, "tsc_compact_rows_dev: null argument");
    TSC_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX && row_bytes > 0 && row_bytes % 8 == 0 && row_bytes / 8 < INT32_MAX, "bad sizes");
    if (n_rows == 0) {
        if (n_kept_host) *n_kept_host = 0;
        return 0;
    }
    DeviceGuard guard(c->device);
    Scratch s(c);
    int32_t *bsum, *act, *total;
    TSC_TRY(s.get(scan_bsum_count(n_rows), &bsum));
    TSC_TRY(s.get(size_t(n_rows), &act));
    TSC_TRY(s.get(1, &total));
    TSC_TRY(scan_mask(c->stream, mask, n_rows, bsum, nullptr, act, nullptr, total));
    int32_t kept = 0;
    TSC_TRY(read_i32(c, total, &kept));
    TSC_TRY(launch_gather_rows(c->stream, src, act, kept, int(row_bytes / 8), nullptr, int(row_bytes / 8), dst));
    if (n_kept_host) *n_kept_host = kept;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_gather_heavy_dev(tsc_ctx *c, const double *coords, const uint8_t *mask, int64_t n_poses, int n_atoms,
                                    const int32_t *heavy_idx, int n_heavy, double *heavy_out, int64_t *n_kept_host) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && heavy_idx && heavy_out, "tsc_gather_heavy_dev: null argument");
    TSC_REQUIRE(n_poses >= 0 && n_poses < INT32_MAX && n_atoms > 0 && n_heavy > 0 && n_heavy <= n_atoms, "bad sizes");
    if (n_poses == 0) {
        if (n_kept_host) *n_kept_host = 0;
        return 0;
    }
    DeviceGuard guard(c->device);
    Scratch s(c);
    std::vector<int32_t> sel(size_t(n_heavy) * 3);
    for (int a = 0; a < n_heavy; ++a) {
        TSC_REQUIRE(heavy_idx[a] >= 0 && heavy_idx[a] < n_atoms, "heavy_idx[%d] out of range", a);
        for (int k = 0; k < 3; ++k) sel[size_t(a) * 3 + k] = heavy_idx[a] * 3 + k;
    }
    int32_t *d_sel, *act = nullptr;
    TSC_TRY(upload(c, s, sel.data(), sel.size(), &d_sel));
    int32_t kept = int32_t(n_poses);
    if (mask) {
        int32_t *bsum, *total;
        TSC_TRY(s.get(scan_bsum_count(n_poses), &bsum));
        TSC_TRY(s.get(size_t(n_poses), &act));
        TSC_TRY(s.get(1, &total));
        TSC_TRY(scan_mask(c->stream, mask, n_poses, bsum, nullptr, act, nullptr, total));
        TSC_TRY(read_i32(c, total, &kept));
    } else {
        TSC_HIP(hipStreamSynchronize(c->stream));  // sel upload reads a stack-local vector
    }
    TSC_TRY(launch_gather_rows(c->stream, coords, act, kept, n_atoms * 3, d_sel, n_heavy * 3, heavy_out));
    if (!mask) TSC_HIP(hipStreamSynchronize(c->stream));
    if (n_kept_host) *n_kept_host = kept;
    return 0;
    TSC_API_GUARD_END
}

