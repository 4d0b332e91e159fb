// tscode_hip.hip -- the C ABI of include/tscode_hip.h on top of the kernels in this directory.
// gfx950 only.  There is deliberately no CPU implementation behind these entry points.
#include <hip/hip_ext.h>
#include "common.hpp"
#include "embed_clash.hpp"
#include "rmsd.hpp"
#include "scan.hpp"
#include "sieve.hpp"
#include "local_pass.hpp"
#include "cull.hpp"
#include "group_filter.hpp"
#include "csearch.hpp"
#include "tfd.hpp"
#include "moi.hpp"
#include "host_order.hpp"

#include <algorithm>

using namespace tsc;

// --------------------------------------------------------------------------------------------------
// library / context

extern "C" __attribute__((visibility("default"))) int tsc_version(void) { return TSC_VERSION; }
extern "C" __attribute__((visibility("default"))) const char *tsc_last_error(void) { return g_err; }
#ifndef TSC_CSRC_DIGEST
#define TSC_CSRC_DIGEST "unrecorded"
#endif
extern "C" __attribute__((visibility("default"))) const char *tsc_build_digest(void) { return TSC_CSRC_DIGEST; }

extern "C" __attribute__((visibility("default"))) int tsc_device_count(void) {
    TSC_API_GUARD_BEGIN
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(TSC_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    return n;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_ctx_create(int device, tsc_ctx **out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(out != nullptr, "tsc_ctx_create: out is null");
    *out = nullptr;
    int n = tsc_device_count();
    if (n < 0) return n;
    if (n == 0) return fail(TSC_ERR_NO_DEVICE, "no HIP device visible: libtscode_hip has no CPU path");
    TSC_REQUIRE(device >= 0 && device < n, "tsc_ctx_create: device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    TSC_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(TSC_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    DeviceGuard guard(device);
    tsc_ctx *c = new (std::nothrow) tsc_ctx();
    if (!c) return fail(TSC_ERR_NOMEM, "out of host memory");
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        // the side chain of the pipeline (sample embed, moments, basis) is three tiny kernels running beside the clash kernel, which
        // fills the device: at the highest priority their workgroups are placed as soon as any of the clash kernel's retire
        int lo = 0, hi = 0;
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->basis_stream, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) {
        c->pinned_bytes = 16384;
        e = hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault);
    }
    if (e != hipSuccess) {
        (void)tsc_ctx_destroy(c);       // streams, events and the pinned buffer created so far
        return fail(TSC_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_ctx_destroy(tsc_ctx *c) {
    TSC_API_GUARD_BEGIN
    if (!c) return 0;
    DeviceGuard guard(c->device);
    // everything enqueued on any of the context's streams ends before what it uses is freed (also the teardown of a context
    // whose creation failed half way: whatever exists by then is released here)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->basis_stream) (void)hipStreamSynchronize(c->basis_stream);
    for (auto &kv : c->cache) (void)hipFree(kv.second);
    for (auto &kv : c->live) (void)hipFree(kv.first);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev_sync) (void)hipEventDestroy(c->ev_sync);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->basis_stream) (void)hipStreamDestroy(c->basis_stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
    TSC_API_GUARD_END
}

static hipStream_t g_dummy;
extern "C" __attribute__((visibility("default"))) int tsc_ctx_set_stream(tsc_ctx *c, void *hip_stream) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c != nullptr, "null context");
    DeviceGuard guard(c->device);
    TSC_HIP(hipStreamSynchronize(c->stream));
    if (hip_stream) {
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        c->stream = static_cast<hipStream_t>(hip_stream);
        c->own_stream = false;
    } else if (!c->own_stream) {
        TSC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    (void)g_dummy;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_ctx_synchronize(tsc_ctx *c) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c != nullptr, "null context");
    DeviceGuard guard(c->device);
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_malloc(tsc_ctx *c, size_t bytes, void **dptr) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && dptr, "null argument");
    DeviceGuard guard(c->device);
    TSC_HIP(hipMalloc(dptr, bytes ? bytes : 8));
    return 0;
    TSC_API_GUARD_END
}
extern "C" __attribute__((visibility("default"))) int tsc_free(tsc_ctx *c, void *dptr) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c != nullptr, "null context");
    DeviceGuard guard(c->device);
    TSC_HIP(hipStreamSynchronize(c->stream));
    if (dptr) TSC_HIP(hipFree(dptr));
    return 0;
    TSC_API_GUARD_END
}
extern "C" __attribute__((visibility("default"))) int tsc_memcpy_h2d(tsc_ctx *c, void *dst, const void *src, size_t bytes) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && (bytes == 0 || (dst && src)), "null argument");
    DeviceGuard guard(c->device);
    if (bytes) {
        TSC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        TSC_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
    TSC_API_GUARD_END
}
extern "C" __attribute__((visibility("default"))) int tsc_memcpy_d2h(tsc_ctx *c, void *dst, const void *src, size_t bytes) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && (bytes == 0 || (dst && src)), "null argument");
    DeviceGuard guard(c->device);
    if (bytes) {
        TSC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        TSC_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
    TSC_API_GUARD_END
}
extern "C" __attribute__((visibility("default"))) int tsc_timer_begin(tsc_ctx *c) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c != nullptr, "null context");
    DeviceGuard guard(c->device);
    TSC_HIP(hipEventRecord(c->ev0, c->stream));
    return 0;
    TSC_API_GUARD_END
}
extern "C" __attribute__((visibility("default"))) int tsc_timer_end(tsc_ctx *c, float *ms) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && ms, "null argument");
    DeviceGuard guard(c->device);
    TSC_HIP(hipEventRecord(c->ev1, c->stream));
    TSC_HIP(hipEventSynchronize(c->ev1));
    TSC_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// helpers

static int make_frag_table(const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf, int n_mols, FragTable *ft) {
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
static int upload(tsc_ctx *c, Scratch &s, const T *host, size_t count, T **dev) {
    TSC_TRY(s.get(count ? count : 1, dev));
    if (count) TSC_HIP(hipMemcpyAsync(*dev, host, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}

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

static int64_t frags_total_doubles(const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf, int n_mols) {
    int64_t end = 0;
    for (int m = 0; m < n_mols; ++m) end = std::max<int64_t>(end, frag_off[m] + int64_t(n_conf[m]) * n_atoms[m] * 3);
    return end;
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

// Fetch a device scalar that the work enqueued so far has produced WITHOUT waiting for what is enqueued after this call:
// the copy goes to the auxiliary stream behind an event; read_i32_finish waits for that stream only.
static int read_i32_begin(tsc_ctx *c, const int32_t *dev) {
    TSC_HIP(hipEventRecord(c->ev_sync, c->stream));
    TSC_HIP(hipStreamWaitEvent(c->aux_stream, c->ev_sync, 0));
    TSC_HIP(hipMemcpyAsync(c->pinned, dev, sizeof(int32_t), hipMemcpyDeviceToHost, c->aux_stream));
    return 0;
}
static int read_i32_finish(tsc_ctx *c, int32_t *host_out) {
    TSC_HIP(hipStreamSynchronize(c->aux_stream));
    *host_out = *static_cast<int32_t *>(c->pinned);
    return 0;
}

static int read_i32(tsc_ctx *c, const int32_t *dev, int32_t *host_out) {
    TSC_HIP(hipMemcpyAsync(c->pinned, dev, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    *host_out = *static_cast<int32_t *>(c->pinned);
    return 0;
}

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

// --------------------------------------------------------------------------------------------------
// K3: prune_conformers_rmsd
//
// A run is a sequence of passes over the schedule of rmsd_pruning.py:186-188.  Nothing in a pass waits for the
// host: the gate of :192 is evaluated on the device (k_pass_step) and every pass that COULD run (20 k < N) is
// enqueued with grids sized for N structures; kernels of a pass that is gated off, and blocks beyond the number
// of still-active structures, return at once.  The host reads the per-pass records once, at the end.

static const double KS[TSC_MAX_PASSES] = {5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1};  // :186-188

static_assert(MAX_SLOTS == TSC_MAX_PASSES, "one cache view per schedule slot");
constexpr int TILE_ROWS = 16;
constexpr int MAX_HP = 32;  // register-tiled kernel only; the sieve kernel takes any h

enum { ALGO_AUTO = 0, ALGO_TILE = 1, ALGO_SIEVE = 2, ALGO_LOCAL = 3 /* reported only: a pass run by the chunk-local kernel */ };

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
    double *Gall = nullptr;
    struct Tickets {
        PassTickets pass;
        LocalTickets local;
    } *tickets = nullptr;  // arrival counters of the fused pair kernel and of the chunk-local pass kernel
    bool cur_local = false;           // the open pass ran (whole) in tsc_prune_pass_local
    // culled passes (cull.hpp): allocated when the first one comes up
    int32_t *morton_order = nullptr, *rank_of = nullptr, *crank = nullptr, *cbase = nullptr, *cfill = nullptr, *blk_cnt = nullptr;
    float *Ds = nullptr, *cbox = nullptr, *rbox = nullptr;
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
    int flag_slot = 0;         // this run's word in the context's pinned buffer (the culled-or-walked verdict of a candidate pass)
};

template <typename T>
static int palloc(tsc_prune *p, size_t count, T **out) {
    void *q = nullptr;
    TSC_TRY(p->ctx->alloc(count * sizeof(T), &q));
    p->blocks.push_back(q);
    *out = static_cast<T *>(q);
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_prune_destroy(tsc_prune *p) {
    TSC_API_GUARD_BEGIN
    if (!p) return 0;
    DeviceGuard guard(p->ctx->device);
    if (p->borrows_xd && p->ctx->xd_borrowers > 0) --p->ctx->xd_borrowers;
    for (void *q : p->blocks) p->ctx->release(q);
    for (auto &slot : p->ev)
        for (hipEvent_t e : slot)
            if (e) p->ctx->event_pool.push_back(e);
    delete p;
    return 0;
    TSC_API_GUARD_END
}

// Doubles of a descriptor basis: KD rows per feature family, then the DW projections of the mean feature vector (+ 1 spare)
// ... and, behind the DW projections, the two families' mean squared descriptor distance over the sample (k_descriptor_basis)
static size_t basis_doubles(int h) { return size_t(KD) * (n_features(h, 0) + n_features(h, 1)) + DW + NFAM + 1; }
static size_t basis_spread_offset(int h) { return size_t(KD) * (n_features(h, 0) + n_features(h, 1)) + DW; }
constexpr size_t PINNED_SPREAD_OFFSET = 8192;  // where a basis' two spread values land in the context's pinned buffer
constexpr size_t PINNED_FLAG_OFFSET = PINNED_SPREAD_OFFSET + 128;  // ... and the culled-or-walked verdicts of the context's runs, one 64-byte line each
constexpr int PINNED_FLAG_SLOTS = 64;
constexpr int64_t AUTO_TILE_MIN_N = 30000;     // a prune of its own (no pipeline around it) spends a synchronisation on the question from here on

// "Would the screen let (almost) every pair through?"  Two structures of the sample lie 2 sum_k lambda_k apart, on average, in a
// family's squared descriptor distance; the screen drops a pair only where a family's distance exceeds h thr^2.  Where BOTH families'
// averages stay below that limit most pairs reach H = p^T q whatever the screen does, and the register-tiled all-pairs kernel, which
// forms H for every pair at 2.5e10 pairs/s, beats the sieve's evaluation stage at 5e9 (profiles/r03_hard_workloads.json: 8 ms
// against 46 on 100 000 structures whose descriptors coincide, cache-free mode).  Either kernel gives the same verdicts.
// Asked in the cache-free mode only: in the reference-exact mode the cache's stop columns end nearly every row early on such an ensemble
// (13 M pair evaluations where the cache-free mode makes 191 M), and with so few pairs the sieve's cheaper passes win (2.6 against 3.6 ms).
static bool screen_is_useless(const double *spread, int h, double thr) {
    const double limit = double(h) * thr * thr;
    return spread[0] < limit && spread[1] < limit;  // (false for NaN / +inf: no estimate)
}

// Basis of the descriptors: leading principal axes of the two feature families (sieve.hpp) over `n_samples` structures
// heavy[stride * i], into d_Q (basis_doubles(h)).  Enqueued on `st`; the scratch it takes from `s` must outlive the kernels.
static size_t moment_doubles(int h) {  // (MOM_BLOCKS partial matrices per family: k_feature_moments)
    const size_t a = size_t(n_features(h, 0) + 1), b = size_t(n_features(h, 1) + 1);
    return size_t(MOM_BLOCKS) * (a * a + b * b) + 1;  // (+ the two arrival counters of k_feature_moments, in the last double)
}

// d_moments (optional): moment_doubles(h) doubles already zeroed on `st` by the caller; otherwise taken from `s` and cleared here
static int build_basis(tsc_ctx *c, hipStream_t st, Scratch &s, const double *heavy, int h, int n_samples, int64_t stride, double *d_Q,
                       unsigned *zero_word = nullptr, double *d_moments = nullptr, double *spread_host = nullptr) {
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

// Descriptors built outside the prune (tsc_pipeline_dev: by the kernel that embeds the passing poses)
struct ExternalDescriptors {
    float *D = nullptr;
    double *G = nullptr;
    unsigned *dmax_bits = nullptr;
    float *heavy32 = nullptr;  // (optional) the float32 copy of the heavy atoms, written by the kernel that embedded them
};

// Descriptors of every structure for the sieve.  `basis` (optional): a basis already enqueued elsewhere (any orthonormal
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

static int get_event(tsc_ctx *c, hipEvent_t *e) {
    if (!c->event_pool.empty()) {
        *e = c->event_pool.back();
        c->event_pool.pop_back();
        return 0;
    }
    TSC_HIP(hipEventCreate(e));
    return 0;
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
    p->flag_slot = c->next_flag_slot++ % PINNED_FLAG_SLOTS;
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
            delete p;
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
        // the float32 copy stage 1 reads (sieve.hpp, pair_stage1): from the embedding kernel where there was one, else converted here
        // (it pays where the gathers come from HBM: 41 MB of heavy atoms at C3 sit in the 256 MB infinity cache and the conversions cost the
        // VALU-bound kernel 2 %; at C4's 348 MB a step goes from 12.1 to 10.6 ms.  "stage1_f32": 0 never, 1 from 128 MB on, 2 always)
        if (!rc && (c->stage1_f32 == 2 || (c->stage1_f32 == 1 && double(n) * h * 24.0 >= 128e6))) {
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

static const double *pending_basis(const tsc_ctx *c, int h);  // (with the pipeline's helpers, below)

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

template <int HP>
static void launch_tile(hipStream_t st, dim3 grid, const tsc_prune *p, const TileArgs &a, hipEvent_t e0, hipEvent_t e1) {
    hipExtLaunchKernelGGL((k_rmsd_tile<HP, TILE_ROWS>), grid, dim3(256), 0, st, e0, e1, 0, (const double *)p->Xr, (const double *)p->Xc,
                          (const double *)p->G, (const int32_t *)p->cend, p->best, p->counters, (const PruneState *)p->state, a);
}

// The pass that follows the open one (what tsc_prune_next_pass will hand out next; not consumed here): the kernel that
// finishes a pass also closes it and opens this one on the device.
static StepArgs next_step_args(const tsc_prune *p, int *next_slot) {
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
static StepCtx step_ctx(const tsc_prune *p, bool range_close = false) {
    return StepCtx{p->state, p->counters, p->records, p->bsum, p->boff, p->n_blocks, reinterpret_cast<unsigned *>(p->tickets),
                   int(sizeof(*p->tickets) / 128), range_close ? p->exch + p->bit_words : nullptr};
}

// Chunks [c_lo, c_hi) of a pass of k chunks that START inside rank's block [n rank / world, n (rank + 1) / world) of the
// structure axis, and the structures [s_lo, s_hi) they cover (the last chunk of the pass runs to n, rmsd_pruning.py:141-144).
static void partition_bounds(int64_t n, int64_t k, int rank, int world, int64_t *c_lo, int64_t *c_hi, int64_t *s_lo, int64_t *s_hi) {
    const int64_t cs = n / k;
    auto first_chunk = [&](int r) { return r <= 0 ? int64_t(0) : (r >= world ? k : std::min<int64_t>(ceil_div<int64_t>(n * r / world, cs), k)); };
    *c_lo = first_chunk(rank), *c_hi = first_chunk(rank + 1);
    *s_lo = *c_lo < k ? *c_lo * cs : n, *s_hi = *c_hi < k ? *c_hi * cs : n;
}
static bool pass_is_partitioned(const tsc_prune *p, int64_t k) {
    return p->part_world > 1 && p->exch && p->algo == ALGO_SIEVE && k >= int64_t(p->part_min_chunks) * p->part_world;
}

// Views of the passes AFTER the open one (where the rows it removes leave their cache keys); none in cache-free mode.
static CacheViews later_views(const tsc_prune *p) {
    CacheViews cv;
    cv.views = p->views, cv.stride = (long long)(p->bit_words + p->dsum_words), cv.bit_words = int(p->bit_words), cv.n = int(p->n);
    cv.first = p->mode == 0 ? p->view_of_slot[p->cur_slot] + 1 : 0;
    cv.count = p->mode == 0 ? p->n_views : 0;
    cv.pass = p->view_pass;
    return cv;
}

static const unsigned long long *view_of_open_pass(const tsc_prune *p) {
    return p->mode == 0 ? p->views + size_t(p->view_of_slot[p->cur_slot]) * (p->bit_words + p->dsum_words) : p->views;
}

static ApplyArgs apply_args(const tsc_prune *p) {
    ApplyArgs a;
    a.g = PassGeom{int(p->n), int(p->cur_k), int(p->n / p->cur_k)};
    a.act = p->act, a.cend = p->cend, a.best = p->best, a.mask = p->mask, a.bits = p->bits, a.bit_words = int(p->bit_words);
    a.bsum = p->bsum, a.block_items = SCAN_TILE, a.cv = later_views(p);
    a.exch = p->cur_range ? p->exch : nullptr;
    return a;
}

// The pair search of one rank's row tiles of the open pass (step 3 of a pass; steps 1-2 have run).
// rows_ub: upper bound of the rows of the pass on this device (n; in a rank-partitioned pass the structures of this rank's chunks)
static int launch_pair_search(tsc_prune *p, int rank, int world, int64_t rows_ub) {
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
    const int n_seg = ceil_div(max_range + 64, seg_cols);  // + 64: a segment starts at the 64-aligned column below r0 + 1
    const int my_tiles = (n_tiles - rank + world - 1) / world;
    dim3 grid(std::max(1, ceil_div(my_tiles, 4)), n_seg);
    // the pair kernel's own start / stop events ride on its dispatch packet (no extra packets in the stream; a
    // hipEventRecord before and after it costs about 4 us each on MI355X)
    hipEvent_t e0 = c->pass_timing >= 1 ? p->ev[slot][1] : nullptr, e1 = c->pass_timing >= 1 ? p->ev[slot][2] : nullptr;
    if (p->algo == ALGO_TILE) {
        TileArgs a;
        a.ld = p->npad, a.h = p->h;
        a.tile_begin = rank, a.tile_stride = world, a.seg_cols = seg_cols;
        a.thr = p->thr, a.maxdev_thr = 2 * p->thr;  // :95
        a.half_h_thr2 = 0.5 * double(p->h) * p->thr * p->thr;
        switch (p->hp) {
            case 4: launch_tile<4>(st, grid, p, a, e0, e1); break;
            case 8: launch_tile<8>(st, grid, p, a, e0, e1); break;
            case 12: launch_tile<12>(st, grid, p, a, e0, e1); break;
            case 16: launch_tile<16>(st, grid, p, a, e0, e1); break;
            case 20: launch_tile<20>(st, grid, p, a, e0, e1); break;
            case 24: launch_tile<24>(st, grid, p, a, e0, e1); break;
            case 28: launch_tile<28>(st, grid, p, a, e0, e1); break;
            case 32: launch_tile<32>(st, grid, p, a, e0, e1); break;
            default: return fail(TSC_ERR_INVALID, "unsupported padded atom count %d", p->hp);
        }
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
#define TSC_LAUNCH_SIEVE_F(CPL, TRIM, FUSED, F32)                                                                                                 \
    hipExtLaunchKernelGGL((k_rmsd_sieve<TILE_ROWS, CPL, TRIM, FUSED, F32>), grid, dim3(256), 0, st, e0, e1, 0, p->heavy, (const int32_t *)p->act, \
                          (const double *)p->Gall, (const float *)p->Dc, (const int32_t *)p->cend, p->best, p->counters,                          \
                          (const PruneState *)p->state, a, fa)
#define TSC_LAUNCH_SIEVE(CPL, TRIM, FUSED) TSC_LAUNCH_SIEVE_F(CPL, TRIM, FUSED, false)
        const bool trim = c->sieve_cpl == 2 && c->sieve_trim;
        // (stage 1 on the float32 copy exists for the default shape of the kernel only)
        if (trim && a.heavy32) {
            if (p->cur_fused) TSC_LAUNCH_SIEVE_F(2, true, true, true);
            else TSC_LAUNCH_SIEVE_F(2, true, false, true);
        } else if (p->cur_fused) {
            if (c->sieve_cpl == 1) TSC_LAUNCH_SIEVE(1, false, true);
            else if (trim) TSC_LAUNCH_SIEVE(2, true, true);
            else if (c->sieve_cpl == 2) TSC_LAUNCH_SIEVE(2, false, true);
            else TSC_LAUNCH_SIEVE(4, false, true);
        } else {
            if (c->sieve_cpl == 1) TSC_LAUNCH_SIEVE(1, false, false);
            else if (trim) TSC_LAUNCH_SIEVE(2, true, false);
            else if (c->sieve_cpl == 2) TSC_LAUNCH_SIEVE(2, false, false);
            else TSC_LAUNCH_SIEVE(4, false, false);
        }
#undef TSC_LAUNCH_SIEVE
#undef TSC_LAUNCH_SIEVE_F
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
        hipExtLaunchKernelGGL(k_pass_chunks, dim3(unsigned(blocks)), dim3(LP_THREADS), 0, st, e0, e1, 0, g, a, p->state, p->mask, p->bits, int(p->bit_words),
                              view_of_open_pass(p), p->heavy, (const double *)p->Gall, (const float *)p->Dall, later_views(p), p->counters, p->bsum,
                              SCAN_TILE, step_ctx(p, range), sa, &p->tickets->local);
        TSC_HIP(hipGetLastError());
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
        if (!rc) rc = palloc(p, (size_t(n) / CULL_COLS + 2) * CULL_BOX, &p->cbox);
        if (!rc) rc = palloc(p, (size_t(n) / CULL_COLS + 2) * 8 * CULL_BOX, &p->rbox);
        if (rc) return rc;
    }
    // 1. per row: which structure it is, its stop column, best[] = none, its descriptor by position (k_open_rows, rmsd.hpp)
    p->cur_fused = p->algo == ALGO_SIEVE && world == 1 && (c->fused_apply != 0 || range);
    {
        OpenArgs oa;
        oa.use_cache = use_cache, oa.fused = p->cur_fused ? 1 : 0, oa.lds_cap = std::min(c->open_lds_blocks, OPEN_LDS_BLOCKS);
        oa.view = view_of_open_pass(p), oa.bits = p->bits, oa.bit_words = int(p->bit_words);
        oa.boff = p->boff, oa.n_blocks = p->n_blocks, oa.block_items = SCAN_TILE;
        oa.n_tiles = unsigned(ceil_div(A, 16)), oa.tickets = &p->tickets->pass;
        oa.rank_of = culled ? p->rank_of : nullptr;
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
        hipLaunchKernelGGL(k_open_rows, dim3(ceil_div(ceil_div(A, 16), 4)), dim3(256), 0, st, g, oa, step_ctx(p, range), sa, p->act, p->cend, p->best, p->tile_cmax,
                           (const float *)p->Dall, p->Dc);
    }
    if (p->algo == ALGO_TILE) {
        const int hp3 = p->hp * 3;
        size_t lds = size_t(64) * (hp3 + 1) * sizeof(double);
        hipLaunchKernelGGL(k_compact_coords, dim3(ceil_div(A, 64)), dim3(256), lds, st, p->heavy, p->h, hp3, p->act, (const PruneState *)p->state,
                           p->Xr, p->Xc, p->npad, p->G);
    }
    bool run_culled = false;
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
    if (run_culled) {
        p->cur_fused = false;  // rows collect verdicts as columns of other tiles too: the pass is applied behind the pair kernel (k_apply_pass)
        const int n_lb = int(ceil_div<int64_t>(n, CULL_LAYOUT_ITEMS));
        const LayoutRange lr{int(s_lo), int(s_hi)};
        hipLaunchKernelGGL(k_layout_count, dim3(unsigned(n_lb)), dim3(256), 0, st, g, lr, (const PruneState *)p->state, (const int32_t *)p->morton_order,
                           (const unsigned long long *)p->bits, int(p->bit_words), p->blk_cnt);
        hipLaunchKernelGGL(k_layout_scan, dim3(unsigned(k)), dim3(64), 0, st, (const PruneState *)p->state, n_lb, (const int32_t *)p->cbase, p->blk_cnt);
        hipLaunchKernelGGL(k_layout_scatter, dim3(unsigned(n_lb)), dim3(256), 0, st, g, lr, (const PruneState *)p->state, (const int32_t *)p->morton_order,
                           (const unsigned long long *)p->bits, int(p->bit_words), (const int32_t *)p->rank_of, (const float *)p->Dc,
                           (const int32_t *)p->blk_cnt, p->Ds, p->crank);
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
        CullArgs ca{p->Ds, p->crank, p->cbase, p->cbox, p->rbox, int(k), tb};
        const int n_tiles = ceil_div(A, TILE_ROWS);
        // (slots of this rank: one by one, or whole runs of tb tiles -- an upper bound; slots beyond the last tile leave at once)
        const int my_tiles = tb <= 1 ? (n_tiles - rank + world - 1) / world : (n_tiles / (tb * world) + 1) * tb;
        // columns of a row tile: from its own 128-aligned position to the end of its (last row's) chunk -- a chunk and a tile more at most
        const int n_seg = ceil_div(int(std::min<int64_t>(A, longest_chunk)) + 2 * CULL_COLS, a.seg_cols);
        hipEvent_t e0 = c->pass_timing >= 1 ? p->ev[slot][1] : nullptr, e1 = c->pass_timing >= 1 ? p->ev[slot][2] : nullptr;
        const int64_t items = int64_t(ceil_div(my_tiles, 4)) * n_seg;
        const dim3 sgrid(unsigned(std::max<int64_t>(1, std::min<int64_t>(items, c->cull_grid))));
        if (a.heavy32)
            hipExtLaunchKernelGGL(k_rmsd_sieve_sorted<true>, sgrid, dim3(256), 0, st, e0, e1, 0, p->heavy, (const int32_t *)p->act, (const double *)p->Gall,
                                  (const int32_t *)p->cend, p->best, p->counters, (const PruneState *)p->state, a, ca, my_tiles, n_seg);
        else
            hipExtLaunchKernelGGL(k_rmsd_sieve_sorted<false>, sgrid, dim3(256), 0, st, e0, e1, 0, p->heavy, (const int32_t *)p->act, (const double *)p->Gall,
                                  (const int32_t *)p->cend, p->best, p->counters, (const PruneState *)p->state, a, ca, my_tiles, n_seg);
        TSC_HIP(hipGetLastError());
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
    TSC_TRY(launch_pair_search(p, rank, world, A));
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

#ifdef TSC_DBG_STAMPS
// measurement builds only: the time stamps of the last stamped pair-kernel launch, 8 per wavefront (tools/stamps.py)
extern "C" __attribute__((visibility("default"))) int tsc_debug_stamps(tsc_ctx *c, unsigned long long *dst, int64_t max_waves, int64_t *n_waves) {
    TSC_API_GUARD_BEGIN
    DeviceGuard guard(c->device);
    TSC_HIP(hipStreamSynchronize(c->stream));
    const int64_t n = std::min(max_waves, c->dbg_waves);
    if (n > 0) TSC_HIP(hipMemcpy(dst, c->dbg_buf, size_t(n) * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *n_waves = n;
    return 0;
    TSC_API_GUARD_END
}
#endif

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
static int prune_run(tsc_ctx *c, const double *heavy, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask, uint8_t *mask_host,
                     tsc_pass_stats *stats, int *n_passes, const double *basis = nullptr, const ExternalDescriptors *ext = nullptr, int force_algo = -1) {
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

// Tunables: "prune_algo" 0 / 2 = descriptor sieve (any size), 1 = register-tiled all-pairs kernel (h <= 32);
// "seg_cols" = columns per work item.
extern "C" __attribute__((visibility("default"))) int tsc_ctx_set_option(tsc_ctx *c, const char *name, double value) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && name, "null argument");
    if (strcmp(name, "prune_algo") == 0) {
        TSC_REQUIRE(value == 0 || value == 1 || value == 2, "prune_algo must be 0, 1 or 2");
        c->prune_algo = int(value);
        return 0;
    }
    if (strcmp(name, "drain_min") == 0) {
        TSC_REQUIRE(value >= 1 && value <= 64, "drain_min must be in [1, 64]");
        c->drain_min = int(value);
        return 0;
    }
    if (strcmp(name, "pass_timing") == 0) {
        TSC_REQUIRE(value == 0 || value == 1 || value == 2, "pass_timing must be 0, 1 or 2");
        c->pass_timing = int(value);
        return 0;
    }
    if (strcmp(name, "sieve_trim") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "sieve_trim must be 0 or 1");
        c->sieve_trim = int(value);
        return 0;
    }
    if (strcmp(name, "sieve_cpl") == 0) {
        TSC_REQUIRE(value == 1 || value == 2 || value == 4, "sieve_cpl must be 1, 2 or 4");
        c->sieve_cpl = int(value);
        return 0;
    }
    if (strcmp(name, "fuse_descriptors") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "fuse_descriptors must be 0 or 1");
        c->fuse_descriptors = int(value);
        return 0;
    }
    if (strcmp(name, "pca_min_n") == 0) {
        TSC_REQUIRE(value >= 0 && value <= 1e9, "pca_min_n must be in [0, 1e9]");
        c->pca_min_n = int64_t(value);
        return 0;
    }
    if (strcmp(name, "early_basis") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "early_basis must be 0 or 1");
        c->early_basis = int(value);
        return 0;
    }
    if (strcmp(name, "local_max_chunk") == 0) {
        TSC_REQUIRE(value >= 16 && value <= LP_MAX_ROWS, "local_max_chunk must be in [16, %d]", LP_MAX_ROWS);
        c->local_max_chunk = int(value);
        return 0;
    }
#ifdef TSC_DBG_STAMPS
    if (strcmp(name, "dbg_stamp_k") == 0) {
        c->dbg_stamp_k = int64_t(value);
        return 0;
    }
#endif
    if (strcmp(name, "cull") == 0) {
        TSC_REQUIRE(value == 0.0 || value == 1.0 || value == 2.0, "cull must be 0 (off), 1 (the device decides per pass) or 2 (every candidate pass is culled)");
        c->cull = int(value);
        return 0;
    }
    if (strcmp(name, "deterministic_basis") == 0) {
        c->deterministic_basis = value != 0.0 ? 1 : 0;
        return 0;
    }
    if (strcmp(name, "cull_grid") == 0) {
        TSC_REQUIRE(value >= 1.0, "cull_grid must be positive");
        c->cull_grid = int64_t(value);
        return 0;
    }
    if (strcmp(name, "cull_tile_block") == 0) {
        TSC_REQUIRE(value >= 1 && value <= 65536, "cull_tile_block must be in [1, 65536]");
        c->cull_tile_block = int(value);
        return 0;
    }
    if (strcmp(name, "stage1_f32") == 0) {
        c->stage1_f32 = int(value);
        return 0;
    }
    if (strcmp(name, "cull_min_pairs") == 0) {
        TSC_REQUIRE(value >= 0.0, "cull_min_pairs must not be negative");
        c->cull_min_pairs = value;
        return 0;
    }
    if (strcmp(name, "fused_apply") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "fused_apply must be 0 or 1");
        c->fused_apply = int(value);
        return 0;
    }
    if (strcmp(name, "open_lds_blocks") == 0) {
        TSC_REQUIRE(value >= 0, "open_lds_blocks must not be negative");
        c->open_lds_blocks = int(std::min(value, 1073741824.0));
        return 0;
    }
    if (strcmp(name, "local_pass") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "local_pass must be 0 or 1");
        c->local_pass = int(value);
        return 0;
    }
    if (strcmp(name, "clash_first") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "clash_first must be 0 or 1");
        c->clash_first = int(value);
        return 0;
    }
    if (strcmp(name, "clash_lanes") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "clash_lanes must be 0 or 1");
        c->clash_lanes = int(value);
        return 0;
    }
    if (strcmp(name, "clash_fp32") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "clash_fp32 must be 0 or 1");
        c->clash_fp32 = int(value);
        return 0;
    }
    if (strcmp(name, "seg_cols") == 0) {
        TSC_REQUIRE(value == 0 || (value >= 256 && value <= 4096 && int(value) % 256 == 0), "seg_cols must be 0 (automatic) or a multiple of 256 in [256, 4096]");
        c->seg_cols = int(value);
        return 0;
    }
    return fail(TSC_ERR_INVALID, "unknown option '%s'", name);
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// greedy per-group filter

extern "C" __attribute__((visibility("default"))) int tsc_greedy_group_filter_dev(tsc_ctx *c, const double *poses, const int32_t *group_off_dev, int n_groups,
                                                                             int64_t n_poses, int n_atoms, double rmsd_thr, uint8_t *accepted) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && poses && group_off_dev && accepted, "tsc_greedy_group_filter_dev: null argument");
    TSC_REQUIRE(n_groups >= 0 && n_poses >= 0 && n_atoms > 0 && rmsd_thr > 0, "bad sizes");
    if (n_groups == 0 || n_poses == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *G;
    TSC_TRY(s.get(size_t(n_poses), &G));
    hipLaunchKernelGGL(k_greedy_group_filter, dim3(n_groups), dim3(256), 0, c->stream, poses, group_off_dev, n_groups, n_atoms, rmsd_thr,
                       accepted, G);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_greedy_group_filter(tsc_ctx *c, const double *poses, const int32_t *group_off, int n_groups, int n_atoms,
                                                                         double rmsd_thr, uint8_t *accepted) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && poses && group_off && accepted, "tsc_greedy_group_filter: null argument");
    TSC_REQUIRE(n_groups >= 0 && n_atoms > 0 && rmsd_thr > 0, "bad sizes");
    if (n_groups == 0) return 0;
    TSC_REQUIRE(group_off[0] == 0, "group_off must start at 0");
    for (int g = 0; g < n_groups; ++g)
        TSC_REQUIRE(group_off[g + 1] >= group_off[g] && group_off[g + 1] - group_off[g] <= GF_MAX_GROUP,
                    "group %d: sizes must be in [0, %d]", g, GF_MAX_GROUP);
    const int64_t n_poses = group_off[n_groups];
    if (n_poses == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_poses;
    int32_t *d_off;
    uint8_t *d_acc;
    TSC_TRY(upload(c, s, poses, size_t(n_poses) * n_atoms * 3, &d_poses));
    TSC_TRY(upload(c, s, group_off, size_t(n_groups) + 1, &d_off));
    TSC_TRY(s.get(size_t(n_poses), &d_acc));
    TSC_TRY(tsc_greedy_group_filter_dev(c, d_poses, d_off, n_groups, n_poses, n_atoms, rmsd_thr, d_acc));
    TSC_HIP(hipMemcpyAsync(accepted, d_acc, size_t(n_poses), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// conformational-search rotations

static int csearch_args(int n_atoms, int n_tors, int64_t n_cand, double thresh, int64_t max_clashes, CsearchArgs *a) {
    TSC_REQUIRE(n_atoms > 0 && n_tors >= 0 && n_cand >= 0, "bad sizes (%d atoms, %d torsions, %lld candidates)", n_atoms, n_tors, (long long)n_cand);
    TSC_REQUIRE(n_atoms <= 65535 && torsion_lists_bytes(n_tors, n_atoms) + csearch_wave_bytes(n_atoms) <= 150 * 1024,
                "%d atoms x %d torsions exceed the LDS staging of the csearch kernels", n_atoms, n_tors);
    a->n = n_atoms, a->n_tors = n_tors, a->n_cand = n_cand;
    a->sq_bound = clash_sq_bound(thresh), a->max_clashes = max_clashes;
    return 0;
}

// wavefronts per workgroup (4, 2 or 1) that fit the LDS, and the dynamic LDS size of the launch
static int csearch_waves(int n_atoms, int n_tors, size_t *lds) {
    int w = 4;
    while (w > 1 && torsion_lists_bytes(n_tors, n_atoms) + w * csearch_wave_bytes(n_atoms) > 150 * 1024) w >>= 1;
    *lds = torsion_lists_bytes(n_tors, n_atoms) + w * csearch_wave_bytes(n_atoms);
    return w;
}

static int check_torsions(const int32_t *torsions, int n_tors, int n_atoms) {
    for (int t = 0; t < n_tors; ++t)
        for (int q = 0; q < 4; ++q)
            TSC_REQUIRE(torsions[4 * t + q] >= 0 && torsions[4 * t + q] < n_atoms, "torsion %d: atom index %d out of range", t, torsions[4 * t + q]);
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_csearch_rotate_dev(tsc_ctx *c, const double *coords, int n_atoms, const int32_t *torsions,
                                                                             const uint8_t *masks, int n_tors, const int32_t *angles, int64_t n_cand,
                                                                             double thresh, int64_t max_clashes, double *out, int32_t *rotated_bonds) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && out && rotated_bonds && torsions && masks && angles, "tsc_csearch_rotate_dev: null argument");
    CsearchArgs a;
    TSC_TRY(csearch_args(n_atoms, n_tors, n_cand, thresh, max_clashes, &a));
    if (n_cand == 0) return 0;
    DeviceGuard guard(c->device);
    size_t lds;
    const int waves = csearch_waves(n_atoms, n_tors, &lds);
    if (lds > 64 * 1024)
        TSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_csearch_rotate), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k_csearch_rotate, dim3(grid_for(n_cand, waves, 256 * 8)), dim3(64 * waves), lds, c->stream, a, coords, torsions, masks, angles, out,
                       rotated_bonds);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_csearch_rotate(tsc_ctx *c, const double *coords, int n_atoms, const int32_t *torsions,
                                                                         const uint8_t *masks, int n_tors, const int32_t *angles, int64_t n_cand,
                                                                         double thresh, int64_t max_clashes, double *out, int32_t *rotated_bonds) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && out && rotated_bonds && (n_tors == 0 || (torsions && masks && angles)), "tsc_csearch_rotate: null argument");
    TSC_REQUIRE(n_atoms > 0 && n_tors >= 0 && n_cand >= 0, "bad sizes");
    TSC_TRY(check_torsions(torsions, n_tors, n_atoms));
    if (n_cand == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_coords, *d_out;
    int32_t *d_tors, *d_angles, *d_rb;
    uint8_t *d_masks;
    TSC_TRY(upload(c, s, coords, size_t(n_atoms) * 3, &d_coords));
    TSC_TRY(upload(c, s, torsions, size_t(n_tors) * 4, &d_tors));
    TSC_TRY(upload(c, s, masks, size_t(n_tors) * n_atoms, &d_masks));
    TSC_TRY(upload(c, s, angles, size_t(n_cand) * n_tors, &d_angles));
    TSC_TRY(s.get(size_t(n_cand) * n_atoms * 3, &d_out));
    TSC_TRY(s.get(size_t(n_cand), &d_rb));
    TSC_TRY(tsc_csearch_rotate_dev(c, d_coords, n_atoms, d_tors, d_masks, n_tors, d_angles, n_cand, thresh, max_clashes, d_out, d_rb));
    TSC_HIP(hipMemcpyAsync(out, d_out, size_t(n_cand) * n_atoms * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(rotated_bonds, d_rb, size_t(n_cand) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_rotate_dihedral(tsc_ctx *c, const double *coords, int64_t n_structs, int n_atoms, const int32_t *torsion,
                                                                          const uint8_t *mask, const double *angles, double *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && torsion && mask && angles && out, "tsc_rotate_dihedral: null argument");
    TSC_REQUIRE(n_structs >= 0 && n_atoms > 0, "bad sizes");
    TSC_REQUIRE(out != coords, "tsc_rotate_dihedral: out must not alias coords");
    for (int q = 1; q <= 2; ++q) TSC_REQUIRE(torsion[q] >= 0 && torsion[q] < n_atoms, "torsion index %d out of range", torsion[q]);
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_c, *d_o, *d_a;
    uint8_t *d_m;
    TSC_TRY(upload(c, s, coords, size_t(n_structs) * n_atoms * 3, &d_c));
    TSC_TRY(upload(c, s, angles, size_t(n_structs), &d_a));
    TSC_TRY(upload(c, s, mask, size_t(n_atoms), &d_m));
    TSC_TRY(s.get(size_t(n_structs) * n_atoms * 3, &d_o));
    hipLaunchKernelGGL(k_rotate_dihedral, dim3(grid_for(n_structs * n_atoms, 256)), dim3(256), 0, c->stream, (const double *)d_c, n_structs, n_atoms, int(torsion[1]),
                       int(torsion[2]), (const uint8_t *)d_m, (const double *)d_a, d_o);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(out, d_o, size_t(n_structs) * n_atoms * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_torsion_comp_check(tsc_ctx *c, const double *coords, int64_t n_structs, int n_atoms,
                                                                             const int32_t *torsion, const uint8_t *mask, double thresh,
                                                                             int64_t max_clashes, int32_t *ok) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && torsion && mask && ok, "tsc_torsion_comp_check: null argument");
    CsearchArgs a;
    TSC_TRY(csearch_args(n_atoms, 1, n_structs, thresh, max_clashes, &a));
    TSC_TRY(check_torsions(torsion, 1, n_atoms));
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_coords;
    int32_t *d_tors, *d_ok;
    uint8_t *d_mask;
    TSC_TRY(upload(c, s, coords, size_t(n_structs) * n_atoms * 3, &d_coords));
    TSC_TRY(upload(c, s, torsion, size_t(4), &d_tors));
    TSC_TRY(upload(c, s, mask, size_t(n_atoms), &d_mask));
    TSC_TRY(s.get(size_t(n_structs), &d_ok));
    size_t lds;
    const int waves = csearch_waves(n_atoms, 1, &lds);
    if (lds > 64 * 1024)
        TSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_torsion_comp_check), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k_torsion_comp_check, dim3(grid_for(n_structs, waves, 256 * 8)), dim3(64 * waves), lds, c->stream, a, (const double *)d_coords,
                       (const int32_t *)d_tors, (const uint8_t *)d_mask, d_ok);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(ok, d_ok, size_t(n_structs) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// torsion-fingerprint pruning

extern "C" __attribute__((visibility("default"))) int tsc_torsion_fingerprints(tsc_ctx *c, const double *coords, int64_t n_structs, int n_atoms,
                                                                               const int32_t *quads, int n_quads, float *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && quads && out, "tsc_torsion_fingerprints: null argument");
    TSC_REQUIRE(n_structs >= 0 && n_atoms > 0 && n_quads >= 0, "bad sizes");
    for (int q = 0; q < 4 * n_quads; ++q) TSC_REQUIRE(quads[q] >= 0 && quads[q] < n_atoms, "quadruplet atom index %d out of range", quads[q]);
    if (n_structs == 0 || n_quads == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_coords;
    int32_t *d_quads;
    float *d_out;
    TSC_TRY(upload(c, s, coords, size_t(n_structs) * n_atoms * 3, &d_coords));
    TSC_TRY(upload(c, s, quads, size_t(n_quads) * 4, &d_quads));
    TSC_TRY(s.get(size_t(n_structs) * n_quads, &d_out));
    hipLaunchKernelGGL(k_torsion_fingerprints, dim3(grid_for(n_structs * n_quads, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)d_coords,
                       n_structs, n_atoms, (const int32_t *)d_quads, n_quads, d_out);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(out, d_out, size_t(n_structs) * n_quads * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_tfd_first_similar(tsc_ctx *c, const float *tf, int64_t n_structs, int n_quads, int64_t d,
                                                                            int64_t k, int64_t num_active, double thresh, int32_t *first) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && tf && first, "tsc_tfd_first_similar: null argument");
    if (n_structs == 0) return 0;
    TSC_REQUIRE(n_structs >= 0 && n_quads >= 0 && d > 0 && k > 0 && num_active >= 0 && num_active <= n_structs && d * k <= n_structs,
                "bad pass geometry (n = %lld, d = %lld, k = %lld, active = %lld)", (long long)n_structs, (long long)d, (long long)k, (long long)num_active);
    TSC_REQUIRE(n_structs < INT32_MAX, "too many structures");
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    float *d_tf;
    int32_t *d_first;
    TSC_TRY(upload(c, s, tf, size_t(n_structs) * n_quads, &d_tf));
    TSC_TRY(s.get(size_t(n_structs), &d_first));
    hipLaunchKernelGGL(k_tfd_first_similar, dim3(grid_for(n_structs, 4, 256 * 16)), dim3(256), 0, c->stream, (const float *)d_tf, n_structs, n_quads, d, k,
                       num_active, thresh, d_first);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(first, d_first, size_t(n_structs) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// the graph step of the similarity prunings on the host (host_order.hpp): no device involved
extern "C" __attribute__((visibility("default"))) int tsc_host_graph_step(const int64_t *rel_i, const int64_t *rel_j, const int64_t *chunk_ptr,
                                                                          const int64_t *chunk_off, const int64_t *chunk_len, int64_t n_chunks,
                                                                          int64_t n_total, uint8_t *keep) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(rel_i && rel_j && chunk_ptr && chunk_off && chunk_len && keep && n_chunks >= 0 && n_total >= 0, "tsc_host_graph_step: bad argument");
    int64_t longest = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        TSC_REQUIRE(chunk_ptr[c + 1] >= chunk_ptr[c] && chunk_off[c] >= 0 && chunk_len[c] >= 0 && chunk_off[c] + chunk_len[c] <= n_total,
                    "chunk %lld: bad bounds", (long long)c);
        for (int64_t q = chunk_ptr[c]; q < chunk_ptr[c + 1]; ++q)
            TSC_REQUIRE(rel_i[q] >= 0 && rel_i[q] < chunk_len[c] && rel_j[q] >= 0 && rel_j[q] < chunk_len[c] && rel_i[q] != rel_j[q],
                        "match %lld lies outside its chunk", (long long)q);
        longest = std::max(longest, chunk_len[c]);
    }
    std::vector<int32_t> index_of(size_t(longest), -1);
    for (int64_t c = 0; c < n_chunks; ++c)
        tsc_host::graph_step_chunk(rel_i + chunk_ptr[c], rel_j + chunk_ptr[c], chunk_ptr[c + 1] - chunk_ptr[c], chunk_off[c], keep, index_of);
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// moments of inertia, embed scores

extern "C" __attribute__((visibility("default"))) int tsc_inertia_moments(tsc_ctx *c, const double *structures, int64_t n_structs, int n_atoms,
                                                                          const double *masses, double *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && structures && masses && out && n_structs >= 0 && n_atoms > 0, "tsc_inertia_moments: bad argument");
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_s, *d_m, *d_o;
    TSC_TRY(upload(c, s, structures, size_t(n_structs) * n_atoms * 3, &d_s));
    TSC_TRY(upload(c, s, masses, size_t(n_atoms), &d_m));
    TSC_TRY(s.get(size_t(n_structs) * 3, &d_o));
    hipLaunchKernelGGL(k_inertia_moments, dim3(grid_for(n_structs, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)d_s, n_structs, n_atoms,
                       (const double *)d_m, d_o);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(out, d_o, size_t(n_structs) * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_moi_first_similar(tsc_ctx *c, const double *moments, int64_t n_structs, double max_deviation,
                                                                            int32_t *first) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && moments && first && n_structs >= 0 && n_structs < INT32_MAX, "tsc_moi_first_similar: bad argument");
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_m;
    int32_t *d_f;
    TSC_TRY(upload(c, s, moments, size_t(n_structs) * 3, &d_m));
    TSC_TRY(s.get(size_t(n_structs), &d_f));
    hipLaunchKernelGGL(k_moi_first_similar, dim3(grid_for(n_structs, 4, 256 * 16)), dim3(256), 0, c->stream, (const double *)d_m, n_structs, max_deviation, d_f);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(first, d_f, size_t(n_structs) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_embed_scores(tsc_ctx *c, const double *structures, int64_t n_structs, int n_atoms,
                                                                       const int32_t *indices, const double *distances, int n_c, float *scores,
                                                                       double *fitness_error) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && structures && indices && distances && scores && fitness_error && n_structs >= 0 && n_atoms > 0 && n_c >= 0, "tsc_embed_scores: bad argument");
    for (int64_t q = 0; q < n_structs * n_c * 2; ++q) TSC_REQUIRE(indices[q] >= 0 && indices[q] < n_atoms, "constrained index %d out of range", indices[q]);
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_s, *d_d, *d_e;
    int32_t *d_i;
    float *d_sc;
    TSC_TRY(upload(c, s, structures, size_t(n_structs) * n_atoms * 3, &d_s));
    TSC_TRY(upload(c, s, indices, size_t(n_structs) * n_c * 2, &d_i));
    TSC_TRY(upload(c, s, distances, size_t(n_structs) * n_c, &d_d));
    TSC_TRY(s.get(size_t(n_structs), &d_sc));
    TSC_TRY(s.get(size_t(n_structs), &d_e));
    hipLaunchKernelGGL(k_embed_scores, dim3(grid_for(n_structs, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)d_s, n_structs, n_atoms,
                       (const int32_t *)d_i, (const double *)d_d, n_c, d_sc, d_e);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(scores, d_sc, size_t(n_structs) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(fitness_error, d_e, size_t(n_structs) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// string-embed pose parameters

extern "C" __attribute__((visibility("default"))) int tsc_string_embed_params_dev(tsc_ctx *c, const double *p1, const double *p2, const double *ref_vec,
                                                                                  const double *mol_vec, const int32_t *conf_pair, int64_t n_sites,
                                                                                  const double *angles, int n_angles, double *rot, double *pos,
                                                                                  int32_t *conf_idx) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && p1 && p2 && ref_vec && mol_vec && conf_pair && angles && rot && pos && conf_idx, "tsc_string_embed_params_dev: null argument");
    TSC_REQUIRE(n_sites >= 0 && n_angles >= 0, "bad sizes");
    if (n_sites == 0 || n_angles == 0) return 0;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_string_embed_params, dim3(grid_for(n_sites * n_angles, 256, 256 * 8)), dim3(256), 0, c->stream, p1, p2, ref_vec, mol_vec,
                       conf_pair, n_sites, angles, n_angles, rot, pos, conf_idx);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_string_embed_params(tsc_ctx *c, const double *p1, const double *p2, const double *ref_vec,
                                                                              const double *mol_vec, const int32_t *conf_pair, int64_t n_sites,
                                                                              const double *angles, int n_angles, double *rot, double *pos,
                                                                              int32_t *conf_idx) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && p1 && p2 && ref_vec && mol_vec && conf_pair && angles && rot && pos && conf_idx, "tsc_string_embed_params: null argument");
    TSC_REQUIRE(n_sites >= 0 && n_angles >= 0, "bad sizes");
    if (n_sites == 0 || n_angles == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_p1, *d_p2, *d_rv, *d_mv, *d_ang, *d_rot, *d_pos;
    int32_t *d_cp, *d_ci;
    const size_t N = size_t(n_sites) * n_angles;
    TSC_TRY(upload(c, s, p1, size_t(n_sites) * 3, &d_p1));
    TSC_TRY(upload(c, s, p2, size_t(n_sites) * 3, &d_p2));
    TSC_TRY(upload(c, s, ref_vec, size_t(n_sites) * 3, &d_rv));
    TSC_TRY(upload(c, s, mol_vec, size_t(n_sites) * 3, &d_mv));
    TSC_TRY(upload(c, s, conf_pair, size_t(n_sites) * 2, &d_cp));
    TSC_TRY(upload(c, s, angles, size_t(n_angles), &d_ang));
    TSC_TRY(s.get(N * 18, &d_rot));
    TSC_TRY(s.get(N * 6, &d_pos));
    TSC_TRY(s.get(N * 2, &d_ci));
    TSC_TRY(tsc_string_embed_params_dev(c, d_p1, d_p2, d_rv, d_mv, d_cp, n_sites, d_ang, n_angles, d_rot, d_pos, d_ci));
    TSC_HIP(hipMemcpyAsync(rot, d_rot, N * 18 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(pos, d_pos, N * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(conf_idx, d_ci, N * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_cyclical_embed_params(tsc_ctx *c, const double *start, const double *end,
                                                                                const double *direction, const double *pivot, const double *meanpoint,
                                                                                const double *r0, const double *r1, const int32_t *n_reactive,
                                                                                const double *angle, int64_t n, double *rot, double *pos) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && start && end && direction && pivot && meanpoint && r0 && r1 && n_reactive && angle && rot && pos, "tsc_cyclical_embed_params: null argument");
    TSC_REQUIRE(n >= 0, "bad size");
    for (int64_t q = 0; q < n; ++q) TSC_REQUIRE(n_reactive[q] == 1 || n_reactive[q] == 2, "row %lld: n_reactive must be 1 or 2", (long long)q);
    if (n == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    const double *host[7] = {start, end, direction, pivot, meanpoint, r0, r1};
    double *dev[7], *d_angle, *d_rot, *d_pos;
    int32_t *d_nr;
    for (int i = 0; i < 7; ++i) TSC_TRY(upload(c, s, host[i], size_t(n) * 3, &dev[i]));
    TSC_TRY(upload(c, s, n_reactive, size_t(n), &d_nr));
    TSC_TRY(upload(c, s, angle, size_t(n), &d_angle));
    TSC_TRY(s.get(size_t(n) * 9, &d_rot));
    TSC_TRY(s.get(size_t(n) * 3, &d_pos));
    hipLaunchKernelGGL(k_cyclical_embed_params, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)dev[0], (const double *)dev[1],
                       (const double *)dev[2], (const double *)dev[3], (const double *)dev[4], (const double *)dev[5], (const double *)dev[6],
                       (const int32_t *)d_nr, (const double *)d_angle, n, d_rot, d_pos);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(rot, d_rot, size_t(n) * 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(pos, d_pos, size_t(n) * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// the embed loops as drivers: string_embed (tscode/embeds.py:91-120) and cyclical_embed (:636-717, :771-847)

// is_new_structure over fingerprints on the device (tfd.hpp): super-blocks of TG_SUPER candidates, three launches each.
// d_acc u8[n], d_list i32[n], d_nk i32[1] (the number kept, on the device)
static int launch_tfd_greedy(tsc_ctx *c, Scratch &s, const float *d_tf, int64_t n, int T, double thresh, uint8_t *d_acc, int32_t *d_list, int32_t *d_nk) {
    TSC_REQUIRE(T <= 12288, "fingerprints of %d torsions: the greedy filter takes up to 12288", T);
    // per super-block: dead u8[TG_SUPER] and, right behind it, nz u64[TG_WORDS] -- one memset clears both
    uint8_t *d_flags;
    unsigned long long *d_sim;
    float *d_kfp;  // the kept fingerprints once more, compact and in the order they were kept (k_tfd_greedy_prior)
    TSC_TRY(s.get(size_t(TG_SUPER) + TG_WORDS * sizeof(unsigned long long), &d_flags));
    TSC_TRY(s.get(size_t(TG_SUPER) * TG_WORDS, &d_sim));
    TSC_TRY(s.get(std::max<size_t>(size_t(n) * T, 1), &d_kfp));
    int32_t *d_nk_before;
    TSC_TRY(s.get(1, &d_nk_before));
    unsigned long long *d_nz = reinterpret_cast<unsigned long long *>(d_flags + TG_SUPER);
    const size_t lds_pairs = 64 * T <= 12288 ? size_t(64) * std::max(T, 1) * sizeof(float) : 0;
    TSC_HIP(hipMemsetAsync(d_nk, 0, sizeof(int32_t), c->stream));
    for (int64_t base = 0; base < n; base += TG_SUPER) {
        const int nc = int(std::min<int64_t>(TG_SUPER, n - base));
        TSC_HIP(hipMemsetAsync(d_flags, 0, size_t(TG_SUPER) + TG_WORDS * sizeof(unsigned long long), c->stream));
        if (base > 0)
            hipLaunchKernelGGL(k_tfd_greedy_prior, dim3(ceil_div(nc, 256), 64), dim3(256), 0, c->stream, d_tf, base, nc, T, thresh, (const float *)d_kfp,
                               (const int32_t *)d_nk, d_flags);
        hipLaunchKernelGGL(k_tfd_greedy_pairs, dim3(TG_WORDS, ceil_div(nc, 256)), dim3(256), lds_pairs, c->stream, d_tf, base, nc, T, thresh, d_sim, d_nz);
        hipLaunchKernelGGL(k_tfd_greedy_replay, dim3(1), dim3(64), 0, c->stream, (const unsigned long long *)d_sim, (const unsigned long long *)d_nz, base, nc,
                           (const uint8_t *)d_flags, d_acc, d_list, d_nk, d_nk_before);
        if (base + TG_SUPER < n && T > 0)  // (the last super-block's fingerprints are compared with nothing any more)
            hipLaunchKernelGGL(k_tfd_greedy_keep, dim3(1), dim3(256), 0, c->stream, d_tf, T, (const int32_t *)d_list, (const int32_t *)d_nk_before,
                               (const int32_t *)d_nk, d_kfp);
    }
    TSC_HIP(hipGetLastError());
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_tfd_greedy_filter(tsc_ctx *c, const float *tf, int64_t n_structs, int n_quads, double thresh,
                                                                            uint8_t *accepted, int64_t *n_kept) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && tf && accepted, "tsc_tfd_greedy_filter: null argument");
    TSC_REQUIRE(n_structs >= 0 && n_structs < INT32_MAX && n_quads >= 0, "bad sizes");
    if (n_kept) *n_kept = 0;
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    float *d_tf;
    uint8_t *d_acc;
    int32_t *d_list, *d_nk;
    TSC_TRY(upload(c, s, tf, std::max<size_t>(size_t(n_structs) * n_quads, 1), &d_tf));
    TSC_TRY(s.get(size_t(n_structs), &d_acc));
    TSC_TRY(s.get(size_t(n_structs), &d_list));
    TSC_TRY(s.get(1, &d_nk));
    TSC_TRY(launch_tfd_greedy(c, s, d_tf, n_structs, n_quads, thresh, d_acc, d_list, d_nk));
    TSC_HIP(hipMemcpyAsync(accepted, d_acc, size_t(n_structs), hipMemcpyDeviceToHost, c->stream));
    int32_t nk = 0;
    TSC_TRY(read_i32(c, d_nk, &nk));
    if (n_kept) *n_kept = nk;
    return 0;
    TSC_API_GUARD_END
}

// What both drivers share once the pose parameters are on the device: clash verdicts of all candidates, the passing poses
// embedded in candidate order, a filter over them (`filter(d_structs, n_pass, d_pos_scan, d_acc)`), the kept poses compacted
// and copied out, both verdicts per candidate.
template <typename Filter>
static int embed_filter_run(tsc_ctx *c, Scratch &s, const double *d_frags, const FragTable &ft, const int64_t *frag_off, const int32_t *n_atoms,
                            const int32_t *n_conf, const int32_t *d_ci, const double *d_rot, const double *d_pos, int64_t N, double clash_thresh,
                            int64_t max_clashes, uint8_t *clash_ok, uint8_t *kept, double *poses, int64_t poses_capacity, int64_t *n_pass_out,
                            int64_t *n_kept_out, Filter filter) {
    hipStream_t st = c->stream;
    const int n = ft.n_total;
    // copies into the CALLER's host arrays are enqueued long before this function returns: whatever path leaves it -- an error
    // included -- the stream is idle first, so that no copy lands in memory the caller has meanwhile freed
    struct SyncOnExit {
        hipStream_t st;
        ~SyncOnExit() { (void)hipStreamSynchronize(st); }
    } sync_on_exit{st};
    uint8_t *d_mask, *d_kept_full, *d_acc;
    int32_t *bsum, *act, *pos_scan, *total, *act2, *total2;
    TSC_TRY(s.get(size_t(N), &d_mask));
    TSC_TRY(s.get(size_t(N), &d_kept_full));
    TSC_TRY(s.get(scan_bsum_count(N), &bsum));
    TSC_TRY(s.get(size_t(N), &act));
    TSC_TRY(s.get(size_t(N) + 1, &pos_scan));
    TSC_TRY(s.get(1, &total));
    TSC_TRY(s.get(size_t(N), &act2));
    TSC_TRY(s.get(1, &total2));
    TSC_TRY(tsc_embed_clash_mask_dev(c, d_frags, frag_off, n_atoms, n_conf, ft.n_mols, d_ci, d_rot, d_pos, N, clash_thresh, max_clashes, d_mask, nullptr));
    TSC_TRY(scan_mask(st, d_mask, N, bsum, pos_scan, act, nullptr, total));
    TSC_HIP(hipMemcpyAsync(clash_ok, d_mask, size_t(N), hipMemcpyDeviceToHost, st));
    TSC_HIP(hipMemsetAsync(d_kept_full, 0, size_t(N), st));
    int32_t n_pass = 0, n_kept = 0;
    TSC_TRY(read_i32(c, total, &n_pass));
    *n_pass_out = n_pass;
    if (n_pass > 0) {
        double *d_structs, *d_out;
        TSC_TRY(s.get(size_t(n_pass) * n * 3, &d_structs));
        TSC_TRY(s.get(size_t(n_pass), &d_acc));
        hipLaunchKernelGGL(k_transform, dim3(grid_for(n_pass, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), st, d_frags, ft, d_ci, d_rot, d_pos,
                           (const int32_t *)act, int64_t(n_pass), d_structs, (const int32_t *)nullptr, 0, (double *)nullptr, (const int32_t *)nullptr);
        TSC_HIP(hipGetLastError());
        TSC_TRY(filter(d_structs, n_pass, pos_scan, total, d_acc));
        hipLaunchKernelGGL(k_scatter_flags, dim3(grid_for(n_pass, 256, 1024)), dim3(256), 0, st, (const uint8_t *)d_acc, (const int32_t *)act, (const int32_t *)total,
                           d_kept_full);
        TSC_TRY(scan_mask(st, d_acc, n_pass, bsum, nullptr, act2, nullptr, total2));
        TSC_TRY(read_i32(c, total2, &n_kept));
        if (poses && n_kept > 0) {
            TSC_REQUIRE(n_kept <= poses_capacity, "poses holds %lld rows, %d poses were kept", (long long)poses_capacity, n_kept);
            TSC_TRY(s.get(size_t(n_kept) * n * 3, &d_out));
            TSC_TRY(launch_gather_rows(st, d_structs, act2, n_kept, n * 3, nullptr, n * 3, d_out));
            TSC_HIP(hipMemcpyAsync(poses, d_out, size_t(n_kept) * n * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        }
    }
    TSC_HIP(hipMemcpyAsync(kept, d_kept_full, size_t(N), hipMemcpyDeviceToHost, st));
    TSC_HIP(hipStreamSynchronize(st));
    *n_kept_out = n_kept;
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_string_embed(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                                                       const int32_t *n_conf, const double *p1, const double *p2, const double *ref_vec,
                                                                       const double *mol_vec, const int32_t *conf_pair, int64_t n_sites, const double *angles,
                                                                       int n_angles, double clash_thresh, int64_t max_clashes, const int32_t *quads, int n_quads,
                                                                       double tfd_thresh, uint8_t *clash_ok, uint8_t *kept, double *poses,
                                                                       int64_t poses_capacity, int64_t *n_pass, int64_t *n_kept) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && frag_off && n_atoms && n_conf && p1 && p2 && ref_vec && mol_vec && conf_pair && angles && clash_ok && kept && n_pass && n_kept,
                "tsc_string_embed: null argument");
    TSC_REQUIRE(n_sites >= 0 && n_angles >= 0 && n_quads >= 0 && (n_quads == 0 || quads), "bad sizes");
    const int64_t N = n_sites * n_angles;
    TSC_REQUIRE(N < INT32_MAX, "too many candidates");
    *n_pass = *n_kept = 0;
    if (N == 0) return 0;
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, 2, &ft));
    for (int64_t q = 0; q < n_sites; ++q)
        TSC_REQUIRE(conf_pair[2 * q] >= 0 && conf_pair[2 * q] < n_conf[0] && conf_pair[2 * q + 1] >= 0 && conf_pair[2 * q + 1] < n_conf[1],
                    "site %lld: conformer index out of range", (long long)q);
    for (int q = 0; q < 4 * n_quads; ++q) TSC_REQUIRE(quads[q] >= 0 && quads[q] < ft.n_total, "quadruplet atom index %d out of range", quads[q]);
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_frags, *d_p1, *d_p2, *d_rv, *d_mv, *d_ang, *d_rot, *d_pos;
    int32_t *d_cp, *d_ci, *d_quads = nullptr;
    TSC_TRY(upload(c, s, frags, size_t(frags_total_doubles(frag_off, n_atoms, n_conf, 2)), &d_frags));
    TSC_TRY(upload(c, s, p1, size_t(n_sites) * 3, &d_p1));
    TSC_TRY(upload(c, s, p2, size_t(n_sites) * 3, &d_p2));
    TSC_TRY(upload(c, s, ref_vec, size_t(n_sites) * 3, &d_rv));
    TSC_TRY(upload(c, s, mol_vec, size_t(n_sites) * 3, &d_mv));
    TSC_TRY(upload(c, s, conf_pair, size_t(n_sites) * 2, &d_cp));
    TSC_TRY(upload(c, s, angles, size_t(n_angles), &d_ang));
    if (n_quads) TSC_TRY(upload(c, s, quads, size_t(n_quads) * 4, &d_quads));
    TSC_TRY(s.get(size_t(N) * 18, &d_rot));
    TSC_TRY(s.get(size_t(N) * 6, &d_pos));
    TSC_TRY(s.get(size_t(N) * 2, &d_ci));
    TSC_TRY(tsc_string_embed_params_dev(c, d_p1, d_p2, d_rv, d_mv, d_cp, n_sites, d_ang, n_angles, d_rot, d_pos, d_ci));
    const int n = ft.n_total;
    // is_new_structure (:47-69, :119): torsion fingerprints of the passing poses, then the greedy filter over the whole list
    auto filter = [&](const double *d_structs, int32_t np, const int32_t *, const int32_t *, uint8_t *d_acc) -> int {
        float *d_tf;
        int32_t *d_list, *d_nk;
        TSC_TRY(s.get(std::max<size_t>(size_t(np) * n_quads, 1), &d_tf));
        TSC_TRY(s.get(size_t(np), &d_list));
        TSC_TRY(s.get(1, &d_nk));
        if (n_quads)
            hipLaunchKernelGGL(k_torsion_fingerprints, dim3(grid_for(int64_t(np) * n_quads, 256, 256 * 8)), dim3(256), 0, c->stream, d_structs, int64_t(np), n,
                               (const int32_t *)d_quads, n_quads, d_tf);
        TSC_HIP(hipGetLastError());
        return launch_tfd_greedy(c, s, d_tf, int64_t(np), n_quads, tfd_thresh, d_acc, d_list, d_nk);
    };
    return embed_filter_run(c, s, d_frags, ft, frag_off, n_atoms, n_conf, d_ci, d_rot, d_pos, N, clash_thresh, max_clashes, clash_ok, kept, poses, poses_capacity,
                            n_pass, n_kept, filter);
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_cyclical_embed(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                                                         const int32_t *n_conf, int n_mols, const double *start, const double *end,
                                                                         const double *direction, const double *pivot, const double *meanpoint, const double *r0,
                                                                         const double *r1, const int32_t *n_reactive, const double *angle, const int32_t *conf_idx,
                                                                         int64_t n_poses, const int32_t *group_off, int n_groups, double clash_thresh,
                                                                         int64_t max_clashes, double rmsd_thr, uint8_t *clash_ok, uint8_t *kept, double *poses,
                                                                         int64_t poses_capacity, int64_t *n_pass, int64_t *n_kept) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && frag_off && n_atoms && n_conf && start && end && direction && pivot && meanpoint && r0 && r1 && n_reactive && angle && conf_idx &&
                    group_off && clash_ok && kept && n_pass && n_kept,
                "tsc_cyclical_embed: null argument");
    TSC_REQUIRE(n_poses >= 0 && n_poses < INT32_MAX && n_groups >= 0 && rmsd_thr > 0, "bad sizes");
    *n_pass = *n_kept = 0;
    if (n_poses == 0) return 0;
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    const int64_t rows = n_poses * n_mols;
    for (int64_t q = 0; q < rows; ++q) {
        TSC_REQUIRE(n_reactive[q] == 1 || n_reactive[q] == 2, "row %lld: n_reactive must be 1 or 2", (long long)q);
        TSC_REQUIRE(conf_idx[q] >= 0 && conf_idx[q] < n_conf[q % n_mols], "row %lld: conformer index out of range", (long long)q);
    }
    TSC_REQUIRE(n_groups > 0 && group_off[0] == 0 && group_off[n_groups] == n_poses, "group_off must run from 0 to n_poses");
    for (int g = 0; g < n_groups; ++g)
        TSC_REQUIRE(group_off[g + 1] >= group_off[g] && group_off[g + 1] - group_off[g] <= GF_MAX_GROUP, "group %d: sizes must be in [0, %d]", g, GF_MAX_GROUP);
    DeviceGuard guard(c->device);
    Scratch s(c);
    const double *host[7] = {start, end, direction, pivot, meanpoint, r0, r1};
    double *d_frags, *dev[7], *d_angle, *d_rot, *d_pos;
    int32_t *d_nr, *d_ci, *d_goff, *d_goff_pass;
    TSC_TRY(upload(c, s, frags, size_t(frags_total_doubles(frag_off, n_atoms, n_conf, n_mols)), &d_frags));
    for (int i = 0; i < 7; ++i) TSC_TRY(upload(c, s, host[i], size_t(rows) * 3, &dev[i]));
    TSC_TRY(upload(c, s, n_reactive, size_t(rows), &d_nr));
    TSC_TRY(upload(c, s, angle, size_t(rows), &d_angle));
    TSC_TRY(upload(c, s, conf_idx, size_t(rows), &d_ci));
    TSC_TRY(upload(c, s, group_off, size_t(n_groups) + 1, &d_goff));
    TSC_TRY(s.get(size_t(n_groups) + 1, &d_goff_pass));
    TSC_TRY(s.get(size_t(rows) * 9, &d_rot));
    TSC_TRY(s.get(size_t(rows) * 3, &d_pos));
    hipLaunchKernelGGL(k_cyclical_embed_params, dim3(grid_for(rows, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)dev[0], (const double *)dev[1],
                       (const double *)dev[2], (const double *)dev[3], (const double *)dev[4], (const double *)dev[5], (const double *)dev[6],
                       (const int32_t *)d_nr, (const double *)d_angle, rows, d_rot, d_pos);
    TSC_HIP(hipGetLastError());
    const int n = ft.n_total;
    // not _rmsd_similarity(pose, angular_poses, rmsd_thr=1) (:715, :843): greedy inside each group of passing poses
    auto filter = [&](const double *d_structs, int32_t np, const int32_t *pos_scan, const int32_t *total, uint8_t *d_acc) -> int {
        hipLaunchKernelGGL(k_group_offsets_after_filter, dim3(grid_for(n_groups + 1, 256, 1024)), dim3(256), 0, c->stream, (const int32_t *)d_goff, n_groups, pos_scan,
                           n_poses, total, d_goff_pass);
        TSC_HIP(hipGetLastError());
        return tsc_greedy_group_filter_dev(c, d_structs, d_goff_pass, n_groups, np, n, rmsd_thr, d_acc);
    };
    return embed_filter_run(c, s, d_frags, ft, frag_off, n_atoms, n_conf, d_ci, d_rot, d_pos, n_poses, clash_thresh, max_clashes, clash_ok, kept, poses,
                            poses_capacity, n_pass, n_kept, filter);
    TSC_API_GUARD_END
}

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
static const double *pending_basis(const tsc_ctx *c, int h) {
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
        if (c->stage1_f32 == 2 || (c->stage1_f32 == 1 && double(n_poses) * n_heavy * 24.0 >= 128e6)) {
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
            if (c->stage1_f32 == 2 || (c->stage1_f32 == 1 && double(n_poses) * n_heavy * 24.0 >= 128e6))
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
    TSC_TRY(scan_mask(st, clash_mask, n_poses, bsum, nullptr, act, nullptr, total));
    // the passing poses are embedded (all atoms + heavy atoms) by a launch sized for every pose that reads the count on the
    // device: it runs while the host fetches the count it needs to set up the prune (the schedule depends on it)
    TSC_TRY(s.get(size_t(n_poses) * n_heavy * 3, &d_heavy));
    TSC_TRY(read_i32_begin(c, total));
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
    int32_t n_pass = 0;
    TSC_TRY(read_i32_finish(c, &n_pass));
    if (n_pass_host) *n_pass_host = n_pass;
    int64_t n_keep = 0;
    int np = 0;
    if (d_basis) TSC_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
    // automatic kernel choice: where the sample's descriptors hardly differ the screen separates nothing and the all-pairs kernel is the
    // faster route (screen_is_useless).  The side chain finished long ago (it runs beside the clash kernel): no wait in practice
    int force_algo = -1;
    if (d_basis && c->prune_algo == ALGO_AUTO && mode == 1 && n_heavy <= MAX_HP) {
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
