// ctx.hip -- library / context entry points of include/tscode_hip.h and the tunables
// gfx950 only.  There is deliberately no CPU implementation behind these entry points.
#include "prune_host.hpp"

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
    // runs that are still alive go with their context (their blocks are the context's, their events join its pool below): a host whose
    // finalisers come in no particular order -- Python collecting a run and its engine in one cycle -- must not hand a run to
    // tsc_prune_destroy after this
    while (!c->live_runs.empty()) (void)tsc_prune_destroy(c->live_runs.back());
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
    if (strcmp(name, "sieve_mm") == 0) {
        TSC_REQUIRE(value == 0 || value == 1 || value == 2, "sieve_mm must be 0 (never), 1 (large runs) or 2 (always)");
        c->sieve_mm = int(value);
        return 0;
    }
    if (strcmp(name, "sieve_mm16") == 0) {
        TSC_REQUIRE(value == 0 || value == 1, "sieve_mm16 must be 0 or 1");
        c->sieve_mm16 = int(value);
        return 0;
    }
    if (strcmp(name, "mm_min_n") == 0) {
        TSC_REQUIRE(value >= 0 && value <= 4e9, "mm_min_n must be in [0, 4e9]");
        c->mm_min_n = int64_t(value);
        return 0;
    }
    if (strcmp(name, "mm_seg_cols") == 0) {
        TSC_REQUIRE(value == 0 || (value >= 64 && value <= 1024 && int(value) % 64 == 0), "mm_seg_cols must be 0 (automatic) or a multiple of 64 in [64, 1024]");
        c->mm_seg_cols = int(value);
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
    if (strcmp(name, "cull_xcd") == 0) {
        c->cull_xcd = value != 0.0 ? 1 : 0;
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

