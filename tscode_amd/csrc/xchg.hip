// xchg.hip -- the sharded prune's small per-pass messages exchanged INSIDE the library: every rank's receive area is mapped into every
// other rank (hipIpcGetMemHandle / hipIpcOpenMemHandle), a rank writes its contribution straight into all peers over xGMI and raises a
// flag there; a peer that has seen all flags reduces what it received into its own buffer.  A one-shot all-reduce, two launches, no host
// in between -- tsc_xchg_allreduce has the signature of tsc_exchange_fn, so tsc_prune_run_sharded can call it as its exchange function
// with no frame of the host language per collective.  (The library still opens no communicator: the HOST moves the 64-byte handles,
// once, over whatever process group it has.)
// gfx950 only.  There is deliberately no CPU implementation behind these entry points.
//
// What may be exchanged this way (rmsd_pruning.py:149-157: chunks write disjoint out_mask[first:last]; :92,101-113: rows independent):
//   TSC_XCHG_SUM_I64  removed-row bits + statistics of a pass partitioned by chunks (7.6 KB at C3, 60 KB at C4), the cache views once;
//   TSC_XCHG_MIN_I32  best[] of a pass dealt by row tiles (<= 1.9 MB at C4).
//
// Memory: the receive area is FINE-GRAINED (uncached) device memory -- a peer's writes arrive over the fabric behind this device's L2,
// which would keep serving stale lines of ordinary (coarse-grained) memory; flags are system-scope atomics.  Layout of an area:
//   [0, 4096)                                   flag[src] at 128-byte spacing: the number of the last exchange src has delivered here
//   4096 + (parity * world + src) * stride      slot of src for exchanges of that parity (two sets: a rank can be one exchange ahead of
//                                               a peer that is still reducing the previous one, never two -- it needs that peer's flag)
#include "host.hpp"

namespace {

constexpr int XCHG_MAX_RANKS = 16;
constexpr size_t XCHG_HEADER = 4096;
constexpr int XCHG_PUSH_BLOCKS = 64;

struct PeerAreas {
    char *area[XCHG_MAX_RANKS];
};

// Every block copies its share of `bytes` (16-byte units; the tail in 4-byte units) into the slot of `rank` in every peer; the block
// that finishes last raises this rank's flag in every peer.
__global__ __launch_bounds__(256) void k_xchg_push(const char *__restrict__ src, int64_t bytes, PeerAreas peers, int rank, int world, size_t slot_off, unsigned seq,
                                                   unsigned *ticket) {
    const int64_t n16 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) ? bytes / 16 : 0;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = reinterpret_cast<const uint4 *>(src)[i];
        for (int p = 0; p < world; ++p)
            if (p != rank) reinterpret_cast<uint4 *>(peers.area[p] + slot_off)[i] = v;
    }
    const int64_t n4 = (bytes - n16 * 16) / 4;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const unsigned v = reinterpret_cast<const unsigned *>(src + n16 * 16)[i];
        for (int p = 0; p < world; ++p)
            if (p != rank) reinterpret_cast<unsigned *>(peers.area[p] + slot_off + n16 * 16)[i] = v;
    }
    __threadfence_system();  // this thread's stores have reached the peers
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int p = 0; p < world; ++p)
                if (p != rank) __hip_atomic_store(reinterpret_cast<unsigned *>(peers.area[p] + 128 * size_t(rank)), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Waits until every peer has delivered exchange `seq` (its flag here is at least seq, in wrap-around arithmetic), then folds the peers'
// slots into buf.  Every block looks at the flags itself (a few words) -- no second launch, no inter-block signal.  A peer that does not
// deliver within `timeout_ticks` of the 100 MHz clock is reported in *status (pinned host memory) and the buffer is left as it is: the
// exit condition every wave reaches.
template <int KIND>
__global__ __launch_bounds__(256) void k_xchg_reduce(unsigned long long *__restrict__ buf, int64_t words, int odd_tail, const char *area, size_t slot_base,
                                                     size_t slot_stride, int rank, int world, unsigned seq, long long timeout_ticks, int *status) {
    __shared__ int late;
    if (threadIdx.x == 0) late = 0;
    __syncthreads();
    if (threadIdx.x < unsigned(world) && int(threadIdx.x) != rank) {
        const unsigned *flag = reinterpret_cast<const unsigned *>(area + 128 * size_t(threadIdx.x));
        const long long t0 = wall_clock64();
        while (int(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > timeout_ticks) {
                late = 1;
                break;
            }
        }
    }
    __syncthreads();
    if (late) {
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < words; i += stride) {
        unsigned long long v = buf[i];
        for (int s = 0; s < world; ++s) {
            if (s == rank) continue;
            const unsigned long long w =
                __hip_atomic_load(reinterpret_cast<const unsigned long long *>(area + slot_base + size_t(s) * slot_stride) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (KIND == TSC_XCHG_SUM_I64) {
                v += w;
            } else {
                const int lo = min(int(unsigned(v)), int(unsigned(w))), hi = min(int(unsigned(v >> 32)), int(unsigned(w >> 32)));
                v = (unsigned long long)(unsigned(lo)) | ((unsigned long long)(unsigned(hi)) << 32);
            }
        }
        buf[i] = v;
    }
    if (KIND == TSC_XCHG_MIN_I32 && odd_tail && blockIdx.x == 0 && threadIdx.x == 0) {  // an odd count: the last int32 on its own
        int *b = reinterpret_cast<int *>(buf + words);
        int v = *b;
        for (int s = 0; s < world; ++s)
            if (s != rank)
                v = min(v, __hip_atomic_load(reinterpret_cast<const int *>(area + slot_base + size_t(s) * slot_stride + size_t(words) * 8), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM));
        *b = v;
    }
}

}  // namespace

struct tsc_xchg {
    tsc_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    int64_t slot_bytes = 0;
    size_t slot_stride = 0, area_bytes = 0;
    char *area = nullptr;
    char *peer[XCHG_MAX_RANKS] = {nullptr};
    bool opened[XCHG_MAX_RANKS] = {false};
    unsigned *ticket = nullptr;
    int *status = nullptr;  // pinned host word: exchanges that gave up waiting for a peer
    unsigned seq = 0;
    bool connected = false;
    double timeout_s = 5.0;
    int64_t n_exchanges = 0;
};

extern "C" __attribute__((visibility("default"))) int tsc_xchg_slot_bytes(int64_t n, int mode, int64_t *bytes) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(bytes && n > 0, "bad argument");
    int64_t words = 0;
    TSC_TRY(tsc_prune_exchange_words(n, mode, &words));  // bits + statistics + every cache view: the largest SUM message is a part of it
    *bytes = std::max<int64_t>(words * 8, n * 4 + 8);    // ... and best[] of a pass dealt by row tiles
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_xchg_destroy(tsc_xchg *x) {
    TSC_API_GUARD_BEGIN
    if (!x) return 0;
    DeviceGuard guard(x->ctx->device);
    (void)hipStreamSynchronize(x->ctx->stream);
    for (int p = 0; p < x->world; ++p)
        if (x->opened[p] && x->peer[p]) (void)hipIpcCloseMemHandle(x->peer[p]);
    if (x->area) (void)hipFree(x->area);
    if (x->ticket) (void)hipFree(x->ticket);
    if (x->status) (void)hipHostFree(x->status);
    delete x;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_xchg_create(tsc_ctx *c, int rank, int world, int64_t slot_bytes, tsc_xchg **out, void *handle_out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && out && handle_out, "tsc_xchg_create: null argument");
    TSC_REQUIRE(world >= 1 && world <= XCHG_MAX_RANKS && rank >= 0 && rank < world, "bad rank %d / world %d (at most %d ranks)", rank, world, XCHG_MAX_RANKS);
    TSC_REQUIRE(slot_bytes > 0 && slot_bytes < (int64_t(1) << 32), "bad slot size");
    static_assert(sizeof(hipIpcMemHandle_t) == TSC_XCHG_HANDLE_BYTES, "the handle is 64 bytes");
    *out = nullptr;
    DeviceGuard guard(c->device);
    tsc_xchg *x = new (std::nothrow) tsc_xchg();
    if (!x) return fail(TSC_ERR_NOMEM, "out of host memory");
    x->ctx = c, x->rank = rank, x->world = world, x->slot_bytes = slot_bytes;
    x->slot_stride = (size_t(slot_bytes) + 255) & ~size_t(255);
    x->area_bytes = XCHG_HEADER + 2 * size_t(world) * x->slot_stride;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&x->area), x->area_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(reinterpret_cast<void **>(&x->area), x->area_bytes, hipDeviceMallocFinegrained);
    }
    if (e == hipSuccess) e = hipMemset(x->area, 0, x->area_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&x->ticket), 256);
    if (e == hipSuccess) e = hipMemset(x->ticket, 0, 256);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&x->status), 64, hipHostMallocDefault);
    if (e == hipSuccess) {
        *x->status = 0;
        e = hipDeviceSynchronize();
    }
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, x->area);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)tsc_xchg_destroy(x);
        return fail(e == hipErrorOutOfMemory ? TSC_ERR_NOMEM : TSC_ERR_HIP, "tsc_xchg_create: %s (fine-grained allocation of %zu bytes / hipIpcGetMemHandle)",
                    hipGetErrorString(e), size_t(XCHG_HEADER + 2 * size_t(world) * ((size_t(slot_bytes) + 255) & ~size_t(255))));
    }
    memcpy(handle_out, &h, sizeof(h));
    x->peer[rank] = x->area;
    *out = x;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_xchg_connect(tsc_xchg *x, const void *handles) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(x && handles, "tsc_xchg_connect: null argument");
    if (x->connected) return fail(TSC_ERR_STATE, "tsc_xchg_connect: already connected");
    DeviceGuard guard(x->ctx->device);
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char *>(handles) + size_t(p) * TSC_XCHG_HANDLE_BYTES, sizeof(h));
        void *q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(TSC_ERR_HIP, "tsc_xchg_connect: hipIpcOpenMemHandle of rank %d's area failed: %s (multi-process GPU work on this driver needs "
                                     "HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment)", p, hipGetErrorString(e));
        }
        x->peer[p] = static_cast<char *>(q), x->opened[p] = true;
    }
    x->connected = true;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_xchg_set_timeout(tsc_xchg *x, double seconds) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(x && seconds > 0 && seconds <= 600, "bad argument");
    x->timeout_s = seconds;
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_xchg_status(tsc_xchg *x, int64_t *n_exchanges, int *n_timeouts) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(x != nullptr, "null argument");
    if (n_exchanges) *n_exchanges = x->n_exchanges;
    if (n_timeouts) *n_timeouts = *static_cast<volatile int *>(x->status);
    return 0;
    TSC_API_GUARD_END
}

// tsc_exchange_fn: user = the tsc_xchg.  Enqueues on the context's stream; returns without waiting.
extern "C" __attribute__((visibility("default"))) int tsc_xchg_allreduce(void *user, int kind, void *buf_dev, int64_t count) {
    TSC_API_GUARD_BEGIN
    tsc_xchg *x = static_cast<tsc_xchg *>(user);
    TSC_REQUIRE(x && buf_dev && count >= 0, "tsc_xchg_allreduce: null argument");
    TSC_REQUIRE(kind == TSC_XCHG_SUM_I64 || kind == TSC_XCHG_MIN_I32, "tsc_xchg_allreduce: kind %d", kind);
    if (!x->connected && x->world > 1) return fail(TSC_ERR_STATE, "tsc_xchg_allreduce: not connected (tsc_xchg_connect)");
    const int64_t bytes = count * (kind == TSC_XCHG_SUM_I64 ? 8 : 4);
    TSC_REQUIRE(bytes <= x->slot_bytes, "tsc_xchg_allreduce: %lld bytes exceed the slots of %lld bytes this exchange was created with", (long long)bytes,
                (long long)x->slot_bytes);
    TSC_REQUIRE((reinterpret_cast<uintptr_t>(buf_dev) & 7u) == 0, "tsc_xchg_allreduce: the buffer must be 8-byte aligned");
    ++x->n_exchanges;
    if (x->world == 1 || count == 0) return 0;
    tsc_ctx *c = x->ctx;
    DeviceGuard guard(c->device);
    const unsigned seq = ++x->seq;
    const int parity = int(seq & 1u);
    const size_t slot_base = XCHG_HEADER + size_t(parity) * x->world * x->slot_stride;
    PeerAreas peers;
    for (int p = 0; p < XCHG_MAX_RANKS; ++p) peers.area[p] = p < x->world ? x->peer[p] : nullptr;
    const int pb = int(std::max<int64_t>(1, std::min<int64_t>(XCHG_PUSH_BLOCKS, ceil_div<int64_t>(bytes, 16 * 256 * 4))));
    hipLaunchKernelGGL(k_xchg_push, dim3(pb), dim3(256), 0, c->stream, static_cast<const char *>(buf_dev), bytes, peers, x->rank, x->world,
                       slot_base + size_t(x->rank) * x->slot_stride, seq, x->ticket);
    const int64_t words = bytes / 8;
    const int rb = int(std::max<int64_t>(1, std::min<int64_t>(256, ceil_div<int64_t>(words, 256 * 4))));
    const long long ticks = (long long)(x->timeout_s * 1.0e8);  // wall_clock64: 100 MHz
    if (kind == TSC_XCHG_SUM_I64)
        hipLaunchKernelGGL(k_xchg_reduce<TSC_XCHG_SUM_I64>, dim3(rb), dim3(256), 0, c->stream, static_cast<unsigned long long *>(buf_dev), words, 0, (const char *)x->area,
                           slot_base, x->slot_stride, x->rank, x->world, seq, ticks, x->status);
    else
        hipLaunchKernelGGL(k_xchg_reduce<TSC_XCHG_MIN_I32>, dim3(rb), dim3(256), 0, c->stream, static_cast<unsigned long long *>(buf_dev), words, int(count & 1), (const char *)x->area,
                           slot_base, x->slot_stride, x->rank, x->world, seq, ticks, x->status);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}
