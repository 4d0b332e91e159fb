// pairs_mm.hip -- the pair kernel with the descriptor screen on the matrix cores, k_rmsd_sieve_mm<FUSED, F32> (mm.hpp), and its form for the culled passes, k_rmsd_sieve_sorted_mm<F32> (cull_mm.hpp).  gfx950 only.
#include "prune_host.hpp"

int launch_rmsd_sieve_mm(bool fused, bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                         const float *Dc, const _Float16 *Dh, const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state, const SieveArgs &a,
                         const FusedApply &fa) {
#define TSC_LAUNCH_MM(FUSED, F32) \
    hipExtLaunchKernelGGL((k_rmsd_sieve_mm<FUSED, F32>), grid, dim3(64 * MM_WAVES), 0, st, e0, e1, 0, heavy, act, Gall, Dc, Dh, cend, best, counters, state, a, fa)
    if (fused && f32) TSC_LAUNCH_MM(true, true);
    else if (fused) TSC_LAUNCH_MM(true, false);
    else if (f32) TSC_LAUNCH_MM(false, true);
    else TSC_LAUNCH_MM(false, false);
#undef TSC_LAUNCH_MM
    TSC_HIP(hipGetLastError());
    return 0;
}

int launch_rmsd_sieve_sorted_mm(bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                                const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state, const SieveArgs &a, const CullArgs &ca,
                                const CullMmArgs &cm, int n_groups, int n_seg) {
    if (f32) hipExtLaunchKernelGGL(k_rmsd_sieve_sorted_mm<true>, grid, dim3(64), 0, st, e0, e1, 0, heavy, act, Gall, cend, best, counters, state, a, ca, cm, n_groups, n_seg);
    else hipExtLaunchKernelGGL(k_rmsd_sieve_sorted_mm<false>, grid, dim3(64), 0, st, e0, e1, 0, heavy, act, Gall, cend, best, counters, state, a, ca, cm, n_groups, n_seg);
    TSC_HIP(hipGetLastError());
    return 0;
}

int launch_rmsd_sieve_mm16(bool fused, bool f32, int waves, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                           const _Float16 *Dh, const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state,
                           const SieveArgs &a, const FusedApply &fa) {
#define TSC_LAUNCH_MM16(FUSED, F32, W) \
    hipExtLaunchKernelGGL((k_rmsd_sieve_mm16<FUSED, F32, W>), grid, dim3(64 * W), 0, st, e0, e1, 0, heavy, act, Gall, Dh, cend, best, counters, state, a, fa)
#define TSC_LAUNCH_MM16_W(FUSED, F32) \
    do {                              \
        if (waves == 2) TSC_LAUNCH_MM16(FUSED, F32, 2); \
        else TSC_LAUNCH_MM16(FUSED, F32, 4);            \
    } while (0)
    TSC_REQUIRE(waves == 2 || waves == 4, "k_rmsd_sieve_mm16: 2 or 4 wavefronts per workgroup");
    if (fused && f32) TSC_LAUNCH_MM16_W(true, true);
    else if (fused) TSC_LAUNCH_MM16_W(true, false);
    else if (f32) TSC_LAUNCH_MM16_W(false, true);
    else TSC_LAUNCH_MM16_W(false, false);
#undef TSC_LAUNCH_MM16_W
#undef TSC_LAUNCH_MM16
    TSC_HIP(hipGetLastError());
    return 0;
}
