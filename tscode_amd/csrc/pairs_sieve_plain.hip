// pairs_sieve_plain.hip -- the descriptor sieve k_rmsd_sieve<16, CPL, TRIM, false, F32> (sieve.hpp): the shapes that leave the verdicts to k_apply_pass (row tiles dealt to several ranks; option fused_apply = 0).
// gfx950 only.  The ten instantiations of the sieve are the longest part of the build: two translation units of five, side by side.
#include "prune_host.hpp"

int launch_rmsd_sieve_plain(int cpl, bool trim, bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act,
                            const double *Gall, const float *Dc, const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state,
                            const SieveArgs &a, const FusedApply &fa) {
#define TSC_LAUNCH_SIEVE(CPL, TRIM, F32) \
    hipExtLaunchKernelGGL((k_rmsd_sieve<TILE_ROWS, CPL, TRIM, false, F32>), grid, dim3(256), 0, st, e0, e1, 0, heavy, act, Gall, Dc, cend, best, counters, state, a, fa)
    // (stage 1 on the float32 copy exists for the default shape of the kernel only)
    if (f32) {
        TSC_REQUIRE(cpl == 2 && trim, "the float32 stage 1 exists for the trimmed two-column shape only");
        TSC_LAUNCH_SIEVE(2, true, true);
    } else if (cpl == 1) {
        TSC_LAUNCH_SIEVE(1, false, false);
    } else if (cpl == 2 && trim) {
        TSC_LAUNCH_SIEVE(2, true, false);
    } else if (cpl == 2) {
        TSC_LAUNCH_SIEVE(2, false, false);
    } else {
        TSC_LAUNCH_SIEVE(4, false, false);
    }
#undef TSC_LAUNCH_SIEVE
    TSC_HIP(hipGetLastError());
    return 0;
}
