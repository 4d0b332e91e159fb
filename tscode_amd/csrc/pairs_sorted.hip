// pairs_sorted.hip -- the pair kernel of the culled passes, k_rmsd_sieve_sorted<F32> (cull.hpp), and the one-launch pass of short chunks,
// k_pass_chunks (local_pass.hpp).  gfx950 only.
#include "prune_host.hpp"

int launch_rmsd_sieve_sorted(bool f32, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *heavy, const int32_t *act, const double *Gall,
                             const int32_t *cend, int32_t *best, PassCounters *counters, const PruneState *state, const SieveArgs &a, const CullArgs &ca,
                             int my_tiles, int n_seg) {
    if (f32) hipExtLaunchKernelGGL(k_rmsd_sieve_sorted<true>, grid, dim3(256), 0, st, e0, e1, 0, heavy, act, Gall, cend, best, counters, state, a, ca, my_tiles, n_seg);
    else hipExtLaunchKernelGGL(k_rmsd_sieve_sorted<false>, grid, dim3(256), 0, st, e0, e1, 0, heavy, act, Gall, cend, best, counters, state, a, ca, my_tiles, n_seg);
    TSC_HIP(hipGetLastError());
    return 0;
}

int launch_pass_chunks(hipStream_t st, unsigned blocks, hipEvent_t e0, hipEvent_t e1, const PassGeom &g, const LocalPassArgs &a, PruneState *state, uint8_t *mask,
                       unsigned long long *bits, int bit_words, const unsigned long long *view, const double *heavy, const double *Gall, const float *Dall,
                       const CacheViews &cv, PassCounters *counters, int32_t *bsum, int block_items, const StepCtx &sc, const StepArgs &sa, LocalTickets *tickets) {
    hipExtLaunchKernelGGL(k_pass_chunks, dim3(blocks), dim3(LP_THREADS), 0, st, e0, e1, 0, g, a, state, mask, bits, bit_words, view, heavy, Gall, Dall, cv, counters,
                          bsum, block_items, sc, sa, tickets);
    TSC_HIP(hipGetLastError());
    return 0;
}
