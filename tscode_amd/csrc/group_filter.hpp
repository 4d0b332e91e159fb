// group_filter.hpp -- greedy per-group similarity filter of the embed loops.
//
// tscode/embeds.py:715 and :843 keep a pose of an angular group only if
//     not _rmsd_similarity(pose, poses_accepted_so_far_in_this_group, rmsd_thr=1)
// (tscode/rmsd_pruning.py:208-224: all atoms, no cache, similar = rmsd < thr and maxdev < 2 thr).  The filter is
// sequential inside a group by definition (a pose is tested against the poses ACCEPTED before it), groups are
// independent.  One wavefront per group: for pose s the lanes test it against up to 64 accepted poses at a time
// (sign test, then the exact path for the few that need it); the accepted list lives in LDS.
#pragma once
#include "common.hpp"
#include "sieve.hpp"

namespace tsc {

constexpr int GF_MAX_GROUP = 1024;  // poses per group (the reference's groups hold (steps+1)^n_mols <= 216 by default)

__global__ __launch_bounds__(256) void k_greedy_group_filter(const double *__restrict__ poses, const int32_t *__restrict__ group_off,
                                                              int n_groups, int n_atoms, double thr, uint8_t *__restrict__ accepted,
                                                              double *__restrict__ Gscratch) {
    __shared__ int s_kept[4][GF_MAX_GROUP];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wid;
    if (g >= n_groups) return;
    const int lo = group_off[g], hi = group_off[g + 1];
    const int n3 = n_atoms * 3;
    int *kept = s_kept[wid];
    // squared norms of the group's poses (the sign test needs them)
    for (int s = lo + lane; s < hi; s += 64) {
        const double *x = poses + int64_t(s) * n3;
        double gsum = 0.0;
        for (int e = 0; e < n3; ++e) gsum += x[e] * x[e];
        Gscratch[s] = gsum;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int nk = 0;
    const double half_h_thr2 = 0.5 * double(n_atoms) * thr * thr;
    for (int s = lo; s < hi; ++s) {
        const double *ps = poses + int64_t(s) * n3;
        const double Gs = Gscratch[s];
        bool similar = false;
        for (int base = 0; base < nk && !similar; base += 64) {
            bool sim = false;
            if (base + lane < nk) {
                const int j = kept[base + lane];
                bool exact;
                // argument order as in the reference: rmsd_and_max_numba(ref = pose s, structure = accepted pose)
                sim = pair_is_similar(ps, poses + int64_t(j) * n3, n_atoms, Gs, Gscratch[j], half_h_thr2, thr, 2.0 * thr, exact, 0, 1);
            }
            similar = __ballot(sim) != 0;
        }
        if (lane == 0) accepted[s] = similar ? 0 : 1;
        if (!similar) {
            if (lane == 0) kept[nk] = s;
            ++nk;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace tsc
