// group_filter.hpp -- greedy per-group similarity filter of the embed loops.
//
// tscode/embeds.py:715 and :843 keep a pose of an angular group only if
//     not _rmsd_similarity(pose, poses_accepted_so_far_in_this_group, rmsd_thr=1)
// (tscode/rmsd_pruning.py:208-224: all atoms, no cache, similar = rmsd < thr and maxdev < 2 thr).  The filter is
// sequential inside a group by definition (a pose is tested against the poses ACCEPTED before it), groups are
// independent.
#pragma once
#include "common.hpp"
#include "sieve.hpp"

namespace tsc {

constexpr int GF_MAX_GROUP = 8192;  // poses per group: the reference's groups hold (rotation_steps + 1)^n_mols poses -- 36 or 216 by
                                    // default, 8100 at STEPS = 89 for two molecules; the list of kept poses sits in LDS (32 KB)

// One workgroup per group.  Groups of up to 64 poses (the reference's are 36 or 216 wide before the clash filter, usually far
// fewer after it): all P (P - 1) / 2 pair tests run at once, one per thread, into a P x P bit matrix in LDS, and one thread
// replays the greedy order on the bits -- accepted[s] = no accepted j < s is similar to s.  That evaluates pairs the
// sequential filter would skip (those against rejected poses), at most twice as many, but in one round instead of P
// dependent ones (96 000 poses in 4 000 groups of 24: 0.99 -> 0.47 ms; what is left is every thread walking its own two
// structures in global memory).  Larger groups: wavefront 0
// walks the poses in order and tests each against up to 64 accepted poses at a time.
inline __global__ __launch_bounds__(256) void k_greedy_group_filter(const double *__restrict__ poses, const int32_t *__restrict__ group_off,
                                                              int n_groups, int n_atoms, double thr, uint8_t *__restrict__ accepted,
                                                              double *__restrict__ Gscratch) {
    __shared__ int s_kept[GF_MAX_GROUP];
    __shared__ unsigned long long s_sim[64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, tid = threadIdx.x;
    const int g = blockIdx.x;
    if (g >= n_groups) return;
    const int lo = group_off[g], hi = group_off[g + 1], P = hi - lo;
    const int n3 = n_atoms * 3;
    // squared norms of the group's poses (the sign test needs them)
    for (int s = lo + tid; s < hi; s += 256) {
        const double *x = poses + int64_t(s) * n3;
        double gsum = 0.0;
        for (int e = 0; e < n3; ++e) gsum += x[e] * x[e];
        Gscratch[s] = gsum;
    }
    if (tid < 64) s_sim[tid] = 0ull;
    __syncthreads();
    const double half_h_thr2 = 0.5 * double(n_atoms) * thr * thr;
    if (P <= 64) {
        const int n_pairs = P * (P - 1) / 2;
        for (int q = tid; q < n_pairs; q += 256) {
            int s = int((1.0f + sqrtf(1.0f + 8.0f * float(q))) * 0.5f);  // q = s (s - 1) / 2 + j, j < s
            while (s * (s - 1) / 2 > q) --s;
            while ((s + 1) * s / 2 <= q) ++s;
            const int j = q - s * (s - 1) / 2;
            bool exact;
            // argument order as in the reference: rmsd_and_max_numba(ref = the later pose s, structure = the earlier pose j)
            if (pair_is_similar(poses + int64_t(lo + s) * n3, poses + int64_t(lo + j) * n3, n_atoms, Gscratch[lo + s], Gscratch[lo + j], half_h_thr2, thr,
                                2.0 * thr, exact, 0, 1))
                atomicOr(&s_sim[s], 1ull << j);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long acc = 0ull;
            for (int s = 0; s < P; ++s) {
                const bool ok = (s_sim[s] & acc) == 0ull;
                accepted[lo + s] = ok ? 1 : 0;
                if (ok) acc |= 1ull << s;
            }
        }
        return;
    }
    if (wid != 0) return;
    int nk = 0;
    for (int s = lo; s < hi; ++s) {
        const double *ps = poses + int64_t(s) * n3;
        const double Gs = Gscratch[s];
        bool similar = false;
        for (int base = 0; base < nk && !similar; base += 64) {
            bool sim = false;
            if (base + lane < nk) {
                const int j = s_kept[base + lane];
                bool exact;
                // argument order as in the reference: rmsd_and_max_numba(ref = pose s, structure = accepted pose)
                sim = pair_is_similar(ps, poses + int64_t(j) * n3, n_atoms, Gs, Gscratch[j], half_h_thr2, thr, 2.0 * thr, exact, 0, 1);
            }
            similar = __ballot(sim) != 0;
        }
        if (lane == 0) accepted[s] = similar ? 0 : 1;
        if (!similar) {
            if (lane == 0) s_kept[nk] = s;
            ++nk;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace tsc
