// tfd.hpp -- SURVEY.md 8(f) N2: torsion fingerprints and the pair search of prune_conformers_tfd.
//
// tscode/numba_functions.py:142-231.  The fingerprint of a structure is the float32 vector of its dihedral angles over a
// list of atom quadruplets (:255-264, algebra.py:24-55); two structures are similar when the wrapped absolute differences
// sum to less than `thresh` degrees (:242-253).  A pass cuts the structure list into k chunks and, inside a chunk, every
// row i looks for the first j > i that is similar (:181-199).  Rows are independent and the reference's cache only skips
// pairs already found dissimilar (it cannot change a verdict), so a pass is one launch: one wavefront per row, 64 columns
// per step, ballot for the first hit.  What happens to the matches (networkx components, "first of the cluster") stays on
// the host, with the reference's own Python objects (tscode_amd/numba_functions.py).
#pragma once
#include "common.hpp"

namespace tsc {

// algebra.py:24-55 in fp64, result in degrees; fp contraction off: the fixture values were produced without FMA and the
// result is rounded to float32 right after
__device__ inline double dihedral_deg_dev(const double *__restrict__ p0, const double *__restrict__ p1, const double *__restrict__ p2,
                                          const double *__restrict__ p3) {
#pragma clang fp contract(off)
    double b0[3], b1[3], b2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) b0[k] = -1.0 * (p1[k] - p0[k]), b1[k] = p2[k] - p1[k], b2[k] = p3[k] - p2[k];
    const double n1 = sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) b1[k] /= n1;
    const double d0 = b0[0] * b1[0] + b0[1] * b1[1] + b0[2] * b1[2], d2 = b2[0] * b1[0] + b2[1] * b1[1] + b2[2] * b1[2];
    double v[3], w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = b0[k] - d0 * b1[k], w[k] = b2[k] - d2 * b1[k];
    const double x = v[0] * w[0] + v[1] * w[1] + v[2] * w[2];
    const double c0 = b1[1] * v[2] - b1[2] * v[1], c1 = b1[2] * v[0] - b1[0] * v[2], c2 = b1[0] * v[1] - b1[1] * v[0];
    const double y = c0 * w[0] + c1 * w[1] + c2 * w[2];
    return atan2(y, x) * (180.0 / 3.14159265358979323846);
}

// _get_tf_mat (:233-240): out f32[N][T]
__global__ __launch_bounds__(256) void k_torsion_fingerprints(const double *__restrict__ coords, int64_t N, int n, const int32_t *__restrict__ quads,
                                                               int T, float *__restrict__ out) {
    const int64_t total = N * T;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t s = e / T;
        const int t = int(e - s * T);
        const double *c = coords + s * n * 3;
        const int32_t *q = quads + 4 * t;
        out[e] = float(dihedral_deg_dev(c + 3 * q[0], c + 3 * q[1], c + 3 * q[2], c + 3 * q[3]));
    }
}

// One pass of the pair search (:171-199).  first[i] = absolute index of the first similar j > i inside i's chunk, -1 if
// none or if i lies in no chunk (the last chunk ends at num_active, :175-178).
__global__ __launch_bounds__(256) void k_tfd_first_similar(const float *__restrict__ tf, int64_t N, int T, int64_t d, int64_t k, int64_t num_active,
                                                            double thresh, int32_t *__restrict__ first) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); i < N; i += int64_t(gridDim.x) * 4) {
        int64_t step = i / d;
        if (step > k - 1) step = k - 1;
        const int64_t start = d * step;
        int64_t len = (step == k - 1) ? num_active - start : d;
        if (len < 0) len = 0;
        const int64_t i_rel = i - start;
        int32_t found = -1;
        if (i_rel < len) {
            const float *a = tf + i * T;
            for (int64_t j0 = i_rel + 1; j0 < len && found < 0; j0 += 64) {
                const int64_t j = j0 + lane;
                bool sim = false;
                if (j < len) {
                    const float *b = tf + (start + j) * T;
                    double sum = 0.0;
                    for (int t = 0; t < T; ++t) {
                        const float d32 = fabsf(a[t] - b[t]);          // float32, like the reference's arrays
                        double dd = double(d32);
                        if (d32 > 180.0f) dd -= 360.0;                  // the integer term makes the rest float64
                        sum += fabs(dd);
                    }
                    sim = sum < thresh;
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(sim);
                if (m) found = int32_t(start + j0 + (__ffsll((long long)m) - 1));
            }
        }
        if (lane == 0) first[i] = found;
    }
}

}  // namespace tsc
