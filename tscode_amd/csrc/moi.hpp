// moi.hpp -- SURVEY.md 8(f) N4: moments of inertia (tscode/algebra.py:165-213), their pair search, embed scores.
//
// get_moi_similarity_matches recomputes the moments of structure j for every pair (i, j); here they are computed once per
// structure (one thread each: centre of mass, inertia tensor, cyclic Jacobi, eigenvalues ordered by absolute value as
// diagonalize does), then one wavefront per row looks for the first j > i whose three moments all lie within
// max_deviation (relative to row i's) -- the same launch shape as the torsion-fingerprint search (tfd.hpp).  The graph
// step of prune_by_moment_of_inertia stays on the host (tscode_amd/optimization_methods.py).
#pragma once
#include "common.hpp"

namespace tsc {

inline __global__ __launch_bounds__(256) void k_inertia_moments(const double *__restrict__ structures, int64_t N, int n, const double *__restrict__ masses,
                                                          double *__restrict__ out) {
    for (int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; s < N; s += int64_t(gridDim.x) * blockDim.x) {
        const double *c = structures + s * n * 3;
        double tot = 0, com0 = 0, com1 = 0, com2 = 0;
        for (int a = 0; a < n; ++a) {  // algebra.py:215-223
            const double m = masses[a];
            tot += m, com0 += c[3 * a] * m, com1 += c[3 * a + 1] * m, com2 += c[3 * a + 2] * m;
        }
        com0 /= tot, com1 /= tot, com2 /= tot;
        double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int a = 0; a < n; ++a) {  // :177-182
            const double x = c[3 * a] - com0, y = c[3 * a + 1] - com1, z = c[3 * a + 2] - com2, m = masses[a];
            const double r2 = x * x + y * y + z * z;
            A[0][0] += m * (r2 - x * x), A[0][1] += m * (-x * y), A[0][2] += m * (-x * z);
            A[1][1] += m * (r2 - y * y), A[1][2] += m * (-y * z), A[2][2] += m * (r2 - z * z);
        }
        A[1][0] = A[0][1], A[2][0] = A[0][2], A[2][1] = A[1][2];
        for (int sweep = 0; sweep < 60; ++sweep) {  // cyclic Jacobi, constant indices only
            const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
            const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
            if (!(off > 1e-32 * dia)) break;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = p + 1; q < 3; ++q) {
                    const double apq = A[p][q];
                    if (apq != 0.0) {
                        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                        double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                        t = theta < 0.0 ? -t : t;
                        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const double akp = A[k][p], akq = A[k][q];
                            A[k][p] = cs * akp - sn * akq, A[k][q] = sn * akp + cs * akq;
                        }
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const double apk = A[p][k], aqk = A[q][k];
                            A[p][k] = cs * apk - sn * aqk, A[q][k] = sn * apk + cs * aqk;
                        }
                    }
                }
        }
        double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2], t;  // order by |eigenvalue| (:209)
        if (fabs(e0) > fabs(e1)) t = e0, e0 = e1, e1 = t;
        if (fabs(e1) > fabs(e2)) t = e1, e1 = e2, e2 = t;
        if (fabs(e0) > fabs(e1)) t = e0, e0 = e1, e1 = t;
        out[3 * s] = e0, out[3 * s + 1] = e1, out[3 * s + 2] = e2;
    }
}

// algebra.py:188-205: first[i] = first j > i with all(|im_i - im_j| / im_i < max_deviation), -1 if none
inline __global__ __launch_bounds__(256) void k_moi_first_similar(const double *__restrict__ mo, int64_t N, double max_deviation, int32_t *__restrict__ first) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); i < N; i += int64_t(gridDim.x) * 4) {
        const double a0 = mo[3 * i], a1 = mo[3 * i + 1], a2 = mo[3 * i + 2];
        int32_t found = -1;
        for (int64_t j0 = i + 1; j0 < N && found < 0; j0 += 64) {
            const int64_t j = j0 + lane;
            bool sim = false;
            if (j < N) sim = fabs(a0 - mo[3 * j]) / a0 < max_deviation && fabs(a1 - mo[3 * j + 1]) / a1 < max_deviation && fabs(a2 - mo[3 * j + 2]) / a2 < max_deviation;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(sim);
            if (m) found = int32_t(j0 + (__ffsll((long long)m) - 1));
        }
        if (lane == 0) first[i] = found;
    }
}

// numba_functions.py:273-288 _score_embed_poses (float32 accumulator, as the reference's array) and the signed error of
// fitness_check (optimization_methods.py:544-557; a NaN target stands for None and is skipped)
inline __global__ __launch_bounds__(256) void k_embed_scores(const double *__restrict__ structures, int64_t N, int n, const int32_t *__restrict__ indices,
                                                       const double *__restrict__ distances, int n_c, float *__restrict__ scores,
                                                       double *__restrict__ fitness_error) {
    for (int64_t s = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; s < N; s += int64_t(gridDim.x) * blockDim.x) {
        const double *c = structures + s * n * 3;
        float sc = 0.0f;
        double err = 0.0;
        for (int i = 0; i < n_c; ++i) {
            const int a = indices[(s * n_c + i) * 2], b = indices[(s * n_c + i) * 2 + 1];
            const double dx = c[3 * a] - c[3 * b], dy = c[3 * a + 1] - c[3 * b + 1], dz = c[3 * a + 2] - c[3 * b + 2];
            const double dist = sqrt(dx * dx + dy * dy + dz * dz), target = distances[s * n_c + i];
            if (target == target) {
                sc = float(double(sc) + fabs(dist - target));
                err += dist - target;
            }
        }
        scores[s] = sc;
        fitness_error[s] = err;
    }
}

}  // namespace tsc
