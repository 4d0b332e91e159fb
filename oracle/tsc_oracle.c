/*
 * tsc_oracle.c -- CPU restatement of TSCoDe's geometry hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP engine in tscode_amd/.  It restates, in plain C,
 * the algorithm of the reference (ntampellini/TSCoDe v0.4.16) for the path named in
 * BASELINE.json; every function cites the reference file:line it follows.  It is pinned against
 * golden vectors produced by running the reference's own Python in the build container
 * (tests/golden/gen_golden.py -> tests/golden/G*.npz; tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (tscode_amd) never links, imports or calls it.
 *
 * Third-party arithmetic on the path: the reference reaches LAPACK dgesdd/dgetrf and BLAS dgemm
 * through np.linalg.svd / np.linalg.det / "@" (rmsd_pruning.py:15,19,20,26,29; algebra.py:275-282).
 * Those sources are not under /root/reference; the 3x3 SVD here is a one-sided Jacobi (Hestenes)
 * SVD, which yields the same factorisation up to the usual sign/ordering freedom; the golden
 * vectors pin the results (rmsd, max deviation, rotation matrices) to <= 1e-9.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */

/* algebra.py:89-96  norm_of(vec) = sqrt(x*x + y*y + z*z) */
static inline double norm_of3(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

static inline double det3(const double m[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

static void matmul3(const double a[9], const double b[9], double c[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

/*
 * 3x3 SVD  A = U diag(s) Vt  with s[0] >= s[1] >= s[2] >= 0 (the np.linalg.svd contract used at
 * rmsd_pruning.py:19 and algebra.py:275).  One-sided Jacobi: right-rotate column pairs of A until
 * they are mutually orthogonal; column norms are the singular values.  Null columns (rank-deficient
 * input) are completed to an orthonormal basis by cross products.
 */
static void svd3(const double A[9], double U[9], double s[3], double Vt[9]) {
    double a[3][3], v[3][3]; /* a[col][row], v[col][row] */
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            a[c][r] = A[3 * r + c];
            v[c][r] = (r == c) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += a[i][r] * a[i][r];
                    beta += a[j][r] * a[j][r];
                    gamma += a[i][r] * a[j][r];
                }
                if (gamma == 0.0 || fabs(gamma) <= 2.3e-16 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int r = 0; r < 3; ++r) {
                    double x = a[i][r], y = a[j][r];
                    a[i][r] = cs * x - sn * y;
                    a[j][r] = sn * x + cs * y;
                    x = v[i][r], y = v[j][r];
                    v[i][r] = cs * x - sn * y;
                    v[j][r] = sn * x + cs * y;
                }
            }
        if (!rotated) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) nrm[c] = sqrt(a[c][0] * a[c][0] + a[c][1] * a[c][1] + a[c][2] * a[c][2]);
    for (int i = 0; i < 2; ++i) /* sort descending */
        for (int j = i + 1; j < 3; ++j)
            if (nrm[ord[j]] > nrm[ord[i]]) {
                int t = ord[i];
                ord[i] = ord[j];
                ord[j] = t;
            }
    /* Left vectors: u0 from the largest column; u1 by Gram-Schmidt of the second column against u0;
     * u2 = +-(u0 x u1), the sign taken from the third column.  This keeps U orthonormal when the
     * input is rank deficient (columns of norm ~1e-17 are rounding noise, not directions). */
    double u[3][3];
    for (int k = 0; k < 3; ++k) {
        s[k] = nrm[ord[k]];
        for (int r = 0; r < 3; ++r) Vt[3 * k + r] = v[ord[k]][r];
    }
    if (!(s[0] > 0.0)) { /* A == 0 */
        for (int k = 0; k < 3; ++k)
            for (int r = 0; r < 3; ++r) u[k][r] = (k == r) ? 1.0 : 0.0;
    } else {
        const double *a0 = a[ord[0]], *a1 = a[ord[1]], *a2 = a[ord[2]];
        for (int r = 0; r < 3; ++r) u[0][r] = a0[r] / s[0];
        double d = a1[0] * u[0][0] + a1[1] * u[0][1] + a1[2] * u[0][2];
        double w[3] = {a1[0] - d * u[0][0], a1[1] - d * u[0][1], a1[2] - d * u[0][2]};
        double n = norm_of3(w);
        if (!(n > 1e-13 * s[0])) { /* any unit vector orthogonal to u0 */
            int m = 0;
            if (fabs(u[0][1]) < fabs(u[0][m])) m = 1;
            if (fabs(u[0][2]) < fabs(u[0][m])) m = 2;
            d = u[0][m];
            w[0] = -d * u[0][0], w[1] = -d * u[0][1], w[2] = -d * u[0][2];
            w[m] += 1.0;
            n = norm_of3(w);
        }
        for (int r = 0; r < 3; ++r) u[1][r] = w[r] / n;
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
        u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
        u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        if (u[2][0] * a2[0] + u[2][1] * a2[1] + u[2][2] * a2[2] < 0.0)
            for (int r = 0; r < 3; ++r) u[2][r] = -u[2][r];
    }
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 3; ++r) U[3 * r + k] = u[k][r];
}

/* ------------------------------------------------------------------------------------------ */
/* rmsd_pruning.py:6-41  rmsd_and_max_numba(p, q)                                              */

ORC_API void orc_rmsd_and_max(const double *p, const double *q, int h, double *rmsd_out, double *maxdev_out) {
    /* :15  cov_mat = p.T @ q */
    double cov[9] = {0};
    for (int a = 0; a < h; ++a)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) cov[3 * i + j] += p[3 * a + i] * q[3 * a + j];
    /* :19  v, _, w = svd(cov_mat) */
    double v[9], s[3], w[9];
    svd3(cov, v, s, w);
    /* :20-23  d = det(v)*det(w) < 0  ->  v[:, -1] = -v[:, -1] */
    if (det3(v) * det3(w) < 0.0) {
        v[2] = -v[2];
        v[5] = -v[5];
        v[8] = -v[8];
    }
    /* :26  rot_mat = v @ w */
    double rot[9];
    matmul3(v, w, rot);
    /* :29-39  p = p @ rot_mat; diff = p - q; rmsd = sqrt(sum(diff^2)/len); max_delta = max ||diff_a|| */
    double ss = 0.0, mx = 0.0;
    for (int a = 0; a < h; ++a) {
        double d[3];
        for (int j = 0; j < 3; ++j)
            d[j] = p[3 * a] * rot[j] + p[3 * a + 1] * rot[3 + j] + p[3 * a + 2] * rot[6 + j] - q[3 * a + j];
        ss += d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        double n = norm_of3(d);
        if (a == 0 || n > mx) mx = n;
    }
    *rmsd_out = sqrt(ss / (double)h);
    *maxdev_out = mx;
}

ORC_API void orc_rmsd_pairs(const double *heavy, int h, const int64_t *pairs, int64_t n_pairs, double *rmsd, double *maxdev) {
    for (int64_t k = 0; k < n_pairs; ++k)
        orc_rmsd_and_max(heavy + (size_t)pairs[2 * k] * h * 3, heavy + (size_t)pairs[2 * k + 1] * h * 3, h, rmsd + k, maxdev + k);
}

/* rmsd_pruning.py:208-224  _rmsd_similarity(ref, structures, rmsd_thr): all atoms, no cache */
ORC_API int orc_rmsd_similarity(const double *ref, const double *structures, int64_t count, int n, double rmsd_thr) {
    for (int64_t s = 0; s < count; ++s) {
        double r, m;
        orc_rmsd_and_max(ref, structures + (size_t)s * n * 3, n, &r, &m);
        if (r < rmsd_thr && m < 2 * rmsd_thr) return 1;
    }
    return 0;
}

/* embeds.py:715 / :843 greedy use of _rmsd_similarity inside one angular group:
 * a pose is accepted iff it is not similar to any pose accepted before it. */
ORC_API void orc_greedy_group_filter(const double *poses, int64_t count, int n, double rmsd_thr, uint8_t *accepted) {
    int64_t *kept = (int64_t *)malloc(sizeof(int64_t) * (size_t)(count > 0 ? count : 1));
    int64_t nk = 0;
    for (int64_t s = 0; s < count; ++s) {
        int sim = 0;
        for (int64_t k = 0; k < nk && !sim; ++k) {
            double r, m;
            orc_rmsd_and_max(poses + (size_t)s * n * 3, poses + (size_t)kept[k] * n * 3, n, &r, &m);
            sim = (r < rmsd_thr && m < 2 * rmsd_thr);
        }
        accepted[s] = (uint8_t)!sim;
        if (!sim) kept[nk++] = s;
    }
    free(kept);
}

/* ------------------------------------------------------------------------------------------ */
/* algebra.py:98-157  all_dists(A, B): C[i,j] = sqrt(sum_k (A[i,k]-B[j,k])^2)                  */
/* (the 32x32 blocking of the reference only reorders independent entries)                    */

ORC_API void orc_all_dists(const double *A, int na, const double *B, int nb, double *C) {
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) {
                double d = A[3 * i + k] - B[3 * j + k];
                acc += d * d;
            }
            C[(size_t)i * nb + j] = sqrt(acc);
        }
}

static int64_t count_below(const double *A, int na, const double *B, int nb, double thresh) {
    int64_t c = 0;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) {
                double d = A[3 * i + k] - B[3 * j + k];
                acc += d * d;
            }
            c += sqrt(acc) < thresh;
        }
    return c;
}

/* numba_functions.py:49-56  count_clashes: ordered self pairs with 0 < d < 0.5 */
ORC_API int64_t orc_count_clashes(const double *coords, int n) {
    int64_t c = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) {
                double d = coords[3 * i + k] - coords[3 * j + k];
                acc += d * d;
            }
            double d = sqrt(acc);
            c += (d < 0.5) && (d > 0.0);
        }
    return c;
}

/* numba_functions.py:59-105  compenetration_check(coords, ids, thresh, max_clashes) -> 0/1
 * n_ids == 0 stands for ids=None.  counts_out (optional, 3 entries) receives the per-fragment-pair
 * counts that were accumulated before the function returned (-1 = not evaluated: early exit). */
ORC_API int orc_compenetration_check(const double *coords, int n, const int64_t *ids, int n_ids, double thresh,
                                     int64_t max_clashes, int64_t *counts_out) {
    if (counts_out) counts_out[0] = counts_out[1] = counts_out[2] = -1;
    if (n_ids == 0) { /* :71-72 */
        int64_t c = orc_count_clashes(coords, n);
        if (counts_out) counts_out[0] = c;
        return c > max_clashes ? 0 : 1;
    }
    if (n_ids == 2) { /* :74-81  m1 = coords[0:ids[0]], m2 = coords[ids[0]:]; all_dists(m2, m1) */
        int n1 = (int)ids[0];
        int64_t c = count_below(coords + 3 * n1, n - n1, coords, n1, thresh);
        if (counts_out) counts_out[0] = c;
        return c > max_clashes ? 0 : 1;
    }
    /* :85-105 three fragments, cumulative count, early exits */
    int n1 = (int)ids[0], n2 = (int)ids[1], n3 = n - n1 - n2;
    const double *m1 = coords, *m2 = coords + 3 * n1, *m3 = coords + 3 * (n1 + n2);
    int64_t clashes = 0, c;
    c = count_below(m2, n2, m1, n1, thresh);
    if (counts_out) counts_out[0] = c;
    clashes += c;
    if (clashes > max_clashes) return 0;
    c = count_below(m3, n3, m2, n2, thresh);
    if (counts_out) counts_out[1] = c;
    clashes += c;
    if (clashes > max_clashes) return 0;
    c = count_below(m1, n1, m3, n3, thresh);
    if (counts_out) counts_out[2] = c;
    clashes += c;
    if (clashes > max_clashes) return 0;
    return 1;
}

/* embedder.py:1243-1248  compenetration_refining loop: mask[s] = compenetration_check(structure_s, ...) */
ORC_API void orc_compenetration_mask(const double *coords, int64_t n_poses, int n, const int64_t *ids, int n_ids,
                                     double thresh, int64_t max_clashes, uint8_t *mask) {
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < n_poses; ++s)
        mask[s] = (uint8_t)orc_compenetration_check(coords + (size_t)s * n * 3, n, ids, n_ids, thresh, max_clashes, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* embeds.py:961-969  get_embed, batched: out[s] = concat_m (R[s,m] @ X_m[c[s,m]].T).T + t[s,m]  */
/* algebra.py:390-400 transform_coords is the single-fragment case.                            */

ORC_API void orc_transform_batch(const double *const *frag_coords, const int64_t *frag_natoms, int n_mols,
                                 const int32_t *conf_idx, const double *rot, const double *pos, int64_t n_poses, double *out) {
    int n = 0;
    for (int m = 0; m < n_mols; ++m) n += (int)frag_natoms[m];
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < n_poses; ++s) {
        double *o = out + (size_t)s * n * 3;
        for (int m = 0; m < n_mols; ++m) {
            const double *X = frag_coords[m] + (size_t)conf_idx[s * n_mols + m] * frag_natoms[m] * 3;
            const double *R = rot + ((size_t)s * n_mols + m) * 9;
            const double *t = pos + ((size_t)s * n_mols + m) * 3;
            for (int a = 0; a < frag_natoms[m]; ++a, o += 3)
                for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * X[3 * a] + R[3 * i + 1] * X[3 * a + 1] + R[3 * i + 2] * X[3 * a + 2] + t[i];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* rotation helpers (pose parameters)                                                         */

/* algebra.py:284-323  quaternion (x, y, z, w) scalar-last -> matrix */
ORC_API void orc_quaternion_to_rotation_matrix(const double Q[4], double out[9]) {
    double q0 = Q[3], q1 = Q[0], q2 = Q[1], q3 = Q[2];
    out[0] = 2 * (q0 * q0 + q1 * q1) - 1;
    out[1] = 2 * (q1 * q2 - q0 * q3);
    out[2] = 2 * (q1 * q3 + q0 * q2);
    out[3] = 2 * (q1 * q2 + q0 * q3);
    out[4] = 2 * (q0 * q0 + q2 * q2) - 1;
    out[5] = 2 * (q2 * q3 - q0 * q1);
    out[6] = 2 * (q1 * q3 - q0 * q2);
    out[7] = 2 * (q2 * q3 + q0 * q1);
    out[8] = 2 * (q0 * q0 + q3 * q3) - 1;
}

/* algebra.py:325-344  rot_mat_from_pointer(pointer, angle_deg) */
ORC_API void orc_rot_mat_from_pointer(const double pointer[3], double angle_deg, double out[9]) {
    double n = norm_of3(pointer); /* algebra.py:80-87 norm() */
    double u[3] = {pointer[0] / n, pointer[1] / n, pointer[2] / n};
    double ang = angle_deg * (M_PI / 180);
    double s = sin(ang / 2), c = cos(ang / 2);
    double quat[4] = {s * u[0], s * u[1], s * u[2], c};
    orc_quaternion_to_rotation_matrix(quat, out);
}

/* algebra.py:258-282  align_vec_pair(ref, tgt): B[i,k] = sum_j ref[j][i]*tgt[j][k]; SVD; det fix; u @ vh */
ORC_API void orc_align_vec_pair(const double ref[6], const double tgt[6], double out[9]) {
    double B[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) B[3 * i + k] = ref[i] * tgt[k] + ref[3 + i] * tgt[3 + k];
    double u[9], s[3], vh[9], uv[9];
    svd3(B, u, s, vh);
    matmul3(u, vh, uv);
    if (det3(uv) < 0) {
        u[2] = -u[2];
        u[5] = -u[5];
        u[8] = -u[8];
    }
    matmul3(u, vh, out);
}

/* utils.py:183-208  rotation_matrix_from_vectors(vec1, vec2) (Rodrigues; exact-zero tests kept) */
ORC_API void orc_rotation_matrix_from_vectors(const double v1[3], const double v2[3], double out[9]) {
    double n1 = norm_of3(v1), n2 = norm_of3(v2);
    double a[3] = {v1[0] / n1, v1[1] / n1, v1[2] / n1}, b[3] = {v2[0] / n2, v2[1] / n2, v2[2] / n2};
    double v[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    double s = norm_of3(v);
    if (s != 0) {
        double c = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        double k[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}, kk[9];
        matmul3(k, k, kk);
        double f = (1 - c) / (s * s);
        for (int i = 0; i < 9; ++i) out[i] = ((i % 4 == 0) ? 1.0 : 0.0) + k[i] + kk[i] * f;
        return;
    }
    double ab[3] = {a[0] + b[0], a[1] + b[1], a[2] + b[2]};
    if (norm_of3(ab) == 0) {
        double z[3] = {0, 0, 1};
        orc_rot_mat_from_pointer(z, 180, out);
        return;
    }
    for (int i = 0; i < 9; ++i) out[i] = (i % 4 == 0) ? 1.0 : 0.0;
}

/* algebra.py:58-62  vec_angle(v1, v2) in degrees */
ORC_API double orc_vec_angle(const double v1[3], const double v2[3]) {
    double n1 = norm_of3(v1), n2 = norm_of3(v2);
    double d = (v1[0] / n1) * (v2[0] / n2) + (v1[1] / n1) * (v2[1] / n2) + (v1[2] / n1) * (v2[2] / n2);
    if (d > 1.0) d = 1.0;
    if (d < -1.0) d = -1.0;
    return acos(d) * 180 / M_PI;
}

/* ------------------------------------------------------------------------------------------ */
/* rmsd_pruning.py:43-206  prune_conformers_rmsd                                               */

typedef struct {
    int64_t k;               /* number of chunks of this pass */
    int64_t n_active_before; /* count_nonzero(mask) entering the pass */
    int64_t n_active_after;
    int64_t pairs_evaluated; /* calls of rmsd_and_max_numba (rmsd_pruning.py:70) */
    int64_t cache_hit_exits; /* rows that returned at :66-67 */
    int64_t new_keys;        /* keys appended at :76 */
    double seconds;
} orc_pass_stats;

static const double ORC_KS[18] = {5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1};

/*
 * heavy: f64[N, h, 3]  (structures[:, atomnos != 1], rmsd_pruning.py:178-179)
 * mode 0: reference-exact, including the pair cache of :65-67/:75-77
 * mode 1: cache-free (the cache test is skipped; labelled non-parity wherever reported)
 * mask_out: u8[N]; stats: up to 18 entries; pass_masks (optional): u8[18, N] mask after each pass
 * keys_out (optional): i64[N, 2] the cache keys (first, first+delta) in creation order; n_keys_out
 * row_parallel != 0: rows of a chunk are also spread over threads (NOT the reference's parallelisation,
 *   results identical because rows of a pass are independent: SURVEY.md F4)
 *
 * The cache lookup "hash_value in cache" (a linear scan of a typed List in the reference, :66) is done
 * through a per-pass bitmap: a key (a, b) can only be hit in a pass where a is a chunk start and b lies
 * in that chunk, and then it is hit exactly by the pairs (i, j) of that chunk with a + (j - i) == b.
 * For N > 200 000 the reference fails (float k reaches range(): SURVEY.md F6); int(k) is used here.
 */
ORC_API int orc_prune_rmsd(const double *heavy, int64_t N, int h, double rmsd_thr, int mode, int row_parallel,
                           uint8_t *mask_out, orc_pass_stats *stats, int *n_passes_out, uint8_t *pass_masks,
                           int64_t *keys_out, int64_t *n_keys_out) {
    if (N <= 0 || h <= 0) return -1;
    const double maxdev_thr = 2 * rmsd_thr; /* :95 */
    uint8_t *mask = (uint8_t *)malloc((size_t)N), *out = (uint8_t *)malloc((size_t)N), *dbit = (uint8_t *)calloc((size_t)N + 1, 1);
    int64_t *key_a = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *key_b = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *row_key = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t n_keys = 0;
    memset(mask, 1, (size_t)N); /* :182 */
    int n_passes = 0;

    for (int ks = 0; ks < 18; ++ks) {
        int64_t k = (int64_t)ORC_KS[ks];
        int64_t n_active = 0;
        for (int64_t i = 0; i < N; ++i) n_active += mask[i];
        if (!(k == 1 || 20 * k < n_active)) continue; /* :192 */
#ifdef _OPENMP
        double t0 = omp_get_wtime();
#endif
        int64_t cs = N / k; /* :136 */
        /* per-pass view of the cache */
        memset(dbit, 0, (size_t)N + 1);
        if (mode == 0)
            for (int64_t q = 0; q < n_keys; ++q) {
                int64_t a = key_a[q], b = key_b[q];
                if (cs <= 0 || a % cs != 0) continue;
                int64_t c = a / cs;
                if (c >= k) continue;
                int64_t last = (c == k - 1) ? N : cs * (c + 1);
                if (b < last) dbit[b] = 1;
            }
        int64_t evals = 0, hits = 0;
        /* :139 prange over chunks; each chunk is serial over rows (:98) unless row_parallel */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : evals, hits) if (!row_parallel)
        for (int64_t chunk = 0; chunk < k; ++chunk) {
            int64_t first = chunk * cs;                              /* :140 */
            int64_t last = (chunk == k - 1) ? N : cs * (chunk + 1); /* :141-144 */
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : evals, hits) if (row_parallel)
            for (int64_t i = first; i < last; ++i) {
                row_key[i] = -1;
                if (!mask[i]) { /* :118-119 */
                    out[i] = 0;
                    continue;
                }
                int similar = 0;
                const double *ref = heavy + (size_t)i * h * 3;
                for (int64_t j = i + 1; j < last; ++j) { /* :57 over structures[i+1:] of the chunk */
                    if (!mask[j]) continue;              /* :60 */
                    int64_t b = first + (j - i);         /* :65 hash_value = (first, first+1+(j-i-1)) */
                    if (dbit[b]) {                       /* :66-67 */
                        ++hits;
                        break;
                    }
                    double r, m;
                    orc_rmsd_and_max(ref, heavy + (size_t)j * h * 3, h, &r, &m); /* :70 */
                    ++evals;
                    if (r < rmsd_thr && m < maxdev_thr) { /* :75-77 */
                        row_key[i] = b;
                        similar = 1;
                        break;
                    }
                }
                out[i] = (uint8_t)!similar; /* :113 */
            }
        }
        /* :204 cache.extend(computed_pairs) -- after the pass, in row order */
        int64_t new_keys = 0;
        for (int64_t i = 0; i < N; ++i)
            if (row_key[i] >= 0) {
                int64_t c = i / cs;
                if (c >= k) c = k - 1;
                key_a[n_keys] = c * cs;
                key_b[n_keys] = row_key[i];
                ++n_keys;
                ++new_keys;
            }
        int64_t after = 0;
        for (int64_t i = 0; i < N; ++i) after += out[i];
        memcpy(mask, out, (size_t)N);
        if (stats) {
            stats[n_passes].k = k;
            stats[n_passes].n_active_before = n_active;
            stats[n_passes].n_active_after = after;
            stats[n_passes].pairs_evaluated = evals;
            stats[n_passes].cache_hit_exits = hits;
            stats[n_passes].new_keys = new_keys;
#ifdef _OPENMP
            stats[n_passes].seconds = omp_get_wtime() - t0;
#else
            stats[n_passes].seconds = 0;
#endif
        }
        if (pass_masks) memcpy(pass_masks + (size_t)n_passes * N, mask, (size_t)N);
        ++n_passes;
    }
    memcpy(mask_out, mask, (size_t)N);
    if (n_passes_out) *n_passes_out = n_passes;
    if (keys_out)
        for (int64_t q = 0; q < n_keys; ++q) {
            keys_out[2 * q] = key_a[q];
            keys_out[2 * q + 1] = key_b[q];
        }
    if (n_keys_out) *n_keys_out = n_keys;
    free(mask);
    free(out);
    free(dbit);
    free(key_a);
    free(key_b);
    free(row_key);
    return 0;
}

/* Smallest margin of any pair evaluated by the reference-exact schedule to either threshold, and of any
 * inter-fragment distance to the clash threshold: the guard band of SURVEY.md 8(d).  Evaluates the same
 * pairs as orc_prune_rmsd(mode) and returns min |rmsd - thr| and min |maxdev - 2 thr| (the latter only over
 * pairs whose rmsd test passed, since the reference's `and` short-circuits on values, not on evaluation). */
ORC_API int orc_prune_margins(const double *heavy, int64_t N, int h, double rmsd_thr, int mode, double *min_rmsd_margin,
                              double *min_maxdev_margin) {
    const double maxdev_thr = 2 * rmsd_thr;
    uint8_t *mask = (uint8_t *)malloc((size_t)N), *out = (uint8_t *)malloc((size_t)N), *dbit = (uint8_t *)calloc((size_t)N + 1, 1);
    int64_t *key_a = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *key_b = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *row_key = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t n_keys = 0;
    memset(mask, 1, (size_t)N);
    double mr = INFINITY, mm = INFINITY;
    for (int ks = 0; ks < 18; ++ks) {
        int64_t k = (int64_t)ORC_KS[ks], n_active = 0;
        for (int64_t i = 0; i < N; ++i) n_active += mask[i];
        if (!(k == 1 || 20 * k < n_active)) continue;
        int64_t cs = N / k;
        memset(dbit, 0, (size_t)N + 1);
        if (mode == 0)
            for (int64_t q = 0; q < n_keys; ++q) {
                int64_t a = key_a[q], b = key_b[q];
                if (a % cs != 0 || a / cs >= k) continue;
                int64_t c = a / cs, last = (c == k - 1) ? N : cs * (c + 1);
                if (b < last) dbit[b] = 1;
            }
#pragma omp parallel for schedule(dynamic, 16) reduction(min : mr, mm)
        for (int64_t i = 0; i < N; ++i) {
            int64_t chunk = i / cs;
            if (chunk >= k) chunk = k - 1;
            int64_t first = chunk * cs, last = (chunk == k - 1) ? N : cs * (chunk + 1);
            row_key[i] = -1;
            if (!mask[i]) {
                out[i] = 0;
                continue;
            }
            int similar = 0;
            for (int64_t j = i + 1; j < last; ++j) {
                if (!mask[j]) continue;
                int64_t b = first + (j - i);
                if (dbit[b]) break;
                double r, m;
                orc_rmsd_and_max(heavy + (size_t)i * h * 3, heavy + (size_t)j * h * 3, h, &r, &m);
                double dr = fabs(r - rmsd_thr), dm = fabs(m - maxdev_thr);
                if (dr < mr) mr = dr;
                if (r < rmsd_thr && dm < mm) mm = dm;
                if (r < rmsd_thr && m < maxdev_thr) {
                    row_key[i] = b;
                    similar = 1;
                    break;
                }
            }
            out[i] = (uint8_t)!similar;
        }
        for (int64_t i = 0; i < N; ++i)
            if (row_key[i] >= 0) {
                int64_t c = i / cs;
                if (c >= k) c = k - 1;
                key_a[n_keys] = c * cs;
                key_b[n_keys++] = row_key[i];
            }
        memcpy(mask, out, (size_t)N);
    }
    *min_rmsd_margin = mr;
    *min_maxdev_margin = mm;
    free(mask);
    free(out);
    free(dbit);
    free(key_a);
    free(key_b);
    free(row_key);
    return 0;
}

/* min over poses and inter-fragment atom pairs of |d - thresh| (guard band for the clash mask) */
ORC_API double orc_clash_margin(const double *coords, int64_t n_poses, int n, const int64_t *ids, int n_ids, double thresh) {
    double best = INFINITY;
    int off[4] = {0, 0, 0, 0};
    for (int m = 0; m < n_ids; ++m) off[m + 1] = off[m] + (int)ids[m];
#pragma omp parallel for schedule(static) reduction(min : best)
    for (int64_t s = 0; s < n_poses; ++s) {
        const double *c = coords + (size_t)s * n * 3;
        for (int fa = 0; fa < n_ids; ++fa)
            for (int fb = fa + 1; fb < n_ids; ++fb)
                for (int i = off[fa]; i < off[fa + 1]; ++i)
                    for (int j = off[fb]; j < off[fb + 1]; ++j) {
                        double acc = 0;
                        for (int k = 0; k < 3; ++k) {
                            double d = c[3 * i + k] - c[3 * j + k];
                            acc += d * d;
                        }
                        double m = fabs(sqrt(acc) - thresh);
                        if (m < best) best = m;
                    }
    }
    return best;
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* One pass of rmsd_pruning.py:82-121 restricted to a subset of rows, for the multi-rank protocol tests.
 * Rows of a pass are independent (in_mask is read-only during a pass, :92,:101-113), so a rank may own
 * any subset of them.  The active rows are numbered 0..A-1 in index order ("compacted rank"); the row of
 * rank r belongs to this caller iff (r / tile_rows) % world == rank.  best[r] receives the compacted rank
 * of the first similar column the reference would stop at (:75-77), INT32_MAX if the row survives, and
 * is left untouched for rows of other ranks.  keys: the cache so far, as (first, first + delta) pairs. */
ORC_API int orc_prune_pass_rows(const double *heavy, int64_t N, int h, double rmsd_thr, int mode, const uint8_t *mask,
                                const int64_t *keys, int64_t n_keys, int64_t k, int rank, int world, int tile_rows, int32_t *best) {
    const double maxdev_thr = 2 * rmsd_thr;
    int64_t cs = N / k;
    uint8_t *dbit = (uint8_t *)calloc((size_t)N + 1, 1);
    int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * ((size_t)N + 1));
    int32_t run = 0;
    for (int64_t i = 0; i < N; ++i) {
        pos[i] = run;
        run += mask[i] != 0;
    }
    pos[N] = run;
    if (mode == 0)
        for (int64_t q = 0; q < n_keys; ++q) {
            int64_t a = keys[2 * q], b = keys[2 * q + 1];
            if (a % cs != 0 || a / cs >= k) continue;
            int64_t c = a / cs, last = (c == k - 1) ? N : cs * (c + 1);
            if (b < last) dbit[b] = 1;
        }
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < N; ++i) {
        if (!mask[i]) continue;
        int32_t r = pos[i];
        if ((r / tile_rows) % world != rank) continue;
        int64_t chunk = i / cs;
        if (chunk >= k) chunk = k - 1;
        int64_t first = chunk * cs, last = (chunk == k - 1) ? N : cs * (chunk + 1);
        int32_t found = INT32_MAX;
        for (int64_t j = i + 1; j < last; ++j) {
            if (!mask[j]) continue;
            if (dbit[first + (j - i)]) break;
            double rr, mm;
            orc_rmsd_and_max(heavy + (size_t)i * h * 3, heavy + (size_t)j * h * 3, h, &rr, &mm);
            if (rr < rmsd_thr && mm < maxdev_thr) {
                found = pos[j];
                break;
            }
        }
        best[r] = found;
    }
    free(dbit);
    free(pos);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* csearch dihedral rotations (torsion_module.py:463-500)                  */

/* utils.py:389-414  rotate_dihedral(coords, dihedral, angle, mask): IN PLACE.
 * axis = coords[i2] - coords[i3]; mat = rot_mat_from_pointer(axis, angle); center = coords[i3];
 * coords[mask] = (mat @ (coords[mask] - center).T).T + center */
ORC_API void orc_rotate_dihedral(double *coords, int n, const int32_t torsion[4], double angle_deg, const uint8_t *mask) {
    const int i2 = torsion[1], i3 = torsion[2];
    double axis[3], center[3], R[9];
    for (int k = 0; k < 3; ++k) axis[k] = coords[3 * i2 + k] - coords[3 * i3 + k], center[k] = coords[3 * i3 + k];
    orc_rot_mat_from_pointer(axis, angle_deg, R);
    for (int a = 0; a < n; ++a) {
        if (!mask[a]) continue;
        double v[3] = {coords[3 * a] - center[0], coords[3 * a + 1] - center[1], coords[3 * a + 2] - center[2]};
        for (int i = 0; i < 3; ++i) coords[3 * a + i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2] + center[i];
    }
}

/* numba_functions.py:26-47  torsion_comp_check(coords, torsion, mask, thresh, max_clashes) -> 1 / 0:
 * m1 = coords[mask]; m2 = coords[~mask minus {i2, i3}]; 0 if count(all_dists(m2, m1) < thresh) > max_clashes else 1.
 * min_margin (optional): smallest |d - thresh| met (guard band of the tests). */
ORC_API int orc_torsion_comp_check(const double *coords, int n, const int32_t torsion[4], const uint8_t *mask, double thresh,
                                   int64_t max_clashes, double *min_margin) {
    const int i2 = torsion[1], i3 = torsion[2];
    int64_t count = 0;
    for (int b = 0; b < n; ++b) {
        if (mask[b] || b == i2 || b == i3) continue;
        for (int a = 0; a < n; ++a) {
            if (!mask[a]) continue;
            double dx = coords[3 * b] - coords[3 * a], dy = coords[3 * b + 1] - coords[3 * a + 1], dz = coords[3 * b + 2] - coords[3 * a + 2];
            double d = sqrt(dx * dx + dy * dy + dz * dz); /* algebra.py:133-155 */
            if (d < thresh) ++count;
            if (min_margin && fabs(d - thresh) < *min_margin) *min_margin = fabs(d - thresh);
        }
    }
    return count > max_clashes ? 0 : 1;
}

/* torsion_module.py:463-500, the body of the loop over angle sets, for every candidate (the reference stops at n_out
 * accepted candidates; the caller takes the first n_out rows with rotated_bonds != 0):
 *   new_coords = copy(coords); rotated_bonds = 0
 *   for t, angle in enumerate(angle_set):
 *     if angle != 0:
 *        rotate_dihedral(new_coords, torsion, angle, mask)            (in place: temp_coords IS new_coords)
 *        if not torsion_comp_check(...):  for _ in range(angle // 5): rotate by -5; if check passes: rotated_bonds += 1; break
 *        else: rotated_bonds += 1
 * angle // 5 is Python floor division; a negative count runs no back-off step (and the clashing rotation stays).
 * out [m][n][3], rotated_bonds [m]; min_margin (optional, in/out) over every check made. */
ORC_API void orc_csearch_rotate(const double *coords, int n, const int32_t *torsions, const uint8_t *masks, int n_tors,
                                const int32_t *angles, int64_t n_cand, double thresh, int64_t max_clashes, double *out,
                                int32_t *rotated_bonds, double *min_margin) {
    double margin = min_margin ? *min_margin : 0.0;
#pragma omp parallel for schedule(dynamic, 8) reduction(min : margin)
    for (int64_t m = 0; m < n_cand; ++m) {
        double *c = out + (size_t)m * n * 3;
        memcpy(c, coords, sizeof(double) * (size_t)n * 3);
        int rotated = 0;
        double mm = margin;
        for (int t = 0; t < n_tors; ++t) {
            const int angle = angles[m * n_tors + t];
            if (angle == 0) continue;
            const int32_t *tor = torsions + 4 * t;
            const uint8_t *mask = masks + (size_t)t * n;
            orc_rotate_dihedral(c, n, tor, (double)angle, mask);
            if (!orc_torsion_comp_check(c, n, tor, mask, thresh, max_clashes, min_margin ? &mm : NULL)) {
                int steps = angle >= 0 ? angle / 5 : -((-angle + 4) / 5); /* floor division */
                for (int rep = 0; rep < steps; ++rep) {
                    orc_rotate_dihedral(c, n, tor, -5.0, mask);
                    if (orc_torsion_comp_check(c, n, tor, mask, thresh, max_clashes, min_margin ? &mm : NULL)) {
                        ++rotated;
                        break;
                    }
                }
            } else {
                ++rotated;
            }
        }
        rotated_bonds[m] = rotated;
        if (mm < margin) margin = mm;
    }
    if (min_margin) *min_margin = margin;
}

/* ------------------------------------------------------------------------------------------ */
/* pose parameters of the string embed (embeds.py:98-116)                   */
/* For site s (one conformer pair x one reactive-centre pair) and angle a:
 *   R0 = rotation_matrix_from_vectors(mol_vec, -ref_vec)              (:108)
 *   R  = rot_mat_from_pointer(ref_vec, angle) @ R0  if angle != 0     (:110-112)
 *   t  = p1 - R @ p2                                                  (:114)
 * pose index = s * n_angles + a; molecule 0 stays at identity / origin.
 * rot [n_sites * n_angles][2][9], pos [..][2][3], conf_idx [..][2]. */
ORC_API void orc_string_embed_params(const double *p1, const double *p2, const double *ref_vec, const double *mol_vec,
                                     const int32_t *conf_pair, int64_t n_sites, const double *angles, int n_angles, double *rot,
                                     double *pos, int32_t *conf_idx) {
    for (int64_t s = 0; s < n_sites; ++s) {
        const double neg_ref[3] = {-ref_vec[3 * s], -ref_vec[3 * s + 1], -ref_vec[3 * s + 2]};
        double R0[9];
        orc_rotation_matrix_from_vectors(mol_vec + 3 * s, neg_ref, R0);
        for (int a = 0; a < n_angles; ++a) {
            const int64_t q = s * n_angles + a;
            double R[9];
            if (angles[a] != 0) {
                double dR[9];
                orc_rot_mat_from_pointer(ref_vec + 3 * s, angles[a], dR);
                matmul3(dR, R0, R);
            } else {
                memcpy(R, R0, sizeof(R));
            }
            double *ro = rot + q * 18, *po = pos + q * 6;
            for (int i = 0; i < 9; ++i) ro[i] = (i % 4 == 0) ? 1.0 : 0.0, ro[9 + i] = R[i];
            for (int i = 0; i < 3; ++i) {
                po[i] = 0.0;
                po[3 + i] = p1[3 * s + i] - (R[3 * i] * p2[3 * s] + R[3 * i + 1] * p2[3 * s + 1] + R[3 * i + 2] * p2[3 * s + 2]);
            }
            conf_idx[2 * q] = conf_pair[2 * s], conf_idx[2 * q + 1] = conf_pair[2 * s + 1];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* torsion fingerprints and the pair search of prune_conformers_tfd         */

/* algebra.py:24-55 dihedral(p) in degrees (Praxeolitic formula) */
static double dihedral_deg(const double *p0, const double *p1, const double *p2, const double *p3) {
    double b0[3], b1[3], b2[3];
    for (int k = 0; k < 3; ++k) b0[k] = -1.0 * (p1[k] - p0[k]), b1[k] = p2[k] - p1[k], b2[k] = p3[k] - p2[k];
    const double n1 = norm_of3(b1);
    for (int k = 0; k < 3; ++k) b1[k] /= n1;
    const double d0 = b0[0] * b1[0] + b0[1] * b1[1] + b0[2] * b1[2], d2 = b2[0] * b1[0] + b2[1] * b1[1] + b2[2] * b1[2];
    double v[3], w[3];
    for (int k = 0; k < 3; ++k) v[k] = b0[k] - d0 * b1[k], w[k] = b2[k] - d2 * b1[k];
    const double x = v[0] * w[0] + v[1] * w[1] + v[2] * w[2];
    const double c[3] = {b1[1] * v[2] - b1[2] * v[1], b1[2] * v[0] - b1[0] * v[2], b1[0] * v[1] - b1[1] * v[0]};
    const double y = c[0] * w[0] + c[1] * w[1] + c[2] * w[2];
    return atan2(y, x) * (180.0 / M_PI);
}

/* numba_functions.py:233-240, :255-264  _get_tf_mat: float32 [N][T] */
ORC_API void orc_torsion_fingerprints(const double *coords, int64_t N, int n, const int32_t *quads, int T, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < N; ++s) {
        const double *c = coords + (size_t)s * n * 3;
        for (int t = 0; t < T; ++t) {
            const int32_t *q = quads + 4 * t;
            out[s * T + t] = (float)dihedral_deg(c + 3 * q[0], c + 3 * q[1], c + 3 * q[2], c + 3 * q[3]);
        }
    }
}

/* the same angles before the float32 store: lets a test measure how far each value is from a float32 rounding tie (a
 * fingerprint that differs in one float32 bit can flip a `sum < thresh` verdict when the sum sits exactly on the threshold,
 * as it does for poses that differ by one 10-degree step of the string embed) */
ORC_API void orc_torsion_angles_f64(const double *coords, int64_t N, int n, const int32_t *quads, int T, double *out) {
    for (int64_t s = 0; s < N; ++s) {
        const double *c = coords + (size_t)s * n * 3;
        for (int t = 0; t < T; ++t) {
            const int32_t *q = quads + 4 * t;
            out[s * T + t] = dihedral_deg(c + 3 * q[0], c + 3 * q[1], c + 3 * q[2], c + 3 * q[3]);
        }
    }
}

/* numba_functions.py:242-253 tfd_similarity: deltas = |tfp1 - tfp2| (float32); deltas = |deltas - (deltas > 180) * 360|
 * (float64 after the integer term); True iff sum(deltas) < thresh.  *sum_out (optional) = the sum. */
ORC_API int orc_tfd_similarity(const float *a, const float *b, int T, double thresh, double *sum_out) {
    double sum = 0.0;
    for (int t = 0; t < T; ++t) {
        const float d32 = fabsf(a[t] - b[t]);
        double d = (double)d32;
        if (d32 > 180.0f) d -= 360.0;
        sum += fabs(d);
    }
    if (sum_out) *sum_out = sum;
    return sum < thresh;
}

/* numba_functions.py:171-199, the pair search of one pass: chunk `step` covers [d*step, d*(step+1)), the last one
 * [d*(k-1), num_active) (:175-178 -- yes, the count of active structures as an index); inside a chunk row i scans j > i
 * (removed structures included: the mask is never consulted) and stops at the first similar one.  The reference's cache
 * only skips pairs it already found dissimilar, so it cannot change a verdict and is not needed here.
 * first[i] = absolute index of the first similar j, -1 if none or if i lies in no chunk.  min_margin (optional, in/out). */
ORC_API void orc_tfd_first_similar(const float *tf, int64_t N, int T, int64_t d, int64_t k, int64_t num_active, double thresh,
                                   int32_t *first, double *min_margin) {
    double margin = min_margin ? *min_margin : 0.0;
    for (int64_t i = 0; i < N; ++i) first[i] = -1;
#pragma omp parallel for schedule(dynamic, 1) reduction(min : margin)
    for (int64_t step = 0; step < k; ++step) {
        const int64_t start = d * step;
        int64_t len = (step == k - 1) ? num_active - start : d;
        if (len < 0) len = 0;
        for (int64_t i = 0; i < len; ++i)
            for (int64_t j = i + 1; j < len; ++j) {
                double sum;
                const int sim = orc_tfd_similarity(tf + (start + i) * T, tf + (start + j) * T, T, thresh, &sum);
                if (min_margin && fabs(sum - thresh) < margin) margin = fabs(sum - thresh);
                if (sim) {
                    first[start + i] = (int32_t)(start + j);
                    break;
                }
            }
    }
    if (min_margin) *min_margin = margin;
}

/* embeds.py:47-69 is_new_structure over an ordered list of fingerprints: structure s is accepted iff tfd_similarity(tfp_s,
 * ref) is False for every fingerprint accepted before it (:58-60); an accepted fingerprint joins the list (:63) and is never
 * dropped: `lru_cache = lru_cache[1:]` (:66-67) rebinds the local name only, the caller's list keeps growing.
 * accepted u8[N]; min_margin (optional, in/out) = smallest |sum - thresh| met. */
ORC_API int64_t orc_tfd_greedy_filter(const float *tf, int64_t N, int T, double thresh, uint8_t *accepted, double *min_margin) {
    int64_t *kept = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1)), nk = 0;
    double margin = min_margin ? *min_margin : 0.0;
    for (int64_t s = 0; s < N; ++s) {
        int is_new = 1;
        for (int64_t k = 0; k < nk; ++k) {
            double sum;
            const int sim = orc_tfd_similarity(tf + s * T, tf + kept[k] * T, T, thresh, &sum);
            if (min_margin && fabs(sum - thresh) < margin) margin = fabs(sum - thresh);
            if (sim) {
                is_new = 0;
                break;
            }
        }
        accepted[s] = (uint8_t)is_new;
        if (is_new) kept[nk++] = s;
    }
    free(kept);
    if (min_margin) *min_margin = margin;
    return nk;
}

/* utils.py:210-261 polygonize.  n = 2: out f64[2][2][2][3] (two orientations of two centred superposed segments);
 * n = 3: out f64[8][3][2][3] (the eight orientation patterns of a triangle's sides).  Returns 0, or -1 where the reference
 * raises TriangleError (:236-237). */
ORC_API int orc_polygonize(const double *lengths, int n, double *out) {
    double arr[3][2][3];
    memset(arr, 0, sizeof(arr));
    if (n == 2) {
        arr[0][0][0] = -lengths[0] / 2, arr[0][1][0] = +lengths[0] / 2;
        arr[1][0][0] = -lengths[1] / 2, arr[1][1][0] = +lengths[1] / 2;
        for (int t = 0; t < 2; ++t) memcpy(out + t * 12, arr, sizeof(double) * 12);
        for (int i = 0; i < 6; ++i) out[12 + 6 + i] *= -1;      /* vertices_out[1,1] *= -1 */
        return 0;
    }
    for (int i = 0; i < 3; ++i)
        if (!(lengths[i] < lengths[(i + 2) % 3] + lengths[(i + 1) % 3])) return -1;
    arr[0][1][0] = lengths[0], arr[1][0][0] = lengths[0];
    const double a = lengths[0] * lengths[0], b = lengths[1] * lengths[1], c = lengths[2] * lengths[2];
    const double x = (a - b + c) / (2 * sqrt(a)), y = sqrt(c - x * x);
    arr[1][1][0] = x, arr[1][1][1] = y, arr[2][0][0] = x, arr[2][0][1] = y;
    for (int t = 0; t < 8; ++t) memcpy(out + t * 18, arr, sizeof(double) * 18);
    static const int swaps[12][2] = {{1, 2}, {2, 1}, {3, 1}, {3, 2}, {4, 0}, {5, 0}, {5, 1}, {6, 0}, {6, 2}, {7, 0}, {7, 1}, {7, 2}};
    for (int q = 0; q < 12; ++q) {
        double *side = out + swaps[q][0] * 18 + swaps[q][1] * 6;
        for (int i = 0; i < 3; ++i) {
            const double tmp = side[i];
            side[i] = side[3 + i], side[3 + i] = tmp;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* pose parameters of the cyclical embed (embeds.py:676-713), one (pose, molecule) per row:
 *   atomic_pivot_mean = mean(reactive_coords)                                   (1 or 2 reactive atoms)
 *   mol_direction = meanpoint - atomic_pivot_mean, or meanpoint if that is exactly zero   (:678-681)
 *   A = align_vec_pair([end - start, direction], [pivot, mol_direction])        (:691-692)
 *   axis = A @ (r0 - r1) for two reactive atoms, else A @ pivot                 (:697-700)
 *   S = rot_mat_from_pointer(axis, angle); centre = A @ atomic_pivot_mean       (:704-708)
 *   rotation = S @ A; position = centre - S @ centre + mean(start, end) - A @ meanpoint   (:711-714)
 * in: start, end, direction, pivot, meanpoint, r0, r1 f64[n][3]; n_reactive i32[n]; angle f64[n]; out rot [n][9], pos [n][3]. */
ORC_API void orc_cyclical_embed_params(const double *start, const double *end, const double *direction, const double *pivot,
                                       const double *meanpoint, const double *r0, const double *r1, const int32_t *n_reactive,
                                       const double *angle, int64_t n, double *rot, double *pos) {
    for (int64_t q = 0; q < n; ++q) {
        const double *st = start + 3 * q, *en = end + 3 * q, *dir = direction + 3 * q, *pv = pivot + 3 * q, *mp = meanpoint + 3 * q;
        const double *a0 = r0 + 3 * q, *a1 = r1 + 3 * q;
        double apm[3], md[3], ref[6], tgt[6], A[9], axis_in[3], axis[3], S[9], centre[3];
        int zero = 1;
        for (int i = 0; i < 3; ++i) {
            apm[i] = n_reactive[q] == 2 ? (a0[i] + a1[i]) / 2.0 : a0[i];
            md[i] = mp[i] - apm[i];
            if (md[i] != 0.) zero = 0;
        }
        if (zero)
            for (int i = 0; i < 3; ++i) md[i] = mp[i];
        for (int i = 0; i < 3; ++i) ref[i] = en[i] - st[i], ref[3 + i] = dir[i], tgt[i] = pv[i], tgt[3 + i] = md[i];
        orc_align_vec_pair(ref, tgt, A);
        for (int i = 0; i < 3; ++i) axis_in[i] = n_reactive[q] == 2 ? a0[i] - a1[i] : pv[i];
        for (int i = 0; i < 3; ++i) {
            axis[i] = A[3 * i] * axis_in[0] + A[3 * i + 1] * axis_in[1] + A[3 * i + 2] * axis_in[2];
            centre[i] = A[3 * i] * apm[0] + A[3 * i + 1] * apm[1] + A[3 * i + 2] * apm[2];
        }
        orc_rot_mat_from_pointer(axis, angle[q], S);
        matmul3(S, A, rot + 9 * q);
        for (int i = 0; i < 3; ++i) {
            const double s_c = S[3 * i] * centre[0] + S[3 * i + 1] * centre[1] + S[3 * i + 2] * centre[2];
            const double a_m = A[3 * i] * mp[0] + A[3 * i + 1] * mp[1] + A[3 * i + 2] * mp[2];
            pos[3 * q + i] = centre[i] - s_c + ((st[i] + en[i]) / 2.0 - a_m);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* moments of inertia (algebra.py:165-213) and embed scores                 */

/* eigenvalues of a symmetric 3x3 by cyclic Jacobi, sorted by absolute value (algebra.py:207-212 diagonalize:
 * np.linalg.eig, columns ordered by |eigenvalue|, diag(B^-1 A B)) */
static void sym3_eigvals_by_abs(double A[9], double out[3]) {
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5], dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (!(off > 1e-32 * dia)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = A[3 * p + q];
                if (apq == 0.0) continue;
                double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
                double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq, A[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk, A[3 * q + k] = s * apk + c * aqk;
                }
            }
    }
    double e[3] = {A[0], A[4], A[8]};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (fabs(e[j]) > fabs(e[j + 1])) {
                double t = e[j];
                e[j] = e[j + 1], e[j + 1] = t;
            }
    out[0] = e[0], out[1] = e[1], out[2] = e[2];
}

/* algebra.py:165-186 get_inertia_moments for every structure: out [N][3] */
ORC_API void orc_inertia_moments(const double *structures, int64_t N, int n, const double *masses, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < N; ++s) {
        const double *c = structures + (size_t)s * n * 3;
        double tot = 0, com[3] = {0, 0, 0}, I[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < n; ++a) {
            tot += masses[a];
            for (int k = 0; k < 3; ++k) com[k] += c[3 * a + k] * masses[a];
        }
        for (int k = 0; k < 3; ++k) com[k] /= tot;
        for (int a = 0; a < n; ++a) {
            double x[3] = {c[3 * a] - com[0], c[3 * a + 1] - com[1], c[3 * a + 2] - com[2]};
            double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) I[3 * i + j] += masses[a] * ((i == j ? r2 : 0.0) - x[i] * x[j]);
        }
        sym3_eigvals_by_abs(I, out + 3 * s);
    }
}

/* algebra.py:188-205 get_moi_similarity_matches: first[i] = first j > i with all(|im_i - im_j| / im_i < max_deviation), -1 */
ORC_API void orc_moi_first_similar(const double *moments, int64_t N, double max_deviation, int32_t *first, double *min_margin) {
    double margin = min_margin ? *min_margin : 0.0;
#pragma omp parallel for schedule(dynamic, 16) reduction(min : margin)
    for (int64_t i = 0; i < N; ++i) {
        first[i] = -1;
        for (int64_t j = i + 1; j < N; ++j) {
            int all = 1;
            for (int k = 0; k < 3; ++k) {
                double rel = fabs(moments[3 * i + k] - moments[3 * j + k]) / moments[3 * i + k];
                if (min_margin && fabs(rel - max_deviation) < margin) margin = fabs(rel - max_deviation);
                if (!(rel < max_deviation)) all = 0;
            }
            if (all) {
                first[i] = (int32_t)j;
                break;
            }
        }
    }
    if (min_margin) *min_margin = margin;
}

/* numba_functions.py:273-288 _score_embed_poses (float32 accumulator) and optimization_methods.py:544-557 fitness_check's
 * signed error (float64; a NaN target = None is skipped): indices i32[N][n_c][2], distances f64[N][n_c] */
ORC_API void orc_embed_scores(const double *structures, int64_t N, int n, const int32_t *indices, const double *distances, int n_c,
                              float *scores, double *fitness_error) {
    for (int64_t s = 0; s < N; ++s) {
        const double *c = structures + (size_t)s * n * 3;
        float sc = 0.0f;
        double err = 0.0;
        for (int i = 0; i < n_c; ++i) {
            const int a = indices[(s * n_c + i) * 2], b = indices[(s * n_c + i) * 2 + 1];
            const double d[3] = {c[3 * a] - c[3 * b], c[3 * a + 1] - c[3 * b + 1], c[3 * a + 2] - c[3 * b + 2]};
            const double dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), target = distances[s * n_c + i];
            if (target == target) {
                sc = (float)((double)sc + fabs(dist - target));
                err += dist - target;
            }
        }
        if (scores) scores[s] = sc;
        if (fitness_error) fitness_error[s] = err;
    }
}
