// csearch.hpp -- SURVEY.md 8(f) N3: the dihedral rotations of the conformational search, batched.
//
// tscode/torsion_module.py:463-500 builds every candidate conformer from the same start structure: for each torsion with
// a non-zero angle, rotate_dihedral (utils.py:389-414: the masked atoms turn about the i2-i3 bond) followed by
// torsion_comp_check (numba_functions.py:26-47: no atom of the moved side within `thresh` of an atom of the other side,
// the bond atoms i2, i3 aside); a clashing rotation is walked back in 5-degree steps until it stops clashing
// (`for _ in range(angle // 5)`, Python floor division, so a negative angle is never walked back).  Candidates are
// independent; the torsions of one candidate are not.  One wavefront owns one candidate: its coordinates live in LDS,
// lanes are atoms, the torsions run in order with the data-dependent back-off loop inside.
#pragma once
#include "common.hpp"
#include "rmsd.hpp"

namespace tsc {

struct CsearchArgs {
    int n;       // atoms
    int n_tors;
    long long n_cand;
    double sq_bound;  // d < thresh  <=>  d2 < sq_bound  (clash_sq_bound, embed_clash.hpp)
    long long max_clashes;
};

// algebra.py:325-344 rot_mat_from_pointer(axis, angle_deg) via algebra.py:284-323 (quaternion, scalar last)
__device__ inline void rot_mat_from_pointer_dev(const double ax[3], double angle_deg, double R[9]) {
    const double nrm = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // algebra.py:89-96
    const double u0 = ax[0] / nrm, u1 = ax[1] / nrm, u2 = ax[2] / nrm;
    const double half = angle_deg * (3.14159265358979323846 / 180) / 2;
    const double s = sin(half), q0 = cos(half);
    const double q1 = s * u0, q2 = s * u1, q3 = s * u2;
    R[0] = 2 * (q0 * q0 + q1 * q1) - 1, R[1] = 2 * (q1 * q2 - q0 * q3), R[2] = 2 * (q1 * q3 + q0 * q2);
    R[3] = 2 * (q1 * q2 + q0 * q3), R[4] = 2 * (q0 * q0 + q2 * q2) - 1, R[5] = 2 * (q2 * q3 - q0 * q1);
    R[6] = 2 * (q1 * q3 - q0 * q2), R[7] = 2 * (q2 * q3 + q0 * q1), R[8] = 2 * (q0 * q0 + q3 * q3) - 1;
}

// utils.py:389-414 on a structure in LDS: every lane forms the (wave-uniform) matrix, lanes with a masked atom apply it.
// If i3 itself is masked it maps onto itself exactly (its offset from the centre is zero), so no lane reads a value that
// another lane is changing.
__device__ inline void rotate_dihedral_lds(double *c, int n, int i2, int i3, double angle_deg, const uint8_t *__restrict__ mask, int lane) {
    const double ax[3] = {c[3 * i2] - c[3 * i3], c[3 * i2 + 1] - c[3 * i3 + 1], c[3 * i2 + 2] - c[3 * i3 + 2]};
    const double cen[3] = {c[3 * i3], c[3 * i3 + 1], c[3 * i3 + 2]};
    double R[9];
    rot_mat_from_pointer_dev(ax, angle_deg, R);
    __builtin_amdgcn_wave_barrier();
    for (int a = lane; a < n; a += 64) {
        if (!mask[a]) continue;
        const double v0 = c[3 * a] - cen[0], v1 = c[3 * a + 1] - cen[1], v2 = c[3 * a + 2] - cen[2];
        c[3 * a] = R[0] * v0 + R[1] * v1 + R[2] * v2 + cen[0];
        c[3 * a + 1] = R[3] * v0 + R[4] * v1 + R[5] * v2 + cen[1];
        c[3 * a + 2] = R[6] * v0 + R[7] * v1 + R[8] * v2 + cen[2];
    }
    __builtin_amdgcn_wave_barrier();
}

// numba_functions.py:26-47 on a structure in LDS (wave-uniform result): 1 = passes
__device__ inline int torsion_comp_check_lds(const double *c, int n, int i2, int i3, const uint8_t *__restrict__ mask, double sq_bound,
                                             long long max_clashes, int lane) {
    int cnt = 0;
    for (int a = lane; a < n; a += 64) {
        if (!mask[a]) continue;
        const double x = c[3 * a], y = c[3 * a + 1], z = c[3 * a + 2];
        for (int b = 0; b < n; ++b) {
            if (mask[b] || b == i2 || b == i3) continue;
            const double dx = c[3 * b] - x, dy = c[3 * b + 1] - y, dz = c[3 * b + 2] - z;
            cnt += (dx * dx + dy * dy + dz * dz < sq_bound) ? 1 : 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    return (long long)cnt > max_clashes ? 0 : 1;
}

// out [n_cand][n][3], rotated_bonds [n_cand]; angles [n_cand][n_tors] int32 degrees; masks [n_tors][n]; torsions [n_tors][4]
__global__ __launch_bounds__(256) void k_csearch_rotate(CsearchArgs a, const double *__restrict__ base, const int32_t *__restrict__ tors,
                                                         const uint8_t *__restrict__ masks, const int32_t *__restrict__ angles,
                                                         double *__restrict__ out, int32_t *__restrict__ rotated_bonds) {
    extern __shared__ __attribute__((aligned(16))) double s_c[];  // [4][n * 3]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = a.n;
    double *c = s_c + size_t(wid) * n * 3;
    for (int64_t m = int64_t(blockIdx.x) * 4 + wid; m < a.n_cand; m += int64_t(gridDim.x) * 4) {
        for (int e = lane; e < n * 3; e += 64) c[e] = base[e];  // new_coords = np.copy(coords), :473
        __builtin_amdgcn_wave_barrier();
        int rotated = 0;
        for (int t = 0; t < a.n_tors; ++t) {
            const int angle = angles[m * a.n_tors + t];
            if (angle == 0) continue;  // :482
            const int i2 = tors[4 * t + 1], i3 = tors[4 * t + 2];
            const uint8_t *mask = masks + size_t(t) * n;
            rotate_dihedral_lds(c, n, i2, i3, double(angle), mask, lane);  // :484
            if (!torsion_comp_check_lds(c, n, i2, i3, mask, a.sq_bound, a.max_clashes, lane)) {  // :487
                const int steps = angle >= 0 ? angle / 5 : -((-angle + 4) / 5);  // angle // 5
                for (int rep = 0; rep < steps; ++rep) {  // :490-498
                    rotate_dihedral_lds(c, n, i2, i3, -5.0, mask, lane);
                    if (torsion_comp_check_lds(c, n, i2, i3, mask, a.sq_bound, a.max_clashes, lane)) {
                        ++rotated;
                        break;
                    }
                }
            } else {
                ++rotated;  // :501
            }
        }
        double *o = out + m * n * 3;
        for (int e = lane; e < n * 3; e += 64) o[e] = c[e];
        if (lane == 0) rotated_bonds[m] = rotated;
        __builtin_amdgcn_wave_barrier();
    }
}

// torsion_comp_check for a batch of structures sharing torsion and mask: ok[s] = 1 / 0
__global__ __launch_bounds__(256) void k_torsion_comp_check(CsearchArgs a, const double *__restrict__ coords, const int32_t *__restrict__ tors,
                                                             const uint8_t *__restrict__ mask, int32_t *__restrict__ ok) {
    extern __shared__ __attribute__((aligned(16))) double s_c[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = a.n;
    double *c = s_c + size_t(wid) * n * 3;
    for (int64_t m = int64_t(blockIdx.x) * 4 + wid; m < a.n_cand; m += int64_t(gridDim.x) * 4) {
        const double *src = coords + m * n * 3;
        for (int e = lane; e < n * 3; e += 64) c[e] = src[e];
        __builtin_amdgcn_wave_barrier();
        const int r = torsion_comp_check_lds(c, n, tors[1], tors[2], mask, a.sq_bound, a.max_clashes, lane);
        if (lane == 0) ok[m] = r;
        __builtin_amdgcn_wave_barrier();
    }
}


// ---------------------------------------------------------------------------------------------------
// pose parameters of the string embed (tscode/embeds.py:98-116), one thread per pose.
//   R0 = rotation_matrix_from_vectors(mol_vec, -ref_vec)   (utils.py:183-208: Rodrigues, exact-zero tests kept)
//   R  = rot_mat_from_pointer(ref_vec, angle) @ R0 for angle != 0;   t = p1 - R @ p2
// pose = site * n_angles + a; molecule 0 stays at identity.  rot [N][2][9], pos [N][2][3], conf_idx [N][2].
__device__ inline void rotation_matrix_from_vectors_dev(const double v1[3], const double v2[3], double out[9]) {
    // no fused multiply-add here: the reference tests the cross product of the normalised vectors against exact zero, and
    // for (anti)parallel inputs whether that test fires depends on the last bit of each product (NumPy multiplies, then
    // subtracts)
#pragma clang fp contract(off)
    const double n1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]), n2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    const double a[3] = {v1[0] / n1, v1[1] / n1, v1[2] / n1}, b[3] = {v2[0] / n2, v2[1] / n2, v2[2] / n2};
    const double v[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double s = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (s != 0) {
        const double c = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        const double k[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
        const double f = (1 - c) / (s * s);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double kk = k[3 * i] * k[j] + k[3 * i + 1] * k[3 + j] + k[3 * i + 2] * k[6 + j];
                out[3 * i + j] = ((i == j) ? 1.0 : 0.0) + k[3 * i + j] + kk * f;
            }
        return;
    }
    const double ab[3] = {a[0] + b[0], a[1] + b[1], a[2] + b[2]};
    if (sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]) == 0) {  // antiparallel: half a turn about z
        const double z[3] = {0, 0, 1};
        rot_mat_from_pointer_dev(z, 180.0, out);
        return;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] = (i % 4 == 0) ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void k_string_embed_params(const double *__restrict__ p1, const double *__restrict__ p2,
                                                              const double *__restrict__ ref_vec, const double *__restrict__ mol_vec,
                                                              const int32_t *__restrict__ conf_pair, int64_t n_sites,
                                                              const double *__restrict__ angles, int n_angles, double *__restrict__ rot,
                                                              double *__restrict__ pos, int32_t *__restrict__ conf_idx) {
    const int64_t total = n_sites * n_angles;
    for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int64_t s = q / n_angles;
        const double angle = angles[q - s * n_angles];
        const double rv[3] = {ref_vec[3 * s], ref_vec[3 * s + 1], ref_vec[3 * s + 2]};
        const double neg[3] = {-rv[0], -rv[1], -rv[2]};
        const double mv[3] = {mol_vec[3 * s], mol_vec[3 * s + 1], mol_vec[3 * s + 2]};
        double R0[9], R[9];
        rotation_matrix_from_vectors_dev(mv, neg, R0);
        if (angle != 0) {
            double dR[9];
            rot_mat_from_pointer_dev(rv, angle, dR);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) R[3 * i + j] = dR[3 * i] * R0[j] + dR[3 * i + 1] * R0[3 + j] + dR[3 * i + 2] * R0[6 + j];
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = R0[i];
        }
        double *ro = rot + q * 18, *po = pos + q * 6;
#pragma unroll
        for (int i = 0; i < 9; ++i) ro[i] = (i % 4 == 0) ? 1.0 : 0.0, ro[9 + i] = R[i];
        const double x = p2[3 * s], y = p2[3 * s + 1], z = p2[3 * s + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            po[i] = 0.0;
            po[3 + i] = p1[3 * s + i] - (R[3 * i] * x + R[3 * i + 1] * y + R[3 * i + 2] * z);
        }
        conf_idx[2 * q] = conf_pair[2 * s], conf_idx[2 * q + 1] = conf_pair[2 * s + 1];
    }
}

// ---------------------------------------------------------------------------------------------------
// pose parameters of the cyclical embed (tscode/embeds.py:676-713), one thread per (pose, molecule).
// align_vec_pair (algebra.py:258-282: SVD of B = sum_j ref_j tgt_j^T, improper-rotation fix, U V^T) is the proper rotation
// R that best maps tgt_j onto ref_j; here it is taken as the top eigenvector of Horn's quaternion matrix of
// S = sum_j tgt_j ref_j^T (cyclic Jacobi, top_eigvec4 of rmsd.hpp) -- the same rotation wherever it is unique.
__device__ inline void align_vec_pair_dev(const double ref0[3], const double ref1[3], const double tgt0[3], const double tgt1[3], double R[9]) {
    double S[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) S[3 * a + b] = tgt0[a] * ref0[b] + tgt1[a] * ref1[b];
    double N[4][4];
    N[0][0] = S[0] + S[4] + S[8];
    N[1][1] = S[0] - S[4] - S[8];
    N[2][2] = -S[0] + S[4] - S[8];
    N[3][3] = -S[0] - S[4] + S[8];
    N[0][1] = N[1][0] = S[5] - S[7];
    N[0][2] = N[2][0] = S[6] - S[2];
    N[0][3] = N[3][0] = S[1] - S[3];
    N[1][2] = N[2][1] = S[1] + S[3];
    N[1][3] = N[3][1] = S[6] + S[2];
    N[2][3] = N[3][2] = S[5] + S[7];
    double e[4];
    top_eigvec4(N, e);
    const double nn = 1.0 / sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
    const double w = e[0] * nn, x = e[1] * nn, y = e[2] * nn, z = e[3] * nn;
    R[0] = w * w + x * x - y * y - z * z, R[1] = 2 * (x * y - w * z), R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z), R[4] = w * w - x * x + y * y - z * z, R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y), R[7] = 2 * (y * z + w * x), R[8] = w * w - x * x - y * y + z * z;
}

__global__ __launch_bounds__(256) void k_cyclical_embed_params(const double *__restrict__ start, const double *__restrict__ end,
                                                                const double *__restrict__ direction, const double *__restrict__ pivot,
                                                                const double *__restrict__ meanpoint, const double *__restrict__ r0,
                                                                const double *__restrict__ r1, const int32_t *__restrict__ n_reactive,
                                                                const double *__restrict__ angle, int64_t n, double *__restrict__ rot,
                                                                double *__restrict__ pos) {
    for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < n; q += int64_t(gridDim.x) * blockDim.x) {
        double st[3], en[3], dir[3], pv[3], mp[3], a0[3], a1[3], apm[3], md[3], ref0[3];
        const bool two = n_reactive[q] == 2;
        bool zero = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            st[i] = start[3 * q + i], en[i] = end[3 * q + i], dir[i] = direction[3 * q + i], pv[i] = pivot[3 * q + i], mp[i] = meanpoint[3 * q + i];
            a0[i] = r0[3 * q + i], a1[i] = r1[3 * q + i];
            apm[i] = two ? (a0[i] + a1[i]) / 2.0 : a0[i];  // np.mean(reactive_coords, axis=0), :673
            md[i] = mp[i] - apm[i];                         // :676
            zero = zero && md[i] == 0.0;
            ref0[i] = en[i] - st[i];
        }
        if (zero) {  // :677-678
#pragma unroll
            for (int i = 0; i < 3; ++i) md[i] = mp[i];
        }
        double A[9], S[9], axis[3], centre[3];
        align_vec_pair_dev(ref0, dir, pv, md, A);  // :691-692
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double v0 = two ? a0[0] - a1[0] : pv[0], v1 = two ? a0[1] - a1[1] : pv[1], v2 = two ? a0[2] - a1[2] : pv[2];
            axis[i] = A[3 * i] * v0 + A[3 * i + 1] * v1 + A[3 * i + 2] * v2;                  // :697-700
            centre[i] = A[3 * i] * apm[0] + A[3 * i + 1] * apm[1] + A[3 * i + 2] * apm[2];    // :708
        }
        rot_mat_from_pointer_dev(axis, angle[q], S);  // :704
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) rot[9 * q + 3 * i + j] = S[3 * i] * A[j] + S[3 * i + 1] * A[3 + j] + S[3 * i + 2] * A[6 + j];  // :711
            const double s_c = S[3 * i] * centre[0] + S[3 * i + 1] * centre[1] + S[3 * i + 2] * centre[2];
            const double a_m = A[3 * i] * mp[0] + A[3 * i + 1] * mp[1] + A[3 * i + 2] * mp[2];
            pos[3 * q + i] = centre[i] - s_c + ((st[i] + en[i]) / 2.0 - a_m);  // :713-714
        }
    }
}

}  // namespace tsc
