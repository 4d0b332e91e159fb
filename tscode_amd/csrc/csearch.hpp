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
#include "embed_clash.hpp"
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

// Per torsion the atoms split into the side that turns (`moved`: mask set) and the side it is checked against (`fixed`:
// mask clear, the bond atoms i2, i3 aside -- numba_functions.py:36-39).  A workgroup compacts both index lists of every
// torsion into LDS once; the kernels then walk lists, not masks.
struct TorsionLists {
    uint16_t *moved, *fixed;  // [n_tors][n]
    int *count;               // [n_tors][4] = (n_moved, n_fixed, whether i2 is among the moved atoms, -)
};

__host__ __device__ inline size_t torsion_lists_bytes(int n_tors, int n) {
    return (size_t(n_tors) * n * 2 * sizeof(uint16_t) + size_t(n_tors) * 4 * sizeof(int) + 15) & ~size_t(15);
}

// LDS of one wavefront: the structure in fp64, the fixed side of the current torsion as three packed fp32 arrays, and the matrix
// and centre of a walk-back step (12 doubles: wave-uniform values that would otherwise sit in 24 vector registers)
__host__ __device__ inline size_t csearch_wave_bytes(int n) {
    return size_t(n) * 3 * sizeof(double) + size_t(3) * ((n + 2) & ~1) * sizeof(float) + 12 * sizeof(double);
}

__device__ inline TorsionLists torsion_lists_at(void *lds, int n_tors, int n) {
    TorsionLists L;
    L.moved = static_cast<uint16_t *>(lds);
    L.fixed = L.moved + size_t(n_tors) * n;
    L.count = reinterpret_cast<int *>(L.fixed + size_t(n_tors) * n);
    return L;
}

__device__ inline void build_torsion_lists(TorsionLists L, const uint8_t *__restrict__ masks, const int32_t *__restrict__ tors, int n_tors, int n) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int t = wid; t < n_tors; t += nw) {
        const uint8_t *mask = masks + size_t(t) * n;
        const int i2 = tors[4 * t + 1], i3 = tors[4 * t + 2];
        int nm = 0, nf = 0;
        for (int a0 = 0; a0 < n; a0 += 64) {
            const int a = a0 + lane;
            const bool mv = a < n && mask[a] != 0;
            const bool fx = a < n && !mv && a != i2 && a != i3;
            const unsigned long long bm = __ballot(mv), bf = __ballot(fx), below = (1ull << lane) - 1ull;
            if (mv) L.moved[size_t(t) * n + nm + __popcll(bm & below)] = uint16_t(a);
            if (fx) L.fixed[size_t(t) * n + nf + __popcll(bf & below)] = uint16_t(a);
            nm += __popcll(bm), nf += __popcll(bf);
        }
        if (lane == 0) L.count[4 * t] = nm, L.count[4 * t + 1] = nf, L.count[4 * t + 2] = mask[i2] != 0 ? 1 : 0, L.count[4 * t + 3] = 0;
    }
    __syncthreads();
}

// utils.py:389-414 on a structure in LDS: every lane forms the (wave-uniform) matrix, lanes apply it to the moved atoms.
// If i3 itself is masked it maps onto itself exactly (its offset from the centre is zero), so no lane reads a value that
// another lane is changing.
__device__ inline void apply_rotation_lds(double *c, const double R[9], const double cen[3], const uint16_t *moved, int nm, int lane) {
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < nm; r += 64) {
        const int a = moved[r];
        const double v0 = c[3 * a] - cen[0], v1 = c[3 * a + 1] - cen[1], v2 = c[3 * a + 2] - cen[2];
        c[3 * a] = R[0] * v0 + R[1] * v1 + R[2] * v2 + cen[0];
        c[3 * a + 1] = R[3] * v0 + R[4] * v1 + R[5] * v2 + cen[1];
        c[3 * a + 2] = R[6] * v0 + R[7] * v1 + R[8] * v2 + cen[2];
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ inline void dihedral_rotation(const double *c, int i2, int i3, double angle_deg, double R[9], double cen[3]) {
    const double ax[3] = {c[3 * i2] - c[3 * i3], c[3 * i2 + 1] - c[3 * i3 + 1], c[3 * i2 + 2] - c[3 * i3 + 2]};
    cen[0] = c[3 * i3], cen[1] = c[3 * i3 + 1], cen[2] = c[3 * i3 + 2];
    rot_mat_from_pointer_dev(ax, angle_deg, R);
}

__device__ inline void rotate_dihedral_lds(double *c, int i2, int i3, double angle_deg, const uint16_t *moved, int nm, int lane) {
    double R[9], cen[3];
    dihedral_rotation(c, i2, i3, angle_deg, R, cen);
    apply_rotation_lds(c, R, cen, moved, nm, lane);
}

// The fixed side of a torsion as packed fp32 (F = X[npad] Y[npad] Z[npad], an odd tail padded far away); returns the largest
// |coordinate| among the fixed atoms (wave-uniform).  The fixed atoms do not move while the torsion is walked back.
__device__ inline double stage_fixed_f32(const double *c, const uint16_t *fixed, int nf, float *F, int npad, int lane) {
    double cmax = 0.0;
    for (int j = lane; j < nf; j += 64) {
        const int b = fixed[j];
        const double x = c[3 * b], y = c[3 * b + 1], z = c[3 * b + 2];
        F[j] = float(x), F[npad + j] = float(y), F[2 * npad + j] = float(z);
        cmax = fmax(cmax, fmax(fabs(x), fmax(fabs(y), fabs(z))));
    }
    if (lane == 0 && (nf & 1)) F[nf] = F[npad + nf] = F[2 * npad + nf] = 1.0e18f;
    for (int off = 32; off > 0; off >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, off));
    __builtin_amdgcn_wave_barrier();
    return cmax;
}

// numba_functions.py:26-47 on a structure in LDS (wave-uniform result): 1 = passes.  Lanes are laid out as A moved atoms x
// G slices of the fixed list (A the power of two covering min(n_moved, 64)), so that a small rotating group still fills
// the wavefront.  With max_clashes == 0 (the reference's only use) the verdict is "no distance below thresh": the minimum
// is taken in packed fp32 against the staged fixed side and decided with the rigorous band of k_clash (embed_clash.hpp:
// fp32_min_band); a minimum inside the band, or max_clashes != 0, takes the fp64 count.
__device__ inline int torsion_comp_check_lds(const double *c, const uint16_t *moved, int nm, const uint16_t *fixed, int nf, const float *F,
                                             int npad, double cmax_fixed, double sq_bound, long long max_clashes, int lane, int *hint = nullptr) {
    if (hint) *hint = -1;
    int A = 1;
    while (A < nm && A < 64) A <<= 1;
    const int G = 64 / A, ai = lane & (A - 1), g = lane / A;
    const int per = (((nf + G - 1) / G) + 1) & ~1;  // even, so that packed pairs never straddle two slices
    const int b0 = min(nf, g * per), b1 = min(nf, b0 + per);
    if (max_clashes == 0 && nm > 0 && nf > 0) {
        double cmax = cmax_fixed;
        for (int r = ai; r < nm; r += A) {
            const int a = moved[r];
            cmax = fmax(cmax, fmax(fabs(c[3 * a]), fmax(fabs(c[3 * a + 1]), fabs(c[3 * a + 2]))));
        }
        for (int off = 32; off > 0; off >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, off));
        float lo, hi;
        if (fp32_min_band(sq_bound, cmax, &lo, &hi)) {
            const float *X = F, *Y = F + npad, *Z = F + 2 * npad;
            float m = __builtin_inff();
            bool found = false;
            for (int r = ai; r < nm && !found; r += A) {
                const int a = moved[r];
                const float xi = float(c[3 * a]), yi = float(c[3 * a + 1]), zi = float(c[3 * a + 2]);
                const clash_f32x2 x2 = {xi, xi}, y2 = {yi, yi}, z2 = {zi, zi};
                // sixteen columns at a time; most rotations a conformational search tries do clash, and a certain clash ends the check
                for (int jb = b0; jb < b1 && !found; jb += 16) {
                    const int je = min(jb + 16, b1);
#pragma unroll 4
                    for (int j = jb; j < je; j += 2) {
                        const clash_f32x2 dx = x2 - *reinterpret_cast<const clash_f32x2 *>(X + j);
                        const clash_f32x2 dy = y2 - *reinterpret_cast<const clash_f32x2 *>(Y + j);
                        const clash_f32x2 dz = z2 - *reinterpret_cast<const clash_f32x2 *>(Z + j);
                        clash_f32x2 s2 = dx * dx;
                        s2 = __builtin_elementwise_fma(dy, dy, s2);
                        s2 = __builtin_elementwise_fma(dz, dz, s2);
                        m = fminf(m, fminf(s2.x, s2.y));
                    }
                    found = __any(m < lo);
                }
            }
            const unsigned long long certain = __ballot(m < lo);
            if (certain) {  // some distance certainly < thresh; the lane that saw it is where the next step of a walk-back looks first
                if (hint) *hint = __ffsll((long long)certain) - 1;
                return 0;
            }
            for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
            if (m >= hi) return 1;  // every distance certainly >= thresh (NaN: falls through)
        }
    }
    int cnt = 0;
    for (int r = ai; r < nm; r += A) {
        const int a = moved[r];
        const double x = c[3 * a], y = c[3 * a + 1], z = c[3 * a + 2];
        for (int j = b0; j < b1; ++j) {
            const int b = fixed[j];
            const double dx = c[3 * b] - x, dy = c[3 * b + 1] - y, dz = c[3 * b + 2] - z;
            cnt += (dx * dx + dy * dy + dz * dz < sq_bound) ? 1 : 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    return (long long)cnt > max_clashes ? 0 : 1;
}

// The walk-back of torsion_module.py:490-498 turns the moved side by 5 degrees at a time and asks again; the atoms that clashed
// a step ago mostly still do.  `hint` = the lane of torsion_comp_check_lds that held the smallest distance: its moved atoms
// against its slice of the fixed side, spread over the wavefront.  true = one of those distances is certainly below thresh (the
// same rigorous fp32 band, for a coordinate bound that holds through the whole walk-back) -- the check would fail; false = not
// decided here, the full check runs.
__device__ inline bool hint_still_clashes(const double *c, const uint16_t *moved, int nm, int nf, const float *F, int npad, float lo, int hint, int lane) {
    int A = 1;
    while (A < nm && A < 64) A <<= 1;
    const int G = 64 / A, ai = hint & (A - 1), g = hint / A;
    const int per = (((nf + G - 1) / G) + 1) & ~1;
    const int b0 = min(nf, g * per), b1 = min(nf, b0 + per);
    const float *X = F, *Y = F + npad, *Z = F + 2 * npad;
    float m = __builtin_inff();
    for (int r = ai; r < nm; r += A) {
        const int a = moved[r];
        const float xi = float(c[3 * a]), yi = float(c[3 * a + 1]), zi = float(c[3 * a + 2]);
        for (int j = b0 + lane; j < b1; j += 64) {
            const float dx = xi - X[j], dy = yi - Y[j], dz = zi - Z[j];
            m = fminf(m, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
    }
    return __any(m < lo);
}

// Coordinate bound of the moved side that holds for every rotation about the axis through `cen`: |cen|_inf + the atom's distance
// from cen (a rotation keeps it; a part in 10^6 covers the rounding of many steps)
__device__ inline double moved_side_bound(const double *c, const double cen[3], const uint16_t *moved, int nm, int lane) {
    const double c_inf = fmax(fabs(cen[0]), fmax(fabs(cen[1]), fabs(cen[2])));
    double b = 0.0;
    for (int r = lane; r < nm; r += 64) {
        const int a = moved[r];
        const double v0 = c[3 * a] - cen[0], v1 = c[3 * a + 1] - cen[1], v2 = c[3 * a + 2] - cen[2];
        b = fmax(b, c_inf + sqrt(v0 * v0 + v1 * v1 + v2 * v2) * 1.000001);
    }
    for (int off = 32; off > 0; off >>= 1) b = fmax(b, __shfl_xor(b, off));
    return b;
}

// rotate_dihedral (tscode/utils.py:389-414) for a batch of structures that share torsion and mask: structure s turns its masked atoms by
// angles[s] DEGREES (any real number: tscode/torsion_module.py:984-1005 searches fractional corrections) about its own i2 - i3 bond,
// centre i3; no clash check, no walk-back.  One thread per (structure, atom); out must not alias coords (the axis atoms are read by
// every thread of a structure).
inline __global__ __launch_bounds__(256) void k_rotate_dihedral(const double *__restrict__ coords, int64_t n_structs, int n, int i2, int i3,
                                                          const uint8_t *__restrict__ mask, const double *__restrict__ angles, double *__restrict__ out) {
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < n_structs * n; e += int64_t(gridDim.x) * 256) {
        const int64_t s = e / n;
        const int a = int(e - s * n);
        const double *c = coords + s * n * 3;
        double v[3] = {c[3 * a], c[3 * a + 1], c[3 * a + 2]};
        if (mask[a]) {
            double R[9], cen[3];
            dihedral_rotation(c, i2, i3, angles[s], R, cen);
            const double d0 = v[0] - cen[0], d1 = v[1] - cen[1], d2 = v[2] - cen[2];  // (mat @ (coords[mask] - center).T).T + center, :412
            v[0] = (R[0] * d0 + R[1] * d1 + R[2] * d2) + cen[0];
            v[1] = (R[3] * d0 + R[4] * d1 + R[5] * d2) + cen[1];
            v[2] = (R[6] * d0 + R[7] * d1 + R[8] * d2) + cen[2];
        }
        out[e * 3] = v[0], out[e * 3 + 1] = v[1], out[e * 3 + 2] = v[2];
    }
}

// out [n_cand][n][3], rotated_bonds [n_cand]; angles [n_cand][n_tors] int32 degrees; masks [n_tors][n]; torsions [n_tors][4]
// dynamic LDS: torsion lists, then one csearch_wave_bytes(n) area per wavefront of the block
inline __global__ __launch_bounds__(256) void k_csearch_rotate(CsearchArgs a, const double *__restrict__ base, const int32_t *__restrict__ tors,
                                                         const uint8_t *__restrict__ masks, const int32_t *__restrict__ angles,
                                                         double *__restrict__ out, int32_t *__restrict__ rotated_bonds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, n = a.n, npad = (n + 2) & ~1;
    const TorsionLists L = torsion_lists_at(s_raw, a.n_tors, n);
    build_torsion_lists(L, masks, tors, a.n_tors, n);
    double *c = reinterpret_cast<double *>(s_raw + torsion_lists_bytes(a.n_tors, n) + size_t(wid) * csearch_wave_bytes(n));
    float *F = reinterpret_cast<float *>(c + size_t(n) * 3);
    double *step_rot = reinterpret_cast<double *>(F + size_t(3) * npad);
    for (int64_t m = int64_t(blockIdx.x) * nw + wid; m < a.n_cand; m += int64_t(gridDim.x) * nw) {
        for (int e = lane; e < n * 3; e += 64) c[e] = base[e];  // new_coords = np.copy(coords), :473
        __builtin_amdgcn_wave_barrier();
        int rotated = 0;
        for (int t = 0; t < a.n_tors; ++t) {
            const int angle = angles[m * a.n_tors + t];
            if (angle == 0) continue;  // :482
            const int i2 = tors[4 * t + 1], i3 = tors[4 * t + 2];
            const uint16_t *moved = L.moved + size_t(t) * n, *fixed = L.fixed + size_t(t) * n;
            const int nm = L.count[4 * t], nf = L.count[4 * t + 1];
            const double cmax_fixed = a.max_clashes == 0 ? stage_fixed_f32(c, fixed, nf, F, npad, lane) : 0.0;
            // :484-501 as one loop with a single check site: step -1 is the rotation by `angle`, steps 0 .. angle // 5 - 1 the walk-back
            const int steps = angle >= 0 ? angle / 5 : 0;  // range(angle // 5): a negative angle is never walked back
            // every step of the walk-back is the same rotation: the axis atoms do not move (i3 is the centre; i2 unless the mask turns
            // it, in which case the matrix is formed anew each step as the reference does)
            const bool same_matrix = L.count[4 * t + 2] == 0;
            int hint = -1;
            bool hint_ok = false;
            float lo_h = 0.0f;
            for (int rep = -1; rep < steps; ++rep) {
                if (rep <= 0 || !same_matrix) {  // the matrix of this step -> LDS (the one place the trigonometry is done)
                    double R[9], cen[3];
                    dihedral_rotation(c, i2, i3, rep < 0 ? double(angle) : -5.0, R, cen);
                    if (rep == 0) {  // the walk-back begins: the band its hints are decided with
                        float hi_h;
                        hint_ok = a.max_clashes == 0 && fp32_min_band(a.sq_bound, fmax(cmax_fixed, moved_side_bound(c, cen, moved, nm, lane)), &lo_h, &hi_h);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) step_rot[i] = R[i];
#pragma unroll
                        for (int i = 0; i < 3; ++i) step_rot[9 + i] = cen[i];
                    }
                }
                apply_rotation_lds(c, step_rot, step_rot + 9, moved, nm, lane);  // :484, :491
                if (rep >= 0 && hint_ok && hint >= 0 && hint_still_clashes(c, moved, nm, nf, F, npad, lo_h, hint, lane)) continue;  // still clashing
                if (torsion_comp_check_lds(c, moved, nm, fixed, nf, F, npad, cmax_fixed, a.sq_bound, a.max_clashes, lane, &hint)) {  // :487, :492
                    ++rotated;  // :494, :501
                    break;
                }
            }
        }
        double *o = out + m * n * 3;
        for (int e = lane; e < n * 3; e += 64) o[e] = c[e];
        if (lane == 0) rotated_bonds[m] = rotated;
        __builtin_amdgcn_wave_barrier();
    }
}

// torsion_comp_check for a batch of structures sharing torsion and mask: ok[s] = 1 / 0
inline __global__ __launch_bounds__(256) void k_torsion_comp_check(CsearchArgs a, const double *__restrict__ coords, const int32_t *__restrict__ tors,
                                                             const uint8_t *__restrict__ mask, int32_t *__restrict__ ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, n = a.n, npad = (n + 2) & ~1;
    const TorsionLists L = torsion_lists_at(s_raw, 1, n);
    build_torsion_lists(L, mask, tors, 1, n);
    double *c = reinterpret_cast<double *>(s_raw + torsion_lists_bytes(1, n) + size_t(wid) * csearch_wave_bytes(n));
    float *F = reinterpret_cast<float *>(c + size_t(n) * 3);
    const int nm = L.count[0], nf = L.count[1];
    for (int64_t m = int64_t(blockIdx.x) * nw + wid; m < a.n_cand; m += int64_t(gridDim.x) * nw) {
        const double *src = coords + m * n * 3;
        for (int e = lane; e < n * 3; e += 64) c[e] = src[e];
        __builtin_amdgcn_wave_barrier();
        const double cmax_fixed = a.max_clashes == 0 ? stage_fixed_f32(c, L.fixed, nf, F, npad, lane) : 0.0;
        const int r = torsion_comp_check_lds(c, L.moved, nm, L.fixed, nf, F, npad, cmax_fixed, a.sq_bound, a.max_clashes, lane);
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

inline __global__ __launch_bounds__(256) void k_string_embed_params(const double *__restrict__ p1, const double *__restrict__ p2,
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

inline __global__ __launch_bounds__(256) void k_cyclical_embed_params(const double *__restrict__ start, const double *__restrict__ end,
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
