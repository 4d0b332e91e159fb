// embed_clash.hpp -- K1 (batched rigid-body embedding) and K2 (compenetration mask) kernels.
//
// K1  out[s] = concat_m (R[s,m] @ X_m[c[s,m]].T).T + t[s,m]         reference embeds.py:961-969
// K2  mask[s] = count(all_dists(frag_b, frag_a) < thresh) <= max     reference numba_functions.py:59-105
//
// Both are HBM-streaming kernels: K1 writes n*24 B per pose, K2 reads n*24 B per pose.  K2 stages each
// pose once in LDS (one pose per LP-lane group of a wavefront), every lane owns one atom of the
// "later" fragments and walks the atoms of the earlier fragments, which all lanes of the group read
// from the same LDS address (a broadcast); the per-pose count is reduced with cross-lane shuffles.
#pragma once
#include "common.hpp"

namespace tsc {

constexpr int MAX_MOLS = 8;

struct FragTable {
    int n_mols;
    int n_total;                 // atoms per pose
    long long frag_off[MAX_MOLS];  // offset of fragment m in `frags`, in doubles
    int n_atoms[MAX_MOLS];
    int n_conf[MAX_MOLS];
    int atom_off[MAX_MOLS + 1];  // first atom of fragment m inside a pose
};

__device__ inline int frag_of_atom(const FragTable &ft, int a) {
    int m = 0;
#pragma unroll
    for (int k = 1; k < MAX_MOLS; ++k)
        if (k < ft.n_mols && a >= ft.atom_off[k]) m = k;
    return m;
}

// one atom of one pose: R x + t
__device__ inline void embed_atom(const double *__restrict__ frags, const FragTable &ft, const int32_t *__restrict__ conf_idx,
                                  const double *__restrict__ rot, const double *__restrict__ pos, int64_t s, int a, double out[3]) {
    int m = frag_of_atom(ft, a);
    int64_t sm = s * ft.n_mols + m;
    const double *X = frags + ft.frag_off[m] + (int64_t(conf_idx[sm]) * ft.n_atoms[m] + (a - ft.atom_off[m])) * 3;
    const double *R = rot + sm * 9;
    const double *t = pos + sm * 3;
    double x0 = X[0], x1 = X[1], x2 = X[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = R[3 * i] * x0 + R[3 * i + 1] * x1 + R[3 * i + 2] * x2 + t[i];
}

// one atom of one pose from pose parameters staged in LDS: R [entries][9], t [entries][3], conformer [entries],
// entry l = local pose * n_mols + fragment.  (Fetching R and t per atom from global memory costs 12 vector loads per
// atom, all hitting the same few lines: the address unit, not the bandwidth, then bounds the embedding kernels.)
__device__ inline void embed_atom_staged(const double *__restrict__ frags, const FragTable &ft, const double *sR, const double *sT, const int *sC,
                                         int local_pose, int a, double out[3]) {
    const int m = frag_of_atom(ft, a), l = local_pose * ft.n_mols + m;
    const double *X = frags + ft.frag_off[m] + (int64_t(sC[l]) * ft.n_atoms[m] + (a - ft.atom_off[m])) * 3;
    const double *R = sR + l * 9, *t = sT + l * 3;
    const double x0 = X[0], x1 = X[1], x2 = X[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = R[3 * i] * x0 + R[3 * i + 1] * x1 + R[3 * i + 2] * x2 + t[i];
}

// K1: a workgroup embeds TR_POSES poses at a time: their rotations, translations and conformer indices go to LDS first,
// then one thread per (pose, atom); consecutive threads write consecutive 24-byte triples.
// idx (optional): only the listed poses are embedded, out row r <- pose idx[r] (used after the clash
// filter); heavy_sel/heavy_out (optional): additionally write the heavy-atom subset of each pose.
constexpr int TR_POSES = 32;
__host__ __device__ inline size_t transform_lds_bytes(int n_mols) { return size_t(TR_POSES) * n_mols * (12 * sizeof(double) + sizeof(int)) + TR_POSES * sizeof(int64_t); }

inline __global__ __launch_bounds__(256) void k_transform(const double *__restrict__ frags, FragTable ft,
                                                    const int32_t *__restrict__ conf_idx, const double *__restrict__ rot,
                                                    const double *__restrict__ pos, const int32_t *__restrict__ idx,
                                                    int64_t n_out, double *__restrict__ out,
                                                    const int32_t *__restrict__ heavy_slot, int n_heavy,
                                                    double *__restrict__ heavy_out, const int32_t *__restrict__ n_out_dev,
                                                    double *__restrict__ zero_me = nullptr, int n_zero = 0) {
    // n_out_dev (optional): the row count lives on the device (the total of the scan that made idx); the grid is then
    // sized for an upper bound and the host need not wait for the count before launching
    // zero_me (optional): n_zero doubles cleared on the way (the moment accumulators of the basis estimate that follows
    // the sample embed: a memset in that short chain is two more launches)
    extern __shared__ __attribute__((aligned(16))) double s_tr[];
    const int n = ft.n_total, nm = ft.n_mols, tid = threadIdx.x;
    for (int e = blockIdx.x * 256 + tid; e < n_zero; e += gridDim.x * 256) zero_me[e] = 0.0;
    if (n_out_dev) n_out = *n_out_dev;
    int64_t *sP = reinterpret_cast<int64_t *>(s_tr);  // pose of every local row
    double *sR = s_tr + TR_POSES, *sT = sR + TR_POSES * nm * 9;
    int *sC = reinterpret_cast<int *>(sT + TR_POSES * nm * 3);
    for (int64_t r0 = int64_t(blockIdx.x) * TR_POSES; r0 < n_out; r0 += int64_t(gridDim.x) * TR_POSES) {
        const int np = int(min<int64_t>(TR_POSES, n_out - r0));
        if (tid < np) sP[tid] = idx ? int64_t(idx[r0 + tid]) : r0 + tid;
        __syncthreads();
        for (int q = tid; q < np * nm * 9; q += 256) {
            const int row = q / (nm * 9), w = q - row * nm * 9;
            sR[q] = rot[sP[row] * nm * 9 + w];
        }
        for (int q = tid; q < np * nm * 3; q += 256) {
            const int row = q / (nm * 3), w = q - row * nm * 3;
            sT[q] = pos[sP[row] * nm * 3 + w];
        }
        for (int q = tid; q < np * nm; q += 256) {
            const int row = q / nm, w = q - row * nm;
            sC[q] = conf_idx[sP[row] * nm + w];
        }
        __syncthreads();
        for (int e = tid; e < np * n; e += 256) {
            const int row = e / n, a = e - row * n;
            const int64_t r = r0 + row;
            double v[3];
            embed_atom_staged(frags, ft, sR, sT, sC, row, a, v);
            if (out) {
                double *o = out + (r * n + a) * 3;
                o[0] = v[0];
                o[1] = v[1];
                o[2] = v[2];
            }
            if (heavy_out) {
                const int hs = heavy_slot[a];  // position of atom a among the heavy atoms, or -1
                if (hs >= 0) {
                    double *h = heavy_out + (r * n_heavy + hs) * 3;
                    h[0] = v[0];
                    h[1] = v[1];
                    h[2] = v[2];
                }
            }
        }
        __syncthreads();
    }
}

struct ClashArgs {
    int64_t n_poses;
    int n;          // atoms per pose
    int first_row;  // first atom that owns a row of the count (atoms of fragments >= 1; 0 for ids=None)
    int self_mode;  // 1: ids=None -> count_clashes semantics (all ordered pairs i != j with 0 < d < 0.5)
    int lp;         // lanes per pose (power of two, <= 64)
    int n_mols;
    int atom_off[MAX_MOLS + 1];
    double sq_bound;  // d < thresh  <=>  d2 < sq_bound   (exact: see clash_sq_bound)
    long long max_clashes;
};

// Smallest double x with sqrt(x) >= thresh, so that (sqrt(d2) < thresh) == (d2 < x) for every d2:
// IEEE sqrt is correctly rounded and monotone.  Lets the kernels compare squared distances while
// giving the verdict of the reference's `all_dists(...) < thresh` (algebra.py:133-155, sqrt then <).
inline double clash_sq_bound(double thresh) {
    if (!(thresh > 0)) return 0.0;
    double x = thresh * thresh;
    while (std::sqrt(x) >= thresh) x = std::nextafter(x, 0.0);
    while (std::sqrt(x) < thresh) x = std::nextafter(x, INFINITY);
    return x;
}

typedef float clash_f32x2 __attribute__((ext_vector_type(2)));

// LDS bytes one wavefront of k_clash needs: the fp64 pose(s) and, for MINMODE, an fp32 copy as three arrays
__host__ __device__ inline size_t clash_lds_per_wave(int n, int lp, bool minmode) {
    const int ppw = 64 / lp, npad = (n + 1) & ~1;
    return size_t(ppw) * n * 3 * sizeof(double) + (minmode ? size_t(ppw) * 3 * npad * sizeof(float) : 0);
}

// MINMODE (max_clashes == 0, no counts wanted -- the pipeline's case): the verdict is "no inter-fragment distance
// below thresh", i.e. min d^2 >= sq_bound.  The minimum is taken in packed fp32 (two partner atoms per instruction,
// v_pk_add_f32 / v_pk_fma_f32 / v_min3_f32: 3.5 instructions per pair against 8 in fp64) on an fp32 copy of the pose,
// and decided with a rigorous error bound: with C = max |coordinate| of the pose, u = 2^-24 and e = 4 u C (1 + u) the
// computed s32 differs from the exact d^2 by at most 4 u s32 + 2 sqrt(3) e d + 3 e^2, so s32 >= hi = x0 + M means
// d^2 >= x0 and s32 < lo = x0 - M means d^2 < x0 (M below).  A pose whose minimum falls between lo and hi (a few in
// 10^4) is decided by the fp64 count like everything else.
// lo / hi of the comment above for the squared bound x0 and the coordinate bound cmax; false = no usable band (take fp64)
__device__ inline bool fp32_min_band(double x0, double cmax, float *lo_out, float *hi_out) {
    constexpr double U = 5.9604644775390625e-08;  // 2^-24
    const double e = 4.0 * U * cmax * (1.0 + U);
    const double M = 1.01 * (8.0 * U * x0 + (2.0 * 1.7320508075688774 * e * sqrt(x0) + 3.0 * e * e) * (1.0 + 4.0 * U));
    float lo = float(x0 - M), hi = float(x0 + M);
    if (double(lo) > x0 - M) lo = __uint_as_float(__float_as_uint(lo) - 1u);  // round towards the safe side (both > 0)
    if (double(hi) < x0 + M) hi = __uint_as_float(__float_as_uint(hi) + 1u);
    *lo_out = lo, *hi_out = hi;
    return cmax < 1.0e15 && x0 - M > 0.0;
}

template <bool FUSED, bool SELF, bool MINMODE>
inline __global__ __launch_bounds__(256) void k_clash(ClashArgs a, const double *__restrict__ coords,
                                                const double *__restrict__ frags, FragTable ft,
                                                const int32_t *__restrict__ conf_idx, const double *__restrict__ rot,
                                                const double *__restrict__ pos, uint8_t *__restrict__ mask,
                                                int32_t *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) double s_xyz[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n = a.n, lp = a.lp, ppw = 64 / lp, npad = (n + 1) & ~1;
    const size_t wave_doubles = clash_lds_per_wave(n, lp, MINMODE) / sizeof(double);
    double *w_xyz = s_xyz + size_t(wid) * wave_doubles;
    float *w_f32 = reinterpret_cast<float *>(w_xyz + size_t(ppw) * n * 3);  // [ppw][3][npad]
    const int64_t waves_total = int64_t(gridDim.x) * 4;
    for (int64_t wv = int64_t(blockIdx.x) * 4 + wid; wv * ppw < a.n_poses; wv += waves_total) {
        const int64_t pose0 = wv * ppw;
        const int np = int(min<int64_t>(ppw, a.n_poses - pose0));
        // ---- stage np poses in LDS
        double cmax = 0.0;
        if (FUSED) {
            // (R, t and the conformer index come straight from global memory here: staging them in LDS first, as k_transform
            // does, puts one more dependent round trip in front of a wavefront that is a chain of them -- measured 14 us slower)
            for (int e = lane; e < np * n; e += 64) {
                int sub = e / n, at = e - sub * n;
                double v[3];
                embed_atom(frags, ft, conf_idx, rot, pos, pose0 + sub, at, v);
                w_xyz[e * 3 + 0] = v[0];
                w_xyz[e * 3 + 1] = v[1];
                w_xyz[e * 3 + 2] = v[2];
                if (MINMODE) {
                    float *f = w_f32 + size_t(sub) * 3 * npad;
                    f[at] = float(v[0]), f[npad + at] = float(v[1]), f[2 * npad + at] = float(v[2]);
                    cmax = fmax(cmax, fmax(fabs(v[0]), fmax(fabs(v[1]), fabs(v[2]))));
                }
            }
        } else {
            const double *src = coords + pose0 * n * 3;
            for (int e = lane; e < np * n * 3; e += 64) {
                const double v = src[e];
                w_xyz[e] = v;
                if (MINMODE) {
                    const int sub = e / (n * 3), r = e - sub * n * 3, at = r / 3, k = r - at * 3;
                    w_f32[size_t(sub) * 3 * npad + k * npad + at] = float(v);
                    cmax = fmax(cmax, fabs(v));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int sub = lane / lp, li = lane - sub * lp;
        const double *p = w_xyz + size_t(sub) * n * 3;
        bool decided = false;
        int verdict = 0;
        if (MINMODE) {
            // one bound for the whole wavefront (the largest coordinate of any of its poses): simple and still tiny
            for (int off = 32; off > 0; off >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, off));
            float lo, hi;
            const bool banded = fp32_min_band(a.sq_bound, cmax, &lo, &hi);
            const float *X = w_f32 + size_t(sub) * 3 * npad, *Y = X + npad, *Z = Y + npad;
            float m = __builtin_inff();
            if (sub < np && banded) {
                for (int ia = a.first_row + li; ia < n; ia += lp) {
                    const float xi = X[ia], yi = Y[ia], zi = Z[ia];
                    int jend = 0;
#pragma unroll
                    for (int k = 1; k < MAX_MOLS; ++k)
                        if (k < a.n_mols && ia >= a.atom_off[k]) jend = a.atom_off[k];
                    const clash_f32x2 x2 = {xi, xi}, y2 = {yi, yi}, z2 = {zi, zi};
                    const int j2 = jend & ~1;
#pragma unroll 4
                    for (int j = 0; j < j2; j += 2) {
                        const clash_f32x2 dx = x2 - *reinterpret_cast<const clash_f32x2 *>(X + j);
                        const clash_f32x2 dy = y2 - *reinterpret_cast<const clash_f32x2 *>(Y + j);
                        const clash_f32x2 dz = z2 - *reinterpret_cast<const clash_f32x2 *>(Z + j);
                        clash_f32x2 s2 = dx * dx;
                        s2 = __builtin_elementwise_fma(dy, dy, s2);
                        s2 = __builtin_elementwise_fma(dz, dz, s2);
                        m = fminf(m, fminf(s2.x, s2.y));
                    }
                    if (jend & 1) {
                        const float dx = xi - X[j2], dy = yi - Y[j2], dz = zi - Z[j2];
                        m = fminf(m, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
                    }
                    // a pose that owns the whole wavefront stops at its first certain clash (most poses of a dense embed clash)
                    if (lp == 64 && __any(m < lo)) break;
                }
                for (int off = lp >> 1; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
                if (m >= hi) decided = true, verdict = 1;       // every distance certainly >= thresh (NaN: undecided)
                else if (m < lo) decided = true, verdict = 0;   // some distance certainly < thresh
            } else {
                for (int off = lp >> 1; off > 0; off >>= 1) (void)__shfl_xor(m, off);  // keep the shuffles convergent
            }
            if (sub >= np) decided = true;
        }
        // ---- count close pairs in fp64: lane owns atom ia of pose `sub`, walks earlier-fragment atoms j
        int cnt = 0;
        if (!MINMODE || __any(!decided)) {
            if (sub < np && !decided) {
                for (int ia = a.first_row + li; ia < n; ia += lp) {
                    const double x = p[ia * 3], y = p[ia * 3 + 1], z = p[ia * 3 + 2];
                    int jend = n;
                    if (!SELF) {
                        jend = 0;
#pragma unroll
                        for (int k = 1; k < MAX_MOLS; ++k)
                            if (k < a.n_mols && ia >= a.atom_off[k]) jend = a.atom_off[k];
                    }
                    // four independent partner atoms per trip: their LDS reads are issued together
#pragma unroll 4
                    for (int j = 0; j < jend; ++j) {
                        const double dx = x - p[j * 3], dy = y - p[j * 3 + 1], dz = z - p[j * 3 + 2];
                        const double d2 = dx * dx + dy * dy + dz * dz;
                        const bool hit = SELF ? (d2 < a.sq_bound && d2 > 0.0) : (d2 < a.sq_bound);
                        cnt += hit ? 1 : 0;
                    }
                }
            }
            for (int off = lp >> 1; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            if (!decided) verdict = (long long)cnt <= a.max_clashes ? 1 : 0;
        }
        if (sub < np && li == 0) {
            mask[pose0 + sub] = uint8_t(verdict);
            if (counts) counts[pose0 + sub] = cnt;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// K1 + K2 fused, ONE POSE PER LANE (two fragments, verdict only, no clash allowed: the pipeline's case at 100k - 1M poses of 50 atoms).
// k_clash gives a pose to a group of lanes: every wavefront is then a chain of dependent round trips -- the pose's parameters, its
// fragments' coordinates, the pose through LDS, the group's shuffles -- for two poses of work, and 60 us pass over 19 MB of input
// (0.3 TB/s) with the VALU a tenth busy.  Here a lane owns a pose from its 200 bytes of parameters to its verdict byte: the atoms of the
// SMALLER fragment ("A", at most 2 NA2 of them) are embedded once and stay in registers as packed fp32 pairs, the atoms of the other are
// embedded one after the other and each is compared with all of A (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 / min: 3.5 instructions per
// distance, nothing but registers), so a wavefront issues some 2 500 independent-enough instructions for 64 poses and the kernel is
// bound by VALU issue, not by latency.  Same arithmetic and the same rigorous band as k_clash's MINMODE (fp64 embedding, fp32 copy,
// s32 = dx^2 + dy^2 + dz^2 by mul + 2 fma; fp32_min_band with the POSE's own largest coordinate): a pose whose minimum falls inside the
// band is recounted in fp64 by the whole wavefront, all its n_A n_B distances at once (d^2 < sq_bound: exactly the reference's
// sqrt-then-compare verdict, numba_functions.py:77-86).
template <int NA2>
inline __global__ __launch_bounds__(256) void k_clash_lanes(int64_t n_poses, const double *__restrict__ frags, FragTable ft, int mA, int mB,
                                                      const int32_t *__restrict__ conf_idx, const double *__restrict__ rot,
                                                      const double *__restrict__ pos, double sq_bound, uint8_t *__restrict__ mask) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nA = ft.n_atoms[mA], nB = ft.n_atoms[mB], nm = ft.n_mols;
    for (int64_t s0 = (int64_t(blockIdx.x) * 4 + wid) * 64; s0 < n_poses; s0 += int64_t(gridDim.x) * 256) {
        const bool valid = s0 + lane < n_poses;
        const int64_t s = valid ? s0 + lane : n_poses - 1;  // (idle lanes of the last wavefront walk along with the last pose)
        double RA[9], tA[3], RB[9], tB[3];
        {
            const double *r = rot + (s * nm + mA) * 9, *t = pos + (s * nm + mA) * 3;
#pragma unroll
            for (int q = 0; q < 9; ++q) RA[q] = r[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) tA[q] = t[q];
            r = rot + (s * nm + mB) * 9, t = pos + (s * nm + mB) * 3;
#pragma unroll
            for (int q = 0; q < 9; ++q) RB[q] = r[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) tB[q] = t[q];
        }
        const double *XA = frags + ft.frag_off[mA] + int64_t(conf_idx[s * nm + mA]) * nA * 3;
        const double *XB = frags + ft.frag_off[mB] + int64_t(conf_idx[s * nm + mB]) * nB * 3;
        clash_f32x2 ax[NA2], ay[NA2], az[NA2];
        double cmax = 0.0;
#pragma unroll
        for (int k = 0; k < NA2; ++k) {
            float v[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int a = 2 * k + u;
                if (a < nA) {
                    const double x0 = XA[3 * a], x1 = XA[3 * a + 1], x2 = XA[3 * a + 2];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double w = RA[3 * i] * x0 + RA[3 * i + 1] * x1 + RA[3 * i + 2] * x2 + tA[i];  // embed_atom
                        cmax = fmax(cmax, fabs(w));
                        v[u][i] = float(w);
                    }
                } else {
                    v[u][0] = v[u][1] = v[u][2] = 1.0e18f;  // (padding: an atom far from everything; its squared distances stay finite in fp32)
                }
            }
            ax[k] = clash_f32x2{v[0][0], v[1][0]}, ay[k] = clash_f32x2{v[0][1], v[1][1]}, az[k] = clash_f32x2{v[0][2], v[1][2]};
        }
        float m = __builtin_inff();
        for (int b = 0; b < nB; ++b) {
            const double x0 = XB[3 * b], x1 = XB[3 * b + 1], x2 = XB[3 * b + 2];
            float bf[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double w = RB[3 * i] * x0 + RB[3 * i + 1] * x1 + RB[3 * i + 2] * x2 + tB[i];
                cmax = fmax(cmax, fabs(w));
                bf[i] = float(w);
            }
            const clash_f32x2 bx = {bf[0], bf[0]}, by = {bf[1], bf[1]}, bz = {bf[2], bf[2]};
            // two running minima: consecutive pairs of A do not wait for each other
            float m0 = m, m1 = __builtin_inff();
#pragma unroll
            for (int k = 0; k < NA2; ++k) {
                const clash_f32x2 dx = bx - ax[k], dy = by - ay[k], dz = bz - az[k];
                clash_f32x2 s2 = dx * dx;
                s2 = __builtin_elementwise_fma(dy, dy, s2);
                s2 = __builtin_elementwise_fma(dz, dz, s2);
                if (k & 1) m1 = fminf(m1, fminf(s2.x, s2.y));
                else m0 = fminf(m0, fminf(s2.x, s2.y));
            }
            m = fminf(m0, m1);
        }
        float lo, hi;
        const bool banded = fp32_min_band(sq_bound, cmax, &lo, &hi);
        int verdict = 1;
        bool decided = !valid;
        if (banded && m >= hi) decided = true, verdict = 1;      // every distance certainly >= thresh (a NaN minimum: undecided)
        else if (banded && m < lo) decided = true, verdict = 0;  // some distance certainly < thresh
        // ---- poses inside the band (a few in 10^4), or without a usable band: every distance in fp64, by the whole wavefront
        for (unsigned long long und = __builtin_amdgcn_ballot_w64(!decided); und; und &= und - 1) {
            const int l = __ffsll((long long)und) - 1;
            const int64_t sp = s0 + l;
            bool hit = false;
            for (int e = lane; e < nA * nB; e += 64) {
                const int a = e / nB, b = e - a * nB;
                double p[3], q[3];
                embed_atom(frags, ft, conf_idx, rot, pos, sp, ft.atom_off[mA] + a, p);
                embed_atom(frags, ft, conf_idx, rot, pos, sp, ft.atom_off[mB] + b, q);
                const double dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                hit = hit || (dx * dx + dy * dy + dz * dz < sq_bound);
            }
            const bool any = __builtin_amdgcn_ballot_w64(hit) != 0;
            if (lane == l) verdict = any ? 0 : 1;  // count(D < thresh) <= 0
        }
        if (valid) mask[s0 + lane] = uint8_t(verdict);
    }
}

// K1 + K2 fused, one pose per lane, ANY two or three fragments (verdict only, no clash allowed): what k_clash_lanes does for two fragments with
// a small one, for fragments of any size and for the three fragment pairs of a trimolecular pose (numba_functions.py:87-105: (m2, m1), (m3, m2),
// (m1, m3); with max_clashes = 0 the verdict is "no distance of any pair below thresh", whatever the order and the reference's early exits).
//   * the "A" fragment of a pair goes through the registers 2 NA2 atoms at a time (packed fp32 pairs); for every such tile the atoms of
//     the "B" fragment are embedded one after the other (fp64, 15 instructions against the tile's 3.5 per distance) and each meets the
//     whole tile -- registers only, two running minima;
//   * a WORKGROUP owns CLM_POSES poses and takes the fragment pairs one after the other: after a pair the poses that certainly clash
//     (minimum below the band) leave, the others are packed again (their number, running minimum and coordinate bound in LDS), so that the
//     wavefronts of the next pair are full of poses still worth looking at.  On C5 (500k poses of 70 + 70 + 60 atoms, 75 % of them
//     clashing) that is a quarter less work than three pairs for every pose; what a lane of a clashing pose would otherwise still do cannot
//     be skipped inside a wavefront, 64 poses wide;
//   * same rigorous band as k_clash's MINMODE (fp32_min_band with the POSE's own largest coordinate so far: the bound only has to cover
//     the atoms that entered the minimum); a pose that ends inside the band is recounted in fp64 by a whole wavefront.
#ifndef TSC_CLM_OCC
#define TSC_CLM_OCC 2
#endif
#ifndef TSC_CLM_POSES
#define TSC_CLM_POSES 512
#endif
constexpr int CLM_POSES = TSC_CLM_POSES;   // poses per workgroup and round (two per thread in the first pair)
template <int NA2>
inline __global__ __launch_bounds__(256, TSC_CLM_OCC) void k_clash_lanes_multi(int64_t n_poses, const double *__restrict__ frags, FragTable ft,
                                                            const int32_t *__restrict__ conf_idx, const double *__restrict__ rot,
                                                            const double *__restrict__ pos, double sq_bound, uint8_t *__restrict__ mask) {
    __shared__ int s_id[2][CLM_POSES];        // poses still undecided, packed (two lists: the one being read, the one being made)
    __shared__ float s_m[2][CLM_POSES];       // ... their running minimum of squared distances
    __shared__ float s_c[2][CLM_POSES];       // ... and the largest |coordinate| met so far (rounded up)
    __shared__ int s_n[2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nm = ft.n_mols, n_pairs = nm == 2 ? 1 : 3;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t base = int64_t(blockIdx.x) * CLM_POSES; base < n_poses; base += int64_t(gridDim.x) * CLM_POSES) {
        const int n0 = int(min<int64_t>(CLM_POSES, n_poses - base));
        __syncthreads();                       // (the lists of the round before are no longer read)
        if (tid == 0) s_n[0] = n0, s_n[1] = 0;
        for (int e = tid; e < n0; e += 256) s_id[0][e] = e, s_m[0][e] = __builtin_inff(), s_c[0][e] = 0.0f;
        __syncthreads();
        for (int pr = 0; pr < n_pairs; ++pr) {
            const int cur = pr & 1, nxt = cur ^ 1;
            // numba_functions.py:87-105: (m2, m1), (m3, m2), (m1, m3); two fragments: (m2, m1)
            const int mA = nm == 2 ? 1 : (pr == 0 ? 1 : (pr == 1 ? 2 : 0)), mB = nm == 2 ? 0 : (pr == 0 ? 0 : (pr == 1 ? 1 : 2));
            const int nA = ft.n_atoms[mA], nB = ft.n_atoms[mB];
            const int n_cur = s_n[cur];
            const bool last_pair = pr == n_pairs - 1;
            for (int e0 = wid * 64; e0 < n_cur; e0 += 256) {        // (wave-uniform: 64 packed poses per trip)
                const bool valid = e0 + lane < n_cur;
                const int e = valid ? e0 + lane : n_cur - 1;          // (idle lanes walk along with the last pose)
                const int lid = s_id[cur][e];
                const int64_t s = base + lid;
                float m = s_m[cur][e];
                double cmax = double(s_c[cur][e]);
                double RA[9], tA[3], RB[9], tB[3];
                {
                    const double *r = rot + (s * nm + mA) * 9, *t = pos + (s * nm + mA) * 3;
#pragma unroll
                    for (int q = 0; q < 9; ++q) RA[q] = r[q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) tA[q] = t[q];
                    r = rot + (s * nm + mB) * 9, t = pos + (s * nm + mB) * 3;
#pragma unroll
                    for (int q = 0; q < 9; ++q) RB[q] = r[q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) tB[q] = t[q];
                }
                const double *XA = frags + ft.frag_off[mA] + int64_t(conf_idx[s * nm + mA]) * nA * 3;
                const double *XB = frags + ft.frag_off[mB] + int64_t(conf_idx[s * nm + mB]) * nB * 3;
                for (int a0 = 0; a0 < nA; a0 += 2 * NA2) {           // the A fragment, a register tile at a time
                    clash_f32x2 ax[NA2], ay[NA2], az[NA2];
#pragma unroll
                    for (int k = 0; k < NA2; ++k) {
                        float v[2][3];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int a = a0 + 2 * k + u;
                            if (a < nA) {
                                const double x0 = XA[3 * a], x1 = XA[3 * a + 1], x2 = XA[3 * a + 2];
#pragma unroll
                                for (int i = 0; i < 3; ++i) {
                                    const double w = RA[3 * i] * x0 + RA[3 * i + 1] * x1 + RA[3 * i + 2] * x2 + tA[i];  // embed_atom
                                    cmax = fmax(cmax, fabs(w));
                                    v[u][i] = float(w);
                                }
                            } else {
                                v[u][0] = v[u][1] = v[u][2] = 1.0e18f;  // (padding: an atom far from everything; its squared distances stay finite in fp32)
                            }
                        }
                        ax[k] = clash_f32x2{v[0][0], v[1][0]}, ay[k] = clash_f32x2{v[0][1], v[1][1]}, az[k] = clash_f32x2{v[0][2], v[1][2]};
                    }
                    for (int b = 0; b < nB; ++b) {
                        const double x0 = XB[3 * b], x1 = XB[3 * b + 1], x2 = XB[3 * b + 2];
                        float bf[3];
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const double w = RB[3 * i] * x0 + RB[3 * i + 1] * x1 + RB[3 * i + 2] * x2 + tB[i];
                            cmax = fmax(cmax, fabs(w));
                            bf[i] = float(w);
                        }
                        const clash_f32x2 bx = {bf[0], bf[0]}, by = {bf[1], bf[1]}, bz = {bf[2], bf[2]};
                        float m0 = m, m1 = __builtin_inff();           // two running minima: consecutive pairs of A do not wait for each other
#pragma unroll
                        for (int k = 0; k < NA2; ++k) {
                            const clash_f32x2 dx = bx - ax[k], dy = by - ay[k], dz = bz - az[k];
                            clash_f32x2 s2 = dx * dx;
                            s2 = __builtin_elementwise_fma(dy, dy, s2);
                            s2 = __builtin_elementwise_fma(dz, dz, s2);
                            if (k & 1) m1 = fminf(m1, fminf(s2.x, s2.y));
                            else m0 = fminf(m0, fminf(s2.x, s2.y));
                        }
                        m = fminf(m0, m1);
                    }
                }
                // after this pair: certainly a clash -> the pose leaves with verdict 0; else it goes on to the next pair, or to its verdict
                float lo, hi;
                const bool banded = fp32_min_band(sq_bound, cmax, &lo, &hi);
                const bool clash = valid && banded && m < lo;           // some distance certainly < thresh (a NaN minimum: never certain)
                if (clash) mask[s] = 0;
                const bool goes_on = valid && !clash;
                if (!last_pair) {
                    const unsigned long long go = __builtin_amdgcn_ballot_w64(goes_on);
                    int slot0 = 0;
                    if (lane == 0 && go) slot0 = atomicAdd(&s_n[nxt], __popcll(go));
                    slot0 = __builtin_amdgcn_readfirstlane(slot0);
                    if (goes_on) {
                        const int d = slot0 + __popcll(go & lt_mask);
                        s_id[nxt][d] = lid, s_m[nxt][d] = m;
                        s_c[nxt][d] = __double2float_ru(cmax);
                    }
                } else {
                    // every distance certainly >= thresh: passes.  Inside the band (a few poses in 10^4), or without a usable band: every
                    // distance of every pair in fp64, by the whole wavefront
                    bool decided = !goes_on;
                    if (goes_on && banded && m >= hi) decided = true, mask[s] = 1;
                    for (unsigned long long und = __builtin_amdgcn_ballot_w64(!decided); und; und &= und - 1) {
                        const int l = __ffsll((long long)und) - 1;
                        const int64_t sp = base + __builtin_amdgcn_readlane(lid, l);
                        bool hit = false;
                        for (int q = 0; q < n_pairs; ++q) {
                            const int qa = nm == 2 ? 1 : (q == 0 ? 1 : (q == 1 ? 2 : 0)), qb = nm == 2 ? 0 : (q == 0 ? 0 : (q == 1 ? 1 : 2));
                            const int na = ft.n_atoms[qa], nb = ft.n_atoms[qb];
                            for (int el = lane; el < na * nb; el += 64) {
                                const int a = el / nb, b = el - a * nb;
                                double p3[3], q3[3];
                                embed_atom(frags, ft, conf_idx, rot, pos, sp, ft.atom_off[qa] + a, p3);
                                embed_atom(frags, ft, conf_idx, rot, pos, sp, ft.atom_off[qb] + b, q3);
                                const double dx = q3[0] - p3[0], dy = q3[1] - p3[1], dz = q3[2] - p3[2];
                                hit = hit || (dx * dx + dy * dy + dz * dz < sq_bound);
                            }
                        }
                        const bool any = __builtin_amdgcn_ballot_w64(hit) != 0;
                        if (lane == l) mask[sp] = any ? 0 : 1;          // count(D < thresh) <= 0
                    }
                }
            }
            __syncthreads();                   // the list of the next pair is complete
            if (tid == 0) s_n[cur] = 0;        // (it becomes the list the pair after next fills)
            __syncthreads();
        }
    }
}

// all_dists (algebra.py:98-157): out[i, j] = sqrt(sum_k (A[i,k] - B[j,k])^2)
inline __global__ __launch_bounds__(256) void k_all_dists(const double *__restrict__ A, int na, const double *__restrict__ B, int nb,
                                                    double *__restrict__ out) {
    int64_t total = int64_t(na) * nb;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        int i = int(e / nb), j = int(e - int64_t(i) * nb);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double d = A[i * 3 + k] - B[j * 3 + k];
            acc += d * d;
        }
        out[e] = sqrt(acc);
    }
}

inline int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace tsc
