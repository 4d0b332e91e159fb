// mfma_probe.hip -- what v_mfma_f32_16x16x32_f16 does on this device: operand layout, special values, accumulation error.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_probe.hip -o tools/probe/mfma_probe ; run on the GPU box.  Measurement tool, not product code.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// A[16][32], B[32][16] row-major halves in memory; the kernel loads them in the ASSUMED register layout and stores C[16][16]
__global__ void k_probe(const _Float16 *A, const _Float16 *B, float *C, int n_mats) {
    const int lane = threadIdx.x & 63, g = lane >> 4, rc = lane & 15;
    for (int m = blockIdx.x; m < n_mats; m += gridDim.x) {
        h8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = A[(size_t(m) * 16 + rc) * 32 + 8 * g + j];
            b[j] = B[(size_t(m) * 32 + 8 * g + j) * 16 + rc];
        }
        f4 z = {0.f, 0.f, 0.f, 0.f};
        f4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
        for (int i = 0; i < 4; ++i) C[(size_t(m) * 16 + 4 * g + i) * 16 + rc] = c[i];
    }
}

// the K = 16 form the library uses (v_mfma_f32_16x16x16_f16): A[16][16], B[16][16], accumulator started at -limit
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ void k_probe16(const _Float16 *A, const _Float16 *B, float *C, int n_mats, float c0) {
    const int lane = threadIdx.x & 63, g = lane >> 4, rc = lane & 15;
    for (int m = blockIdx.x; m < n_mats; m += gridDim.x) {
        h4 a, b;
        for (int j = 0; j < 4; ++j) {
            a[j] = A[(size_t(m) * 16 + rc) * 16 + 4 * g + j];
            b[j] = B[(size_t(m) * 16 + 4 * g + j) * 16 + rc];
        }
        f4 z = {c0, c0, c0, c0};
        f4 c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, z, 0, 0, 0);
        for (int i = 0; i < 4; ++i) C[(size_t(m) * 16 + 4 * g + i) * 16 + rc] = c[i];
    }
}

static float h2f(_Float16 h) { return float(h); }

int main() {
    const int NM = 4096;
    std::vector<_Float16> A(size_t(NM) * 16 * 32), B(size_t(NM) * 32 * 16);
    std::vector<float> C(size_t(NM) * 256);
    srand(12345);
    auto rnd = []() { return double(rand()) / RAND_MAX * 2.0 - 1.0; };
    // matrix 0: layout (small integers, exact); 1: inf in A; 2: f16 subnormals; 3..: screen-like sums with heavy cancellation
    for (int m = 0; m < NM; ++m) {
        for (int r = 0; r < 16; ++r)
            for (int k = 0; k < 32; ++k) {
                double v;
                if (m == 0) v = (r + 1) * ((k % 5) - 2);
                else if (m == 1) v = (r == 3 && k == 24) ? INFINITY : ((k == 24) ? 100.0 : (k < 8 ? rnd() : 0.0));
                else if (m == 2) v = (k < 16) ? ldexp(1.0 + (r & 3) * 0.25, -20 + (k & 3)) : 0.0;   // subnormal halves (below 2^-14)
                else v = 0.0;
                A[(size_t(m) * 16 + r) * 32 + k] = _Float16(v);
            }
        for (int k = 0; k < 32; ++k)
            for (int c = 0; c < 16; ++c) {
                double v;
                if (m == 0) v = (c + 1) * ((k % 3) - 1) + (k == 31 ? 7 : 0);
                else if (m == 1) v = (k == 24) ? 1.0 : (k < 8 ? rnd() : 0.0);
                else if (m == 2) v = (k < 16) ? 1.0 + c : 0.0;
                else v = 0.0;
                B[(size_t(m) * 32 + k) * 16 + c] = _Float16(v);
            }
        if (m >= 3) {
            // rows x (8 comps up to 64), columns y = x + small: T = |x|^2 + |y|^2 - 2 x.y with split halves, exactly the sieve's slots
            const double M = (m & 1) ? 63.0 : 40.0, spread = (m & 2) ? 0.5 : 6.0;
            double X[16][8], Y[16][8];
            for (int r = 0; r < 16; ++r)
                for (int k = 0; k < 8; ++k) X[r][k] = rnd() * M;
            for (int c = 0; c < 16; ++c)
                for (int k = 0; k < 8; ++k) Y[c][k] = X[c][k] + rnd() * spread;
            auto split = [](double x, _Float16 &h, _Float16 &l) {
                h = _Float16(float(x));
                float rest = float(x) - float(h);
                l = fabsf(rest) < 6.103515625e-05f ? _Float16(0.f) : _Float16(rest);
            };
            auto split3 = [](double n, _Float16 p[3]) {
                double rest = n;
                for (int q = 0; q < 3; ++q) {
                    float f = float(rest);
                    p[q] = fabsf(f) < 6.103515625e-05f ? _Float16(0.f) : _Float16(f);
                    rest -= double(float(p[q]));
                }
            };
            for (int r = 0; r < 16; ++r) {
                _Float16 h[8], l[8], np[3];
                double n = 0;
                for (int k = 0; k < 8; ++k) {
                    split(X[r][k], h[k], l[k]);
                    const double xh = double(h2f(h[k])) + double(h2f(l[k]));
                    n += xh * xh;
                }
                split3(n, np);
                _Float16 *a = &A[(size_t(m) * 16 + r) * 32];
                for (int k = 0; k < 8; ++k) a[k] = h[k], a[8 + k] = h[k], a[16 + k] = l[k];
                a[24] = np[0], a[25] = np[1], a[26] = np[2], a[27] = a[28] = a[29] = _Float16(1.f), a[30] = a[31] = _Float16(0.f);
            }
            for (int c = 0; c < 16; ++c) {
                _Float16 h[8], l[8], np[3];
                double n = 0;
                for (int k = 0; k < 8; ++k) {
                    split(Y[c][k], h[k], l[k]);
                    const double yh = double(h2f(h[k])) + double(h2f(l[k]));
                    n += yh * yh;
                }
                split3(n, np);
                _Float16 *b = &B[size_t(m) * 32 * 16];
                for (int k = 0; k < 8; ++k) {
                    b[(k) * 16 + c] = _Float16(-2.f * h2f(h[k]));
                    b[(8 + k) * 16 + c] = _Float16(-2.f * h2f(l[k]));
                    b[(16 + k) * 16 + c] = _Float16(-2.f * h2f(h[k]));
                }
                b[24 * 16 + c] = b[25 * 16 + c] = b[26 * 16 + c] = _Float16(1.f);
                b[27 * 16 + c] = np[0], b[28 * 16 + c] = np[1], b[29 * 16 + c] = np[2];
                b[30 * 16 + c] = b[31 * 16 + c] = _Float16(0.f);
            }
        }
    }
    _Float16 *dA, *dB;
    float *dC;
    hipMalloc(&dA, A.size() * 2), hipMalloc(&dB, B.size() * 2), hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, 0, dA, dB, dC, NM);
    if (hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
    // exact references
    int layout_bad = 0;
    double worst_rel = 0, worst_abs = 0, worst_terms = 0;
    double sub_expect = 0, sub_got = 0;
    for (int m = 0; m < NM; ++m)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                double s = 0, sa = 0;
                for (int k = 0; k < 32; ++k) {
                    const double a = double(h2f(A[(size_t(m) * 16 + r) * 32 + k])), b = double(h2f(B[(size_t(m) * 32 + k) * 16 + c]));
                    if (a != 0.0 && b != 0.0) s += a * b, sa += fabs(a * b);
                }
                const double got = C[(size_t(m) * 16 + r) * 16 + c];
                if (m == 0 && got != s) ++layout_bad;
                if (m == 1 && r < 5 && c < 2) printf("inf test r=%d c=%d: got %g expect %g\n", r, c, got, s);
                if (m == 2 && r == 1 && c == 2) sub_expect = s, sub_got = got;
                if (m >= 3) {
                    const double e = fabs(got - s);
                    if (e / sa > worst_rel) worst_rel = e / sa, worst_abs = e, worst_terms = sa;
                }
            }
    printf("layout mismatches (matrix 0): %d of 256\n", layout_bad);
    printf("subnormal f16 inputs: expect %.10g got %.10g (%s)\n", sub_expect, sub_got, sub_got == 0.0 ? "FLUSHED" : "kept");
    printf("screen-like sums (%d x 256): worst |err| / sum|terms| = %.3g = 2^%.2f  (abs %.3g at sum|terms| %.6g)\n", NM - 3, worst_rel, log2(worst_rel), worst_abs, worst_terms);
    // ---- the K = 16 form: rows x (8 components up to 64, rounded to float16), norms of the rounded vectors in three pieces, -limit in the accumulator
    {
        std::vector<_Float16> A16(size_t(NM) * 256), B16(size_t(NM) * 256);
        std::vector<float> C16(size_t(NM) * 256);
        const float limit = 480.0f;
        for (int m = 0; m < NM; ++m) {
            const double M = (m & 1) ? 63.0 : 40.0, spread = (m & 2) ? 0.5 : 6.0;
            double X[16][8], Y[16][8];
            for (int r = 0; r < 16; ++r)
                for (int k = 0; k < 8; ++k) X[r][k] = rnd() * M;
            for (int c = 0; c < 16; ++c)
                for (int k = 0; k < 8; ++k) Y[c][k] = X[c][k] + rnd() * spread;
            auto split3 = [](double n, _Float16 p[3]) {
                double rest = n;
                for (int q = 0; q < 3; ++q) {
                    float f = float(rest);
                    p[q] = fabsf(f) < 6.103515625e-05f ? _Float16(0.f) : _Float16(f);
                    rest -= double(float(p[q]));
                }
            };
            for (int r = 0; r < 16; ++r) {
                _Float16 np[3], *a = &A16[(size_t(m) * 16 + r) * 16];
                double n = 0;
                for (int k = 0; k < 8; ++k) a[k] = _Float16(float(X[r][k])), n += double(h2f(a[k])) * double(h2f(a[k]));
                split3(n, np);
                a[8] = np[0], a[9] = np[1], a[10] = np[2], a[11] = a[12] = a[13] = _Float16(1.f), a[14] = a[15] = _Float16(0.f);
            }
            for (int c = 0; c < 16; ++c) {
                _Float16 np[3], yh[8], *b = &B16[size_t(m) * 256];
                double n = 0;
                for (int k = 0; k < 8; ++k) yh[k] = _Float16(float(Y[c][k])), n += double(h2f(yh[k])) * double(h2f(yh[k])), b[k * 16 + c] = _Float16(-2.f * h2f(yh[k]));
                split3(n, np);
                b[8 * 16 + c] = b[9 * 16 + c] = b[10 * 16 + c] = _Float16(1.f);
                b[11 * 16 + c] = np[0], b[12 * 16 + c] = np[1], b[13 * 16 + c] = np[2], b[14 * 16 + c] = b[15 * 16 + c] = _Float16(0.f);
            }
        }
        _Float16 *dA16, *dB16;
        float *dC16;
        (void)hipMalloc(&dA16, A16.size() * 2), (void)hipMalloc(&dB16, B16.size() * 2), (void)hipMalloc(&dC16, C16.size() * 4);
        (void)hipMemcpy(dA16, A16.data(), A16.size() * 2, hipMemcpyHostToDevice), (void)hipMemcpy(dB16, B16.data(), B16.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe16, dim3(256), dim3(64), 0, 0, dA16, dB16, dC16, NM, -limit);
        if (hipMemcpy(C16.data(), dC16, C16.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
        double w_rel = 0, w_abs = 0, w_terms = 0;
        for (int m = 0; m < NM; ++m)
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c) {
                    double s = -double(limit), sa = double(limit);
                    for (int k = 0; k < 16; ++k) {
                        const double a = double(h2f(A16[(size_t(m) * 16 + r) * 16 + k])), b = double(h2f(B16[(size_t(m) * 16 + k) * 16 + c]));
                        s += a * b, sa += fabs(a * b);
                    }
                    const double e = fabs(double(C16[(size_t(m) * 16 + r) * 16 + c]) - s);
                    if (e / sa > w_rel) w_rel = e / sa, w_abs = e, w_terms = sa;
                }
        printf("K = 16 form with the accumulator at -limit (%d x 256 sums): worst |err| / sum|terms| = %.3g = 2^%.2f  (abs %.3g at sum|terms| %.6g)\n", NM, w_rel, log2(w_rel), w_abs, w_terms);
    }
    return layout_bad ? 2 : 0;
}
