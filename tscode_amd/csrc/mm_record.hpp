// mm_record.hpp -- the float16 records of a structure's descriptor that the matrix-core screen reads (mm.hpp has the scheme and its
// error bound); written by k_open_rows (rmsd.hpp) by position, every pass.
#pragma once
#include "common.hpp"

namespace tsc {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int MM_KD = 8;                    // components per family (sieve.hpp: KD), two families
constexpr int MM_REC_HALVES = 32;           // float16 per structure in the column records (64 bytes): per family [-2 x0 .. -2 x7 | 1 1 1 n0 | n1 n2 0 0] in four
                                            // chunks of 4, the two families' chunks SIDE BY SIDE (chunk g of family 0, chunk g of family 1: one 16-byte load
                                            // gives a lane its K slots 4 g .. 4 g + 3 of both)
constexpr float MM_FLUSH = 6.103515625e-05f;  // 2^-14, the smallest normal float16: smaller norm pieces are dropped (and bounded) rather than left to subnormals

// sigma: the power of two that takes the largest |component| of the run (bit pattern of a non-negative float) into [32, 64)
__host__ __device__ inline float mm_scale(unsigned dmax_bits) {
    int field = 259 - int((dmax_bits >> 23) & 0xffu);   // 127 + 5 - (e - 127)
    field = field < 1 ? 1 : (field > 254 ? 254 : field);
    unsigned bits = unsigned(field) << 23;
    float f;
    memcpy(&f, &bits, sizeof(f));
    return f;
}

// The records of one structure from its 16 stored components (d[2k + fam], sieve.hpp): x = the nearest float16 of sigma * d, and
// |x|^2 (of the ROUNDED vector: exact in float64) in three float16 pieces that leave 2^-14 at most.
__device__ inline void mm_write_record(const float d[2 * MM_KD], float sigma, _Float16 *__restrict__ col_rec) {
#pragma unroll
    for (int fam = 0; fam < 2; ++fam) {
        f16x4 c0, c1, c2, c3;
        double n = 0.0;
#pragma unroll
        for (int k = 0; k < MM_KD; ++k) {
            const _Float16 h = _Float16(sigma * d[2 * k + fam]);
            const double xr = double(float(h));
            n = fma(xr, xr, n);
            const _Float16 m2 = _Float16(-2.0f * float(h));   // (exact: a power of two; |2 x| <= 128)
            if (k < 4) c0[k] = m2;
            else c1[k - 4] = m2;
        }
        _Float16 np[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {   // n <= 8 * 64^2 = 32768
            const float f = float(n);
            np[q] = fabsf(f) < MM_FLUSH ? _Float16(0.0f) : _Float16(f);
            n -= double(float(np[q]));
        }
        const _Float16 one = _Float16(1.0f), zero = _Float16(0.0f);
        c2 = f16x4{one, one, one, np[0]};
        c3 = f16x4{np[1], np[2], zero, zero};
        f16x4 *o = reinterpret_cast<f16x4 *>(col_rec + 4 * fam);
        o[0] = c0, o[2] = c1, o[4] = c2, o[6] = c3;
    }
}

}  // namespace tsc
