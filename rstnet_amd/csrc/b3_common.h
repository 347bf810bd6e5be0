// Three-plane bf16 arithmetic shared by the kernels that run fp32 contractions on the bf16 matrix instruction (gemm_win.hip,
// resblock_b3.hip): an fp32 number is EXACTLY hi + mid + lo with hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = rne_bf16(x - hi - mid)
// (8 + 8 + 8 significand bits; both subtractions are exact in fp32), so an fp32 product is the sum of nine bf16 products, each exact in
// the matrix core's fp32 accumulator.  Six are kept -- lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, added smallest first -- and the
// three dropped (mid*lo, lo*mid, lo*lo) are below 2^-23 |x||w|, one fp32 rounding of the product.
#pragma once
#include "rst_common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// two fp32 -> the two packed bf16 (round to nearest even) and the exact remainders
__device__ __forceinline__ unsigned b3_peel(f32x2& v) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    v[0] -= __uint_as_float(h << 16);
    v[1] -= __uint_as_float(h & 0xffff0000u);
    return h;
}

// plane pairs of the six products in the order they are accumulated (smallest terms first): A-side plane, B-side plane
#define B3_QA {2, 0, 1, 1, 0, 0}
#define B3_QB {0, 2, 1, 0, 1, 0}
