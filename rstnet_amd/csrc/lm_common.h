// Helpers shared by the LM decode-step translation units (lm_step.hip, lm_attn.hip, lm_sample.hip, lm_skinny.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "rst_common.h"
#include "rst_kernels.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }

// Ring slot -> position map of RingKVCache.complete (modules/transformer.py:254-278) incl. the `delta <= 0` quirk (Q1);
// returns whether `slot` is visible to the query at position `pos` (= the step just appended).
__device__ __forceinline__ bool ring_visible(int slot, long pos, int cap, int context, long end_offset) {
    const int end_index = (int)(end_offset % cap);
    const int delta = slot - end_index;
    long pk = delta <= 0 ? end_offset + delta : end_offset + delta - cap;
    if (slot >= end_offset) pk = -1;
    const long dl = pos - pk;
    bool ok = slot < cap && pk >= 0 && dl >= 0;
    if (context > 0) ok = ok && dl < context;
    return ok;
}

// ---- MFMA-ordered operand layout of the skinny GEMM (lm_skinny.hip) -- shared with the producers that emit it directly
// [tile of 32 rows][K/16][64 lanes = 32 * ((k / 8) % 2) + row % 32][8 bf16]
__device__ __forceinline__ long packed_index(int row, int k, int K) {
    return ((((long)(row >> 5) * (K >> 4) + (k >> 4)) * 64) + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
}

// 8 fp32 -> bf16 hi (truncated: an exact fp32 prefix, so x - hi is exact) and bf16 lo (residual rounded half-up): x = hi + lo
// to 2^-17.  v_perm_b32 packs the upper halves of two dwords in one instruction.
__device__ __forceinline__ void split_hi_lo8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned ua = __float_as_uint(v[2 * j]), ub = __float_as_uint(v[2 * j + 1]);
        hi[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                 // {hi16(b), hi16(a)}
        const float ra = v[2 * j] - __uint_as_float(ua & 0xffff0000u), rb = v[2 * j + 1] - __uint_as_float(ub & 0xffff0000u);
        lo[j] = __builtin_amdgcn_perm(__float_as_uint(rb) + 0x8000u, __float_as_uint(ra) + 0x8000u, 0x07060302u);
    }
}

// writes 8 consecutive k (k % 8 == 0) of batch row `row` into the hi and lo planes of a packed activation buffer
__device__ __forceinline__ void store_packed8(unsigned short* xp, long plane_elems, int row, int k, int K, const float (&v)[8]) {
    u32x4 hi, lo;
    split_hi_lo8(v, hi, lo);
    const long at = packed_index(row, k, K);
    *reinterpret_cast<u32x4*>(xp + at) = hi;
    *reinterpret_cast<u32x4*>(xp + plane_elems + at) = lo;
}

inline unsigned cap_grid(long g, long cap) { return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g)); }

}  // namespace
