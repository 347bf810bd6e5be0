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
// DPP forms (no LDS crossbar round trip per step: a __shfl_xor butterfly is 6 dependent ds_bpermute, ~0.3 us for a lone wave).
// row16_*: over the 16 lanes of a DPP row, result in every lane of the row; wave_*_fast: over the wave, result uniform.  Call with
// all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
#define DPP_QUAD_XOR1 0xB1        /* quad_perm [1, 0, 3, 2] */
#define DPP_QUAD_XOR2 0x4E        /* quad_perm [2, 3, 0, 1] */
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<DPP_QUAD_XOR1>(v);
    v += dpp_f<DPP_QUAD_XOR2>(v);
    v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_sum_fast(float v) {
    v = row16_sum(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max_fast(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}
// sum over aligned groups of GS lanes (4 <= GS <= 64, a power of two), result in every lane of the group
__device__ __forceinline__ float group_sum(float v, int GS) {
    v += dpp_f<DPP_QUAD_XOR1>(v);
    v += dpp_f<DPP_QUAD_XOR2>(v);
    if (GS >= 8) v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
    if (GS >= 16) v += dpp_f<DPP_ROW_MIRROR>(v);
    if (GS >= 32) v += __shfl_xor(v, 16);
    if (GS >= 64) v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }

// Ring slot -> position map of RingKVCache.complete (modules/transformer.py:254-278) incl. the `delta <= 0` quirk (Q1);
// returns whether `slot` is visible to the query at position `pos` (= the step just appended).
__device__ __forceinline__ bool ring_visible_at(int slot, long pos, int cap, int context, long end_offset, int end_index) {
    const int delta = slot - end_index;
    long pk = delta <= 0 ? end_offset + delta : end_offset + delta - cap;
    if (slot >= end_offset) pk = -1;
    const long dl = pos - pk;
    bool ok = slot < cap && pk >= 0 && dl >= 0;
    if (context > 0) ok = ok && dl < context;
    return ok;
}
__device__ __forceinline__ bool ring_visible(int slot, long pos, int cap, int context, long end_offset) {
    return ring_visible_at(slot, pos, cap, context, end_offset, (int)(end_offset % cap));
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
