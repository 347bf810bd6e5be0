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

inline unsigned cap_grid(long g, long cap) { return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g)); }

}  // namespace
