// Shared helpers for the librstnet_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RST_OK 0
#define RST_ERR_INVALID_ARG (-1)
#define RST_ERR_UNSUPPORTED (-2)
#define RST_ERR_LAUNCH (-3)

// error string of the calling thread (set by rst_fail, read by rst_last_error)
void rst_set_error(const char* fmt, ...);
int rst_check_launch(const char* what);

#define RST_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            rst_set_error(__VA_ARGS__);   \
            return RST_ERR_INVALID_ARG;   \
        }                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RST_WAVE 64

__device__ __forceinline__ float rst_elu(float v) {
    // ATen CPU elu: x > 0 ? x : exp(x) - 1   (alpha = 1)
    return v > 0.0f ? v : (expf(v) - 1.0f);
}

__device__ __forceinline__ float rst_gelu(float v) {
    // exact (erf) GELU, F.gelu default
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// row of element r (0..15) of a 32x32 MFMA accumulator held by `lane` (column = lane & 31)
__device__ __forceinline__ int rst_mfma32_row(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
