// Shared helpers for the librstnet_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

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

// Measurement knobs (A/B switches between kernel forms, ablation variants that skip work) exist ONLY in the tools build
// (`make ablation` -> librstnet_hip_ablation.so, -DRST_ABLATION, loaded by tools/ via RSTNET_LIB): the shipped library reads no
// environment variable, so nothing outside its arguments can change what it computes.
#ifdef RST_ABLATION
#include <cstdlib>
static inline int rst_knob(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
#else
static inline constexpr int rst_knob(const char*, int dflt) { return dflt; }
#endif

// ---- per-device host-side state ------------------------------------------------------------------------------------------------
// A process may drive several GPUs (one rank per GPU is the deployment, but the library does not assume it): kernel attributes
// (hipFuncSetAttribute applies to the current device's copy of a kernel), CU counts and residency answers are cached PER DEVICE.
#define RST_MAX_DEVICES 64
// cache slot of the current device, or -1 for an ordinal the tables do not cover: such a device is never cached (its attributes are
// set and its properties queried on every call) instead of sharing device 0's answers
static inline int rst_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev >= 0 && dev < RST_MAX_DEVICES ? dev : -1;
}
// `static RstOncePerDevice once; if (once.first()) { ...opt-in... }`: true the first time it is asked on each device.  Host threads
// driving different GPUs may get here together: the mask is one atomic fetch_or (two threads on the SAME device can at worst both
// see "first" -- the opt-ins are idempotent -- never neither).
struct RstOncePerDevice {
    std::atomic<unsigned long long> mask{0};
    bool first() {
        const int slot = rst_device_slot();
        if (slot < 0) return true;
        const unsigned long long bit = 1ull << slot;
        if (mask.load(std::memory_order_acquire) & bit) return false;
        return !(mask.fetch_or(bit, std::memory_order_acq_rel) & bit);
    }
};
// CUs of the current device
static inline int rst_cu_count() {
    static std::atomic<int> n[RST_MAX_DEVICES];
    const int slot = rst_device_slot();
    int v = slot >= 0 ? n[slot].load(std::memory_order_relaxed) : 0;
    if (v == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
        if (v <= 0) v = 256;
        if (slot >= 0) n[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}
// per-device cache cell of a residency answer (0 = not asked yet): `scratch` serves devices the tables do not cover (asked every time)
static inline signed char& rst_device_cell(signed char* table, int stride, signed char& scratch) {
    const int slot = rst_device_slot();
    if (slot < 0) { scratch = 0; return scratch; }
    return table[slot * stride];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RST_WAVE 64

__device__ __forceinline__ float rst_elu(float v) {
    // ELU, alpha = 1: x > 0 ? x : e^x - 1.  e^x as one v_exp_f32 of the fp32 product x * log2(e): for x <= 0 the product's rounding
    // moves e^x by <= |x| * 2^-24, so e^x - 1 stays within 7e-8 absolute of the exact value -- the same bound as a correctly rounded
    // expf(x) - 1 (4.5e-8; the subtraction's own rounding dominates both) at 5 VALU instructions instead of 14: the codec's
    // residual blocks spend more issue slots on ELU than on anything else but the MFMAs.
    return v > 0.0f ? v : (__builtin_amdgcn_exp2f(v * 1.4426950408889634f) - 1.0f);
}

__device__ __forceinline__ float rst_gelu(float v) {
    // exact (erf) GELU, F.gelu default
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// element (row, k) of a few-row fp32 operand in the packed order of skinny_f32.hip: [tile of 32 rows][Kp / 8][64 lanes = 32 (k % 2) + row % 32]
// [4 floats: (k % 8) / 2] -- shared with the producers that write that operand directly (the few-row GEMM's packed output, the
// attention step of codec_attn.hip)
__device__ __forceinline__ long f32_packed_index(int row, int k, int Kp) {
    return ((((long)(row >> 5) * (Kp >> 3) + (k >> 3)) * 64) + (k & 1) * 32 + (row & 31)) * 4 + ((k & 7) >> 1);
}

// row of element r (0..15) of a 32x32 MFMA accumulator held by `lane` (column = lane & 31)
__device__ __forceinline__ int rst_mfma32_row(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// U write-through partials ws[(k0 + u) * stride], u < U, read with relaxed agent-scope (L1-bypassing) loads that are all in flight at
// once; entries at k0 + u >= n repeat the last valid one (callers skip them).  A `for (k) s += atomic_load(...)` loop is compiled as
// one exposed round trip per term -- the last arriver of a split reduction then spends longer summing than the splits spent
// computing.
template <int U>
__device__ __forceinline__ void rst_load_partials(const float* ws, long stride, int k0, int n, float (&t)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
        t[u] = __hip_atomic_load(ws + (long)min(k0 + u, n - 1) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

