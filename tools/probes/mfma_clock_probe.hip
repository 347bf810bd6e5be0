// Sustained fp32-MFMA rate and shader clock of the device under a pure v_mfma_f32_32x32x2_f32 load, with and without a
// background stream of 16-byte global loads (what the matrix pipe can deliver to a GEMM at the clock the power limit allows).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_clock_probe.hip -o /tmp/mfma_clock_probe && /tmp/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LOADS>   // 16-byte loads per 32 MFMAs per lane (the GEMM's k-tile has 4)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void probe(const float* src, long n_f4, float* sink,
                                                                                        unsigned long long* t, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f;
    f32x4 v[LOADS > 0 ? LOADS : 1];
    for (int j = 0; j < (LOADS > 0 ? LOADS : 1); ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    long pos = ((long)blockIdx.x * 256 + threadIdx.x) % n_f4;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LOADS > 0) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) s += v[j][0];
            acc[0][0] += s * 1e-30f;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) {
                v[j] = reinterpret_cast<const f32x4*>(src)[pos];
                pos += 256L * gridDim.x;
                if (pos >= n_f4) pos -= n_f4;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) {
        t[2 * blockIdx.x] = c1 - c0;
        t[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int LOADS>
void run(const float* src, long n_f4, float* sink, unsigned long long* t, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<LOADS><<<grid, 256>>>(src, n_f4, sink, t, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<LOADS><<<grid, 256>>>(src, n_f4, sink, t, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < grid; ++i) {
        cyc += h[2 * i];
        real += h[2 * i + 1];
    }
    const double flops = (double)grid * 4 * iters * 32 * 4096.0;
    // s_memrealtime ticks at 100 MHz
    printf("loads/32mfma=%d  %.3f ms  %.1f TFLOP/s  s_memtime/s_memrealtime = %.3f -> %.0f MHz if memtime counts shader clocks\n", LOADS, ms,
           flops / ms / 1e9, cyc / real, cyc / real * 100.0);
    printf("    MFMA cycles needed per SIMD at 64/instr: %.0f ; wall us %.1f -> %.0f MHz minimum shader clock\n", 3.0 * iters * 32 * 64,
           ms * 1e3, 3.0 * iters * 32 * 64 / (ms * 1e3));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = 3 * prop.multiProcessorCount;
    printf("%s: %d CUs, clockRate %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const long n_f4 = (1L << 30) / 16 * 2;   // 2 GiB stream
    float *src, *sink;
    unsigned long long* t;
    hipMalloc(&src, n_f4 * 16);
    hipMemset(src, 0, n_f4 * 16);
    hipMalloc(&sink, 64);
    hipMalloc(&t, 2 * grid * 8);
    const int iters = 200000;
    run<0>(src, n_f4, sink, t, grid, iters);
    run<4>(src, n_f4, sink, t, grid, iters);
    run<8>(src, n_f4, sink, t, grid, iters);
    run<0>(src, n_f4, sink, t, grid, iters);
    return 0;
}
