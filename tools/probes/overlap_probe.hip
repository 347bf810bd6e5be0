// Do the matrix pipe and the vector ALU of ONE SIMD overlap when they are fed by two different waves?  Workgroups of 512 threads, one per
// CU (two waves per SIMD): waves 0 .. 3 run a pure v_mfma_f32_32x32x16_bf16 stream (4 independent accumulators), waves 4 .. 7 a pure VALU
// stream of one kind.  Times each stream alone and both together; "overlap" = (t_mfma + t_valu - t_both) / min(t_mfma, t_valu).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/overlap_probe.hip -o /tmp/overlap_probe && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// KIND: 0 v_fma_f32, 1 v_pk_add_f32, 2 v_exp_f32, 3 v_cvt_pk_bf16_f32 + shifts (the split), 4 v_cndmask / v_cmp (ELU's select)
template <int KIND>
__device__ __forceinline__ void valu_stream(int iters, float seed, float* sink) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) x[i] = fmaf(x[i], 1.0001f, 0.5f);
                if (KIND == 2) x[i] = __builtin_amdgcn_exp2f(x[i]);
                if (KIND == 4) x[i] = x[i] > 0.25f ? x[(i + 1) & 7] : x[i] + 1.f;
            }
            if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f32x2 v = {x[i], x[i + 1]};
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(v));
                    x[i] = v[0]; x[i + 1] = v[1];
                }
            }
            if (KIND == 3) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    f32x2 v = {x[i], x[i + 1]};
                    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
                    x[i] -= __uint_as_float(h << 16);
                    x[i + 1] -= __uint_as_float(h & 0xffff0000u);
                    x[i] += 1.f;
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) sink[0] = s;
}

template <int KIND>
__global__ __launch_bounds__(512) void probe(int mfma_iters, int valu_iters, float* sink) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (mfma_iters == 0) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x + e); b[e] = (short)(0x3f00 + e); }
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 123.456f) sink[1] = s;
    } else {
        if (valu_iters == 0) return;
        valu_stream<KIND>(valu_iters, threadIdx.x * 1e-3f, sink);
    }
}

// the same wave: K plain VALU instructions (v_fma_f32, independent of the matrix instructions) behind every matrix instruction
template <int K>
__global__ __launch_bounds__(512) void probe_same(int iters, float* sink) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x + e); b[e] = (short)(0x3f00 + e); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) x[k & 7] = fmaf(x[k & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) sink[1] = s;
}

template <int K>
void test_same(float* sink, int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 10000;       // two waves per SIMD: 2 x 10000 x 32 matrix instructions per SIMD
    probe_same<K><<<grid, 512>>>(iters / 10, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe_same<K><<<grid, 512>>>(iters, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("same wave, %d VALU behind every matrix instruction: %.3f ms (%.1f cycles per matrix instruction at 2.1 GHz, two waves per SIMD)\n", K, ms,
           ms * 2.1e6 / (2.0 * iters * 32));
}

template <int KIND>
float run(int mi, int vi, float* sink, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<KIND><<<grid, 512>>>(mi / 10, vi / 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<grid, 512>>>(mi, vi, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND>
void test(const char* name, float* sink, int grid) {
    const int mi = 20000;                          // 32 MFMAs per iteration: 20000 x 32 x 32 cycles = 20.5 M cycles
    // calibrate the VALU stream to about the same duration
    const float t1 = run<KIND>(0, 2000, sink, grid);
    const float tm = run<KIND>(mi, 0, sink, grid);
    const int vi = (int)(2000.0f * tm / t1);
    const float tv = run<KIND>(0, vi, sink, grid);
    const float tb = run<KIND>(mi, vi, sink, grid);
    printf("%-28s mfma alone %.3f ms | valu alone %.3f ms (%d iters, %.2f cycles/instr at 2.1 GHz) | both %.3f ms | overlap %.2f\n", name, tm, tv, vi,
           tv * 2.1e6 / ((double)vi * 128), tb, (tm + tv - tb) / (tm < tv ? tm : tv));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount;
    float* sink;
    hipMalloc(&sink, 64);
    test<0>("v_fma_f32", sink, grid);
    test<1>("v_pk_add_f32", sink, grid);
    test<2>("v_exp_f32", sink, grid);
    test<3>("cvt_pk_bf16 + unpack + sub", sink, grid);
    test<4>("v_cmp + v_cndmask", sink, grid);
    test_same<0>(sink, grid);
    test_same<2>(sink, grid);
    test_same<4>(sink, grid);
    test_same<6>(sink, grid);
    test_same<8>(sink, grid);
    test_same<12>(sink, grid);
    return 0;
}
