// TEST HELPER (not part of librstnet_hip.so): a kernel that holds `wgs` workgroups of `lds_bytes` LDS each on the device for `ms`
// milliseconds of wall clock, so that a test can take CUs away from the persistent frame launches (csrc/persist.h) and check that
// the repair launch keeps the outputs right.  Built by __graft_entry__.build() into tests/helpers/_build/libocc.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, int* sink) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    lds[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += lds[(threadIdx.x + acc) & 255];
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0x7fffffff) *sink = acc;      // never true: keeps the loop
}

extern "C" int occ_cu_count(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    return prop.multiProcessorCount;
}

// 0 on success.  `stream`: a hipStream_t (NOT the stream the persistent launch under test runs on).
extern "C" int occ_launch(int wgs, int lds_bytes, int ms, int* sink_dev, void* stream) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;   // 100 MHz
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) {
            (void)hipGetLastError();
            return -2;
        }
        attr = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (long long)ms * khz, sink_dev);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
