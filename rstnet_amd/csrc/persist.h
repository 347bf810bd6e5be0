// In-launch all-to-all hand-offs of the persistent frame kernels (lm_depth.hip, codec_tr.hip): an op's output vector lives in a
// global array of 8-byte {epoch tag, fp32 value} granules, each written by ONE relaxed agent-scope (sc1, write-through) store;
// a consuming workgroup sweeps the granules it needs with relaxed agent-scope loads until every tag equals the op's epoch and
// stages the values in LDS (cdna_hip_programming.md Guideline 16, form R2: the data is the flag -- no fence, no dispatch-order or
// placement assumption).  The workspace is zeroed by a memset node in front of every launch (epochs count from 1 inside a
// launch); every spin is bounded: a timeout sets the workgroup's `dead` flag (later waits do not spin again) and ORs a code into
// the caller's status word, so the launch always terminates.
#pragma once
#include "lm_common.h"

namespace {

constexpr int DF_THREADS = 256, DF_WAVES = 4;
constexpr unsigned DF_SPIN_LIMIT = 1u << 20;
constexpr int DF_HDR_FLOATS = 512;     // DfShared in the first KB of the header, the sampler's scratch in the second
typedef unsigned long long u64;
#define DF_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ void df_publish(u64* g, unsigned epoch, float v) {
    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), DF_RLX);
}

struct DfShared {
    float red[DF_WAVES];
    long tok[2];
    float tokf[2];
    int dead;              // a wait of this workgroup timed out: later waits do not spin again
    int pad;
};

// Sweep `n` granules (source index of item i = map(i)) until every tag == epoch; values -> dst[i] (LDS).  All threads of the
// workgroup take part (up to GP granules per thread and pass, all requested before any is examined).
template <int GP, typename Map>
__device__ __forceinline__ void df_gather(const u64* g, int n, unsigned epoch, float* dst, Map map, DfShared& sh, unsigned* status, unsigned code) {
    const int tid = threadIdx.x;
    __syncthreads();        // dst may still be read by the rows of the previous op (another wave of this workgroup)
    for (int base = 0; base < n; base += GP * DF_THREADS) {
        u64 v[GP];
        unsigned spins = 0;
        while (true) {
            bool all = true;
#pragma unroll
            for (int j = 0; j < GP; ++j) {
                const int i = base + j * DF_THREADS + tid;
                v[j] = i < n ? __hip_atomic_load(g + map(i), DF_RLX) : ((u64)epoch << 32);
            }
#pragma unroll
            for (int j = 0; j < GP; ++j) all = all && (unsigned)(v[j] >> 32) == epoch;
            if (all) break;
            if (*(volatile int*)&sh.dead || ++spins > DF_SPIN_LIMIT) {
                sh.dead = 1;
                atomicOr(status, code);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            const int i = base + j * DF_THREADS + tid;
            if (i < n) dst[i] = __uint_as_float((unsigned)v[j]);
        }
    }
    __syncthreads();
}

}  // namespace
