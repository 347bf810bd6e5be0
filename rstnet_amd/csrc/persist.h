// In-launch all-to-all hand-offs of the persistent frame kernels (lm_depth.hip, codec_tr.hip): an op's output vector lives in a
// global array of 8-byte {epoch tag, fp32 value} granules, each written by ONE relaxed agent-scope (sc1, write-through) store;
// a consuming workgroup sweeps the granules it needs with relaxed agent-scope loads until every tag equals the op's epoch and
// stages the values in LDS (cdna_hip_programming.md Guideline 16, form R2: the data is the flag -- no fence, no dispatch-order or
// placement assumption).  The workspace is zeroed by a memset node in front of every launch (epochs count from 1 inside a
// launch); every spin is bounded: a timeout sets the workgroup's `dead` flag (later waits do not spin again) and ORs a code into
// the caller's status word, so the launch always terminates.
//
// Safety net (round 3): the launches need every workgroup resident at once, which a shared device does not promise.  Each launcher
// (1) sizes the grid so that EVERY workgroup publishes in the all-to-all ops that separate two writes of a buffer (a workgroup that
// owned no rows there could lag and find its buffer overwritten), (2) refuses shapes the occupancy query says cannot be resident,
// and (3) enqueues, right behind the persistent launch, the SAME kernel as ONE workgroup (`SOLO`): it returns at once when
// status[0] == 0 and otherwise recomputes the whole frame alone -- a single workgroup depends on nobody, so it cannot time out --
// overwriting the outputs, then bumps status[1] (frames repaired), ORs the codes into status[2] and clears status[0].  Wrong
// outputs therefore never leave the stream; the host reads status[1] now and then and retires the persistent path for a device
// that keeps failing (a repaired frame costs the time-outs plus ~30 ms).
#pragma once
#include "lm_common.h"

namespace {

constexpr int DF_THREADS = 256, DF_WAVES = 4;
constexpr long long DF_TIMEOUT_TICKS = 10000000;   // of the 100 MHz wall clock (s_memrealtime): a lost hand-off costs 0.1 s, once per workgroup
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
// workgroup take part (up to GP granules per thread and pass, all requested before any is examined).  ONE sweep in flight per
// workgroup, re-issued when it returns: both variations measured in round 4 were slower (profiles/r04_handoff_polling.txt) -- a
// second sweep in flight (half the detection delay on paper) doubles 2-6 TB/s of fabric traffic that the publishing stores and the
// next op's weight prefetch queue behind (+1 to +4 us per hand-off); watching one sentinel granule per thread first and sweeping
// when it turns adds a round trip (+0.3 to +0.7 us per hand-off).
template <int GP, typename Map>
__device__ __forceinline__ void df_gather(const u64* g, int n, unsigned epoch, float* dst, Map map, DfShared& sh, unsigned* status, unsigned code) {
    const int tid = threadIdx.x;
    __syncthreads();        // dst may still be read by the rows of the previous op (another wave of this workgroup)
    for (int base = 0; base < n; base += GP * DF_THREADS) {
        u64 v[GP];
        long long t0 = 0;
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
            // time-based bound: how long a poll takes depends on what else the chip is doing (a count of polls measured 10x apart)
            if (t0 == 0) t0 = wall_clock64();
            if (*(volatile int*)&sh.dead || wall_clock64() - t0 > DF_TIMEOUT_TICKS) {
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

// End of a SOLO (repair) launch: the frame was recomputed by this one workgroup.
__device__ __forceinline__ void df_solo_done(unsigned* status) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned code = __hip_atomic_load(status, DF_RLX);
        __hip_atomic_fetch_add(status + 1, 1u, DF_RLX);
        __hip_atomic_fetch_or(status + 2, code, DF_RLX);
        __hip_atomic_store(status, 0u, DF_RLX);
    }
}

// Largest grid (<= `cus`) in which every workgroup (4 waves, row r -> wave r mod 4G) owns a row of an op with `min_rows` rows.
inline int df_grid_for_rows(int cus, int min_rows) {
    const int g = (min_rows + DF_WAVES - 1) / DF_WAVES;
    return g < cus ? (g < 1 ? 1 : g) : cus;
}

}  // namespace
