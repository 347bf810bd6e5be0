// What does an in-launch all-to-all hand-off cost at BATCH 32?  (the persistent form of the batch-32 depth chain.)
// The depth transformer of the GPT / Moshi-style models at 32 streams: every op hands a [32 x 1024] fp32 vector (128 KB) from the
// workgroups that produced its column slices to EVERY workgroup (the next op's activations).  At batch 1 the same hand-off is 8 KB of
// {tag, value} granules (1.3 - 1.7 us, tools/probes/temporal_frame_phases.py); the question is whether the 32-row form stays below the
// ~8 - 9 us a graph-replayed launch of these GEMMs costs today (profiles/r05_gpt_kernel_stats.csv: 7.9 - 9.3 us per launch).
// This probe runs ONLY the hand-offs -- a chain of dependent all-to-all exchanges by 256 resident workgroups of 512 threads, no weight
// streaming, no matrix instructions in between -- in three transports, and prints the time per exchange:
//   V1  8-byte {epoch, f32} granules, one per value            (32768 granules = 256 KB swept per workgroup and exchange)
//   V2  16-byte {epoch, f32, f32, f32} granules                 (10923 granules = 171 KB)
//   V3  plain write-through payload + one flag per producer     (1 KB of flags polled, then 128 KB read with 16-byte loads)
// A workgroup's 128 values of exchange e are a function of what it gathered in exchange e - 1 (a true dependence chain); the result is
// checked on the host.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/handoff_b32_probe.hip -o tools/probes/handoff_b32_probe && tools/probes/handoff_b32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int G = 256, NT = 512, ROWS = 32, COLS = 1024, NV = ROWS * COLS, PER_WG = NV / G;      // 128 values per workgroup

// what workgroup `wg` publishes in exchange e, given the vector of exchange e - 1 in LDS (x): a cheap function of a few gathered values
__device__ __forceinline__ float produce(const float* x, int wg, int i, int e) {
    const int j = (wg * PER_WG + i * 97 + e * 13) & (NV - 1);
    return 0.5f * x[j] + 0.25f * x[(j + 4099) & (NV - 1)] + (float)((wg + i) & 7) * 0.125f;
}

template <int V>
__global__ __launch_bounds__(NT) void chain(u64* g8, u32x4* g16, float* payload, unsigned* flags, int iters, float* out, unsigned* fail) {
    extern __shared__ __attribute__((aligned(16))) float x[];     // [NV] the current vector
    const int tid = threadIdx.x, wg = blockIdx.x;
    for (int i = tid; i < NV; i += NT) x[i] = (float)(i & 15) * 0.0625f;
    __syncthreads();
    for (int e = 1; e <= iters; ++e) {
        const int par = e & 1;
        // ---- publish this workgroup's 128 values of exchange e
        if (V == 1) {
            u64* dst = g8 + (long)par * NV + wg * PER_WG;
            if (tid < PER_WG) __hip_atomic_store(dst + tid, ((u64)(unsigned)e << 32) | __float_as_uint(produce(x, wg, tid, e)), RLX);
        } else if (V == 2) {
            // 128 values = 43 granules of 3 (the last one padded)
            u32x4* dst = g16 + (long)par * (G * 43) + wg * 43;
            if (tid < 43) {
                u32x4 gq;
                gq[0] = (unsigned)e;
                for (int k = 0; k < 3; ++k) gq[1 + k] = 3 * tid + k < PER_WG ? __float_as_uint(produce(x, wg, 3 * tid + k, e)) : 0u;
                // one 16-byte write-through store (sc0 sc1): observed untorn on gfx950 (MI355X_MICROARCH.md), the tag rides in the same store
                u32x4* a = dst + tid;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a), "v"(gq) : "memory");
            }
        } else {
            float* dst = payload + (long)par * NV + wg * PER_WG;
            if (tid < PER_WG) __hip_atomic_store(dst + tid, produce(x, wg, tid, e), RLX);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + par * G + wg, (unsigned)e, RLX);
        }
        __syncthreads();        // x is rewritten below
        // ---- gather the whole vector of exchange e
        long long t0 = 0;
        if (V == 1) {
            const u64* src = g8 + (long)par * NV;
            for (int base = 0; base < NV; base += 32 * NT) {
                u64 v[32];
                while (true) {
                    bool all = true;
#pragma unroll
                    for (int j = 0; j < 32; ++j) v[j] = __hip_atomic_load(src + base + j * NT + tid, RLX);
#pragma unroll
                    for (int j = 0; j < 32; ++j) all = all && (unsigned)(v[j] >> 32) == (unsigned)e;
                    if (all) break;
                    if (t0 == 0) t0 = wall_clock64();
                    if (wall_clock64() - t0 > 20000000) { atomicAdd(fail, 1u); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int j = 0; j < 32; ++j) x[base + j * NT + tid] = __uint_as_float((unsigned)v[j]);
            }
        } else if (V == 2) {
            const u32x4* src = g16 + (long)par * (G * 43);
            constexpr int NGR = G * 43;                     // 11008 granules
            u32x4 v[22];
            while (true) {
                bool all = true;
#pragma unroll
                for (int j = 0; j < 22; ++j) {
                    const u32x4* a = src + min(j * NT + tid, NGR - 1);
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[j]) : "v"(a) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 22; ++j) all = all && v[j][0] == (unsigned)e;
                if (all) break;
                if (t0 == 0) t0 = wall_clock64();
                if (wall_clock64() - t0 > 20000000) { atomicAdd(fail, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int j = 0; j < 22; ++j) {
                const int i = j * NT + tid;
                if (i < NGR) {
                    const int w = i / 43, q = i - w * 43;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (3 * q + k < PER_WG) x[w * PER_WG + 3 * q + k] = __uint_as_float(v[j][1 + k]);
                }
            }
        } else {
            while (true) {
                const unsigned f = tid < G ? __hip_atomic_load(flags + par * G + tid, RLX) : (unsigned)e;
                if (__syncthreads_and(f == (unsigned)e)) break;
                if (t0 == 0) t0 = wall_clock64();
                if (wall_clock64() - t0 > 20000000) { atomicAdd(fail, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            // the payload: write-through stores drained before the flag -> L1-bypassing loads see it
            const u32x4* src = reinterpret_cast<const u32x4*>(payload + (long)par * NV);
            u32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const u32x4* a = src + j * NT + tid;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[j]) : "v"(a) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) *reinterpret_cast<u32x4*>(x + 4 * (j * NT + tid)) = v[j];
        }
        __syncthreads();
    }
    if (wg == 0)
        for (int i = tid; i < NV; i += NT) out[i] = x[i];
}

// ---- a model of ONE LAYER-STEP of the batch-32 depth transformer as a persistent chain (the shapes of the GPT / Moshi depth layers:
// E = 1024, hidden 2816): in-projection (12 columns per workgroup) | attention (small hand-off) | out-projection (4) | gated ffn-in (11
// pairs) | ffn-out (4, its 2816-wide input gathered in three chunks).  Per op: the workgroup's weight slice is requested BEFORE the
// hand-off (8 x 16 bytes per lane and wave: they do not depend on activations), the input vector is gathered as 8-byte granules into
// LDS, eight waves split K over v_mfma_f32_32x32x16_bf16 (two per step: the hi and lo halves of the activations), the partial tiles meet
// in LDS, n_c x 32 values are published.  Values are arbitrary (timing only); every op depends on the previous one's granules.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct OpShape { int k_in; int n_out; int steps; };      // values gathered per row / values published per row / K / 16

__global__ __launch_bounds__(NT) void layer_chain(u64* g8, const u32x4* wpool, long wpool_n16, int iters, float* out, unsigned* fail) {
    extern __shared__ __attribute__((aligned(16))) unsigned xs[];     // [32 * 1024] gathered words; the partial tiles alias it afterwards
    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    // ops of a layer-step: {in-proj, attention, out-proj, ffn-in, ffn-out chunk 0, 1, 2}
    const OpShape ops[7] = {{1024, 12, 64}, {12, 4, 0}, {1024, 4, 64}, {1024, 11, 64}, {1024, 0, 64}, {1024, 0, 64}, {768, 4, 48}};
    unsigned epoch = 0;
    long wofs = ((long)wg * NT + tid) % wpool_n16;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float carry = 0.f;
    // the input of the very first op: this workgroup's 4 columns of x, epoch 1
    if (tid < 4 * ROWS) __hip_atomic_store(g8 + (long)(3072 * ROWS) + (long)(wg * 4 + tid / ROWS) * ROWS + tid % ROWS, ((u64)1u << 32) | __float_as_uint(0.5f), RLX);
    for (int it = 0; it < iters; ++it) {
        for (int o = 0; o < 7; ++o) {
            const OpShape op = ops[o];
            // weights of this op's slice for this wave: steps / 8 pieces per lane, requested before the hand-off
            u32x4 w[8];
            const int per = op.steps / 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                w[j] = u32x4{0u, 0u, 0u, 0u};
                if (j < per) { w[j] = __builtin_nontemporal_load(wpool + wofs); wofs += (long)G * NT; if (wofs >= wpool_n16) wofs -= wpool_n16; }
            }
            // hand-off: the input of ops 0, 2, 3 is a fresh vector (epoch + 1 of its producer); chunks 1, 2 of ffn-out re-use epochs
            const int n_in = op.k_in * ROWS;                          // granules to gather
            const bool fresh = o != 5 && o != 6;                      // (chunks 1 / 2 of the hidden vector were published with chunk 0)
            if (fresh) ++epoch;
            const u64* src = g8 + (long)(epoch & 1) * (3072 * ROWS);
            long long t0 = 0;
            for (int base = 0; base < n_in; base += 32 * NT) {
                u64 v[32];
                while (true) {
                    bool all = true;
#pragma unroll
                    for (int j = 0; j < 32; ++j) v[j] = __hip_atomic_load(src + min(base + j * NT + tid, n_in - 1) + (o >= 5 ? (o - 4) * 1024 * ROWS : 0), RLX);
#pragma unroll
                    for (int j = 0; j < 32; ++j) all = all && (unsigned)(v[j] >> 32) == epoch;
                    if (all) break;
                    if (t0 == 0) t0 = wall_clock64();
                    if (wall_clock64() - t0 > 20000000) { atomicAdd(fail, 1u); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (base + j * NT + tid < n_in) xs[base + j * NT + tid] = (unsigned)v[j];
            }
            __syncthreads();
            // multiply: wave `wave` takes steps wave * per ..; A = the gathered words (hi / lo halves emulated by two reads), B = the weights
            for (int j = 0; j < per; ++j) {
                const int st = wave * per + j;
                const u32x4 a = *reinterpret_cast<const u32x4*>(&xs[(st * 64 + lane) * 4 & (32 * 1024 - 4)]);
                const u32x4 b = *reinterpret_cast<const u32x4*>(&xs[((st * 64 + lane) * 4 + 2048) & (32 * 1024 - 4)]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, w[j]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, w[j]), acc, 0, 0, 0);
            }
            if (op.n_out > 0) {
                __syncthreads();                                      // the partial tiles alias the gathered vector
                float* red = reinterpret_cast<float*>(xs);
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[e];
                __syncthreads();
                const int n_pub = op.n_out * ROWS;
                const int next_epoch = epoch + 1;
                // ffn-in publishes the WHOLE hidden vector's slice (11 of 2816 columns): its consumer gathers it in three chunks
                u64* dst = g8 + (long)(next_epoch & 1) * (3072 * ROWS);
                if (tid < n_pub) {
                    float s = carry;
#pragma unroll
                    for (int w8 = 0; w8 < 8; ++w8) s += red[(w8 * 16 + (tid & 15)) * 64 + (tid >> 4 & 63)];
                    carry = s * 1e-30f;
                    const int col = o == 3 ? wg * 11 + tid / ROWS : wg * op.n_out + tid / ROWS;      // this workgroup's columns
                    const int width = o == 0 ? 3072 : (o == 1 ? 1024 : (o == 3 ? 2816 : 1024));
                    const int row = tid % ROWS;
                    // layout [k][row]: value (row, col) at col * 32 + row -- so that the first K * 32 granules are the first K columns
                    if (col < width) __hip_atomic_store(dst + (long)col * ROWS + row, ((u64)(unsigned)next_epoch << 32) | __float_as_uint(s), RLX);
                }
                // columns nobody owns in this model (3072 / 256 = 12 exact; 1024 / 256 = 4 exact; 2816 / 256 = 11 exact) -- none
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            }
            __syncthreads();
        }
    }
    if (wg == 0 && tid == 0) out[0] = carry;
}

static void run_layer(int iters) {
    u64* g8; u32x4* wpool; float* out; unsigned* fail;
    const long wpool_bytes = 1L << 30;                                // 1 GiB: the chain streams it like a frame's 1.27 GB of depth weights
    CK(hipMalloc(&g8, 2L * 3072 * ROWS * 8)); CK(hipMalloc(&wpool, wpool_bytes)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&fail, 4));
    CK(hipMemset(wpool, 0x3c, wpool_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned nfail = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(g8, 0, 2L * 3072 * ROWS * 8)); CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(layer_chain, dim3(G), dim3(NT), 32 * 1024 * 4, 0, g8, wpool, wpool_bytes / 16, iters, out, fail);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned f;
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        nfail += f;
    }
    printf("layer-step model (in-proj | attention | out-proj | ffn-in | ffn-out in 3 chunks): %7.2f us per layer-step (%d layer-steps, best of 4)%s\n",
           best * 1e3f / iters, iters, nfail ? "  TIME-OUTS" : "");
    printf("   the graph-replayed launch chain spends 5 launches on it: 5 x 11.3 us all-in = 56 us (gpt_b32: 3.38 ms per frame / ~300 launches,\n"
           "   profiles/r05_gpt_kernel_stats.csv: 8.97 us average kernel time + boundaries)\n");
    (void)hipFree(g8); (void)hipFree(wpool); (void)hipFree(out); (void)hipFree(fail);
}

static void reference(int iters, std::vector<float>& x) {
    x.resize(NV);
    for (int i = 0; i < NV; ++i) x[i] = (float)(i & 15) * 0.0625f;
    std::vector<float> y(NV);
    for (int e = 1; e <= iters; ++e) {
        for (int wg = 0; wg < G; ++wg)
            for (int i = 0; i < PER_WG; ++i) {
                const int j = (wg * PER_WG + i * 97 + e * 13) & (NV - 1);
                y[wg * PER_WG + i] = 0.5f * x[j] + 0.25f * x[(j + 4099) & (NV - 1)] + (float)((wg + i) & 7) * 0.125f;
            }
        x.swap(y);
    }
}


template <int V>
static void run(const char* name, int iters) {
    u64* g8; u32x4* g16; float* payload; unsigned* flags; float* out; unsigned* fail;
    CK(hipMalloc(&g8, 2L * NV * 8)); CK(hipMalloc(&g16, 2L * G * 43 * 16)); CK(hipMalloc(&payload, 2L * NV * 4)); CK(hipMalloc(&flags, 2 * G * 4));
    CK(hipMalloc(&out, NV * 4)); CK(hipMalloc(&fail, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(chain<V>), hipFuncAttributeMaxDynamicSharedMemorySize, NV * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(g8, 0, 2L * NV * 8)); CK(hipMemset(g16, 0, 2L * G * 43 * 16)); CK(hipMemset(flags, 0, 2 * G * 4)); CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain<V>, dim3(G), dim3(NT), NV * 4, 0, g8, g16, payload, flags, iters, out, fail);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<float> got(NV), ref;
    unsigned nfail;
    CK(hipMemcpy(got.data(), out, NV * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nfail, fail, 4, hipMemcpyDeviceToHost));
    reference(iters, ref);
    int bad = 0;
    for (int i = 0; i < NV; ++i) bad += got[i] != ref[i];
    printf("%-72s %7.2f us per exchange (%d exchanges, best of 4 launches)  %s\n", name, best * 1e3f / iters, iters,
           bad || nfail ? "WRONG RESULT / TIME-OUT" : "result exact");
    if (bad || nfail) printf("   mismatches %d, time-outs %u\n", bad, nfail);
    (void)hipFree(g8); (void)hipFree(g16); (void)hipFree(payload); (void)hipFree(flags); (void)hipFree(out); (void)hipFree(fail);
}

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("%s, %d CUs; all-to-all hand-off of a [32 x 1024] fp32 vector between %d resident workgroups of %d threads\n", prop.name, prop.multiProcessorCount, G, NT);
    const int iters = 400;
    run<1>("V1  8-byte {epoch, f32} granules (256 KB swept per workgroup)", iters);
    run<2>("V2  16-byte {epoch, 3 x f32} granules (172 KB swept per workgroup)", iters);
    run<3>("V3  write-through payload + one flag per producer (1 KB polled, 128 KB read)", iters);
    run_layer(96);
    return 0;
}
