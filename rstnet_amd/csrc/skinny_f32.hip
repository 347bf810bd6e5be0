// Few-row fp32 GEMM of the codec's streaming steps (one 80 ms frame for up to 64 streams: M = B * T_out <= 128 rows):
// every Conv1d / ConvTranspose1d / Linear of a step is weight-bandwidth bound (1 - 33 MB of fp32 weights against a few
// rows), so it runs like the LM's skinny GEMM instead of the tiled gemm_win kernel:
//   * both operands in MFMA operand order [tile of 32 rows][K/8][64 lanes][4 floats] -- lane = 32 * (k % 2) + row % 32,
//     float e of a lane = k = 8 q + 2 e + (k % 2) -- so that one 16-byte lane load feeds four v_mfma_f32_32x32x2_f32 and a
//     wave-level load is one contiguous KB; weights are packed once, the activation WINDOWS (the overlapping-window A
//     operand of gemm_win: history / padding / ELU-on-load included) are gathered and packed per call by a small kernel;
//   * one workgroup = CT tiles of 32 output columns, its 8 waves split K and meet in LDS in a fixed order: deterministic,
//     no cross-workgroup reduction (the split-K hand-off through device-scope atomics costs more than these GEMMs);
//   * fp32 MFMA with k ascending per wave: products are exact fp32 fmaf chains like gemm_win's (sums differ only in order).
#include <hip/hip_runtime.h>

#include "rst_common.h"
#include "rst_kernels.h"

namespace {

// (f32_packed_index, the packed order of both operands: rst_common.h)
__global__ __launch_bounds__(256) void f32_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int K, int Kp) {
    const long total = (long)((N + 31) / 32) * 32 * (Kp / 2);            // (row, k pair) items
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / (Kp / 2)), k = (int)(idx % (Kp / 2)) * 2;
        float v0 = 0.f, v1 = 0.f;
        if (row < N) {
            if (k < K) v0 = w[(long)row * K + k];
            if (k + 1 < K) v1 = w[(long)row * K + k + 1];
        }
        wp[f32_packed_index(row, k, Kp)] = v0;
        wp[f32_packed_index(row, k + 1, Kp)] = v1;
    }
}

// rows of the packed activation buffer: the kernel is instantiated for 1, 2 or 4 row tiles
__host__ __device__ inline int sf_rows(int M) { return M <= 32 ? 32 : (M <= 64 ? 64 : 128); }

__device__ __forceinline__ float sf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// nn.LayerNorm statistics of one row by one wave -- `layernorm_kernel`'s arithmetic (fp32: the lane's sum in column order, the wave's
// sum in butterfly order, variance of the centred values with fmaf) -- and its application to one element.  Shared by the packing launch
// below and by the GEMM's row-major operand path, so that both give the bits of LayerNorm followed by the pack.
__device__ __forceinline__ float sf_ln_apply(float v, float mean, float rstd, float g, float b) { return (v - mean) * rstd * g + b; }

__device__ __forceinline__ void sf_row_stats(const float* xr, int K, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
    for (int i = lane * 4; i < K; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    mean = sf_wave_sum(s) / (float)K;
    float q = 0.f;
    for (int i = lane * 4; i < K; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
    }
    rstd = 1.0f / sqrtf(sf_wave_sum(q) / (float)K + eps);
}

// LayerNorm + pack in ONE launch (round 5): the plain-linear case of the packing below (no window, K = C) with the row's LayerNorm
// applied on the way, so the operand is bit-identical to LayerNorm followed by the pack.
// Every streamed transformer layer of the codec at more than two streams had a LayerNorm launch in front of each of these packs.
// (Round 6: plain linears read their rows row-major inside the GEMM itself -- SkinnyF32Params::xr -- and this launch is the A/B form.)
__global__ __launch_bounds__(256) void f32_pack_ln_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ xp, int M, int K, int Kp, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= sf_rows(M)) return;
    const long tile = (long)(m >> 5) * (Kp >> 3);
    if (m >= M) {            // pad rows of the batch tile: zeros
        for (int k = lane * 4; k < Kp; k += 256)
#pragma unroll
            for (int e = 0; e < 4; ++e) xp[((tile + ((k + e) >> 3)) * 64 + ((k + e) & 1) * 32 + (m & 31)) * 4 + (((k + e) & 7) >> 1)] = 0.f;
        return;
    }
    const float* xr = x + (long)m * K;
    if (K <= 1024) {
        // the row, gamma and beta in registers across the three passes: ONE memory round trip instead of three dependent ones (a launch
        // that lasts 4.7 us per call, 32 calls per 80 ms frame at 32 streams); the arithmetic -- and its order -- is sf_row_stats' and the loop's below
        f32x4 v[4], g4[4], b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = min(lane * 4 + 256 * j, K - 4);          // clamped: unconditional loads
            v[j] = *reinterpret_cast<const f32x4*>(xr + i);
            g4[j] = *reinterpret_cast<const f32x4*>(gamma + i);
            b4[j] = *reinterpret_cast<const f32x4*>(beta + i);
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane * 4 + 256 * j < K) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        const float mean = sf_wave_sum(s) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane * 4 + 256 * j < K) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q = fmaf(d, d, q); }
            }
        const float rstd = 1.0f / sqrtf(sf_wave_sum(q) / (float)K + eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane * 4 + 256 * j;
            if (i < Kp) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (i < K) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = sf_ln_apply(v[j][e], mean, rstd, g4[j][e], b4[j][e]);
                }
                const long base = (tile + (i >> 3)) * 64;
#pragma unroll
                for (int e = 0; e < 4; ++e) xp[(base + (e & 1) * 32 + (m & 31)) * 4 + (((i & 7) + e) >> 1)] = o[e];
            }
        }
        return;
    }
    float mean, rstd;
    sf_row_stats(xr, K, lane, eps, mean, rstd);
    for (int i = lane * 4; i < Kp; i += 256) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (i < K) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = sf_ln_apply(v[e], mean, rstd, g[e], b[e]);
        }
        // k = i .. i + 3 (i % 4 == 0): k & 1 alternates, (k & 7) >> 1 = (i & 7) / 2 + {0, 0, 1, 1}
        const long base = (tile + (i >> 3)) * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) xp[(base + (e & 1) * 32 + (m & 31)) * 4 + (((i & 7) + e) >> 1)] = o[e];
    }
}

// A(b, t, k) = xflat_b[(t * S - P) * C + k] with history / zero / replicate padding (the A operand of gemm_win_kernel), ELU on
// load if asked, written in packed order; item = (row, q, h): the 4 floats of one lane.
__global__ __launch_bounds__(256) void f32_pack_win_kernel(const SkinnyF32PackParams p) {
    const int M = p.B * p.T_out;
    const int M32 = sf_rows(M);          // rows past M are written as zeros
    const int TC = p.T_in * p.C, PC = p.P * p.C;
    const long total = (long)M32 * (p.Kp / 8) * 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int h = (int)(idx & 1);
        const int q = (int)((idx >> 1) % (p.Kp / 8));
        const int m = (int)((idx >> 1) / (p.Kp / 8));
        // the four loads are unconditional (source address chosen per lane, padding cleared by a mask the compiler cannot fold back
        // into a condition): a load under a per-lane condition is waited for on the spot, four round trips instead of one
        f32x4 v;
        {
            const int mm = min(m, M - 1);
            const int b = mm / p.T_out, t = mm - b * p.T_out;
            const long xo = (long)b * p.x_bstride, ho = (long)b * PC;
            const int f0 = (t * p.S - p.P) * p.C;
            const float* src[4];
            int msk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * q + 2 * e + h;
                const int f = f0 + k;
                bool ok = m < M && k < p.K;
                const float* s = p.x + xo + f;
                if (f < 0) {
                    if (p.hist) {
                        s = p.hist + ho + PC + f;
                    } else if (p.pad_mode == 1) {
                        int c = f % p.C;
                        if (c < 0) c += p.C;
                        s = p.x + xo + c;
                    } else {
                        ok = false;
                    }
                } else if (f >= TC) {
                    if (p.pad_mode == 1) s = p.x + xo + TC - p.C + f % p.C;
                    else ok = false;
                }
                src[e] = ok ? s : p.xp;        // any valid address: the value is masked away
                msk[e] = ok ? -1 : 0;
                asm volatile("" : "+v"(msk[e]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = __builtin_bit_cast(float, __builtin_bit_cast(int, *src[e]) & msk[e]);
                if (p.act_in == 1) a = rst_elu(a);     // ELU(0) = 0: padding stays zero
                v[e] = a;
            }
        }
        *reinterpret_cast<f32x4*>(p.xp + ((((long)(m >> 5) * (p.Kp >> 3) + q) * 64) + h * 32 + (m & 31)) * 4) = v;
    }
}

constexpr int SF_WAVES = 8;

// gridDim.y > 1: K is also split across workgroups (blockIdx.y owns a contiguous range of the 8-k chunks) -- a layer of 6-33 MB
// behind N / 32 = 16 or 32 workgroups streams at well under 1 TB/s (measured 62 us for the 33 MB 512 -> 1024 k16 convolution);
// the partial tiles go to `ws` with write-through stores, one arrival counter per column tile, and the LAST workgroup to arrive
// sums them in split order (deterministic) and runs the epilogue (the protocol of gemm_win's split-K: cdna_hip_programming.md G16).
// ROWS (round 6): the activation operand is NOT packed -- the rows of a plain linear are read row-major (p.xr [M][ldx], K % 8 == 0), a
// lane picking its four k of every 8-k chunk out of two 16-byte loads: no packing launch in front of the GEMM (64 rows, 512 x 512: 7.0 us
// against 8.2 us for pack + GEMM; the same bits).  A LayerNorm in front of the linear stays in the packing launch (f32_pack_ln_kernel):
// applying it here was built twice and measured slower both times -- statistics like the packing launch's (one wave per row, rows
// re-read): 15.9 us against 11.3 us for LayerNorm + pack followed by the GEMM; statistics from the K slices the waves hold anyway (two
// LDS exchanges): 10.8 us against 8.1 us after the row tiles had become workgroups.  A packing launch is 2.6 us in a dependent chain;
// nothing that puts a reduction in front of the matrix instructions is cheaper (profiles/r06_few_row_linear_probe.txt).
template <int NB, int CT, bool ROWS = false>
__global__ __launch_bounds__(64 * SF_WAVES) void gemm_skinny_f32_kernel(const SkinnyF32Params p) {
    __shared__ float red[SF_WAVES][NB * 32][33];
    __shared__ int sm_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = ROWS ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int tiles = (p.N + 31) / 32;
    const int tile0 = blockIdx.x * CT;
    const int chunks = p.Kp / 8;
    const int nsplit = gridDim.y;
    const int per_split = (chunks + nsplit - 1) / nsplit;
    const int c_lo = blockIdx.y * per_split, c_hi = min(chunks, c_lo + per_split);
    const int per = (max(c_hi - c_lo, 0) + SF_WAVES - 1) / SF_WAVES;
    const int s0 = c_lo + wave * per, s1 = min(c_hi, s0 + per);
    // gridDim.z > 1: the NB row tiles of this workgroup are tiles blockIdx.z * NB .. of the batch (rows mz .. mz + 32 NB - 1) -- 64 rows
    // as two one-tile workgroups per column tile cost what 32 rows do (the weights cross L2 twice, the launch is a round trip shorter)
    const int mz = blockIdx.z * NB * 32;
    const float* xq = p.xp + (long)blockIdx.z * NB * chunks * 256 + (long)lane * 4;
    const float* wt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wt[c] = p.wp + ((long)min(tile0 + c, tiles - 1) * chunks * 64 + lane) * 4;
    f32x16 acc[NB][CT];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;
    // epilogue operands of this thread's outputs (EP per column tile): requested before the K loop from clamped addresses, used after it
    // -- fetched in the epilogue each group of them is one more exposed round trip of a launch that lasts a handful
    constexpr int EP = NB * 32 * 32 / (64 * SF_WAVES);
    float rpre[CT][EP], bpre[CT][EP], spre[CT][EP];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < EP; ++q) {
            const int idx = tid + q * 64 * SF_WAVES;
            const int m = min(mz + (idx >> 5), p.M - 1), n = min((tile0 + c) * 32 + (idx & 31), p.N - 1);
            rpre[c][q] = p.res ? p.res[(long)m * p.ldy + n] : 0.f;
            bpre[c][q] = p.bias ? p.bias[n] : 0.f;
            spre[c][q] = p.scale ? p.scale[n] : 1.0f;
        }
    constexpr int UN = NB * CT <= 2 ? 4 : 2;
    // One pass of the K loop: UN chunks of weights and activations requested together, then multiplied.  ROWS: the requests of the
    // wave's FIRST pass (weights from HBM, the rows, gamma / beta) go out before the LayerNorm statistics, whose own round trip and
    // reductions then run under them -- the rows' raw values wait in registers for mean / rstd.
    f32x4 a[UN][CT], bx[UN][NB];
    f32x4 xlo[ROWS ? UN : 1][NB], xhi[ROWS ? UN : 1][NB];
    const float* xrow[NB];
    const bool odd = (lane >> 5) != 0;
    if (ROWS) {
#pragma unroll
        for (int t = 0; t < NB; ++t) xrow[t] = p.xr + (long)min(mz + 32 * t + (lane & 31), p.M - 1) * p.ldx;      // rows past M: row M - 1, cleared below
    }
    auto issue = [&](const int s) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = s + u < s1;
            const long so = (long)(s + u) * 256;
#pragma unroll
            for (int c = 0; c < CT; ++c)
                a[u][c] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wt[c] + so)) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (!ROWS) {
#pragma unroll
                for (int t = 0; t < NB; ++t)
                    bx[u][t] = ok ? *reinterpret_cast<const f32x4*>(xq + (long)t * chunks * 256 + so) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                // chunk s + u = columns 8 (s + u) .. + 7; this lane's four are k = 8 (s + u) + 2 e + (lane >> 5).  (Past the wave's range:
                // the last chunk once more -- a valid address, unconditional loads -- against zero weights.)
                const int k0 = 8 * min(s + u, chunks - 1);
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    xlo[u][t] = *reinterpret_cast<const f32x4*>(xrow[t] + k0);
                    xhi[u][t] = *reinterpret_cast<const f32x4*>(xrow[t] + k0 + 4);
                }
            }
        }
    };
    if (ROWS) issue(s0);
    for (int s = s0; s < s1; s += UN) {
        if (!ROWS || s != s0) issue(s);
        if (ROWS) {
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const f32x4 lo = xlo[u][t], hi = xhi[u][t];
                    f32x4 x4 = odd ? f32x4{lo[1], lo[3], hi[1], hi[3]} : f32x4{lo[0], lo[2], hi[0], hi[2]};
                    if (s + u >= s1 || mz + 32 * t + (lane & 31) >= p.M) x4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    bx[u][t] = x4;
                }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NB; ++t)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(bx[u][t][e], a[u][c][e], acc[t][c], 0, 0, 0);
    }
    // activations are the MFMA "A" side: acc[t][c][e] = C[m = 32 t + row(e, lane)][n = 32 (tile0 + c) + (lane & 31)]
    const int i = lane & 31;
    const int M = p.M;
    auto epilogue = [&](float v, int m, int n, int c, int q) {
        if (p.bias) v += bpre[c][q];
        if (p.act_out == 1) v = rst_gelu(v);
        const long o = (long)m * p.ldy + n;
        if (p.res) v = rpre[c][q] + spre[c][q] * v;
        if (p.act_out == 2) v = rst_elu(v);
        // Np_out: the result is the next few-row GEMM's operand -- written straight in its packed order (no packing launch between them)
        if (p.Np_out) p.y[f32_packed_index(m, n, p.Np_out)] = v;
        else p.y[o] = v;
    };
    // packed output: the pad rows of the last batch tile must read as zeros in the consumer
    auto pad_zero = [&](int m, int n) {
        if (p.Np_out && m >= M && m < mz + NB * 32 && n < p.N) p.y[f32_packed_index(m, n, p.Np_out)] = 0.f;
    };
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t * 32 + rst_mfma32_row(e, lane)][i] = acc[t][c][e];
        __syncthreads();
        const int n0 = (tile0 + c) * 32;
#pragma unroll
        for (int q = 0; q < EP; ++q) {
            const int idx = tid + q * 64 * SF_WAVES;
            const int ml = idx >> 5, nl = idx & 31;
            const int m = mz + ml, n = n0 + nl;
            if (m < M && n < p.N) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < SF_WAVES; ++w) v += red[w][ml][nl];
                if (nsplit == 1) epilogue(v, m, n, c, q);
                else __hip_atomic_store(p.ws + ((long)blockIdx.y * M + m) * p.N + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (nsplit == 1) {
                pad_zero(m, n);
            }
        }
    }
    if (nsplit == 1) return;
    // every storing wave drains its write-through stores, ONE lane bumps the tile's counter; the last arriver combines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* const cnt = p.counters + blockIdx.z * gridDim.x + blockIdx.x;       // one counter per (row block, column tile)
        const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm_last = prev == (unsigned)nsplit - 1;
        if (sm_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!sm_last) return;
    // the partials of ALL of this thread's outputs are requested together (the writers' stores left their L2: a round trip to memory)
    // and summed in split order (deterministic), four splits at a time
    float v[CT][EP];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < EP; ++q) v[c][q] = 0.f;
    for (int ks = 0; ks < nsplit; ks += 4) {
        float t[CT][EP][4];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < EP; ++q) {
                const int idx = tid + q * 64 * SF_WAVES;
                const int m = min(mz + (idx >> 5), M - 1), n = min((tile0 + c) * 32 + (idx & 31), p.N - 1);      // past the edge: clamped, unused
                rst_load_partials<4>(p.ws + (long)m * p.N + n, (long)M * p.N, ks, nsplit, t[c][q]);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ks + u < nsplit) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int q = 0; q < EP; ++q) v[c][q] += t[c][q][u];
            }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < EP; ++q) {
            const int idx = tid + q * 64 * SF_WAVES;
            const int m = mz + (idx >> 5), n = (tile0 + c) * 32 + (idx & 31);
            if (m < M && n < p.N) epilogue(v[c][q], m, n, c, q);
            else pad_zero(m, n);
        }
}

inline unsigned sf_grid(long total, long cap) {
    long g = (total + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

int rst_launch_skinny_f32_pack_weight(const float* w, float* wp, int N, int K, hipStream_t stream) {
    RST_REQUIRE(w && wp && N > 0 && K > 0, "skinny_f32_pack_weight: bad arguments");
    const int Kp = (K + 7) / 8 * 8;
    const long total = (long)((N + 31) / 32) * 32 * (Kp / 2);
    hipLaunchKernelGGL(f32_pack_weight_kernel, dim3(sf_grid(total, 8192)), dim3(256), 0, stream, w, wp, N, K, Kp);
    return rst_check_launch("skinny_f32_pack_weight");
}

int rst_launch_skinny_f32_pack_win(const SkinnyF32PackParams& p, hipStream_t stream) {
    RST_REQUIRE(p.x && p.xp && p.B >= 1 && p.T_out >= 1 && p.T_in >= 0 && p.C > 0 && p.K > 0 && p.S > 0 && p.P >= 0 && p.Kp % 8 == 0 &&
                    p.Kp >= p.K && (long)p.B * p.T_out <= 128,
                "skinny_f32_pack_win: bad arguments (M = B * T_out <= 128 required; M=%ld K=%d Kp=%d)", (long)p.B * p.T_out, p.K, p.Kp);
    const long total = (long)sf_rows(p.B * p.T_out) * (p.Kp / 8) * 2;
    hipLaunchKernelGGL(f32_pack_win_kernel, dim3(sf_grid(total, 4096)), dim3(256), 0, stream, p);
    return rst_check_launch("skinny_f32_pack_win");
}

int rst_launch_skinny_f32_pack_ln(const float* x, const float* gamma, const float* beta, float* xp, int M, int K, float eps, hipStream_t stream) {
    RST_REQUIRE(x && gamma && beta && xp && M >= 1 && M <= 128 && K > 0 && K % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)gamma % 16 == 0 &&
                    (uintptr_t)beta % 16 == 0,
                "skinny_f32_pack_ln: bad arguments (1 <= M <= 128, K %% 4 == 0, 16-byte aligned rows; M=%d K=%d)", M, K);
    const int Kp = (K + 7) / 8 * 8;
    hipLaunchKernelGGL(f32_pack_ln_kernel, dim3((sf_rows(M) + 3) / 4), dim3(256), 0, stream, x, gamma, beta, xp, M, K, Kp, eps);
    return rst_check_launch("skinny_f32_pack_ln");
}

int rst_skinny_f32_split_plan_impl(int M, int N, int K) {
    // few workgroups (N / 32 column tiles x M / 32 row tiles) against MBs of weights: split K until ~128-256 workgroups run, each wave
    // keeping >= 4 of the 8-k chunks.  K <= 512 is not split at all: 8 chunks per wave are one pass of the K loop, and the hand-off of
    // the partials (drain, counter, read-back) costs more than it saves (64 rows, 1536 x 512: 8.2 us unsplit, 10.1 us at 4;
    // profiles/r06_few_row_linear_probe.txt has the sweep behind this rule, 16 - 128 rows x 11 shapes x 6 splits).
    if (M < 1 || M > 128) return 1;
    const int tiles = (N + 31) / 32, chunks = (K + 7) / 8, nb = (M + 31) / 32;
    if (tiles >= 512 || chunks <= 64) return 1;
    int s = 1;
    while (tiles * nb * s * 2 <= 256 && chunks / (2 * s * SF_WAVES) >= 4 && s < 32) s *= 2;
    return s;
}

int rst_launch_gemm_skinny_f32(const SkinnyF32Params& p, hipStream_t stream) {
    RST_REQUIRE((p.xp || p.xr) && p.wp && p.y && p.M >= 1 && p.M <= 128 && p.N > 0 && p.Kp > 0 && p.Kp % 8 == 0 && p.act_out >= 0 && p.act_out <= 2,
                "gemm_skinny_f32: bad arguments (1 <= M <= 128, Kp %% 8 == 0; M=%d Kp=%d)", p.M, p.Kp);
    const int tiles = (p.N + 31) / 32;
    const int split = p.split_k > 1 ? p.split_k : 1;
    RST_REQUIRE(split == 1 || (p.ws && p.counters && tiles < 512), "gemm_skinny_f32: split-K needs the scratch buffers (and < 512 column tiles)");
    RST_REQUIRE(p.Np_out == 0 || (p.Np_out == p.N && p.N % 8 == 0 && !p.res), "gemm_skinny_f32: packed output needs N %% 8 == 0, Np_out = N and no residual (N=%d)", p.N);
    const dim3 block(64 * SF_WAVES);
    const int nb = (p.M + 31) / 32;
    // row tiles as workgroups of their own (gridDim.z) instead of a loop inside one (round 6): 64 rows then cost what 32 do -- pack + GEMM
    // 512 x 512: 10.5 -> 8.2 us, 128 rows 13.4 -> 8.4 us.  tools build: RST_SF_ZTILE=0 keeps the round-2 form (all row tiles in one workgroup)
    static const int zt_knob = rst_knob("RST_SF_ZTILE", 1);
    const bool zt = zt_knob != 0 && nb > 1;
    if (p.xr) {
        // row-major rows of a plain linear: no packing launch
        RST_REQUIRE(!p.xp && p.ldx >= p.Kp && p.ldx % 4 == 0 && (uintptr_t)p.xr % 16 == 0,
                    "linear_few_rows: rows must be 16-byte aligned with K %% 8 == 0 and ldx %% 4 == 0 (K=%d ldx=%d)", p.Kp, p.ldx);
        if (zt && tiles < 512) hipLaunchKernelGGL((gemm_skinny_f32_kernel<1, 1, true>), dim3(tiles, split, nb), block, 0, stream, p);
        else if (nb == 1) hipLaunchKernelGGL((gemm_skinny_f32_kernel<1, 1, true>), dim3(tiles, split), block, 0, stream, p);
        else if (nb == 2) hipLaunchKernelGGL((gemm_skinny_f32_kernel<2, 1, true>), dim3(tiles, split), block, 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_f32_kernel<4, 1, true>), dim3(tiles, split), block, 0, stream, p);
        return rst_check_launch("linear_few_rows");
    }
    if (zt && tiles < 512) {
        hipLaunchKernelGGL((gemm_skinny_f32_kernel<1, 1>), dim3(tiles, split, nb), block, 0, stream, p);
    } else if (nb == 1) {
        if (tiles >= 512) hipLaunchKernelGGL((gemm_skinny_f32_kernel<1, 2>), dim3((tiles + 1) / 2), block, 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_f32_kernel<1, 1>), dim3(tiles, split), block, 0, stream, p);
    } else if (nb == 2) {
        hipLaunchKernelGGL((gemm_skinny_f32_kernel<2, 1>), dim3(tiles, split), block, 0, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_skinny_f32_kernel<4, 1>), dim3(tiles, split), block, 0, stream, p);
    }
    return rst_check_launch("gemm_skinny_f32");
}
