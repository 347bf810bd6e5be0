// Batch-1 (small-batch) decode step of the RQ-Transformer: every op of one temporal / depth transformer step at T = 1.
//
// At batch 1 the step is pure weight streaming (14.8 GB of bf16 weights per 80 ms frame for the Moshi-7B shape), so
// the GEMV is built for HBM: 16-byte non-temporal weight loads, four rows per wave in flight, the (normalised /
// gated) activation vector staged ONCE per workgroup in LDS as fp32, fp32 accumulation, wave-level reduction.
// RMSNorm and the SiLU gate are prologues of the consuming GEMV and the residual add is its epilogue, so a
// transformer layer is 5 launches: qkv GEMV | attention (lm_attn.hip) | out-proj GEMV (+res) | ffn-in GEMV | ffn-out GEMV
// (+res).  Activations stay fp32 (>= the reference's bf16), weights bf16.  This file: the GEMV (bf16 weights for the LM, fp32
// weights for the codec's one/two-position linears), the embedding sum and RMSNorm.
#include "lm_common.h"

namespace {

constexpr int GEMV_WAVES = 4;

// 8 consecutive k of one weight row: 16 bytes of bf16, or 32 bytes of fp32 (the codec's weights)
template <bool F32W> struct WChunk;
template <> struct WChunk<false> {
    u32x4 v;
    __device__ __forceinline__ void zero() { v = u32x4{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ void load(const void* w, long elem) {
        v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(static_cast<const unsigned short*>(w) + elem));
    }
    __device__ __forceinline__ float dot(const f32x4& x0, const f32x4& x1, float a) const {
        a = fmaf(bf16_lo(v[0]), x0[0], a); a = fmaf(bf16_hi(v[0]), x0[1], a);
        a = fmaf(bf16_lo(v[1]), x0[2], a); a = fmaf(bf16_hi(v[1]), x0[3], a);
        a = fmaf(bf16_lo(v[2]), x1[0], a); a = fmaf(bf16_hi(v[2]), x1[1], a);
        a = fmaf(bf16_lo(v[3]), x1[2], a); a = fmaf(bf16_hi(v[3]), x1[3], a);
        return a;
    }
};
template <> struct WChunk<true> {
    f32x4 a0, a1;
    __device__ __forceinline__ void zero() { a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0; }
    __device__ __forceinline__ void load(const void* w, long elem) {
        const float* q = static_cast<const float*>(w) + elem;
        a0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
        a1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q + 4));
    }
    __device__ __forceinline__ float dot(const f32x4& x0, const f32x4& x1, float a) const {
        a = fmaf(a0[0], x0[0], a); a = fmaf(a0[1], x0[1], a); a = fmaf(a0[2], x0[2], a); a = fmaf(a0[3], x0[3], a);
        a = fmaf(a1[0], x1[0], a); a = fmaf(a1[1], x1[1], a); a = fmaf(a1[2], x1[2], a); a = fmaf(a1[3], x1[3], a);
        return a;
    }
};

// y[b][n] = (res ? res[b][n] : 0) + (scale ? scale[n] : 1) * act(bias[n] + sum_k xs[b][k] * W[n][k]),   xs = prologue(x)
// FUSED instances (B <= 2, bf16) also carry prologue 4 (short-ring attention) and 5 (embedding + RMSNorm); they are kept apart
// so that their register footprint (the K / V rows of 8 ring slots) does not cost the plain instances their occupancy.
template <int B, int RPW, bool F32W, bool FUSED = false>
__global__ __launch_bounds__(64 * GEMV_WAVES) void gemv_kernel(const GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [B][K]
    __shared__ float red[GEMV_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;

    // ---- the first weight chunk of this workgroup's first row group is requested BEFORE the activation prologue: the two
    // dependent load chains (x -> norm -> LDS, and W) then overlap instead of adding up (small GEMVs are latency-bound)
    // gate_out (RPW == 2): the wave's two rows are (q, N/2 + q) = (u_q, v_q) of a stacked gated layer, one output silu(u) * v
    const int half = p.N / 2;
    const int groups = p.gate_out ? (half + GEMV_WAVES - 1) / GEMV_WAVES : (p.N + RPW * GEMV_WAVES - 1) / (RPW * GEMV_WAVES);
    auto row_of = [&](int grp, int r) {
        const int q = grp * GEMV_WAVES + wave;
        return p.gate_out ? r * half + min(q, half - 1) : min(q * RPW + r, p.N - 1);
    };
    WChunk<F32W> wpre[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        wpre[r].zero();
        if (blockIdx.x < groups && lane * 8 < K) wpre[r].load(p.w, (long)row_of(blockIdx.x, r) * K + lane * 8);
    }
    // the residual of the first row group is requested up front as well: loaded in the epilogue it is one more dependent
    // memory round trip at the very end of a kernel whose whole life is a few microseconds
    float rpre[RPW][B];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) {
            rpre[r][b] = 0.f;
            const int n = (blockIdx.x * GEMV_WAVES + wave) * RPW + r;
            if (p.res && !p.gate_out && lane == 0 && n < p.N) rpre[r][b] = p.res[(long)b * p.ldy + n];
        }

    // ---- prologue: stage the activation vector(s) in LDS
    if (p.prologue == 1) {           // RMSNorm: x * alpha * rsqrt(eps + mean(x^2))   (modules/transformer.py:34-46)
        for (int b = 0; b < B; ++b) {
            constexpr int XR = 16;   // elements kept in registers between the two passes (K <= 4096); the rest is re-read
            float xr[XR];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int k = tid + i * 64 * GEMV_WAVES;
                xr[i] = k < K ? p.x[(long)b * p.ldx + k] : 0.f;
                s = fmaf(xr[i], xr[i], s);
            }
            for (int k = tid + XR * 64 * GEMV_WAVES; k < K; k += 64 * GEMV_WAVES) { const float v = p.x[(long)b * p.ldx + k]; s = fmaf(v, v, s); }
            s = wave_sum(s);
            __syncthreads();
            if (lane == 0) red[wave] = s;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_WAVES; ++w) tot += red[w];
            const float r = 1.0f / sqrtf(p.eps + tot / (float)K);
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int k = tid + i * 64 * GEMV_WAVES;
                if (k < K) xs[b * K + k] = xr[i] * (p.alpha[k] * r);
            }
            for (int k = tid + XR * 64 * GEMV_WAVES; k < K; k += 64 * GEMV_WAVES) xs[b * K + k] = p.x[(long)b * p.ldx + k] * (p.alpha[k] * r);
        }
    } else if (p.prologue == 3) {    // nn.LayerNorm(K, eps) with affine gamma = alpha, beta (two-pass statistics, biased variance)
        for (int b = 0; b < B; ++b) {
            float s = 0.f;
            for (int k = tid; k < K; k += 64 * GEMV_WAVES) s += p.x[(long)b * p.ldx + k];
            s = wave_sum(s);
            __syncthreads();
            if (lane == 0) red[wave] = s;
            __syncthreads();
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_WAVES; ++w) mean += red[w];
            mean /= (float)K;
            float v = 0.f;
            for (int k = tid; k < K; k += 64 * GEMV_WAVES) { const float d = p.x[(long)b * p.ldx + k] - mean; v = fmaf(d, d, v); }
            v = wave_sum(v);
            __syncthreads();
            if (lane == 0) red[wave] = v;
            __syncthreads();
            float var = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_WAVES; ++w) var += red[w];
            const float rstd = 1.0f / sqrtf(var / (float)K + p.eps);
            for (int k = tid; k < K; k += 64 * GEMV_WAVES) xs[b * K + k] = (p.x[(long)b * p.ldx + k] - mean) * rstd * p.alpha[k] + p.beta[k];
        }
    } else if (FUSED && p.prologue == 4) {    // attention over the short ring (modules/transformer.py:376-416 with cap <= 8, no rope)
        // Thread = 4 consecutive dims of one head; the D/4 lanes of a head are neighbours inside a wave, so a score is a
        // butterfly over lane bits < D/4.  K / V of all cap slots are requested before the position scalar is known
        // (they are L2 hits: the ring was written by earlier launches); the new step's k / v come from the qkv row.
        const int D = p.at_D, lph = D >> 2, cap = p.at_cap;
        const float sc_scale = 1.0f / sqrtf((float)D);
        for (int b = 0; b < B; ++b) {
            const float* row = p.x + (long)b * p.ldx;
            for (int e = tid * 4; e < K; e += 4 * 64 * GEMV_WAVES) {
                const int h = e / D, d = e - h * D;
                const f32x4 q = *reinterpret_cast<const f32x4*>(row + e);
                const f32x4 kn = *reinterpret_cast<const f32x4*>(row + K + e);
                const f32x4 vn = *reinterpret_cast<const f32x4*>(row + 2 * K + e);
                const long base = (((long)b * p.at_H + h) * cap) * D + d;
                f32x4 kk[8], vv[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    kk[s] = f32x4{0.f, 0.f, 0.f, 0.f}; vv[s] = kk[s];
                    if (s < cap) {
                        kk[s] = *reinterpret_cast<const f32x4*>(p.at_k + base + (long)s * D);
                        vv[s] = *reinterpret_cast<const f32x4*>(p.at_v + base + (long)s * D);
                    }
                }
                const long pos = *p.at_pos;
                const int slot_cur = (int)(pos % cap);
                float sc[8], m = -INFINITY;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bool cur = s == slot_cur;
                    if (cur) { kk[s] = kn; vv[s] = vn; }
                    float dd = fmaf(kk[s][3], q[3], fmaf(kk[s][2], q[2], fmaf(kk[s][1], q[1], kk[s][0] * q[0])));
                    for (int o = 1; o < lph; o <<= 1) dd += __shfl_xor(dd, o);
                    const bool ok = s < cap && ring_visible(s, pos, cap, p.at_context, pos + 1);
                    sc[s] = ok ? dd * sc_scale : -INFINITY;
                    m = fmaxf(m, sc[s]);
                }
                float l = 0.f;
                f32x4 o4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float pw = sc[s] == -INFINITY ? 0.f : expf(sc[s] - m);
                    l += pw;
                    o4[0] = fmaf(pw, vv[s][0], o4[0]); o4[1] = fmaf(pw, vv[s][1], o4[1]);
                    o4[2] = fmaf(pw, vv[s][2], o4[2]); o4[3] = fmaf(pw, vv[s][3], o4[3]);
                }
                const float inv = 1.0f / l;     // the new step itself is always visible: l > 0
                *reinterpret_cast<f32x4*>(xs + b * K + e) = f32x4{o4[0] * inv, o4[1] * inv, o4[2] * inv, o4[3] * inv};
                if (blockIdx.x == 0) {          // ring append (RingKVCache.complete: index_copy_ at end_offset % capacity)
                    *reinterpret_cast<f32x4*>(p.at_k + base + (long)slot_cur * D) = kn;
                    *reinterpret_cast<f32x4*>(p.at_v + base + (long)slot_cur * D) = vn;
                }
            }
        }
    } else if (FUSED && p.prologue == 5) {    // ScaledEmbedding lookup + add + RMSNorm (models/model.py:411-417 then transformer.py:34-46)
        for (int b = 0; b < B; ++b) {
            constexpr int XR = 16;   // K <= XR * 256 = 4096 (launch check)
            long tok = p.em_tokens[(long)b * p.em_tok_stride + p.em_col];
            const bool zero = tok == -1;
            tok = tok < 0 ? 0 : (tok >= p.em_rows ? p.em_rows - 1 : tok);
            const unsigned short* er = p.em_table + tok * (long)K;
            float xr[XR];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int k = tid + i * 64 * GEMV_WAVES;
                float v = 0.f;
                if (k < K) {
                    v = p.x[(long)b * p.ldx + k];
                    if (!zero) v += __uint_as_float((unsigned)er[k] << 16);
                    if (blockIdx.x == 0) p.em_x_out[(long)b * K + k] = v;
                }
                xr[i] = v;
                s = fmaf(v, v, s);
            }
            s = wave_sum(s);
            __syncthreads();
            if (lane == 0) red[wave] = s;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_WAVES; ++w) tot += red[w];
            const float r = 1.0f / sqrtf(p.eps + tot / (float)K);
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int k = tid + i * 64 * GEMV_WAVES;
                if (k < K) xs[b * K + k] = xr[i] * (p.alpha[k] * r);
            }
        }
    } else if (p.prologue == 2) {    // SiLU gate: x holds [B][2K] = [u ; v], xs = silu(u) * v   (modules/gating.py:12-22)
        for (int b = 0; b < B; ++b)
            for (int k = tid; k < K; k += 64 * GEMV_WAVES)
                xs[b * K + k] = silu(p.x[(long)b * p.ldx + k]) * p.x[(long)b * p.ldx + K + k];
    } else {
        for (int b = 0; b < B; ++b)
            for (int k = tid; k < K; k += 64 * GEMV_WAVES) xs[b * K + k] = p.x[(long)b * p.ldx + k];
    }
    __syncthreads();

    // ---- row groups, grid-strided: RPW rows per wave, 8 bf16 (16 B) per lane per row per iteration, two iterations in flight
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int n0 = (grp * GEMV_WAVES + wave) * RPW;
        float acc[RPW][B];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
        long wrow[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) wrow[r] = (long)row_of(grp, r) * K;
        auto fma8 = [&](const WChunk<F32W> (&wv)[RPW], int k) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + b * K + k);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + b * K + k + 4);
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[r][b] = wv[r].dot(x0, x1, acc[r][b]);
            }
        };
        int k = lane * 8;
        if (grp == (int)blockIdx.x && k < K) {   // the prefetched chunk
            fma8(wpre, k);
            k += 512;
        }
        for (; k + 512 < K; k += 1024) {
            WChunk<F32W> wa[RPW], wb[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                wa[r].load(p.w, wrow[r] + k);
                wb[r].load(p.w, wrow[r] + k + 512);
            }
            fma8(wa, k);
            fma8(wb, k + 512);
        }
        if (k < K) {
            WChunk<F32W> wa[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) wa[r].load(p.w, wrow[r] + k);
            fma8(wa, k);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int b = 0; b < B; ++b) acc[r][b] = wave_sum(acc[r][b]);
        if (p.gate_out) {
            const int q = grp * GEMV_WAVES + wave;
            if (lane == 0 && q < half)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const float u = acc[0][b] + (p.bias ? p.bias[q] : 0.f), v = acc[RPW - 1][b] + (p.bias ? p.bias[half + q] : 0.f);
                    p.y[(long)b * p.ldy + q] = silu(u) * v;
                }
            continue;
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const float s = acc[r][b];
                const int n = n0 + r;
                if (lane == 0 && n < p.N) {
                    const long o = (long)b * p.ldy + n;
                    float sb = p.bias ? s + p.bias[n] : s;
                    if (p.act_out == 1) sb = 0.5f * sb * (1.0f + erff(sb * 0.70710678118654752440f));   // exact GELU (F.gelu)
                    if (p.scale) sb *= p.scale[n];
                    p.y[o] = p.res ? (grp == (int)blockIdx.x ? rpre[r][b] : p.res[o]) + sb : sb;
                }
            }
    }
}

// Batch-1 RMSNorm -> GEMV for the large layers of the temporal transformer (qkv, ffn-in, text head: 100 .. 260 MB of weights
// per launch).  Same arithmetic as gemv_kernel<1, 2, false> with prologue 1; the difference is the schedule.  Vector-memory
// results return in issue order, so the (L2-resident) x and alpha loads go out FIRST, then PRE = 8 weight chunks per row -- all of
// a K = 4096 row -- and the norm runs on data that arrives ahead of the weights: the memory pipe is full from the first cycle
// instead of idling behind the x -> reduce -> LDS chain (measured 22.5 -> 19.2 us for 12288 x 4096, 36.8 -> 32.3 us for
// 22528 x 4096).  GATE: rows (q, N/2 + q) per wave and silu(u) * v as the output (gate_out of gemv_kernel).
template <bool GATE>
__global__ __launch_bounds__(64 * GEMV_WAVES) void gemv_norm_kernel(const GemvParams p) {
    constexpr int RPW = 2, PRE = 8, XR = 16, NT = 64 * GEMV_WAVES;
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [K]
    __shared__ float red[GEMV_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, half = p.N / 2;
    const int groups = GATE ? (half + GEMV_WAVES - 1) / GEMV_WAVES : (p.N + RPW * GEMV_WAVES - 1) / (RPW * GEMV_WAVES);
    const int nchunks = (K + 511) >> 9;
    const int kl = lane * 8;

    float xa[XR], al[XR];            // K <= XR * NT = 4096 (launch check)
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int k = tid + i * NT;
        xa[i] = k < K ? p.x[k] : 0.f;
        al[i] = k < K ? p.alpha[k] : 0.f;
    }
    WChunk<false> buf[PRE][RPW];
    long wrow[RPW];
    auto rows_of = [&](int grp) {
        const int q = grp * GEMV_WAVES + wave;
#pragma unroll
        for (int r = 0; r < RPW; ++r) wrow[r] = (long)(GATE ? r * half + min(q, half - 1) : min(q * RPW + r, p.N - 1)) * K;
    };
    auto issue = [&](int j, int c) {   // chunk c (512 k) of both rows -> slot j
        const int kk = (c << 9) + kl;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (kk < K) buf[j][r].load(p.w, wrow[r] + kk);
            else buf[j][r].zero();
        }
    };
    rows_of(blockIdx.x);
#pragma unroll
    for (int j = 0; j < PRE; ++j) issue(j, j);

    {                                // RMSNorm: x * alpha * rsqrt(eps + mean(x^2))   (modules/transformer.py:34-46)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < XR; ++i) s = fmaf(xa[i], xa[i], s);
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < GEMV_WAVES; ++w) tot += red[w];
        const float r = 1.0f / sqrtf(p.eps + tot / (float)K);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int k = tid + i * NT;
            if (k < K) xs[k] = xa[i] * (al[i] * r);
        }
    }
    __syncthreads();

    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        if (grp != (int)blockIdx.x) {
            rows_of(grp);
#pragma unroll
            for (int j = 0; j < PRE; ++j) issue(j, j);
        }
        float acc[RPW] = {0.f, 0.f};
        for (int base = 0; base < nchunks; base += PRE) {
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                const int c = base + j;
                const int kk = (c << 9) + kl;
                if (kk < K) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + kk);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + kk + 4);
#pragma unroll
                    for (int r = 0; r < RPW; ++r) acc[r] = buf[j][r].dot(x0, x1, acc[r]);
                }
                if (c + PRE < nchunks) issue(j, c + PRE);
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r] = wave_sum(acc[r]);
        const int q = grp * GEMV_WAVES + wave;
        if (lane != 0) continue;
        if (GATE) {
            if (q < half) {
                const float u = acc[0] + (p.bias ? p.bias[q] : 0.f), v = acc[1] + (p.bias ? p.bias[half + q] : 0.f);
                p.y[q] = silu(u) * v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = q * RPW + r;
                if (n < p.N) {
                    float sb = p.bias ? acc[r] + p.bias[n] : acc[r];
                    if (p.scale) sb *= p.scale[n];
                    p.y[n] = p.res ? p.res[n] + sb : sb;
                }
            }
        }
    }
}

// Batch-1 GEMV on a plain activation vector (out-proj 4096 x 4096, ffn-out 4096 x 11264 of the temporal transformer: 33 / 92 MB of
// weights per launch) with K split over the four waves of a workgroup: wave w takes the 512-k chunks c = w, w + 4, ... of RW = 8
// rows, and the eight k of a chunk that a lane multiplies come straight from global memory (L2) into registers -- no activation
// stage in LDS, no barrier in front of the weight stream.  gemv_kernel<1, 2> stages the whole vector first (45 KB per workgroup
// at K = 11264, ~3 us during which only the first chunk of each row is in flight): ffn-out measured 21.4 us = 4.3 TB/s there.
// The four partial sums of a row meet in LDS and are added in wave order (so the rounding differs from gemv_kernel's single
// chain per lane in the last bits; same for every call).
template <int RW>
__global__ __launch_bounds__(64 * GEMV_WAVES) void gemv_ksplit_kernel(const GemvParams p) {
    __shared__ float part[GEMV_WAVES][RW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, nchunks = (K + 511) >> 9;
    const int kl = lane * 8;
    const int groups = (p.N + RW - 1) / RW;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int n0 = grp * RW;
        long wrow[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) wrow[r] = (long)min(n0 + r, p.N - 1) * K;
        float acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r] = 0.f;
        // two chunks in flight: the weights of both are requested before the first is multiplied
        WChunk<false> wa[RW], wb[RW];
        f32x4 xa0, xa1, xb0, xb1;
        auto issue = [&](WChunk<false> (&w)[RW], f32x4& x0, f32x4& x1, int c) {
            const int kk = (c << 9) + kl;
            const bool ok = c < nchunks && kk < K;
            x0 = f32x4{0.f, 0.f, 0.f, 0.f};
            x1 = x0;
            if (ok) {
                x0 = *reinterpret_cast<const f32x4*>(p.x + kk);
                x1 = *reinterpret_cast<const f32x4*>(p.x + kk + 4);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (ok) w[r].load(p.w, wrow[r] + kk);
                else w[r].zero();
            }
        };
        issue(wa, xa0, xa1, wave);
        for (int c = wave; c < nchunks; c += 2 * GEMV_WAVES) {
            issue(wb, xb0, xb1, c + GEMV_WAVES);
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = wa[r].dot(xa0, xa1, acc[r]);
            issue(wa, xa0, xa1, c + 2 * GEMV_WAVES);
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = wb[r].dot(xb0, xb1, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r) part[wave][r] = acc[r];
        }
        __syncthreads();
        if (tid < RW && n0 + tid < p.N) {
            const int n = n0 + tid;
            const float s = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
            float sb = p.bias ? s + p.bias[n] : s;
            if (p.scale) sb *= p.scale[n];
            p.y[n] = p.res ? p.res[n] + sb : sb;
        }
        __syncthreads();          // part is rewritten by the next group
    }
}

// x[b][:] = sum_i table_i[token[b][i]]  (bf16 tables, fp32 sum in table order; id -1 -> zero row, ids clamped at 0)
__global__ __launch_bounds__(256) void embed_sum_kernel(const EmbedSumParams p) {
    const int b = blockIdx.y;
    for (int d = blockIdx.x * 256 + threadIdx.x; d < p.D; d += gridDim.x * 256) {
        float s = p.add ? p.add[(long)b * p.add_stride + d] : 0.f;
        for (int i = 0; i < p.n_tables; ++i) {
            const long tok = p.tokens[(long)b * p.tok_stride + p.tok_index[i]];
            if (tok != -1) {
                // ids outside the table are clamped into it (the reference's F.embedding raises; a kernel cannot, and must not
                // read out of bounds): other negative ids -> row 0, ids >= rows -> the last row
                long row = tok < 0 ? 0 : tok;
                if (p.rows[i] > 0 && row >= p.rows[i]) row = p.rows[i] - 1;
                s += __uint_as_float((unsigned)p.tables[i][row * p.D + d] << 16);
            }
        }
        p.out[(long)b * p.D + d] = s;
    }
}

__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                     float* __restrict__ y, int D, float eps) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (long)blockIdx.x * D;
    float s = 0.f;
    for (int k = tid; k < D; k += 256) s = fmaf(xr[k], xr[k], s);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float r = 1.0f / sqrtf(eps + (red[0] + red[1] + red[2] + red[3]) / (float)D);
    for (int k = tid; k < D; k += 256) y[(long)blockIdx.x * D + k] = xr[k] * (alpha[k] * r);
}

}  // namespace

int rst_launch_gemv(const GemvParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.B <= 4 && p.N > 0 && p.K > 0 && p.K % 8 == 0, "gemv: need 1 <= B <= 4 and K %% 8 == 0 (B=%d K=%d)", p.B, p.K);
    RST_REQUIRE(p.x && p.w && p.y, "gemv: null pointer");
    RST_REQUIRE(p.prologue >= 0 && p.prologue <= 5 && (p.prologue != 1 || p.alpha) && (p.prologue != 3 || (p.alpha && p.beta)),
                "gemv: bad prologue");
    if (p.prologue == 4) {
        RST_REQUIRE(!p.w_f32 && p.at_k && p.at_v && p.at_pos && p.at_cap >= 1 && p.at_cap <= 8 && p.at_H >= 1 && p.at_D >= 4 &&
                    p.at_D <= 256 && (p.at_D & (p.at_D - 1)) == 0 && p.at_H * p.at_D == p.K && p.ldx >= 3 * p.K && p.ldx % 4 == 0,
                    "gemv_attn: needs a ring of <= 8 slots, a power-of-two head dim in 4..256, K = H * D and a [q | k | v] row (H=%d D=%d cap=%d K=%d ldx=%d)",
                    p.at_H, p.at_D, p.at_cap, p.K, p.ldx);
        RST_REQUIRE(((uintptr_t)p.at_k % 16) == 0 && ((uintptr_t)p.at_v % 16) == 0, "gemv_attn: ring pointers must be 16-byte aligned");
    }
    if (p.prologue == 5)
        RST_REQUIRE(!p.w_f32 && p.alpha && p.em_table && p.em_tokens && p.em_x_out && p.em_rows >= 1 && p.K <= 4096 && p.em_tok_stride >= 1 &&
                    p.em_col >= 0 && p.em_col < p.em_tok_stride, "gemv_embed: bad arguments (K=%d rows=%d col=%d stride=%d)", p.K, p.em_rows,
                    p.em_col, p.em_tok_stride);
    RST_REQUIRE(p.act_out == 0 || p.act_out == 1, "gemv: act_out must be 0 (none) or 1 (GELU)");
    RST_REQUIRE(((uintptr_t)p.w % 16) == 0 && ((uintptr_t)p.x % 16) == 0, "gemv: pointers must be 16-byte aligned");
    const size_t lds = (size_t)p.B * p.K * sizeof(float);
    RST_REQUIRE(lds <= 128 * 1024, "gemv: B*K = %d floats do not fit the activation stage (32768)", p.B * p.K);
    RST_REQUIRE(!p.gate_out || (p.N % 2 == 0 && !p.res && !p.scale && !p.act_out), "gemv: gate_out needs an even N and no residual / scale / activation");
    // rows per wave: 4 when that still yields >= 2 workgroups per CU, else 2 (more workgroups -> more loads in flight)
    const bool rpw4 = !p.w_f32 && !p.gate_out && p.prologue < 4 && ((long)p.N + 15) / 16 >= 512;
    // the large batch-1 RMSNorm layers take the x-first streaming schedule (gemv_norm_kernel)
    const bool norm_stream = !p.w_f32 && p.B == 1 && p.prologue == 1 && p.K <= 4096 && !p.act_out && (long)p.N * p.K >= (1L << 24);
    const int rows_per_group = (rpw4 ? 4 : 2) * GEMV_WAVES;
    const long groups = p.gate_out ? ((long)p.N / 2 + GEMV_WAVES - 1) / GEMV_WAVES : ((long)p.N + rows_per_group - 1) / rows_per_group;
    const unsigned grid = cap_grid(groups, norm_stream ? 1024 : (lds > 48 * 1024 ? 512 : 768));
    auto go = [&](auto kern) {
        static RstOncePerDevice attr_once;     // one flag per kernel instantiation (generic lambda)
        if (attr_once.first()) {
            // static LDS (the 16-byte reduction scratch) counts against the 160 KiB budget: ask for what the check above allows
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * GEMV_WAVES), lds, stream, p);
    };
    if (p.w_f32) {
        switch (p.B) {
            case 1: go(gemv_kernel<1, 2, true>); break;
            case 2: go(gemv_kernel<2, 2, true>); break;
            case 3: go(gemv_kernel<3, 2, true>); break;
            default: go(gemv_kernel<4, 2, true>); break;
        }
        return rst_check_launch("gemv_f32");
    }
    if (p.prologue >= 4) {
        RST_REQUIRE(p.B <= 2, "gemv: the fused attention / embedding prologues serve B <= 2 (got %d)", p.B);
        if (p.B == 1) go(gemv_kernel<1, 2, false, true>); else go(gemv_kernel<2, 2, false, true>);
        return rst_check_launch("gemv_fused");
    }
    if (norm_stream) {
        if (p.gate_out) go(gemv_norm_kernel<true>); else go(gemv_norm_kernel<false>);
        return rst_check_launch("gemv_bf16");
    }
    // the large batch-1 layers on a plain vector: K split over the waves, activations straight into registers (gemv_ksplit_kernel)
    if (!p.w_f32 && p.B == 1 && p.prologue == 0 && !p.act_out && !p.gate_out && (long)p.N * p.K >= (1L << 24) && p.K >= 2048) {
        const long g8 = ((long)p.N + 7) / 8;
        hipLaunchKernelGGL(gemv_ksplit_kernel<8>, dim3(cap_grid(g8, 1024)), dim3(64 * GEMV_WAVES), 0, stream, p);
        return rst_check_launch("gemv_bf16");
    }
    switch (p.B * 2 + (rpw4 ? 1 : 0)) {
        case 2: go(gemv_kernel<1, 2, false>); break;
        case 3: go(gemv_kernel<1, 4, false>); break;
        case 4: go(gemv_kernel<2, 2, false>); break;
        case 5: go(gemv_kernel<2, 4, false>); break;
        case 6: go(gemv_kernel<3, 2, false>); break;
        case 7: go(gemv_kernel<3, 4, false>); break;
        case 8: go(gemv_kernel<4, 2, false>); break;
        default: go(gemv_kernel<4, 4, false>); break;
    }
    return rst_check_launch("gemv_bf16");
}

int rst_launch_embed_sum(const EmbedSumParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.D > 0 && p.n_tables >= 0 && p.n_tables <= RST_MAX_TABLES && p.tokens && p.out, "embed_sum: bad arguments");
    hipLaunchKernelGGL(embed_sum_kernel, dim3(cap_grid((p.D + 255) / 256, 64), p.B), dim3(256), 0, stream, p);
    return rst_check_launch("embed_sum");
}

int rst_launch_rmsnorm(const float* x, const float* alpha, float* y, long rows, int D, float eps, hipStream_t stream) {
    RST_REQUIRE(x && alpha && y && rows >= 0 && D > 0, "rmsnorm: bad arguments");
    if (rows == 0) return RST_OK;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, alpha, y, D, eps);
    return rst_check_launch("rmsnorm");
}
