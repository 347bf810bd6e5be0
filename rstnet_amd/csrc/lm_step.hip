// Batch-1 (small-batch) decode step of the RQ-Transformer: every op of one temporal / depth transformer step at T = 1.
//
// At batch 1 the step is pure weight streaming (14.8 GB of bf16 weights per 80 ms frame for the Moshi-7B shape), so
// the GEMV is built for HBM: 16-byte non-temporal weight loads, four rows per wave in flight, the (normalised /
// gated) activation vector staged ONCE per workgroup in LDS as fp32, fp32 accumulation, wave-level reduction.
// RMSNorm and the SiLU gate are prologues of the consuming GEMV and the residual add is its epilogue, so a
// transformer layer is 7 launches: qkv GEMV | rope + KV append | attention partial | attention combine | out-proj
// GEMV (+res) | ffn-in GEMV | ffn-out GEMV (+res).  Activations stay fp32 (>= the reference's bf16), weights bf16.
#include <algorithm>
#include "rst_common.h"
#include "rst_kernels.h"
#include <math.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }

constexpr int GEMV_WAVES = 4;

// Ring slot -> position map of RingKVCache.complete (modules/transformer.py:254-278) incl. the `delta <= 0` quirk (Q1);
// returns whether `slot` is visible to the query at position `pos` (= the step just appended).
__device__ __forceinline__ bool ring_visible(int slot, long pos, int cap, int context, long end_offset) {
    const int end_index = (int)(end_offset % cap);
    const int delta = slot - end_index;
    long pk = delta <= 0 ? end_offset + delta : end_offset + delta - cap;
    if (slot >= end_offset) pk = -1;
    const long dl = pos - pk;
    bool ok = slot < cap && pk >= 0 && dl >= 0;
    if (context > 0) ok = ok && dl < context;
    return ok;
}

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
template <int B, int RPW, bool F32W>
__global__ __launch_bounds__(64 * GEMV_WAVES) void gemv_kernel(const GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [B][K]
    __shared__ float red[GEMV_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;

    // ---- the first weight chunk of this workgroup's first row group is requested BEFORE the activation prologue: the two
    // dependent load chains (x -> norm -> LDS, and W) then overlap instead of adding up (small GEMVs are latency-bound)
    const int groups = (p.N + RPW * GEMV_WAVES - 1) / (RPW * GEMV_WAVES);
    WChunk<F32W> wpre[RPW];
    {
        const int n0 = (blockIdx.x * GEMV_WAVES + wave) * RPW;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            wpre[r].zero();
            if (blockIdx.x < groups && lane * 8 < K) wpre[r].load(p.w, (long)min(n0 + r, p.N - 1) * K + lane * 8);
        }
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
        for (int r = 0; r < RPW; ++r) wrow[r] = (long)min(n0 + r, p.N - 1) * K;
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
            for (int b = 0; b < B; ++b) {
                const float s = wave_sum(acc[r][b]);
                const int n = n0 + r;
                if (lane == 0 && n < p.N) {
                    const long o = (long)b * p.ldy + n;
                    float sb = p.bias ? s + p.bias[n] : s;
                    if (p.act_out == 1) sb = 0.5f * sb * (1.0f + erff(sb * 0.70710678118654752440f));   // exact GELU (F.gelu)
                    if (p.scale) sb *= p.scale[n];
                    p.y[o] = p.res ? p.res[o] + sb : sb;
                }
            }
    }
}

// x[b][:] = sum_i table_i[token[b][i]]  (bf16 tables, fp32 sum in table order; id -1 -> zero row, ids clamped at 0)
__global__ __launch_bounds__(256) void embed_sum_kernel(const EmbedSumParams p) {
    const int b = blockIdx.y;
    for (int d = blockIdx.x * 256 + threadIdx.x; d < p.D; d += gridDim.x * 256) {
        float s = p.add ? p.add[(long)b * p.D + d] : 0.f;
        for (int i = 0; i < p.n_tables; ++i) {
            const long tok = p.tokens[(long)b * p.tok_stride + p.tok_index[i]];
            if (tok != -1) {
                const long row = tok < 0 ? 0 : tok;
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

// qkv [B][T][ldqkv] (T new steps, [q | k | v] with H / G / G heads) -> q_rot [B][H][T][D]; k (rotated) and v written into
// ring slots (pos + t) % cap of [B][G][cap][D].  Work item = one (real, imag) pair of one q head or one k/v head.
__global__ __launch_bounds__(256) void rope_append_kernel(const LmRopeAppendParams p) {
    const int half = p.D / 2;
    const int HG = p.H + p.G;
    const long total = (long)p.B * p.T * HG * half;
    const long pos0 = *p.pos_dev;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx % half);
        const int hh = (int)((idx / half) % HG);
        const int t = (int)((idx / ((long)half * HG)) % p.T);
        const long b = idx / ((long)half * HG * p.T);
        const long pos = pos0 + t;
        const float* row = p.qkv + (b * p.T + t) * (long)p.ldqkv;
        float c = 1.f, s = 0.f;
        if (p.rope && 2 * i < p.rope_dims) {
            const float ang = expf((float)i * p.rope_coef) * (float)pos;
            c = cosf(ang);
            s = sinf(ang);
        }
        if (hh < p.H) {
            const float qr = row[(long)hh * p.D + 2 * i], qi = row[(long)hh * p.D + 2 * i + 1];
            float* qd = p.q + ((b * p.H + hh) * p.T + t) * (long)p.D + 2 * i;
            qd[0] = qr * c - qi * s; qd[1] = qr * s + qi * c;
        } else {
            const int g = hh - p.H;
            const int slot = (int)(pos % p.cap);
            const float* ks = row + (long)p.H * p.D + (long)g * p.D + 2 * i;
            const float* vs = ks + (long)p.G * p.D;
            float* kd = p.k + ((b * p.G + g) * p.cap + slot) * (long)p.D + 2 * i;
            float* vd = p.v + ((b * p.G + g) * p.cap + slot) * (long)p.D + 2 * i;
            kd[0] = ks[0] * c - ks[1] * s; kd[1] = ks[0] * s + ks[1] * c;
            vd[0] = vs[0]; vd[1] = vs[1];
        }
    }
}

// One query per (b, h), read straight from the qkv vector of the new step: RoPE on q (every workgroup) and on the new
// key (the workgroup whose slot range holds the ring slot of this step, which also appends k / v to the ring).
// Slots are split over gridDim.x workgroups; each writes (m, l, o[D]) to the workspace and the last one to arrive combines.
// Lane groups of D/16 lanes own one slot per iteration (each lane 16 contiguous floats of the K / V row: coalesced).
template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(const LmAttnParams p) {
    constexpr int LPS = D / 16;            // lanes per slot
    constexpr int SPW = 64 / LPS;          // slots per wave iteration
    __shared__ float sm_m[4], sm_l[4];
    __shared__ __attribute__((aligned(16))) float sm_o[4][D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPS, grp = lane / LPS;
    const int split = blockIdx.x, h = blockIdx.y;
    const int T = p.q_pre ? p.T : 1;
    const long b = blockIdx.z / T;
    const int tq = blockIdx.z % T;
    const long pos = *p.pos_dev;                 // position of the first new step
    const long pos_q = pos + tq;                 // position of this query
    const long end_offset = pos + T;             // RingKVCache.end_offset after the append of all T new steps
    const int slot_cur = p.q_pre ? -1 : (int)(pos % p.cap);   // fused mode: the slot this launch appends
    const float scale = 1.0f / sqrtf((float)D);
    const int qpk = p.H / p.G, g = h / qpk;      // grouped-query attention: kv head of this query head
    const bool appender = h % qpk == 0;          // one query head per group writes the new step into the ring

    // rotation of this lane's 8 (real, imag) pairs at position `pos` (modules/rope.py:37-62)
    float rc[8], rs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rc[i] = 1.f; rs[i] = 0.f;
        if (p.rope && 2 * (sub * 8 + i) < p.rope_dims) {
            const float ang = expf((float)(sub * 8 + i) * p.rope_coef) * (float)pos;
            rc[i] = cosf(ang); rs[i] = sinf(ang);
        }
    }
    const float* qkv = p.q_pre ? nullptr : p.qkv + b * p.ldqkv + (long)h * D + sub * 16;
    const float* kn = p.q_pre ? nullptr : p.qkv + b * p.ldqkv + ((long)p.H + g) * D + sub * 16;
    const float* vn = p.q_pre ? nullptr : kn + (long)p.G * D;
    float q[16], kcur[16];
    if (p.q_pre) {      // queries already rotated, keys already in the ring (codec transformer: rst_rope_split_f32 ran before)
        const float* qp = p.q_pre + (((b * p.H + h) * T) + tq) * (long)D + sub * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) { q[i] = qp[i]; kcur[i] = 0.f; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float qr = qkv[2 * i], qi = qkv[2 * i + 1], kr = kn[2 * i], ki = kn[2 * i + 1];
            q[2 * i] = qr * rc[i] - qi * rs[i]; q[2 * i + 1] = qr * rs[i] + qi * rc[i];
            kcur[2 * i] = kr * rc[i] - ki * rs[i]; kcur[2 * i + 1] = kr * rs[i] + ki * rc[i];
        }
    }

    const int n_used = (int)min((long)p.cap, end_offset);       // slots >= end_offset are never visible
    const int active = max(1, min((int)gridDim.x, (n_used + 63) / 64));   // splits that have work at this context length
    if (split >= active) return;
    const int per = (n_used + active - 1) / active;
    const int s_lo = split * per, s_hi = min(n_used, s_lo + per);
    float m_run = -INFINITY, l_run = 0.f;
    float o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float* kb = p.k + ((b * p.G + g) * p.cap) * (long)D + sub * 16;
    float* vb = p.v + ((b * p.G + g) * p.cap) * (long)D + sub * 16;

    for (int s0 = s_lo + wave * SPW; s0 < s_hi; s0 += 4 * SPW) {
        const int slot = s0 + grp;
        const bool ok = slot < s_hi && ring_visible(slot, pos_q, p.cap, p.context, end_offset);
        float kv[16], vv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { kv[i] = 0.f; vv[i] = 0.f; }
        if (slot < s_hi && slot == slot_cur) {       // the new step: from qkv, and appended to the ring
#pragma unroll
            for (int i = 0; i < 16; ++i) { kv[i] = kcur[i]; vv[i] = vn[i]; }
            if (appender) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<f32x4*>(kb + (long)slot * D + 4 * i) = f32x4{kv[4 * i], kv[4 * i + 1], kv[4 * i + 2], kv[4 * i + 3]};
                    *reinterpret_cast<f32x4*>(vb + (long)slot * D + 4 * i) = f32x4{vv[4 * i], vv[4 * i + 1], vv[4 * i + 2], vv[4 * i + 3]};
                }
            }
        } else if (ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(kb + (long)slot * D + 4 * i);
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(vb + (long)slot * D + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) { kv[4 * i + e] = k4[e]; vv[4 * i + e] = v4[e]; }
            }
        }
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) d = fmaf(kv[i], q[i], d);
#pragma unroll
        for (int off = LPS / 2; off > 0; off >>= 1) d += __shfl_xor(d, off);   // all lanes take part (ok is per group)
        const float sc = ok ? d * scale : -INFINITY;
        const float m_new = fmaxf(m_run, sc);
        if (m_new != -INFINITY) {
            const float alpha = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
            const float pw = sc == -INFINITY ? 0.f : expf(sc - m_new);
            l_run = l_run * alpha + pw;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = o[i] * alpha + pw * vv[i];
            m_run = m_new;
        }
    }
    // merge the lane groups of the wave (same `sub`), then the 4 waves, into one (m, l, o[D])
    float m_w = m_run;
#pragma unroll
    for (int off = LPS; off < 64; off <<= 1) m_w = fmaxf(m_w, __shfl_xor(m_w, off));
    const float f = (m_run == -INFINITY) ? 0.f : expf(m_run - m_w);
    float l_w = l_run * f;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] *= f;
#pragma unroll
    for (int off = LPS; off < 64; off <<= 1) {
        l_w += __shfl_xor(l_w, off);
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] += __shfl_xor(o[i], off);
    }
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sm_o[wave][sub * 16 + i] = o[i];
        if (sub == 0) { sm_m[wave] = m_w; sm_l[wave] = l_w; }
    }
    __syncthreads();
    if (tid < D) {
        float M = fmaxf(fmaxf(sm_m[0], sm_m[1]), fmaxf(sm_m[2], sm_m[3]));
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float fw = sm_m[w] == -INFINITY ? 0.f : expf(sm_m[w] - M);
            L += sm_l[w] * fw;
            O += sm_o[w][tid] * fw;
        }
        if (active == 1) {                          // single split: finished
            p.out[((b * T + tq) * p.H + h) * (long)D + tid] = L > 0.f ? O / L : 0.f;
        } else {
            // write-through (sc1) partials: visible at agent scope without an L2 write-back fence (cdna_hip_programming.md G16 R1)
            float* ws = p.ws + ((((long)blockIdx.z * p.H + h) * gridDim.x) + split) * (long)(D + 2);
            __hip_atomic_store(ws + 2 + tid, O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                __hip_atomic_store(ws, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ws + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (active == 1) return;
    // In-launch reduction of the splits: every storing wave drains its write-through stores, ONE lane bumps the arrival
    // counter; the LAST arriver of (b, h) reads the partials with agent-scope (L1-bypassing) loads, combines, and re-arms the
    // counter for the next launch.  No dispatch-order / placement assumption.
    __shared__ int sm_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* cnt = p.counters + ((long)blockIdx.z * p.H + h);
        const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm_last = prev == (unsigned)active - 1;
        if (sm_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!sm_last) return;
    if (tid < D) {
        const float* w0 = p.ws + (((long)blockIdx.z * p.H + h) * gridDim.x) * (long)(D + 2);
        float M = -INFINITY;
        for (int s2 = 0; s2 < active; ++s2) M = fmaxf(M, __hip_atomic_load(w0 + (long)s2 * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float L = 0.f, O = 0.f;
        for (int s2 = 0; s2 < active; ++s2) {
            const float* w = w0 + (long)s2 * (D + 2);
            const float ms = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float fw = ms == -INFINITY ? 0.f : expf(ms - M);
            L = fmaf(__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), fw, L);
            O = fmaf(__hip_atomic_load(w + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), fw, O);
        }
        p.out[((b * T + tq) * p.H + h) * (long)D + tid] = L > 0.f ? O / L : 0.f;
    }
}

// Short ring (capacity <= 64, e.g. the depth transformer's 8 steps): one wave per (b, h), no split and no workspace.
// Lane = (slot group g = lane / 8, dim chunk c = lane % 8): a pass covers 8 ring slots, each lane holding D/8 contiguous
// dims of its slot's key and value (coalesced 16-byte loads issued before the position is even known); scores reduce
// over the 8 chunk lanes, the softmax and P.V over the 8 slot groups.  The new step's key / value come straight from qkv
// (and are appended to the ring by the lanes that own its slot).
template <int D>
__global__ __launch_bounds__(64) void attn_small_kernel(const LmAttnParams p) {
    constexpr int DPL = D / 8;                   // dims per lane
    constexpr int MAXP = 8;                      // passes of 8 slots (cap <= 64)
    const int lane = threadIdx.x, g = lane >> 3, c = lane & 7, h = blockIdx.x;
    const long b = blockIdx.y;
    const int cap = p.cap;
    const int qpk = p.H / p.G, kvh = h / qpk;    // grouped-query attention: kv head of this query head
    const bool appender = h % qpk == 0;
    const float* qkv = p.qkv + b * p.ldqkv + (long)h * D + c * DPL;
    const float* knp = p.qkv + b * p.ldqkv + ((long)p.H + kvh) * D + c * DPL;
    const float* vnp = knp + (long)p.G * D;
    float* kc = p.k + ((b * p.G + kvh) * cap) * (long)D + c * DPL;
    float* vc = p.v + ((b * p.G + kvh) * cap) * (long)D + c * DPL;
    const int npass = (cap + 7) >> 3;
    float q[DPL], kn[DPL], vn[DPL];
#pragma unroll
    for (int i = 0; i < DPL; i += 4) {
        *reinterpret_cast<f32x4*>(q + i) = *reinterpret_cast<const f32x4*>(qkv + i);
        *reinterpret_cast<f32x4*>(kn + i) = *reinterpret_cast<const f32x4*>(knp + i);
        *reinterpret_cast<f32x4*>(vn + i) = *reinterpret_cast<const f32x4*>(vnp + i);
    }
    const long pos = *p.pos_dev;
    const int slot_cur = (int)(pos % cap);
    if (p.rope) {
#pragma unroll
        for (int i = 0; i < DPL; i += 2) {
            if (c * DPL + i >= p.rope_dims) continue;
            const float ang = expf((float)((c * DPL + i) >> 1) * p.rope_coef) * (float)pos;
            const float cs = cosf(ang), sn = sinf(ang);
            const float qr = q[i], qi = q[i + 1], kr = kn[i], ki = kn[i + 1];
            q[i] = qr * cs - qi * sn; q[i + 1] = qr * sn + qi * cs;
            kn[i] = kr * cs - ki * sn; kn[i + 1] = kr * sn + ki * cs;
        }
    }
    float sc[MAXP];
    float m = -INFINITY;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
        sc[ps] = -INFINITY;
        if (ps < npass) {
            const int slot = ps * 8 + g;
            const bool cur = slot == slot_cur;
            const bool ok = slot < cap && ring_visible(slot, pos, cap, p.context, pos + 1);
            float kk[DPL];
            if (cur && appender) {
#pragma unroll
                for (int i = 0; i < DPL; i += 4) {
                    *reinterpret_cast<f32x4*>(kc + (long)slot * D + i) = *reinterpret_cast<const f32x4*>(kn + i);
                    *reinterpret_cast<f32x4*>(vc + (long)slot * D + i) = *reinterpret_cast<const f32x4*>(vn + i);
                }
            }
#pragma unroll
            for (int i = 0; i < DPL; i += 4) {
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok && !cur) t = *reinterpret_cast<const f32x4*>(kc + (long)slot * D + i);
                *reinterpret_cast<f32x4*>(kk + i) = t;
            }
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < DPL; ++i) d = fmaf(cur ? kn[i] : kk[i], q[i], d);
            d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
            if (ok) sc[ps] = d / sqrtf((float)D);
            m = fmaxf(m, sc[ps]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 8)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f, o[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) o[i] = 0.f;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
        if (ps < npass) {
            const int slot = ps * 8 + g;
            const bool cur = slot == slot_cur;
            const float pw = sc[ps] == -INFINITY ? 0.f : expf(sc[ps] - m);
            l += pw;
#pragma unroll
            for (int i = 0; i < DPL; i += 4) {
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cur) t = *reinterpret_cast<const f32x4*>(vn + i);
                else if (pw != 0.f) t = *reinterpret_cast<const f32x4*>(vc + (long)slot * D + i);
                o[i] = fmaf(pw, t[0], o[i]); o[i + 1] = fmaf(pw, t[1], o[i + 1]);
                o[i + 2] = fmaf(pw, t[2], o[i + 2]); o[i + 3] = fmaf(pw, t[3], o[i + 3]);
            }
        }
    }
    l += __shfl_xor(l, 8); l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
        o[i] += __shfl_xor(o[i], 8); o[i] += __shfl_xor(o[i], 16); o[i] += __shfl_xor(o[i], 32);
        o[i] = o[i] / l;
    }
    if (g == 0) {
        float* out = p.out + (b * p.H + h) * (long)D + c * DPL;
#pragma unroll
        for (int i = 0; i < DPL; i += 4) *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(o + i);
    }
}

// One workgroup per batch row (256 threads for V <= 4096, 1024 above).  Greedy: argmax (lowest index on ties).
// Sampling (utils/sampling.py:51-105): probs = softmax(logits / temp); (p, idx) = top-k in descending order (ties: lowest
// index first); token = idx[argmax_j p_j / noise_j].  Exact top-k WITHOUT sorting, register resident: each thread keeps
// EPT order-preserving uint keys of the scaled logits; a bit-wise binary search on the key (block-wide counts, early exit
// as soon as exactly k keys lie above the probe) finds the k-th largest key, ties at that value are resolved by a search
// on the index, the exactly-k candidates are compacted into LDS as 64-bit (key, ~index) composites and each computes its
// rank by counting the larger composites.
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void sample_kernel(const LmSampleParams p) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned long long comp[];   // [k rounded up to 8] candidates
    __shared__ float red_v[NW];
    __shared__ int red_i[NW], red_j[NW];
    __shared__ int cnt[52 * NW];
    __shared__ int n_cand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    const float* lg = p.logits + b * p.ld;
    const bool sampling = p.use_sampling && p.temp > 0.f;
    const int V = p.V;

    auto to_key = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto from_key = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    unsigned key[EPT];
    int slot = 0;                                  // every block-wide count uses a fresh row of per-wave LDS cells
    // number of elements in the block satisfying pred(j) (j = the thread's element slot): ballots + scalar popcounts per
    // wave, one LDS cell per wave, one barrier
    auto block_count = [&](auto pred) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < EPT; ++j) c += __popcll(__ballot(pred(j)));
        if (lane == 0) cnt[slot * NW + wave] = c;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += cnt[slot * NW + w];
        ++slot;
        return t;
    };

    if (tid == 0) n_cand = 0;
    // keys of the (scaled) logits, element j of this thread is index j * NT + tid; key 0 (below every real key) pads the tail
    unsigned bk = 0u;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = j * NT + tid;
        float f = i < V ? lg[i] : 0.f;
        if (sampling) f = f / p.temp;
        key[j] = i < V ? to_key(f) : 0u;
        if (key[j] > bk) { bk = key[j]; bi = i; }          // ascending i: the first maximum is kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned ok = __shfl_xor(bk, o);
        const int oi = __shfl_xor(bi, o);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
    }
    if (lane == 0) { red_j[wave] = (int)bk; red_i[wave] = bi; }
    __syncthreads();
    bk = (unsigned)red_j[0]; bi = red_i[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if ((unsigned)red_j[w] > bk || ((unsigned)red_j[w] == bk && red_i[w] < bi)) { bk = (unsigned)red_j[w]; bi = red_i[w]; }
    if (!sampling) {
        if (tid == 0) p.tokens[b * p.tok_stride] = bi;
        return;
    }
    // softmax denominator (fp32, max-subtracted like torch.softmax)
    const float mx = from_key(bk);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) s += key[j] ? expf(from_key(key[j]) - mx) : 0.f;
    s = wave_sum(s);
    if (lane == 0) red_v[wave] = s;
    __syncthreads();
    float denom = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) denom += red_v[w];

    // id blanking of sample_token_audio / sample_token_audio_2048 (utils/sampling.py:107-158): the probabilities of ids >= limit
    // are overwritten after the softmax over ALL ids, so the denominator above is untouched and the ids just leave the race
    int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    limit = limit > 0 && limit < V ? limit : V;
    if (limit < V) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) key[j] = j * NT + tid < limit ? key[j] : 0u;
    }
    // k-th largest key: binary search from the top bit down; stop as soon as a probe isolates exactly k keys
    const int k = min(p.top_k > 0 ? p.top_k : V, limit);
    unsigned thr = 0u;
    bool exact = false;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = thr | (1u << bit);
        const int n = block_count([&](int j) { return key[j] >= cand; });
        if (n >= k) thr = cand;
        if (n == k) { exact = true; break; }
    }
    int idx_lim = 0x7fffffff;            // ties (key == thr) with index <= idx_lim are taken
    if (!exact) {
        const int n_gt = block_count([&](int j) { return key[j] > thr; });
        const int n_eq = block_count([&](int j) { return key[j] == thr; });
        // of the n_eq elements equal to the threshold only the need = k - n_gt with the LOWEST indices belong to the top-k
        const int need = k - n_gt;
        if (need < n_eq) {
            int lim = 0;                 // largest L with count(ties, idx < L) < need, built bit by bit
            for (int bit = 16; bit >= 0; --bit) {
                const int cand = lim | (1 << bit);
                if (block_count([&](int j) { return key[j] == thr && j * NT + tid < cand; }) < need) lim = cand;
            }
            idx_lim = lim;
        }
    }
    // compact the exactly-k candidates (any order: ranks come from comparisons)
    const int kpad = (k + 7) & ~7;
    for (int i = k + tid; i < kpad; i += NT) comp[i] = 0ull;
    int wave_total = 0;
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        wave_total += __popcll(__ballot(key[j] > thr || (key[j] == thr && j * NT + tid <= idx_lim)));
    int base = 0;
    if (lane == 0) base = atomicAdd(&n_cand, wave_total);
    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const bool take = key[j] > thr || (key[j] == thr && j * NT + tid <= idx_lim);
        const unsigned long long mk = __ballot(take);
        const int at = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (take && at < k) comp[at] = ((unsigned long long)key[j] << 32) | (unsigned)(0x7fffffff - (j * NT + tid));
        base += __popcll(mk);
    }
    __syncthreads();
    float win = -INFINITY;
    int win_rank = 0x7fffffff, win_tok = 0;
    for (int c = tid; c < k; c += NT) {
        const unsigned long long mine = comp[c];
        int rank = 0;
        for (int j0 = 0; j0 < kpad; j0 += 8) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = comp[j0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += v[u] > mine ? 1 : 0;
        }
        const float sc = (expf(from_key((unsigned)(mine >> 32)) - mx) / denom) / p.noise[b * p.noise_stride + rank];
        if (sc > win || (sc == win && rank < win_rank)) { win = sc; win_rank = rank; win_tok = 0x7fffffff - (int)(unsigned)mine; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(win, o);
        const int orank = __shfl_xor(win_rank, o);
        const int ot = __shfl_xor(win_tok, o);
        if (ov > win || (ov == win && orank < win_rank)) { win = ov; win_rank = orank; win_tok = ot; }
    }
    __syncthreads();
    if (lane == 0) { red_v[wave] = win; red_i[wave] = win_rank; red_j[wave] = win_tok; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w)
            if (red_v[w] > win || (red_v[w] == win && red_i[w] < win_rank)) { win = red_v[w]; win_rank = red_i[w]; win_tok = red_j[w]; }
        p.tokens[b * p.tok_stride] = win_tok;
    }
}

// Large vocabularies (V > 32768, e.g. the 151 936-entry Qwen head): the keys no longer fit in registers, so the row (which
// is L2-resident: 0.6 MB) is read three times -- (1) arg-max + per-thread maxima, (2) softmax denominator, (3) candidate
// collection -- and the exact top-k is taken over a short candidate list:
//   the k-th largest of the 1024 per-thread maxima is a LOWER bound of the k-th largest key (k distinct elements reach it),
//   so {key >= that bound} contains the top-k and is typically only a little larger than k.
// Every candidate then computes its exact rank (count of larger (key, ~index) composites); ranks < k are the sorted top-k.
// If the candidate list overflows (plateaus of equal logits) the predicate is replaced by the exact one, found by bit-wise
// searches with counting passes over the row (always correct, slower).
constexpr int SAMPLE_BIG_CAP = 4096;
__global__ __launch_bounds__(1024) void sample_big_kernel(const LmSampleParams p) {
    constexpr int NT = 1024, NW = 16;
    __shared__ unsigned long long comp[SAMPLE_BIG_CAP];
    __shared__ float red_v[NW];
    __shared__ int red_i[NW], red_j[NW];
    __shared__ int cnt[96 * NW];
    __shared__ int n_cand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    const float* lg = p.logits + b * p.ld;
    const bool sampling = p.use_sampling && p.temp > 0.f;
    const int V = p.V;
    auto to_key = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto from_key = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    auto key_at = [&](int i) { return to_key(sampling ? lg[i] / p.temp : lg[i]); };
    int slot = 0;
    auto block_count = [&](int c) {            // sum of a per-thread count over the block (fresh LDS row per call)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) cnt[slot * NW + wave] = c;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += cnt[slot * NW + w];
        ++slot;
        return t;
    };
    int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    limit = limit > 0 && limit < V ? limit : V;
    if (tid == 0) n_cand = 0;

    // (1) arg-max over all ids (lowest index on ties) + this thread's maximum over the ids that may be drawn
    unsigned bk = 0u, tk = 0u;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += NT) {
        const unsigned u = key_at(i);
        if (u > bk) { bk = u; bi = i; }
        if (i < limit && u > tk) tk = u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned ok = __shfl_xor(bk, o);
        const int oi = __shfl_xor(bi, o);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
    }
    if (lane == 0) { red_j[wave] = (int)bk; red_i[wave] = bi; }
    __syncthreads();
    bk = (unsigned)red_j[0]; bi = red_i[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if ((unsigned)red_j[w] > bk || ((unsigned)red_j[w] == bk && red_i[w] < bi)) { bk = (unsigned)red_j[w]; bi = red_i[w]; }
    if (!sampling) {
        if (tid == 0) p.tokens[b * p.tok_stride] = bi;
        return;
    }
    // (2) softmax denominator over all ids
    const float mx = from_key(bk);
    float s = 0.f;
    for (int i = tid; i < V; i += NT) s += expf(from_key(key_at(i)) - mx);
    s = wave_sum(s);
    if (lane == 0) red_v[wave] = s;
    __syncthreads();
    float denom = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) denom += red_v[w];

    // lower bound of the k-th largest key: the k-th largest per-thread maximum
    const int k = min(min(p.top_k > 0 ? p.top_k : V, limit), NT);
    unsigned thr = 0u;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = thr | (1u << bit);
        if (block_count(tk >= cand ? 1 : 0) >= k) thr = cand;
    }
    int idx_lim = 0x7fffffff;          // ids equal to thr are taken up to this index
    bool strict_only = false;          // exact predicate: key > thr, or key == thr && i <= idx_lim
    auto take = [&](unsigned u, int i) { return i < limit && u != 0u && (u > thr || (u == thr && i <= idx_lim)); };
    // (3) collect
    auto collect = [&]() {
        for (int i = tid; i < V; i += NT) {
            const unsigned u = key_at(i);
            if (take(u, i)) {
                const int at = atomicAdd(&n_cand, 1);
                if (at < SAMPLE_BIG_CAP) comp[at] = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - i);
            }
        }
        __syncthreads();
        return n_cand;
    };
    int nc = thr != 0u ? collect() : SAMPLE_BIG_CAP + 1;
    if (nc > SAMPLE_BIG_CAP) {
        // exact k-th key by counting passes over the row, then the index bound among its ties
        __syncthreads();
        if (tid == 0) n_cand = 0;
        thr = 0u;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = thr | (1u << bit);
            int c = 0;
            for (int i = tid; i < limit; i += NT) c += key_at(i) >= cand ? 1 : 0;
            if (block_count(c) >= k) thr = cand;
        }
        int c_gt = 0;
        for (int i = tid; i < limit; i += NT) c_gt += key_at(i) > thr ? 1 : 0;
        const int need = k - block_count(c_gt);
        int lim = 0;                   // largest L with count(ties, idx < L) < need
#pragma unroll 1
        for (int bit = 20; bit >= 0; --bit) {
            const int cand = lim | (1 << bit);
            int c = 0;
            for (int i = tid; i < limit && i < cand; i += NT) c += key_at(i) == thr ? 1 : 0;
            if (block_count(c) < need) lim = cand;
        }
        idx_lim = lim;
        (void)strict_only;
        nc = collect();                // exactly k <= 1024 candidates
    }
    // exact ranks among the candidates; ranks < k are the sorted top-k
    float win = -INFINITY;
    int win_rank = 0x7fffffff, win_tok = 0;
    for (int c = tid; c < nc; c += NT) {
        const unsigned long long mine = comp[c];
        int rank = 0;
        for (int j = 0; j < nc; ++j) rank += comp[j] > mine ? 1 : 0;
        if (rank < k) {
            const float sc = (expf(from_key((unsigned)(mine >> 32)) - mx) / denom) / p.noise[b * p.noise_stride + rank];
            if (sc > win || (sc == win && rank < win_rank)) { win = sc; win_rank = rank; win_tok = 0x7fffffff - (int)(unsigned)mine; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(win, o);
        const int orank = __shfl_xor(win_rank, o);
        const int ot = __shfl_xor(win_tok, o);
        if (ov > win || (ov == win && orank < win_rank)) { win = ov; win_rank = orank; win_tok = ot; }
    }
    __syncthreads();
    if (lane == 0) { red_v[wave] = win; red_i[wave] = win_rank; red_j[wave] = win_tok; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w)
            if (red_v[w] > win || (red_v[w] == win && red_i[w] < win_rank)) { win = red_v[w]; win_rank = red_i[w]; win_tok = red_j[w]; }
        p.tokens[b * p.tok_stride] = win_tok;
    }
}

inline unsigned cap_grid(long g, long cap) { return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g)); }

}  // namespace

int rst_launch_gemv(const GemvParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.B <= 4 && p.N > 0 && p.K > 0 && p.K % 8 == 0, "gemv: need 1 <= B <= 4 and K %% 8 == 0 (B=%d K=%d)", p.B, p.K);
    RST_REQUIRE(p.x && p.w && p.y, "gemv: null pointer");
    RST_REQUIRE(p.prologue >= 0 && p.prologue <= 2 && (p.prologue != 1 || p.alpha), "gemv: bad prologue");
    RST_REQUIRE(p.act_out == 0 || p.act_out == 1, "gemv: act_out must be 0 (none) or 1 (GELU)");
    RST_REQUIRE(((uintptr_t)p.w % 16) == 0 && ((uintptr_t)p.x % 16) == 0, "gemv: pointers must be 16-byte aligned");
    const size_t lds = (size_t)p.B * p.K * sizeof(float);
    RST_REQUIRE(lds <= 128 * 1024, "gemv: B*K = %d floats do not fit the activation stage (32768)", p.B * p.K);
    // rows per wave: 4 when that still yields >= 2 workgroups per CU, else 2 (more workgroups -> more loads in flight)
    const bool rpw4 = !p.w_f32 && ((long)p.N + 15) / 16 >= 512;
    const int rows_per_group = (rpw4 ? 4 : 2) * GEMV_WAVES;
    const long groups = ((long)p.N + rows_per_group - 1) / rows_per_group;
    const unsigned grid = cap_grid(groups, lds > 48 * 1024 ? 512 : 768);
    auto go = [&](auto kern) {
        static bool attr_set = false;     // one flag per kernel instantiation (generic lambda)
        if (!attr_set) {
            // static LDS (the 16-byte reduction scratch) counts against the 160 KiB budget: ask for what the check above allows
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipGetLastError();
            attr_set = true;
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

int rst_launch_lm_rope_append(const LmRopeAppendParams& p, hipStream_t stream) {
    RST_REQUIRE(p.qkv && p.q && p.k && p.v && p.pos_dev && p.B >= 1 && p.T >= 1 && p.H > 0 && p.D > 0 && p.D % 2 == 0 && p.cap > 0,
                "lm_rope_append: bad arguments");
    RST_REQUIRE(p.G >= 1 && p.H % p.G == 0 && p.T <= p.cap && p.rope_dims >= 0 && p.rope_dims <= p.D && p.rope_dims % 2 == 0,
                "lm_rope_append: bad kv head count %d for %d heads, %d steps for capacity %d, or rope_dims %d", p.G, p.H, p.T, p.cap,
                p.rope_dims);
    const long total = (long)p.B * p.T * (p.H + p.G) * (p.D / 2);
    hipLaunchKernelGGL(rope_append_kernel, dim3(cap_grid((total + 255) / 256, 1024)), dim3(256), 0, stream, p);
    return rst_check_launch("lm_rope_append");
}

int rst_launch_lm_attn(const LmAttnParams& p, hipStream_t stream) {
    RST_REQUIRE((p.qkv || p.q_pre) && p.k && p.v && p.out && p.pos_dev && p.B >= 1 && p.H > 0 && p.cap > 0 && p.splits >= 1,
                "lm_attn: bad arguments");
    const int T = p.q_pre ? p.T : 1;
    RST_REQUIRE(T >= 1 && (long)p.B * T <= 65535 && p.H <= 65535 && p.D % 2 == 0, "lm_attn: bad sizes");
    RST_REQUIRE(p.G >= 1 && p.H % p.G == 0 && p.rope_dims >= 0 && p.rope_dims <= p.D && p.rope_dims % 2 == 0,
                "lm_attn: bad kv head count %d for %d heads or rope_dims %d", p.G, p.H, p.rope_dims);
    if (!p.q_pre && p.cap <= 64 && p.splits == 1) {
        switch (p.D) {
            case 32: hipLaunchKernelGGL(attn_small_kernel<32>, dim3(p.H, p.B), dim3(64), 0, stream, p); break;
            case 64: hipLaunchKernelGGL(attn_small_kernel<64>, dim3(p.H, p.B), dim3(64), 0, stream, p); break;
            case 128: hipLaunchKernelGGL(attn_small_kernel<128>, dim3(p.H, p.B), dim3(64), 0, stream, p); break;
            default:
                rst_set_error("lm_attn: head dim %d unsupported for short rings (32, 64, 128)", p.D);
                return RST_ERR_UNSUPPORTED;
        }
        return rst_check_launch("lm_attn_small");
    }
    RST_REQUIRE(p.splits == 1 || (p.ws && p.counters), "lm_attn: splits > 1 need the workspace and the (zeroed) counters");
    const dim3 grid(p.splits, p.H, p.B * T);
    switch (p.D) {
        case 64: hipLaunchKernelGGL(attn_decode_kernel<64>, grid, dim3(256), 0, stream, p); break;
        case 128: hipLaunchKernelGGL(attn_decode_kernel<128>, grid, dim3(256), 0, stream, p); break;
        default:
            rst_set_error("lm_attn: head dim %d unsupported for long rings (64, 128)", p.D);
            return RST_ERR_UNSUPPORTED;
    }
    return rst_check_launch("lm_attn");
}

int rst_launch_lm_sample(const LmSampleParams& p, hipStream_t stream) {
    RST_REQUIRE(p.logits && p.tokens && p.B >= 1 && p.V > 0 && p.V <= (1 << 20), "lm_sample: bad arguments");
    RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || p.noise, "lm_sample: sampling needs the exponential noise tensor");
    const int k = p.top_k > 0 && p.top_k < p.V ? p.top_k : p.V;
    RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || k <= 8192, "lm_sample: top-k %d exceeds the 8192 candidate stage", k);
    const size_t lds = (size_t)((k + 7) & ~7) * 8;
    if (p.V <= 2048) hipLaunchKernelGGL((sample_kernel<256, 8>), dim3(p.B), dim3(256), lds, stream, p);
    else if (p.V <= 4096) hipLaunchKernelGGL((sample_kernel<256, 16>), dim3(p.B), dim3(256), lds, stream, p);
    else if (p.V <= 32768) hipLaunchKernelGGL((sample_kernel<1024, 32>), dim3(p.B), dim3(1024), lds, stream, p);
    else {
        RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || k <= 1024, "lm_sample: top-k %d > 1024 for a vocabulary of %d", k, p.V);
        hipLaunchKernelGGL(sample_big_kernel, dim3(p.B), dim3(1024), 0, stream, p);
    }
    return rst_check_launch("lm_sample");
}

// ======================================================================================================================
// Skinny GEMM for 4 < batch <= 64: y[b][n] = (res +) (bias +) sum_k x[b][k] * W[n][k] on the bf16 matrix cores.
// The weight matrix is streamed from HBM exactly once (the step stays bandwidth-bound up to batch ~64); the contraction runs
// on v_mfma_f32_32x32x16_bf16 with the fp32 activations split into bf16 hi + lo parts (two MFMAs per step): products carry
// ~17 mantissa bits of x, i.e. fp32-class accuracy against the fp32 oracle (the reference itself rounds activations to bf16).
// ======================================================================================================================
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

// ---- operand layouts of the skinny GEMM ------------------------------------------------------------------------------
// Both operands are stored in the order the 32x32x16 MFMA consumes them, so that every wave-level load is ONE contiguous
// kilobyte: [tile of 32 rows][step of 16 k][lane = 32 * (k / 8 % 2) + row % 32][8 bf16].
//   weights  Wp : [ceil(N/32)][K/16][64][8]   (rows beyond N zero)        -- packed once per weight (rst_skinny_pack_weight_bf16)
//   activations Xp: [2 = hi, lo][ceil(B/32)][K/16][64][8] (rows beyond B zero) -- packed per call by the (fused) prologue kernel
__device__ __forceinline__ long packed_index(int row, int k, int K) {
    return ((((long)(row >> 5) * (K >> 4) + (k >> 4)) * 64) + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
}

__global__ __launch_bounds__(256) void skinny_pack_weight_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wp,
                                                                int N, int K) {
    const long total = (long)((N + 31) / 32) * 32 * (K / 8);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / (K / 8)), k = (int)(idx % (K / 8)) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < N) v = *reinterpret_cast<const u32x4*>(w + (long)row * K + k);
        *reinterpret_cast<u32x4*>(wp + packed_index(row, k, K)) = v;
    }
}

// Prologue + hi/lo split + packing of one activation row per workgroup (pad rows of the last batch tile are zero-filled).
// hi = fp32 truncated to bf16 (an exact prefix, so x - hi is exact), lo = the residual rounded half-up: x = hi + lo to 2^-17.
// mode 0: identity; 1: RMSNorm x * alpha * rsqrt(eps + mean(x^2)); 2: SiLU gate, x row = [u ; v] -> silu(u) * v.
__global__ __launch_bounds__(256) void skinny_pack_act_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                             unsigned short* __restrict__ xp, int B, int K, int ldx, int mode, float eps) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long half = (long)((B + 31) / 32) * 32 * K;          // elements of the hi (and of the lo) plane
    float scale = 1.f;
    if (mode == 1 && b < B) {
        float s = 0.f;
        for (int k = tid * 4; k < K; k += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k);
            s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
        }
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        scale = 1.0f / sqrtf(eps + (red[0] + red[1] + red[2] + red[3]) / (float)K);
    }
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (b < B) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a0[j]; v[4 + j] = a1[j]; }
            if (mode == 1) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(alpha + k), g1 = *reinterpret_cast<const f32x4*>(alpha + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = v[j] * (g0[j] * scale); v[4 + j] = v[4 + j] * (g1[j] * scale); }
            } else if (mode == 2) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = silu(v[j]) * g0[j]; v[4 + j] = silu(v[4 + j]) * g1[j]; }
            }
        }
        unsigned h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned ua = __float_as_uint(v[2 * j]), ub = __float_as_uint(v[2 * j + 1]);
            h[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                 // {hi16(b), hi16(a)}
            const float ra = v[2 * j] - __uint_as_float(ua & 0xffff0000u), rb = v[2 * j + 1] - __uint_as_float(ub & 0xffff0000u);
            l[j] = __builtin_amdgcn_perm(__float_as_uint(rb) + 0x8000u, __float_as_uint(ra) + 0x8000u, 0x07060302u);
        }
        const long at = packed_index(b, k, K);
        *reinterpret_cast<u32x4*>(xp + at) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(xp + half + at) = u32x4{l[0], l[1], l[2], l[3]};
    }
}

constexpr int SKINNY_WAVES = 8;

// One workgroup = CT adjacent tiles of 32 weight rows; its 8 waves each take an eighth of K.  Per MFMA step a wave loads
// 1 KB of activations (hi), 1 KB (lo) -- L2 hits -- and CT x 1 KB of weights from HBM, all contiguous; no LDS stage and no
// barrier in the main loop.  The activations are the MFMA "A" side, so an accumulator is C[b = row(e, lane)][n = lane & 31]
// and the global stores are 128-byte coalesced.  The 8 partial tiles meet in LDS and are summed in wave order
// (deterministic); there is no cross-workgroup reduction.
template <int NB, int CT>   // batch tiles of 32, weight-row tiles per workgroup
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_skinny_kernel(const SkinnyParams p) {
    __shared__ float red[SKINNY_WAVES][NB * 32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = (p.N + 31) / 32;
    const int tile0 = blockIdx.x * CT;
    const int steps = p.K / 16;
    const int per = (steps + SKINNY_WAVES - 1) / SKINNY_WAVES;
    const int s0 = wave * per, s1 = min(steps, s0 + per);
    const long xplane = (long)NB * 32 * p.K;                        // elements of the hi plane
    const unsigned short* xh = p.xp + (long)lane * 8;
    const unsigned short* xl = xh + xplane;
    const unsigned short* wt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wt[c] = p.w + ((long)min(tile0 + c, tiles - 1) * steps * 64 + lane) * 8;
    f32x16 acc[NB][CT];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;
    constexpr int UN = (NB * 2 + CT) <= 4 ? 4 : 2;
    for (int s = s0; s < s1; s += UN) {
        bf16x8 a[UN][CT], bh[UN][NB], bl[UN][NB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = s + u < s1;
            const long so = (long)(s + u) * 512;
#pragma unroll
            for (int c = 0; c < CT; ++c)
                a[u][c] = ok ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wt[c] + so)) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                bh[u][t] = ok ? *reinterpret_cast<const bf16x8*>(xh + (long)t * steps * 512 + so) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                bl[u][t] = ok ? *reinterpret_cast<const bf16x8*>(xl + (long)t * steps * 512 + so) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[u][t], a[u][c], acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[u][t], a[u][c], acc[t][c], 0, 0, 0);
                }
    }
    const int i = lane & 31;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t * 32 + rst_mfma32_row(e, lane)][i] = acc[t][c][e];
        __syncthreads();
        const int n0 = (tile0 + c) * 32;
        for (int idx = tid; idx < NB * 32 * 32; idx += 64 * SKINNY_WAVES) {
            const int b = idx >> 5, nl = idx & 31;
            const int n = n0 + nl;
            if (b < p.B && n < p.N) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < SKINNY_WAVES; ++w) s += red[w][b][nl];
                const long o = (long)b * p.ldy + n;
                if (p.bias) s += p.bias[n];
                p.y[o] = p.res ? p.res[o] + s : s;
            }
        }
    }
}

}  // namespace

int rst_launch_skinny_pack_weight(const unsigned short* w, unsigned short* wp, int N, int K, hipStream_t stream) {
    RST_REQUIRE(w && wp && N > 0 && K > 0 && K % 16 == 0, "skinny_pack_weight: bad arguments (K %% 16 == 0 required, K=%d)", K);
    const long total = (long)((N + 31) / 32) * 32 * (K / 8);
    hipLaunchKernelGGL(skinny_pack_weight_kernel, dim3(cap_grid((total + 255) / 256, 8192)), dim3(256), 0, stream, w, wp, N, K);
    return rst_check_launch("skinny_pack_weight");
}

int rst_launch_skinny_pack_act(const float* x, const float* alpha, unsigned short* xp, int B, int K, int ldx, int mode, float eps,
                               hipStream_t stream) {
    RST_REQUIRE(x && xp && B >= 1 && B <= 64 && K > 0 && K % 16 == 0 && ldx % 4 == 0, "skinny_pack_act: bad arguments (B=%d K=%d)", B, K);
    RST_REQUIRE(mode == 0 || (mode == 1 && alpha) || mode == 2, "skinny_pack_act: mode 0 / 1 (needs alpha) / 2");
    hipLaunchKernelGGL(skinny_pack_act_kernel, dim3((B + 31) / 32 * 32), dim3(256), 0, stream, x, alpha, xp, B, K, ldx, mode, eps);
    return rst_check_launch("skinny_pack_act");
}

int rst_launch_gemm_skinny(const SkinnyParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.B <= 64 && p.N > 0 && p.K > 0 && p.K % 16 == 0, "gemm_skinny: need 1 <= B <= 64 and K %% 16 == 0 (B=%d K=%d)", p.B, p.K);
    RST_REQUIRE(p.xp && p.w && p.y, "gemm_skinny: null pointer");
    const int tiles = (p.N + 31) / 32;
    const int threads = 64 * SKINNY_WAVES;
    if (p.B <= 32) {
        if (tiles >= 2048) hipLaunchKernelGGL((gemm_skinny_kernel<1, 4>), dim3((tiles + 3) / 4), dim3(threads), 0, stream, p);
        else if (tiles >= 512) hipLaunchKernelGGL((gemm_skinny_kernel<1, 2>), dim3((tiles + 1) / 2), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_kernel<1, 1>), dim3(tiles), dim3(threads), 0, stream, p);
    } else {
        if (tiles >= 512) hipLaunchKernelGGL((gemm_skinny_kernel<2, 2>), dim3((tiles + 1) / 2), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_kernel<2, 1>), dim3(tiles), dim3(threads), 0, stream, p);
    }
    return rst_check_launch("gemm_skinny");
}

// ======================================================================================================================
// fp8 (OCP e4m3) variant of the skinny GEMM -- BASELINE.json configs[4]: "fp8 MFMA GEMMs for temporal attention, batch 32".
// Weights: per-row scale (amax / 448), quantised once; activations: per-row dynamic scale, quantised by the prologue launch.
// v_mfma_f32_32x32x16_fp8_fp8 consumes 8 bytes per lane per step; two steps are packed per 16-byte lane load:
//   [tile of 32 rows][K/32][64 lanes = 32 * ((k / 8) % 2) + row % 32][16 B = step 2p (8 k) | step 2p+1 (8 k)].
// Half the streamed bytes and half the MFMA work of the bf16 hi/lo path, at fp8 accuracy (3 mantissa bits: ~1e-2 on logits);
// opt-in, never the default.
// ======================================================================================================================
namespace {

__device__ __forceinline__ long fp8_packed_index(int row, int k, int K) {     // byte offset
    return ((((long)(row >> 5) * (K >> 5) + (k >> 5)) * 64) + ((k >> 3) & 1) * 32 + (row & 31)) * 16 + ((k >> 4) & 1) * 8 + (k & 7);
}

// v / scale with a true division (bit-compatible with `(t / scale).to(float8_e4m3fn)`), round-to-nearest-even conversion
__device__ __forceinline__ uint2 quant8_fp8(const float (&v)[8], float scale) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] / scale, v[1] / scale, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] / scale, v[3] / scale, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] / scale, v[5] / scale, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] / scale, v[7] / scale, hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

__device__ __forceinline__ float block_max_256(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one workgroup per (padded) weight row
__global__ __launch_bounds__(256) void skinny_pack_weight_fp8_kernel(const unsigned short* __restrict__ w, unsigned char* __restrict__ wp,
                                                                    float* __restrict__ scale, int N, int K) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float amax = 0.f;
    if (row < N)
        for (int k = tid * 8; k < K; k += 2048) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(w + (long)row * K + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16_lo(v[j])), fabsf(bf16_hi(v[j]))));
        }
    amax = block_max_256(amax, red);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) scale[row] = sc;
    const float inv = sc;
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row < N) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(w + (long)row * K + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_lo(q[j]); v[2 * j + 1] = bf16_hi(q[j]); }
        }
        *reinterpret_cast<uint2*>(wp + fp8_packed_index(row, k, K)) = quant8_fp8(v, inv);
    }
}

// one workgroup per (padded) batch row: prologue (0 identity, 1 RMSNorm, 2 SiLU gate) -> per-row amax -> fp8
constexpr int FP8_ACT_CHUNKS = 8;      // K <= 8 * 2048
__global__ __launch_bounds__(256) void skinny_pack_act_fp8_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                                 unsigned char* __restrict__ xp, float* __restrict__ xscale, int B, int K,
                                                                 int ldx, int mode, float eps) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float vals[FP8_ACT_CHUNKS][8];
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < FP8_ACT_CHUNKS; ++c) {
        const int k = tid * 8 + c * 2048;
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[c][j] = 0.f;
        if (b < B && k < K) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { vals[c][j] = a0[j]; vals[c][4 + j] = a1[j]; }
            if (mode == 2) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vals[c][j] = silu(vals[c][j]) * g0[j]; vals[c][4 + j] = silu(vals[c][4 + j]) * g1[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ssq = fmaf(vals[c][j], vals[c][j], ssq);
        }
    }
    if (mode == 1) {
        float s = wave_sum(ssq);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const float r = 1.0f / sqrtf(eps + (red[0] + red[1] + red[2] + red[3]) / (float)K);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < FP8_ACT_CHUNKS; ++c) {
            const int k = tid * 8 + c * 2048;
            if (b < B && k < K) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(alpha + k), g1 = *reinterpret_cast<const f32x4*>(alpha + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vals[c][j] *= g0[j] * r; vals[c][4 + j] *= g1[j] * r; }
            }
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < FP8_ACT_CHUNKS; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(vals[c][j]));
    amax = block_max_256(amax, red);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) xscale[b] = sc;
    const float inv = sc;
#pragma unroll
    for (int c = 0; c < FP8_ACT_CHUNKS; ++c) {
        const int k = tid * 8 + c * 2048;
        if (k < K) *reinterpret_cast<uint2*>(xp + fp8_packed_index(b, k, K)) = quant8_fp8(vals[c], inv);
    }
}

template <int NB, int CT>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_skinny_fp8_kernel(const SkinnyFp8Params p) {
    __shared__ float red[SKINNY_WAVES][NB * 32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = (p.N + 31) / 32;
    const int tile0 = blockIdx.x * CT;
    const int pairs = p.K / 32;
    const int per = (pairs + SKINNY_WAVES - 1) / SKINNY_WAVES;
    const int s0 = wave * per, s1 = min(pairs, s0 + per);
    const unsigned char* xq = p.xp + (long)lane * 16;
    const unsigned char* wt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wt[c] = p.wp + ((long)min(tile0 + c, tiles - 1) * pairs * 64 + lane) * 16;
    f32x16 acc[NB][CT];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;
    typedef long i64x2 __attribute__((ext_vector_type(2)));
    constexpr int UN = 4;
    for (int s = s0; s < s1; s += UN) {
        i64x2 a[UN][CT], bx[UN][NB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = s + u < s1;
            const long so = (long)(s + u) * 1024;
#pragma unroll
            for (int c = 0; c < CT; ++c) a[u][c] = ok ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(wt[c] + so)) : i64x2{0, 0};
#pragma unroll
            for (int t = 0; t < NB; ++t) bx[u][t] = ok ? *reinterpret_cast<const i64x2*>(xq + (long)t * pairs * 1024 + so) : i64x2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bx[u][t][0], a[u][c][0], acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bx[u][t][1], a[u][c][1], acc[t][c], 0, 0, 0);
                }
    }
    const int i = lane & 31;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t * 32 + rst_mfma32_row(e, lane)][i] = acc[t][c][e];
        __syncthreads();
        const int n0 = (tile0 + c) * 32;
        for (int idx = tid; idx < NB * 32 * 32; idx += 64 * SKINNY_WAVES) {
            const int b = idx >> 5, nl = idx & 31;
            const int n = n0 + nl;
            if (b < p.B && n < p.N) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < SKINNY_WAVES; ++w) s += red[w][b][nl];
                s *= p.xscale[b] * p.wscale[n];
                const long o = (long)b * p.ldy + n;
                if (p.bias) s += p.bias[n];
                p.y[o] = p.res ? p.res[o] + s : s;
            }
        }
    }
}

}  // namespace

int rst_launch_skinny_pack_weight_fp8(const unsigned short* w, unsigned char* wp, float* scale, int N, int K, hipStream_t stream) {
    RST_REQUIRE(w && wp && scale && N > 0 && K > 0 && K % 32 == 0, "skinny_pack_weight_fp8: bad arguments (K %% 32 == 0 required, K=%d)", K);
    hipLaunchKernelGGL(skinny_pack_weight_fp8_kernel, dim3((N + 31) / 32 * 32), dim3(256), 0, stream, w, wp, scale, N, K);
    return rst_check_launch("skinny_pack_weight_fp8");
}

int rst_launch_skinny_pack_act_fp8(const float* x, const float* alpha, unsigned char* xp, float* xscale, int B, int K, int ldx, int mode,
                                   float eps, hipStream_t stream) {
    RST_REQUIRE(x && xp && xscale && B >= 1 && B <= 64 && K > 0 && K % 32 == 0 && K <= 2048 * FP8_ACT_CHUNKS && ldx % 4 == 0,
                "skinny_pack_act_fp8: bad arguments (B=%d K=%d; K %% 32 == 0, K <= %d)", B, K, 2048 * FP8_ACT_CHUNKS);
    RST_REQUIRE(mode == 0 || (mode == 1 && alpha) || mode == 2, "skinny_pack_act_fp8: mode 0 / 1 (needs alpha) / 2");
    hipLaunchKernelGGL(skinny_pack_act_fp8_kernel, dim3((B + 31) / 32 * 32), dim3(256), 0, stream, x, alpha, xp, xscale, B, K, ldx, mode, eps);
    return rst_check_launch("skinny_pack_act_fp8");
}

int rst_launch_gemm_skinny_fp8(const SkinnyFp8Params& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.B <= 64 && p.N > 0 && p.K > 0 && p.K % 32 == 0, "gemm_skinny_fp8: need 1 <= B <= 64 and K %% 32 == 0 (B=%d K=%d)", p.B, p.K);
    RST_REQUIRE(p.xp && p.wp && p.xscale && p.wscale && p.y, "gemm_skinny_fp8: null pointer");
    const int tiles = (p.N + 31) / 32;
    const int threads = 64 * SKINNY_WAVES;
    if (p.B <= 32) {
        if (tiles >= 2048) hipLaunchKernelGGL((gemm_skinny_fp8_kernel<1, 4>), dim3((tiles + 3) / 4), dim3(threads), 0, stream, p);
        else if (tiles >= 512) hipLaunchKernelGGL((gemm_skinny_fp8_kernel<1, 2>), dim3((tiles + 1) / 2), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_fp8_kernel<1, 1>), dim3(tiles), dim3(threads), 0, stream, p);
    } else {
        if (tiles >= 512) hipLaunchKernelGGL((gemm_skinny_fp8_kernel<2, 2>), dim3((tiles + 1) / 2), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_fp8_kernel<2, 1>), dim3(tiles), dim3(threads), 0, stream, p);
    }
    return rst_check_launch("gemm_skinny_fp8");
}
