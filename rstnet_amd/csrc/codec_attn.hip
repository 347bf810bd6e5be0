// One streaming step of a codec transformer layer's attention for MANY streams (the layer loop that serves more than two streams per
// step; <= 2 streams run the whole transformer as one persistent launch, codec_tr.hip): split of the in-projection's output, interleaved
// RoPE on q and k, append of k / v to the ring, and the T new queries against the ring -- ONE launch, one workgroup per (stream, head).
// As separate launches this was rope_split (4.9 us) + a split-K decode attention with a last-arriver combine (7.3 us) + their
// boundaries, 16 times per 80 ms frame; at 32 streams x 8 heads the (stream, head) pairs alone fill the 256 CUs, so the ring walk needs
// no split across workgroups and its combine stays in LDS.
//
// The ring walk is the one of codec_tr.hip (round 4): thread (dq, cls) owns dims 4 dq .. 4 dq + 3 of the ring slots of class cls
// (slot % NC == cls, NC = 1024 / D) in the ring append AND in the sweeps, so a new step's row is read back by the very thread that stored
// it (program order); the K and V rows of 16 slots per class are requested together; q . k is summed over the D / 4 lanes of a row
// with DPP steps; the softmax is the reference's two-pass form (row maximum, numerators, row sum) with one slot per thread;
// o[t] += p[t][slot] v accumulates all T queries from one read of V.  Mask and slot -> position map: RingKVCache.complete
// (modules/transformer.py:254-278,404-414) incl. its `delta <= 0` slot.
#include "lm_common.h"

namespace {

constexpr int AS_THREADS = 256, AS_WAVES = 4;

template <int T>
__global__ __launch_bounds__(AS_THREADS) void attn_step_kernel(const AttnStepParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.D, H = p.H, cap = p.cap, E = H * D;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int capS = cap + 16384 / D;           // the ring rounded up to whole batches of 16 slots per slot class
    // LDS carve (floats): qh [T][3][D] | sraw [T][capS] | pB [capS][T] | opart [1024 / D][T][D] | wred [2][4][T]
    float* qh = lds;
    float* sraw = qh + T * 3 * D;
    float* pB = sraw + T * capS;
    float* opart = pB + T * capS;
    float* wred = opart + 1024 * T;
    const long pos = *p.pos_dev;

    float* kring = p.k + ((long)(b * H + h) * cap) * D;
    float* vring = p.v + ((long)(b * H + h) * cap) * D;
    const int GS = D >> 2, NC = AS_THREADS / GS, dq = tid % GS, cls = tid / GS;
    const int slot0 = (int)(pos % cap);            // new step t sits in slot (slot0 + t) % cap
    const long end_offset = pos + T;
    const int n_used = (int)min((long)cap, end_offset);
    const int end_index = (int)(end_offset % cap);
    const float scale = 1.0f / sqrtf((float)D);
    constexpr int JB = 16;                       // slots per class and batch
    const int SPB = JB * NC;                     // slots per batch
    const int nb = (n_used + SPB - 1) / SPB;
    f32x4 kreg[JB], vreg[JB];
    auto k_issue = [&](int base) {
#pragma unroll
        for (int j = 0; j < JB; ++j) kreg[j] = *reinterpret_cast<const f32x4*>(kring + min(base + cls + NC * j, n_used - 1) * D + 4 * dq);
    };
    // (a slot past the used part of the ring is never loaded: a ring the caller did not zero may hold Inf / NaN there)
    auto v_issue = [&](int base) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int sl = base + cls + NC * j;
            vreg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (sl < n_used) vreg[j] = *reinterpret_cast<const f32x4*>(vring + sl * D + 4 * dq);
        }
    };
    // the new steps' own slots hold the PREVIOUS pass of the ring (or nothing yet) when the first batch is requested, which is before the
    // new rows are stored -- the first batch's round trip runs under the load + rotation of q / k / v -- so those registers take the new
    // rows from LDS instead (the thread that owns a slot's dims stores AND patches them: no ordering question)
    auto patch = [&](int base) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int sl = slot0 + t;
            sl -= sl >= cap ? cap : 0;
#pragma unroll
            for (int j = 0; j < JB; ++j)
                if (base + cls + NC * j == sl) {
                    kreg[j] = *reinterpret_cast<const f32x4*>(qh + (t * 3 + 1) * D + 4 * dq);
                    vreg[j] = *reinterpret_cast<const f32x4*>(qh + (t * 3 + 2) * D + 4 * dq);
                }
        }
    };
    // q / k / v of the head for the T new steps: requested FIRST (loads return in order: the rotation below then starts while the ring's
    // first batch is still on its way), rotated (modules/rope.py:37-62, position = pos + t; the expressions of rope_split_kernel) into
    // LDS: item = (t, part, pair)
    const int half = D >> 1;
    const int items = T * 3 * half;
    constexpr int NIT = (T * 3 * 64 + AS_THREADS - 1) / AS_THREADS;       // D <= 128
    float re[NIT], im[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = min(tid + u * AS_THREADS, items - 1);              // clamped: unconditional loads
        const int t = i / (3 * half), j = i - t * 3 * half, part = j / half, pr = j - part * half;
        const float* src = p.qkv + ((long)(b * T + t) * 3 + part) * E + h * D + 2 * pr;
        re[u] = src[0];
        im[u] = src[1];
    }
    k_issue(0);
    v_issue(0);
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = tid + u * AS_THREADS;
        if (i < items) {
            const int t = i / (3 * half), j = i - t * 3 * half, part = j / half, pr = j - part * half;
            float c = 1.0f, sn = 0.0f;
            if (p.rope && part < 2) {
                const float ang = expf((float)pr * p.rope_coef) * ((float)pos + (float)t);
                c = cosf(ang);
                sn = sinf(ang);
            }
            float* dst = qh + (t * 3 + part) * D + 2 * pr;
            dst[0] = re[u] * c - im[u] * sn;
            dst[1] = re[u] * sn + im[u] * c;
        }
    }
    __syncthreads();
    // ring append: thread (dq, cls) stores dims 4 dq .. 4 dq + 3 of the new rows whose slot is of its class
#pragma unroll
    for (int t = 0; t < T; ++t) {
        int sl = slot0 + t;
        sl -= sl >= cap ? cap : 0;
        if ((sl & (NC - 1)) == cls) {
            *reinterpret_cast<f32x4*>(kring + sl * D + 4 * dq) = *reinterpret_cast<const f32x4*>(qh + (t * 3 + 1) * D + 4 * dq);
            *reinterpret_cast<f32x4*>(vring + sl * D + 4 * dq) = *reinterpret_cast<const f32x4*>(qh + (t * 3 + 2) * D + 4 * dq);
        }
    }
    patch(0);
    f32x4 q4[T];
#pragma unroll
    for (int t = 0; t < T; ++t) q4[t] = *reinterpret_cast<const f32x4*>(qh + (t * 3) * D + 4 * dq);
    // (A) scores
    for (int bt = 0; bt < nb; ++bt) {
        if (bt) k_issue(bt * SPB);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            float tot[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float a = kreg[j][0] * q4[t][0];
                a = fmaf(kreg[j][1], q4[t][1], a); a = fmaf(kreg[j][2], q4[t][2], a); a = fmaf(kreg[j][3], q4[t][3], a);
                tot[t] = group_sum(a, GS);
            }
            if (dq == 0) {
#pragma unroll
                for (int t = 0; t < T; ++t) sraw[t * capS + bt * SPB + cls + NC * j] = tot[t];
            }
        }
    }
    __syncthreads();
    // softmax, one slot per thread and sweep
    float tmax[T], M[T], tsum[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { tmax[t] = -INFINITY; tsum[t] = 0.f; }
    for (int sl = tid; sl < nb * SPB; sl += AS_THREADS) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bool ok = sl < n_used && ring_visible_at(sl, pos + t, cap, p.context, end_offset, end_index);
            const float sv = ok ? sraw[t * capS + sl] * scale : -INFINITY;
            sraw[t * capS + sl] = sv;
            tmax[t] = fmaxf(tmax[t], sv);
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float m = wave_max_fast(tmax[t]);
        if (lane == 0) wred[wave * T + t] = m;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t) M[t] = fmaxf(fmaxf(wred[t], wred[T + t]), fmaxf(wred[2 * T + t], wred[3 * T + t]));
    for (int sl = tid; sl < nb * SPB; sl += AS_THREADS) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float sv = sraw[t * capS + sl];
            const float pv = sv == -INFINITY ? 0.f : expf(sv - M[t]);
            pB[sl * T + t] = pv;
            tsum[t] += pv;
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float a = wave_sum_fast(tsum[t]);
        if (lane == 0) wred[(AS_WAVES + wave) * T + t] = a;
    }
    __syncthreads();
    // (B) weighted values
    f32x4 o[T];
#pragma unroll
    for (int t = 0; t < T; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int bt = 0; bt < nb; ++bt) {
        if (bt) v_issue(bt * SPB);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const float* pr = pB + (bt * SPB + cls + NC * j) * T;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float w = pr[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[t][e] = fmaf(w, vreg[j][e], o[t][e]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(opart + (cls * T + t) * D + 4 * dq) = o[t];
    __syncthreads();
    for (int i = tid; i < T * D; i += AS_THREADS) {
        const int t = i / D, d = i - t * D;
        float Oq = 0.f;
        for (int c = 0; c < NC; ++c) Oq += opart[(c * T + t) * D + d];
        const float* ws = wred + AS_WAVES * T + t;
        const float Lq = (ws[0] + ws[T]) + (ws[2 * T] + ws[3 * T]);
        const float r = Lq > 0.f ? Oq / Lq : 0.f;
        // out_rows: the result is the out-projection's operand in the few-row GEMM's packed order (no packing launch, no strided row reads)
        if (p.out_rows) p.out[f32_packed_index(b * T + t, h * D + d, E)] = r;
        else p.out[(long)(b * T + t) * E + h * D + d] = r;
    }
    if (p.out_rows && b == 0) {          // the pad rows of the last 32-row tile read as zeros in the consumer: stream 0's workgroups clear their head's columns
        const int M = p.B * T;
        for (int i = tid; i < (p.out_rows - M) * D; i += AS_THREADS) p.out[f32_packed_index(M + i / D, h * D + i % D, E)] = 0.f;
    }
}

size_t as_lds_bytes(int T, int D, int cap) {
    return ((size_t)T * 3 * D + 2 * (size_t)T * (cap + 16384 / D) + (size_t)(1024 + 2 * AS_WAVES) * T) * sizeof(float);
}

}  // namespace

// Shapes the one-launch step serves: a handful of new positions, the head dims of the hot path, a ring whose scores fit LDS
int rst_attn_step_supported_impl(int T, int D, int cap) {
    if (!(T >= 1 && T <= 4 && (D == 32 || D == 64 || D == 128) && cap >= T)) return 0;
    return as_lds_bytes(T, D, cap) <= 144 * 1024 ? 1 : 0;
}

int rst_launch_attn_step(const AttnStepParams& p, hipStream_t stream) {
    RST_REQUIRE(p.qkv && p.k && p.v && p.out && p.pos_dev && p.B >= 1 && p.H >= 1, "attention_step: null buffers / bad sizes");
    RST_REQUIRE(rst_attn_step_supported_impl(p.T, p.D, p.cap), "attention_step: unsupported shape (T=%d D=%d cap=%d; 1 <= T <= 4, D in 32 / 64 / 128)",
                p.T, p.D, p.cap);
    RST_REQUIRE(p.out_rows == 0 || (p.out_rows >= p.B * p.T && p.out_rows % 32 == 0 && (p.H * p.D) % 8 == 0),
                "attention_step: a packed result needs whole 32-row tiles (out_rows=%d for %d rows) and H * D %% 8 == 0", p.out_rows, p.B * p.T);
    RST_REQUIRE((uintptr_t)p.qkv % 16 == 0 && (uintptr_t)p.k % 16 == 0 && (uintptr_t)p.v % 16 == 0, "attention_step: 16-byte aligned buffers required");
    const size_t lds = as_lds_bytes(p.T, p.D, p.cap);
    auto go = [&](auto kern) {
        if (lds > 48 * 1024) {
            static RstOncePerDevice attr_once;       // one flag per kernel instance
            if (attr_once.first()) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
                (void)hipGetLastError();
            }
        }
        hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(AS_THREADS), lds, stream, p);
    };
    switch (p.T) {
        case 1: go(attn_step_kernel<1>); break;
        case 2: go(attn_step_kernel<2>); break;
        case 3: go(attn_step_kernel<3>); break;
        default: go(attn_step_kernel<4>); break;
    }
    return rst_check_launch("attention_step");
}
