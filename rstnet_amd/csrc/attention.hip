// fp32 attention for the Mimi transformers: RoPE + head split (+ ring-KV append) and a flash-style
// masked attention on the f32 matrix cores.
//
// The attention kernel works on the TRANSPOSED score tile  S^T = K * Q^T  (32 keys x 32 queries per
// v_mfma_f32_32x32x2_f32 accumulator): a lane then owns ONE query column, so the online-softmax row
// reductions are 16 in-register ops + one cross-half shuffle, and the probabilities P^T sit in exactly the
// lane/register positions the second product  O^T = V^T * P^T  needs for its B operand -- no LDS round
// trip for P.  One wave per (batch, head, 32-query tile); K/V tiles are staged through LDS.
#include "rst_common.h"
#include "rst_kernels.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void rope_split_kernel(const RopeSplitParams p) {
    const int half = p.D / 2;
    const long total = (long)p.B * p.T * p.H * half;
    const long pos0 = p.pos_dev ? *p.pos_dev : p.pos0;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx % half);
        const int h = (int)((idx / half) % p.H);
        const int t = (int)((idx / ((long)half * p.H)) % p.T);
        const long b = idx / ((long)half * p.H * p.T);
        const float* src = p.qkv + ((b * p.T + t) * 3) * (long)(p.H * p.D) + (long)h * p.D + 2 * i;
        const float qr = src[0], qi = src[1];
        const float kr = src[(long)p.H * p.D], ki = src[(long)p.H * p.D + 1];
        const float vr = src[2L * p.H * p.D], vi = src[2L * p.H * p.D + 1];
        float c = 1.0f, s = 0.0f;
        if (p.rope) {
            // modules/rope.py:37-40: freqs = exp(ds * coef) (fp32), ts = offset.float() + arange(T).float()
            const float fr = expf((float)i * p.rope_coef);
            const float ang = fr * ((float)pos0 + (float)t);
            c = cosf(ang);
            s = sinf(ang);
        }
        const int slot = p.ring ? (int)((pos0 + t) % p.cap) : t;
        float* qd = p.q + ((b * p.H + h) * p.T + t) * (long)p.D + 2 * i;
        float* kd = p.k + ((b * p.H + h) * p.cap + slot) * (long)p.D + 2 * i;
        float* vd = p.v + ((b * p.H + h) * p.cap + slot) * (long)p.D + 2 * i;
        qd[0] = qr * c - qi * s;
        qd[1] = qr * s + qi * c;
        kd[0] = kr * c - ki * s;
        kd[1] = kr * s + ki * c;
        vd[0] = vr;
        vd[1] = vi;
    }
}

// One workgroup = (batch, head, 4 of the 8 query tiles of a 256-query group), one wave per 32-query tile.  The K / V tiles (32 slots
// x D) of the workgroup's key range are staged through double-buffered LDS by all 256 threads and shared by its four waves (one
// barrier per tile, the next tile's global loads in flight under the MFMAs of this one); a wave only computes on the key tiles its
// own causal / context range reaches.  Under a causal mask tile i of a group needs i + 1 key tiles, so the two workgroups of a
// group take the tiles {0, 2, 5, 7} and {1, 3, 4, 6}: 18 tile products each.  (Waves that are done idle at the barriers; the CU's
// other resident workgroup keeps the matrix pipe busy meanwhile.)  Round 2's form ran one single-wave workgroup per query tile:
// every tile of a head re-staged the head's keys and values on its own (366 MB fetched per launch against 131 MB of q + k + v + out,
// matrix pipe busy 0.24); eight waves on the eight tiles of a group (first version of this round) left each CU with ONE resident
// workgroup whose early tiles finish after a fraction of the loop: 120 -> 102 us per launch only.
constexpr int ATT_WAVES = 4, ATT_GROUP = 8;

// (cos, sin) of every (position, pair) of a whole-utterance pass, with the arithmetic of rope_split_kernel (modules/rope.py:37-62)
__global__ __launch_bounds__(256) void rope_table_kernel_f32(float* __restrict__ tab, int T, int D, float rope_coef, long pos0) {
    const int half = D / 2;
    const long total = (long)T * half;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int i = (int)(idx % half);
        const int t = (int)(idx / half);
        const float fr = expf((float)i * rope_coef);
        const float ang = fr * ((float)pos0 + (float)t);
        tab[(long)t * D + 2 * i] = cosf(ang);
        tab[(long)t * D + 2 * i + 1] = sinf(ang);
    }
}

// two interleaved pairs (x0, x1), (x2, x3) rotated by (c0, s0), (c1, s1): the expressions of rope_split_kernel
__device__ __forceinline__ f32x4 rope_piece(const f32x4 x, const f32x4 t) {
    f32x4 y;
    y[0] = x[0] * t[0] - x[1] * t[1];
    y[1] = x[0] * t[1] + x[1] * t[0];
    y[2] = x[2] * t[2] - x[3] * t[3];
    y[3] = x[2] * t[3] + x[3] * t[2];
    return y;
}

// QKV: the fused form (AttentionParams::row_stride / rope_tab) -- rows are `row_stride` floats apart and are rotated when they are loaded
template <int D, bool QKV = false>
__global__ __launch_bounds__(64 * ATT_WAVES) void attention_kernel(const AttentionParams p) {
    constexpr int DT = D / 32;
    constexpr int KS = D / 8;
    constexpr int KLD = D + 4;
    constexpr int NT = 64 * ATT_WAVES;
    constexpr int PIECES = 32 * D / 4;                   // 16-byte pieces of a K (or V) tile
    constexpr int LPT = (PIECES + NT - 1) / NT;          // per thread
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * D];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int q_first = (blockIdx.x >> 1) * (32 * ATT_GROUP);   // first query of the 8-tile group this workgroup shares with its twin
    const int local = (blockIdx.x & 1) == 0 ? (wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 5 : 7)
                                            : (wave == 0 ? 1 : wave == 1 ? 3 : wave == 2 ? 4 : 6);
    const int q0 = q_first + local * 32;                 // first query of this wave's tile
    const bool has_q = q0 < p.T;
    const int head = blockIdx.y;
    const long b = blockIdx.z;
    const long pos0 = p.pos_dev ? *p.pos_dev : p.pos0;
    const float scale = 1.0f / sqrtf((float)D);
    const float NEG_INF = -INFINITY;

    const long RS = QKV ? (long)p.row_stride : (long)D;
    const float* qb = QKV ? p.q + b * p.T * RS + (long)head * D : p.q + ((b * p.H + head) * p.T) * (long)D;
    const float* kb = QKV ? p.k + b * p.T * RS + (long)head * D : p.k + ((b * p.H + head) * p.cap) * (long)D;
    const float* vb = QKV ? p.v + b * p.T * RS + (long)head * D : p.v + ((b * p.H + head) * p.cap) * (long)D;
    const bool rot = QKV && p.rope_tab != nullptr;
    // (unconditional loads: without a table the same reads go to a valid address -- the first query row -- and are never used)
    const float* const tabp = rot ? p.rope_tab : qb;

    f32x4 qf[KS];
    {
        const int qrow = min(q0 + j, p.T - 1);
        f32x4 qt[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {   // rows past T: a clamped (valid) row, never stored -- unconditional loads, all in flight at once
            qf[s] = *reinterpret_cast<const f32x4*>(qb + (long)qrow * RS + 8 * s + 4 * h);
            if (QKV) qt[s] = *reinterpret_cast<const f32x4*>(tabp + (rot ? (long)qrow * D + 8 * s + 4 * h : 0));
        }
        if (QKV && rot) {
#pragma unroll
            for (int s = 0; s < KS; ++s) qf[s] = rope_piece(qf[s], qt[s]);
        }
    }

    // slots that can be visible: to this wave's tile [lo, hi], to any tile of the workgroup [lo_wg, hi_wg]
    // the last real tile of this workgroup's set ({0, 2, 5, 7} or {1, 3, 4, 6}): its end bounds the keys to stage
    const int t_last = (p.T - 1) / 32 - (blockIdx.x >> 1) * ATT_GROUP;                  // last real tile of the sequence, group-local
    int l_last = -1;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) {
        const int lw = (blockIdx.x & 1) == 0 ? (w == 0 ? 0 : w == 1 ? 2 : w == 2 ? 5 : 7) : (w == 0 ? 1 : w == 1 ? 3 : w == 2 ? 4 : 6);
        if (lw <= t_last) l_last = max(l_last, lw);
    }
    const int q_last = q_first + 32 * max(l_last, 0);
    int lo = 0, hi = p.cap - 1, lo_wg = 0, hi_wg = p.cap - 1;
    if (!p.ring) {
        hi = min(p.cap - 1, q0 + 31);
        hi_wg = min(p.cap - 1, q_last + 31);
        if (p.context > 0) { lo = max(0, q0 - p.context + 1); lo_wg = max(0, q_first + 32 * (int)(blockIdx.x & 1) - p.context + 1); }
        if (l_last < 0) hi_wg = -1;                      // (the twin of the last group may have no tile at all)
    }
    if (!has_q) hi = -1;                                 // a wave past the end of the sequence only helps staging
    const long end_offset = pos0 + p.T;                  // RingKVCache.end_offset after the append
    const int end_index = (int)(end_offset % p.cap);
    const long pq = pos0 + q0 + j;                       // this lane's query position

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;

    // K / V tile at slots s0 .. s0 + 31 -> registers.  Slots beyond cap read the last slot again (finite data): their scores are masked
    // to -inf below, so the duplicate takes weight 0 -- every load is unconditional (a load under a per-lane condition is waited for
    // inside its branch); threads beyond the tile (D = 32) re-read its last piece and do not store it.
    f32x4 kreg[LPT], vreg[LPT], treg[QKV ? LPT : 1];
    auto request = [&](int s0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = min(tid + NT * i, PIECES - 1);
            const int row = min(s0 + idx / (D / 4), p.cap - 1), c4 = (idx % (D / 4)) * 4;
            kreg[i] = *reinterpret_cast<const f32x4*>(kb + (long)row * RS + c4);
            vreg[i] = *reinterpret_cast<const f32x4*>(vb + (long)row * RS + c4);
            if (QKV) treg[i] = *reinterpret_cast<const f32x4*>(tabp + (rot ? (long)row * D + c4 : 0));      // (slot = position: pos0 == 0)
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = tid + NT * i;
            if (idx < PIECES) {
                const int row = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
                if (QKV && rot) kreg[i] = rope_piece(kreg[i], treg[i]);
                *reinterpret_cast<f32x4*>(Ks[buf] + row * KLD + c4) = kreg[i];
                *reinterpret_cast<f32x4*>(Vs[buf] + row * D + c4) = vreg[i];
            }
        }
    };

    const int s_begin = (lo_wg / 32) * 32;
    request(s_begin);
    int buf = 0;
    for (int s0 = s_begin; s0 <= hi_wg; s0 += 32) {
        // tile s0 -> LDS[buf] (the buffer last read two iterations ago: every wave has passed the barrier that followed), then the
        // next tile's loads go out and stay in flight under this tile's MFMAs
        stage(buf);
        __syncthreads();
        if (s0 + 32 <= hi_wg) request(s0 + 32);
        const float* Kt = Ks[buf];
        const float* Vt = Vs[buf];
        buf ^= 1;
        if (s0 + 31 < lo || s0 > hi) continue;           // wave-uniform: outside this tile's causal / context range

        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Kt + j * KLD + 8 * s + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[s][e], sacc, 0, 0, 0);
        }

        // mask + online softmax; lane owns query column j, rows (keys) rst_mfma32_row(r, lane).  Interior tiles -- every slot of the
        // tile visible to every query of the wave's tile: all but the diagonal (and a context edge) in the batch pass -- skip the
        // position arithmetic (64-bit compares per score: as many VALU cycles as the tile's MFMAs took); e^x runs on v_exp_f32
        // (exp2 of x * log2 e, ~1e-7 relative on weights in [0, 1]: the softmax costs a quarter of the libm form's instructions).
        const bool interior = !p.ring && s0 + 31 <= q0 && s0 + 31 < p.cap && (p.context <= 0 || q0 + 31 - s0 < p.context);
        float mt = NEG_INF;
        if (interior) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] *= scale;
                mt = fmaxf(mt, sacc[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int slot = s0 + rst_mfma32_row(r, lane);
                long pk;
                if (!p.ring) {
                    pk = pos0 + slot;
                } else {
                    // RingKVCache.complete (modules/transformer.py:254-278), including the `delta <= 0` quirk
                    const int delta = slot - end_index;
                    pk = delta <= 0 ? end_offset + delta : end_offset + delta - p.cap;
                    if (slot >= end_offset) pk = -1;
                }
                const long dlt = pq - pk;
                bool ok = slot < p.cap && pk >= 0 && dlt >= 0;
                if (p.context > 0) ok = ok && dlt < p.context;
                const float sv = ok ? sacc[r] * scale : NEG_INF;
                sacc[r] = sv;
                mt = fmaxf(mt, sv);
            }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        constexpr float LOG2E = 1.4426950408889634f;
        const float alpha = (m_run == NEG_INF) ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = (sacc[r] == NEG_INF) ? 0.0f : __builtin_amdgcn_exp2f((sacc[r] - m_new) * LOG2E);
            sacc[r] = pv;
            psum += pv;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = rst_mfma32_row(r, lane);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const float vf = Vt[key * D + d * 32 + j];
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sacc[r], oacc[d], 0, 0, 0);
            }
        }
    }

    // O^T[dim][query]: this lane holds, for its query j, dims d * 32 + 8 g + 4 h + {0..3} in accumulator elements 4 g .. 4 g + 3 ->
    // 16-byte stores straight from the registers (both lane halves of a query are adjacent: 32-byte runs per row)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (has_q && q0 + j < p.T) {
        float* orow = p.out + ((b * p.T + q0 + j) * p.H + head) * (long)D + 4 * h;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(orow + d * 32 + 8 * g) = v;
            }
    }
}

}  // namespace

int rst_launch_rope_split(const RopeSplitParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 0 && p.T >= 0 && p.H > 0 && p.D > 0 && p.D % 2 == 0 && p.cap > 0, "rope_split: bad sizes");
    if (p.B == 0 || p.T == 0) return RST_OK;
    RST_REQUIRE(p.qkv && p.q && p.k && p.v, "rope_split: null pointer");
    RST_REQUIRE(p.ring || p.cap >= p.T, "rope_split: cap (%d) < T (%d) without ring", p.cap, p.T);
    RST_REQUIRE(!p.ring || p.T <= p.cap, "rope_split: T (%d) > ring capacity (%d)", p.T, p.cap);
    const long total = (long)p.B * p.T * p.H * (p.D / 2);
    if (total == 0) return RST_OK;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rope_split_kernel, dim3((unsigned)g), dim3(256), 0, stream, p);
    return rst_check_launch("rope_split");
}

int rst_launch_rope_table(float* tab, int T, int D, float rope_coef, long pos0, hipStream_t stream) {
    RST_REQUIRE(tab && T > 0 && D > 0 && D % 2 == 0, "rope_table: bad arguments");
    const long total = (long)T * (D / 2);
    long g = (total + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(rope_table_kernel_f32, dim3((unsigned)g), dim3(256), 0, stream, tab, T, D, rope_coef, pos0);
    return rst_check_launch("rope_table");
}

int rst_launch_attention(const AttentionParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 0 && p.T >= 0 && p.H > 0 && p.cap > 0, "attention: bad sizes");
    RST_REQUIRE(p.B <= 65535 && p.H <= 65535, "attention: grid too large");
    if (p.B == 0 || p.T == 0) return RST_OK;
    RST_REQUIRE(p.q && p.k && p.v && p.out, "attention: null pointer");
    const dim3 grid(2 * ((p.T + 32 * ATT_GROUP - 1) / (32 * ATT_GROUP)), p.H, p.B);
    if (p.row_stride > 0) {
        RST_REQUIRE(!p.ring && p.pos0 == 0 && !p.pos_dev && p.cap == p.T && p.row_stride >= p.D && p.row_stride % 4 == 0,
                    "attention (fused qkv): whole-utterance passes only (ring = 0, pos0 = 0, cap = T)");
        switch (p.D) {
            case 32: hipLaunchKernelGGL((attention_kernel<32, true>), grid, dim3(64 * ATT_WAVES), 0, stream, p); break;
            case 64: hipLaunchKernelGGL((attention_kernel<64, true>), grid, dim3(64 * ATT_WAVES), 0, stream, p); break;
            case 128: hipLaunchKernelGGL((attention_kernel<128, true>), grid, dim3(64 * ATT_WAVES), 0, stream, p); break;
            default:
                rst_set_error("attention: head dim %d unsupported (32, 64, 128)", p.D);
                return RST_ERR_UNSUPPORTED;
        }
        return rst_check_launch("attention_qkv");
    }
    switch (p.D) {
        case 32: hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(64 * ATT_WAVES), 0, stream, p); break;
        case 64: hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(64 * ATT_WAVES), 0, stream, p); break;
        case 128: hipLaunchKernelGGL(attention_kernel<128>, grid, dim3(64 * ATT_WAVES), 0, stream, p); break;
        default:
            rst_set_error("attention: head dim %d unsupported (32, 64, 128)", p.D);
            return RST_ERR_UNSUPPORTED;
    }
    return rst_check_launch("attention");
}
