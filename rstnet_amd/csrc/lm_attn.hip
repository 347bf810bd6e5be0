// Ring-KV attention of the LM decode step: RoPE + append + single-query attention in one launch (long rings: slots split over
// workgroups with an in-launch combine; short rings: one wave per head), the multi-query form used by streaming codec steps
// and prompt prefill, and the T-step rope / append kernel.  All reproduce RingKVCache.complete's slot -> position map (Q1).
#include "lm_common.h"

namespace {

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
// KV16: the ring holds bf16 (the reference's cache precision: RingKVCache(..., dtype=torch.bfloat16), modules/transformer.py:228,
// loaders.py:144) -- keys / values are rounded to nearest-even when appended, the new step's own key / value take part in its
// attention at that precision too (the reference reads them back from the cache), and the long-context frame reads half the bytes.
__device__ __forceinline__ unsigned short df_bf16_rne(float f) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float df_bf16_round(float f) { return __uint_as_float((unsigned)df_bf16_rne(f) << 16); }

template <bool KV16> struct KvRow;
template <> struct KvRow<false> {       // fp32 ring: 16 floats of a row
    static __device__ __forceinline__ void load(const void* base, long elem, float (&o)[16]) {
        const float* q = static_cast<const float*>(base) + elem;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(q + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * i + e] = t[e];
        }
    }
    static __device__ __forceinline__ void store(void* base, long elem, const float (&v)[16]) {
        float* q = static_cast<float*>(base) + elem;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(q + 4 * i) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    }
    static __device__ __forceinline__ float round(float f) { return f; }
};
template <> struct KvRow<true> {        // bf16 ring: the same 16 elements in 32 bytes
    static __device__ __forceinline__ void load(const void* base, long elem, float (&o)[16]) {
        const unsigned short* q = static_cast<const unsigned short*>(base) + elem;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(q + 8 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[8 * i + 2 * e] = bf16_lo(t[e]); o[8 * i + 2 * e + 1] = bf16_hi(t[e]); }
        }
    }
    static __device__ __forceinline__ void store(void* base, long elem, const float (&v)[16]) {
        unsigned short* q = static_cast<unsigned short*>(base) + elem;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (unsigned)df_bf16_rne(v[8 * i + 2 * e]) | ((unsigned)df_bf16_rne(v[8 * i + 2 * e + 1]) << 16);
            *reinterpret_cast<u32x4*>(q + 8 * i) = t;
        }
    }
    static __device__ __forceinline__ float round(float f) { return df_bf16_round(f); }
};

template <int D, bool KV16, int UNR>
__device__ __forceinline__ void attn_decode_body(const LmAttnParams& p) {
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
    // (with a table of the step's (cos, sin) pairs -- computed once per frame by rope_table_kernel with this very arithmetic -- the
    // launch skips 24 libm calls per lane: at short context they were most of its duration)
    float rc[8], rs[8];
    if (p.rope_cs) {
        const f32x4* t4 = reinterpret_cast<const f32x4*>(p.rope_cs + sub * 16);
        const f32x4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
        rc[0] = t0[0]; rs[0] = t0[1]; rc[1] = t0[2]; rs[1] = t0[3]; rc[2] = t1[0]; rs[2] = t1[1]; rc[3] = t1[2]; rs[3] = t1[3];
        rc[4] = t2[0]; rs[4] = t2[1]; rc[5] = t2[2]; rs[5] = t2[3]; rc[6] = t3[0]; rs[6] = t3[1]; rc[7] = t3[2]; rs[7] = t3[3];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            rc[i] = 1.f; rs[i] = 0.f;
            if (p.rope && 2 * (sub * 8 + i) < p.rope_dims) {
                const float ang = expf((float)(sub * 8 + i) * p.rope_coef) * (float)pos;
                rc[i] = cosf(ang); rs[i] = sinf(ang);
            }
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
    const long kv_row0 = ((b * p.G + g) * p.cap) * (long)D + sub * 16;      // element offset of this lane's 16 dims in slot 0

    // UNR slots per lane group and iteration: the K / V rows of all of them are requested before the first score is formed (a
    // wave that waits out one memory latency per slot streams the ring at a fraction of the bandwidth).  The scores are folded
    // into the running softmax one slot at a time, in slot order, as before.
    for (int s0 = s_lo + wave * SPW; s0 < s_hi; s0 += 4 * SPW * UNR) {
        float kv[UNR][16], vv[UNR][16];
        bool ok[UNR];
        // every K / V row of the iteration is requested unconditionally (slot clamped into the ring) before any is looked at; rows
        // that do not take part are cleared afterwards through a mask the compiler cannot fold back into a condition on the load
        // (a load under a per-lane condition is waited for inside its branch: one exposed round trip per slot instead of per iteration)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int slot = min(s0 + u * 4 * SPW + grp, p.cap - 1);
            KvRow<KV16>::load(p.k, kv_row0 + (long)slot * D, kv[u]);
            KvRow<KV16>::load(p.v, kv_row0 + (long)slot * D, vv[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int slot = s0 + u * 4 * SPW + grp;
            const bool cur = slot < s_hi && slot == slot_cur;
            ok[u] = slot < s_hi && ring_visible(slot, pos_q, p.cap, p.context, end_offset);
            unsigned msk = (ok[u] && !cur) ? 0xffffffffu : 0u;
            asm volatile("" : "+v"(msk));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                kv[u][i] = __uint_as_float(__float_as_uint(kv[u][i]) & msk);
                vv[u][i] = __uint_as_float(__float_as_uint(vv[u][i]) & msk);
            }
            if (cur) {       // the new step: from qkv (at the ring's precision), and appended to the ring
#pragma unroll
                for (int i = 0; i < 16; ++i) { kv[u][i] = KvRow<KV16>::round(kcur[i]); vv[u][i] = KvRow<KV16>::round(vn[i]); }
                if (appender) {
                    KvRow<KV16>::store(p.k, kv_row0 + (long)slot * D, kv[u]);
                    KvRow<KV16>::store(p.v, kv_row0 + (long)slot * D, vv[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) d = fmaf(kv[u][i], q[i], d);
#pragma unroll
            for (int off = LPS / 2; off > 0; off >>= 1) d += __shfl_xor(d, off);   // all lanes take part (ok is per group)
            const float sc = ok[u] ? d * scale : -INFINITY;
            const float m_new = fmaxf(m_run, sc);
            if (m_new != -INFINITY) {
                const float alpha = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
                const float pw = sc == -INFINITY ? 0.f : expf(sc - m_new);
                l_run = l_run * alpha + pw;
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = o[i] * alpha + pw * vv[u][i];
                m_run = m_new;
            }
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
            const float r = L > 0.f ? O / L : 0.f;
            if (p.out_packed) sm_o[0][tid] = r;     // each thread reads / writes only its own column of sm_o[0]
            else p.out[((b * T + tq) * p.H + h) * (long)D + tid] = r;
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
    // packed output (the operand layout of the skinny GEMM that consumes the attention result): 8 consecutive dims per thread
    auto emit_packed = [&]() {
        __syncthreads();
        if (tid < D / 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = sm_o[0][tid * 8 + j];
            store_packed8(p.out_packed, p.out_plane, (int)b, h * D + tid * 8, p.H * D, v);
        }
    };
    if (active == 1) {
        if (p.out_packed) emit_packed();
        return;
    }
    // In-launch reduction of the splits: every storing wave drains its write-through stores, ONE lane bumps the arrival
    // counter; the LAST arriver of (b, h) reads the partials with agent-scope (L1-bypassing) loads, combines, and re-arms the
    // counter for the next launch.  No dispatch-order / placement assumption.
    __shared__ int sm_last;
    // publish form: relaxed agent-scope stores are `sc1` write-through stores (they leave the XCD's L2), every storing wave drains
    // them with the asm wait below (inline asm: the compiler cannot drop it), ONE lane then bumps the counter; the reader uses
    // relaxed agent-scope (`sc1`, L1-bypassing) loads -- the "sc1 payload -> asm vmcnt(0) -> sc1 flag / sc1 loads" form that
    // MI355X_MICROARCH.md lists as valid without a release / acquire fence pair (its cost rows: handoff-flag, splitk-seam).
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
        // eight splits at a time: their (max, sum, out[tid]) triples are all in flight before the first is used
        float M = -INFINITY;
        for (int s2 = 0; s2 < active; s2 += 8) {
            float t[8];
            rst_load_partials<8>(w0, D + 2, s2, active, t);
#pragma unroll
            for (int u = 0; u < 8; ++u) M = fmaxf(M, t[u]);      // entries past `active` repeat the last one: harmless for a max
        }
        float L = 0.f, O = 0.f;
        for (int s2 = 0; s2 < active; s2 += 8) {
            float ms[8], ls[8], os[8];
            rst_load_partials<8>(w0, D + 2, s2, active, ms);
            rst_load_partials<8>(w0 + 1, D + 2, s2, active, ls);
            rst_load_partials<8>(w0 + 2 + tid, D + 2, s2, active, os);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (s2 + u < active) {
                    const float fw = ms[u] == -INFINITY ? 0.f : expf(ms[u] - M);
                    L = fmaf(ls[u], fw, L);
                    O = fmaf(os[u], fw, O);
                }
            }
        }
        const float r = L > 0.f ? O / L : 0.f;
        if (p.out_packed) sm_o[0][tid] = r;
        else p.out[((b * T + tq) * p.H + h) * (long)D + tid] = r;
    }
    if (p.out_packed) emit_packed();
}

template <int D, bool KV16 = false>
__global__ __launch_bounds__(256) void attn_decode_kernel(const LmAttnParams p) {
    attn_decode_body<D, KV16, 2>(p);
}
// The same for launches of more workgroups than three per CU hold (batch x heads x splits > 3 x CUs): one slot per lane group
// in flight instead of two keeps the kernel under 128 VGPRs, i.e. FOUR workgroups per CU -- at 32 streams x 32 heads the 1024
// workgroups ran as two rounds (768 + 256) of a launch that is one dependent chain (18 us; `profiles/r04_kernel_resources.txt`:
// 161 VGPRs, occupancy 3).
template <int D, bool KV16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_decode_dense_kernel(const LmAttnParams p) {
    attn_decode_body<D, KV16, 1>(p);
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
    float* kc = static_cast<float*>(p.k) + ((b * p.G + kvh) * cap) * (long)D + c * DPL;
    float* vc = static_cast<float*>(p.v) + ((b * p.G + kvh) * cap) * (long)D + c * DPL;
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
        if constexpr (DPL % 8 == 0) {
            if (p.out_packed) {
#pragma unroll
                for (int i = 0; i < DPL; i += 8) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = o[i + j];
                    store_packed8(p.out_packed, p.out_plane, (int)b, h * D + c * DPL + i, p.H * D, v);
                }
                return;
            }
        }
        float* out = p.out + (b * p.H + h) * (long)D + c * DPL;
#pragma unroll
        for (int i = 0; i < DPL; i += 4) *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(o + i);
    }
}

}  // namespace

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

namespace {
__global__ __launch_bounds__(128) void rope_table_kernel(const long* pos_dev, float* out, int D, int rope_dims, float rope_coef) {
    const int i = threadIdx.x;
    if (i >= D / 2) return;
    float c = 1.f, s = 0.f;
    if (2 * i < rope_dims) {
        const float ang = expf((float)i * rope_coef) * (float)*pos_dev;
        c = cosf(ang); s = sinf(ang);
    }
    out[2 * i] = c;
    out[2 * i + 1] = s;
}
}  // namespace

int rst_launch_lm_rope_table(const long* pos_dev, float* out, int D, int rope_dims, float rope_coef, hipStream_t stream) {
    RST_REQUIRE(pos_dev && out && D >= 2 && D <= 256 && D % 2 == 0 && rope_dims >= 0 && rope_dims <= D, "lm_rope_table: bad arguments (D=%d rope_dims=%d)", D, rope_dims);
    hipLaunchKernelGGL(rope_table_kernel, dim3(1), dim3(128), 0, stream, pos_dev, out, D, rope_dims, rope_coef);
    return rst_check_launch("lm_rope_table");
}

int rst_launch_lm_attn(const LmAttnParams& p, hipStream_t stream) {
    RST_REQUIRE((p.qkv || p.q_pre) && p.k && p.v && (p.out || p.out_packed) && p.pos_dev && p.B >= 1 && p.H > 0 && p.cap > 0 && p.splits >= 1,
                "lm_attn: bad arguments");
    RST_REQUIRE(!p.out_packed || (!p.q_pre && (p.D == 64 || p.D == 128)), "lm_attn: packed output needs the single-step form and head dim 64 / 128");
    const int T = p.q_pre ? p.T : 1;
    RST_REQUIRE(T >= 1 && (long)p.B * T <= 65535 && p.H <= 65535 && p.D % 2 == 0, "lm_attn: bad sizes");
    RST_REQUIRE(p.G >= 1 && p.H % p.G == 0 && p.rope_dims >= 0 && p.rope_dims <= p.D && p.rope_dims % 2 == 0,
                "lm_attn: bad kv head count %d for %d heads or rope_dims %d", p.G, p.H, p.rope_dims);
    RST_REQUIRE(!p.kv_bf16 || (!p.q_pre && !(p.cap <= 64 && p.splits == 1)), "lm_attn: bf16 rings are served by the long-ring single-step form only");
    RST_REQUIRE(!p.rope_cs || (uintptr_t)p.rope_cs % 16 == 0, "lm_attn: the rope table must be 16-byte aligned");
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
    const bool dense = (long)p.splits * p.H * p.B * T > 3L * rst_cu_count();
    if (dense && (p.D == 64 || p.D == 128)) {
        if (p.kv_bf16) {
            if (p.D == 64) hipLaunchKernelGGL((attn_decode_dense_kernel<64, true>), grid, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((attn_decode_dense_kernel<128, true>), grid, dim3(256), 0, stream, p);
            return rst_check_launch("lm_attn_kv16");
        }
        if (p.D == 64) hipLaunchKernelGGL(attn_decode_dense_kernel<64>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(attn_decode_dense_kernel<128>, grid, dim3(256), 0, stream, p);
        return rst_check_launch("lm_attn");
    }
    if (p.kv_bf16) {
        switch (p.D) {
            case 64: hipLaunchKernelGGL((attn_decode_kernel<64, true>), grid, dim3(256), 0, stream, p); break;
            case 128: hipLaunchKernelGGL((attn_decode_kernel<128, true>), grid, dim3(256), 0, stream, p); break;
            default:
                rst_set_error("lm_attn: head dim %d unsupported for long rings (64, 128)", p.D);
                return RST_ERR_UNSUPPORTED;
        }
        return rst_check_launch("lm_attn_kv16");
    }
    switch (p.D) {
        case 64: hipLaunchKernelGGL(attn_decode_kernel<64>, grid, dim3(256), 0, stream, p); break;
        case 128: hipLaunchKernelGGL(attn_decode_kernel<128>, grid, dim3(256), 0, stream, p); break;
        default:
            rst_set_error("lm_attn: head dim %d unsupported for long rings (64, 128)", p.D);
            return RST_ERR_UNSUPPORTED;
    }
    return rst_check_launch("lm_attn");
}
