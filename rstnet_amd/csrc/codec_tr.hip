// One streaming step of a Mimi transformer (encoder_transformer / decoder_transformer: 8 layers of LayerNorm -> fused-QKV attention
// with interleaved RoPE and a ring KV cache -> LayerScale residual -> LayerNorm -> GELU FFN -> LayerScale residual, fp32;
// modules/transformer.py:376-423,434-592,595-690) for the few rows of one 80 ms frame (R = streams x positions <= 4) as ONE
// persistent launch.  As separate launches a layer is 6 dependent kernels of 1-4 MB each (LN + in-proj GEMV | rope / split |
// ring attention | out-proj GEMV | LN + linear1 GEMV | linear2 GEMV), ~5-7 us apiece whatever their size; here the 5 op
// boundaries of a layer are in-launch hand-offs (persist.h) and the weights of the next op are requested before each wait.
//
// Work split: G workgroups x 4 waves, weight row r of an op belongs to wave (r mod 4G); every workgroup keeps the residual stream
// x [R][E] in LDS and recomputes the LayerNorms itself.  Attention of (stream b, head h) is owned by workgroup b * H + h: it
// gathers the head's q / k / v of the new steps, rotates q and k (modules/rope.py:37-62, position = *pos_dev + t), appends k / v to
// the ring in HBM (slot (pos + t) % cap: the ring persists across frames; only this workgroup index ever touches the head's
// ring inside a launch) and runs the T queries against the ring with the slot -> position map and mask of
// RingKVCache.complete (transformer.py:254-278,404-414, incl. the `delta <= 0` slot), the four waves sharing each query's slots.
#include "persist.h"

namespace {

// ---- rows of an fp32 weight matrix: RU rows x CU chunks of 512 k (8 floats = two 16-byte loads per lane and chunk)
template <int RU, int CU> struct CtPre { f32x4 wv[RU][CU][2]; };

template <int RU, int CU>
__device__ __forceinline__ void ct_rows_load(CtPre<RU, CU>& pre, const float* w, int N, int K, int r0, int kb, int W, int lane) {
#pragma unroll
    for (int j = 0; j < RU; ++j)
#pragma unroll
        for (int c = 0; c < CU; ++c) {
            const int r = r0 + j * W, kk = kb + c * 512 + lane * 8;
            pre.wv[j][c][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            pre.wv[j][c][1] = pre.wv[j][c][0];
            if (r < N && kk < K) {
                const float* q = w + (long)r * K + kk;
                pre.wv[j][c][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
                pre.wv[j][c][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q + 4));
            }
        }
}

template <int RU, int CU>
__device__ __forceinline__ void ct_rows_issue(CtPre<RU, CU>& pre, const float* w, int N, int K, int gw, int W, int lane) {
    ct_rows_load<RU, CU>(pre, w, N, K, gw, 0, W, lane);
}

// y[r][0..R) for rows r = gw, gw + W, ...; the first block (rows gw + j W, k < CU * 512) was requested by ct_rows_issue
template <int R, int RU, int CU, typename Epi>
__device__ __forceinline__ void ct_rows(CtPre<RU, CU>& pre, const float* w, int N, int K, const float* xs, int gw, int W, int lane, Epi epi) {
    for (int r0 = gw; r0 < N; r0 += RU * W) {
        float acc[RU][R];
#pragma unroll
        for (int j = 0; j < RU; ++j)
#pragma unroll
            for (int b = 0; b < R; ++b) acc[j][b] = 0.f;
        for (int kb = 0; kb < K; kb += CU * 512) {
            if (r0 != gw || kb != 0) ct_rows_load<RU, CU>(pre, w, N, K, r0, kb, W, lane);
#pragma unroll
            for (int c = 0; c < CU; ++c) {
                const int kk = kb + c * 512 + lane * 8;
                if (kk < K) {
#pragma unroll
                    for (int b = 0; b < R; ++b) {
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + b * K + kk);
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + b * K + kk + 4);
#pragma unroll
                        for (int j = 0; j < RU; ++j) {
                            const f32x4 a0 = pre.wv[j][c][0], a1 = pre.wv[j][c][1];
                            float a = acc[j][b];
                            a = fmaf(a0[0], x0[0], a); a = fmaf(a0[1], x0[1], a); a = fmaf(a0[2], x0[2], a); a = fmaf(a0[3], x0[3], a);
                            a = fmaf(a1[0], x1[0], a); a = fmaf(a1[1], x1[1], a); a = fmaf(a1[2], x1[2], a); a = fmaf(a1[3], x1[3], a);
                            acc[j][b] = a;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            const int r = r0 + j * W;
            float s[R];
#pragma unroll
            for (int b = 0; b < R; ++b) s[b] = wave_sum(acc[j][b]);
            if (lane == 0 && r < N) epi(r, s);
        }
    }
}

// nn.LayerNorm(E, eps) with affine gamma / beta over every row of x [R][E] -> xs (two-pass statistics, biased variance: the
// arithmetic of gemv_kernel's prologue 3)
template <int R>
__device__ __forceinline__ void ct_layernorm(const float* x, const float* gamma, const float* beta, float eps, int E, float* xs, DfShared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = 0; b < R; ++b) {
        float s = 0.f;
        for (int i = tid; i < E; i += DF_THREADS) s += x[b * E + i];
        s = wave_sum(s);
        __syncthreads();
        if (lane == 0) sh.red[wave] = s;
        __syncthreads();
        const float mean = (sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3]) / (float)E;
        float v = 0.f;
        for (int i = tid; i < E; i += DF_THREADS) { const float d = x[b * E + i] - mean; v = fmaf(d, d, v); }
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) sh.red[wave] = v;
        __syncthreads();
        const float var = sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3];
        const float rstd = 1.0f / sqrtf(var / (float)E + eps);
        for (int i = tid; i < E; i += DF_THREADS) xs[b * E + i] = (x[b * E + i] - mean) * rstd * gamma[i] + beta[i];
    }
    __syncthreads();
}

// SOLO: the one-workgroup repair launch enqueued behind the persistent one (persist.h): a no-op unless a hand-off of that launch timed
// out, else the whole step recomputed by this workgroup alone (every (stream, head) pair in turn; the ring slots of the new steps are
// rewritten with the right values before anything reads them -- reads of a new step's own slots come from LDS either way).
template <int R, bool SOLO>
__global__ __launch_bounds__(DF_THREADS) void codec_tr_kernel(const CodecTrParams p) {
    if (SOLO && __hip_atomic_load(p.status, DF_RLX) == 0u) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    DfShared& sh = *reinterpret_cast<DfShared*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x, G = gridDim.x;
    const int gw = wg * DF_WAVES + wave, W = G * DF_WAVES;
    const int E = p.E, F = p.F, D = p.D, H = p.H, T = p.T, cap = p.cap;
    const int XW = E > F ? E : F;
    // LDS carve (floats): header | xs [R][max(E, F)] | xres [R][E] | qh [T][3][D] | att_part [4][D + 2]
    float* xs = lds + DF_HDR_FLOATS;
    float* xres = xs + R * XW;
    float* qh = xres + R * E;
    float* att_part = qh + T * 3 * D;             // [DF_WAVES][D + 2] partial (max, sum, out) of a query per wave
    u64* gX = p.gran + (SOLO ? (long)R * (5L * E + F) : 0L);
    u64* gQKV = gX + (long)R * E;
    u64* gATT = gQKV + (long)R * 3 * E;
    u64* gH = gATT + (long)R * E;
    if (tid == 0) sh.dead = 0;
    CtPre<2, 1> pq;         // in-projection rows of this wave
    CtPre<1, 1> po;         // out-projection
    CtPre<2, 1> p1;         // linear1
    CtPre<1, 4> p2;         // linear2
    ct_rows_issue<2, 1>(pq, p.in_proj[0], 3 * E, E, gw, W, lane);
    for (int i = tid; i < R * E; i += DF_THREADS) xres[i] = p.x[i];
    const long pos = *p.pos_dev;
    __syncthreads();
    unsigned eX = 0, eQKV = 0, eATT = 0, eH = 0;

    for (int l = 0; l < p.L; ++l) {
        // ---- in-projection of the normed rows
        ct_layernorm<R>(xres, p.n1g[l], p.n1b[l], p.eps, E, xs, sh);
        ++eQKV;
        ct_rows<R, 2, 1>(pq, p.in_proj[l], 3 * E, E, xs, gw, W, lane, [&](int r, float (&s)[R]) {
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gQKV + (long)b * 3 * E + r, eQKV, s[b]);
        });
        ct_rows_issue<1, 1>(po, p.out_proj[l], E, E, gw, W, lane);
        // ---- attention of (stream, head) = workgroup index
        ++eATT;
        for (int bh = wg; bh < p.B * H; bh += G) {
            const int b = bh / H, h = bh - b * H;
            // q / k / v of the head for the T new steps of stream b: item i = (t, part, d)
            df_gather<2>(gQKV, T * 3 * D, eQKV, qh, [&](int i) { const int t = i / (3 * D), j = i - t * 3 * D, part = j / D;
                                                              return ((long)(b * T + t) * 3 + part) * E + h * D + (j - part * D); }, sh, p.status, 2u);
            // interleaved RoPE on q and k at position pos + t (modules/rope.py:37-62), then the ring append
            for (int i = tid; i < T * D; i += DF_THREADS) {      // item = (t, which in {q, k}, pair)
                const int t = i / D, j = i - t * D, which = j / (D / 2), pr = j - which * (D / 2);
                if (p.rope) {
                    const float ang = expf((float)pr * p.rope_coef) * ((float)pos + (float)t);
                    const float c = cosf(ang), sn = sinf(ang);
                    float* v2 = qh + (t * 3 + which) * D + 2 * pr;
                    const float re = v2[0], im = v2[1];
                    v2[0] = re * c - im * sn;
                    v2[1] = re * sn + im * c;
                }
            }
            __syncthreads();
            float* kring = p.kc[l] + ((long)(b * H + h) * cap) * D;
            float* vring = p.vc[l] + ((long)(b * H + h) * cap) * D;
            for (int i = tid; i < T * D; i += DF_THREADS) {
                const int t = i / D, d = i - t * D;
                const int slot = (int)((pos + t) % cap);
                kring[(long)slot * D + d] = qh[(t * 3 + 1) * D + d];
                vring[(long)slot * D + d] = qh[(t * 3 + 2) * D + d];
            }
            // The T queries one after the other, each by ALL four waves: lane group `grp` (LPS = D / 16 lanes, 16 dims each) of wave w owns
            // slot (pass * SPW + grp) of the passes w, w + 4, ..., and the K / V rows of UB passes are requested before the first is
            // used.  (Round 2 gave each query ONE wave that walked the ring pass by pass: with the 250-slot ring of a session past
            // 10 s that is 16 dependent memory round trips per layer -- ~24 us of a ~35 us layer.)  The four partial (max, sum, out)
            // triples meet in LDS and are combined in wave order.
            {
                const int LPS = D >> 4, SPW = 64 / LPS;
                const int sub = lane % LPS, grp = lane / LPS;
                const long end_offset = pos + T;
                const int n_used = (int)min((long)cap, end_offset);
                const int npass = (n_used + SPW - 1) / SPW;
                const float scale = 1.0f / sqrtf((float)D);
                constexpr int UB = 4;
                for (int t = 0; t < T; ++t) {
                    const long pos_q = pos + t;
                    float q[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) q[i] = qh[(t * 3) * D + sub * 16 + i];
                    float m_run = -INFINITY, l_run = 0.f, o[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[i] = 0.f;
                    for (int p0 = wave; p0 < npass; p0 += DF_WAVES * UB) {
                        f32x4 kq[UB][4], vq[UB][4];
#pragma unroll
                        for (int u = 0; u < UB; ++u) {        // unconditional loads from a clamped slot; masked below
                            const int slot = min((p0 + u * DF_WAVES) * SPW + grp, cap - 1);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                kq[u][i] = *reinterpret_cast<const f32x4*>(kring + (long)slot * D + sub * 16 + 4 * i);
                                vq[u][i] = *reinterpret_cast<const f32x4*>(vring + (long)slot * D + sub * 16 + 4 * i);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int pass = p0 + u * DF_WAVES;
                            const int slot = pass * SPW + grp;
                            const bool ok = pass < npass && slot < n_used && ring_visible(slot, pos_q, cap, p.context, end_offset);
                            // a slot written by this very step comes from LDS (the stores above may not have landed for these loads)
                            int tn = -1;
                            for (int t2 = 0; t2 < T; ++t2) tn = (int)((pos + t2) % cap) == slot ? t2 : tn;
                            float kv[16], vv[16];
                            // values of a slot that is masked / not yet written are CLEARED, not multiplied by a zero weight: a ring the
                            // caller did not zero may hold Inf / NaN there (0 * NaN would poison the row; the reference masks such slots).
                            // The mask goes through an opaque register so that the loads above stay unconditional.
                            int vmask = ok ? -1 : 0;
                            asm volatile("" : "+v"(vmask));
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                kv[i] = tn >= 0 ? qh[(tn * 3 + 1) * D + sub * 16 + i] : kq[u][i >> 2][i & 3];
                                const float vraw = tn >= 0 ? qh[(tn * 3 + 2) * D + sub * 16 + i] : vq[u][i >> 2][i & 3];
                                vv[i] = __int_as_float(__float_as_int(vraw) & vmask);
                            }
                            float d = 0.f;
#pragma unroll
                            for (int i = 0; i < 16; ++i) d = fmaf(kv[i], q[i], d);
                            for (int off = LPS / 2; off > 0; off >>= 1) d += __shfl_xor(d, off);
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
                    }
                    // merge the lane groups of the wave (same `sub`)
                    float m_w = m_run;
                    for (int off = LPS; off < 64; off <<= 1) m_w = fmaxf(m_w, __shfl_xor(m_w, off));
                    const float f = (m_run == -INFINITY) ? 0.f : expf(m_run - m_w);
                    float l_w = l_run * f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[i] *= f;
                    for (int off = LPS; off < 64; off <<= 1) {
                        l_w += __shfl_xor(l_w, off);
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[i] += __shfl_xor(o[i], off);
                    }
                    // the waves' partials -> LDS, combined in wave order by D threads
                    float* part = att_part + wave * (D + 2);
                    if (grp == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) part[2 + sub * 16 + i] = o[i];
                        if (sub == 0) { part[0] = m_w; part[1] = l_w; }
                    }
                    __syncthreads();
                    if (tid < D) {
                        float M = -INFINITY;
#pragma unroll
                        for (int w = 0; w < DF_WAVES; ++w) M = fmaxf(M, att_part[w * (D + 2)]);
                        float Lq = 0.f, Oq = 0.f;
#pragma unroll
                        for (int w = 0; w < DF_WAVES; ++w) {
                            const float mw = att_part[w * (D + 2)];
                            const float fw = mw == -INFINITY ? 0.f : expf(mw - M);
                            Lq = fmaf(att_part[w * (D + 2) + 1], fw, Lq);
                            Oq = fmaf(att_part[w * (D + 2) + 2 + tid], fw, Oq);
                        }
                        df_publish(gATT + (long)(b * T + t) * E + h * D + tid, eATT, Lq > 0.f ? Oq / Lq : 0.f);
                    }
                    __syncthreads();          // att_part is rewritten by the next query
                }
            }
        }
        // ---- out-projection, LayerScale, residual
        df_gather<4>(gATT, R * E, eATT, xs, [](int i) { return i; }, sh, p.status, 4u);
        ++eX;
        ct_rows<R, 1, 1>(po, p.out_proj[l], E, E, xs, gw, W, lane, [&](int r, float (&s)[R]) {
            const float sc1 = p.ls1[l] ? p.ls1[l][r] : 1.0f;
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gX + (long)b * E + r, eX, xres[b * E + r] + sc1 * s[b]);
        });
        ct_rows_issue<2, 1>(p1, p.lin1[l], F, E, gw, W, lane);
        df_gather<4>(gX, R * E, eX, xres, [](int i) { return i; }, sh, p.status, 8u);
        // ---- FFN: x + scale2 * W2 gelu(W1 LN(x))
        ct_layernorm<R>(xres, p.n2g[l], p.n2b[l], p.eps, E, xs, sh);
        ++eH;
        ct_rows<R, 2, 1>(p1, p.lin1[l], F, E, xs, gw, W, lane, [&](int r, float (&s)[R]) {
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gH + (long)b * F + r, eH, rst_gelu(s[b]));
        });
        ct_rows_issue<1, 4>(p2, p.lin2[l], E, F, gw, W, lane);
        df_gather<8>(gH, R * F, eH, xs, [](int i) { return i; }, sh, p.status, 16u);
        ++eX;
        ct_rows<R, 1, 4>(p2, p.lin2[l], E, F, xs, gw, W, lane, [&](int r, float (&s)[R]) {
            const float sc2 = p.ls2[l] ? p.ls2[l][r] : 1.0f;
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gX + (long)b * E + r, eX, xres[b * E + r] + sc2 * s[b]);
        });
        if (l + 1 < p.L) ct_rows_issue<2, 1>(pq, p.in_proj[l + 1], 3 * E, E, gw, W, lane);
        df_gather<4>(gX, R * E, eX, xres, [](int i) { return i; }, sh, p.status, 32u);
    }
    if (wg == 0)
        for (int i = tid; i < R * E; i += DF_THREADS) p.y[i] = xres[i];
    if (SOLO) df_solo_done(p.status);
}

int ct_cu_count() {
    return rst_cu_count();       // per device (rst_common.h)
}

}  // namespace

long rst_codec_tr_workspace_granules(int R, int E, int F) { return (long)R * (5L * E + F); }

// Workgroups of the persistent launch for a shape, 0 if it is not served: every workgroup must own a row of the in-projection
// (3E rows) and of linear1 (F rows) -- the all-to-all ops between two writes of a hand-off buffer (persist.h) -- every (stream,
// head) pair needs a workgroup and one workgroup of that LDS footprint must fit a CU.
int rst_codec_tr_grid(int B, int T, int E, int H, int F, int L, int cap) {
    const int R = B * T;
    if (!(B >= 1 && T >= 1 && T <= DF_WAVES && R <= 4 && E > 0 && E % 8 == 0 && F > 0 && F % 8 == 0 && H > 0 && E % H == 0 && L >= 1 &&
          L <= RST_CTR_MAX_L && cap >= T))
        return 0;
    const int D = E / H;
    if (!(D >= 16 && D % 16 == 0 && D <= 256 && (64 % (D / 16)) == 0)) return 0;
    const int XW = E > F ? E : F;
    const size_t lds = ((size_t)DF_HDR_FLOATS + (size_t)R * XW + (size_t)R * E + (size_t)T * 3 * D + (size_t)DF_WAVES * (D + 2)) * sizeof(float);
    if (lds > 150 * 1024) return 0;
    const int G = df_grid_for_rows(ct_cu_count(), min(3 * E, F));
    if (B * H > G) return 0;
    static signed char fits_dev[RST_MAX_DEVICES];        // per device: 0 = not asked yet, 1 = fits, -1 = does not
    signed char& fits = fits_dev[rst_current_device()];
    if (fits == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(codec_tr_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        int nb = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, codec_tr_kernel<4, false>, DF_THREADS, 150 * 1024);
        (void)hipGetLastError();
        fits = (e == hipSuccess && nb >= 1) ? 1 : -1;
    }
    return fits > 0 ? G : 0;
}

int rst_launch_codec_tr(const CodecTrParams& p, hipStream_t stream) {
    const int R = p.B * p.T;
    const int G = rst_codec_tr_grid(p.B, p.T, p.E, p.H, p.F, p.L, p.cap);
    RST_REQUIRE(G > 0 && p.H * p.D == p.E, "codec_tr: unsupported shape (B=%d T=%d E=%d F=%d H=%d D=%d L=%d cap=%d) or no resident grid for it", p.B, p.T,
                p.E, p.F, p.H, p.D, p.L, p.cap);
    RST_REQUIRE(p.x && p.y && p.pos_dev && p.gran && p.status, "codec_tr: null buffers");
    for (int l = 0; l < p.L; ++l)
        RST_REQUIRE(p.in_proj[l] && p.out_proj[l] && p.lin1[l] && p.lin2[l] && p.n1g[l] && p.n1b[l] && p.n2g[l] && p.n2b[l] && p.kc[l] && p.vc[l],
                    "codec_tr: layer %d pointers", l);
    const int XW = p.E > p.F ? p.E : p.F;
    const size_t lds = ((size_t)DF_HDR_FLOATS + (size_t)R * XW + (size_t)R * p.E + (size_t)p.T * 3 * p.D + (size_t)DF_WAVES * (p.D + 2)) * sizeof(float);
    // both granule sets (persistent launch | repair launch) start at zero in every call
    if (hipMemsetAsync(p.gran, 0, (size_t)rst_codec_tr_workspace_granules(R, p.E, p.F) * 16, stream) != hipSuccess) {
        rst_set_error("codec_tr: workspace memset failed");
        return RST_ERR_LAUNCH;
    }
    auto go = [&](auto kern, int grid) {
        static RstOncePerDevice attr_once;       // one flag per kernel instance
        if (attr_once.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(DF_THREADS), lds, stream, p);
    };
    // the persistent launch, then its one-workgroup repair launch (a no-op unless a hand-off timed out: persist.h)
    switch (R) {
        case 1: go(codec_tr_kernel<1, false>, G); go(codec_tr_kernel<1, true>, 1); break;
        case 2: go(codec_tr_kernel<2, false>, G); go(codec_tr_kernel<2, true>, 1); break;
        case 3: go(codec_tr_kernel<3, false>, G); go(codec_tr_kernel<3, true>, 1); break;
        default: go(codec_tr_kernel<4, false>, G); go(codec_tr_kernel<4, true>, 1); break;
    }
    return rst_check_launch("codec_tr");
}
