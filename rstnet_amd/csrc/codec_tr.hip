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

// tools build only: wall-clock stamps (100 MHz) of workgroup 0 and of the last workgroup at the op boundaries of every layer
// (tools/probes/codec_tr_phases.py prints them)
#ifdef RST_ABLATION
#define CT_STAMPS_PER_LAYER 20
__device__ unsigned long long ct_stamps[2][CT_STAMPS_PER_LAYER * RST_CTR_MAX_L + 2];
// (kept in LDS while the launch runs: a global store per stamp would put its round trip into the next barrier)
#define CT_STAMP(i) do { if (!SOLO && tid == 0) ct_lds_stamps[(i)] = wall_clock64(); } while (0)
#else
#define CT_STAMP(i) do {} while (0)
#endif

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
            for (int b = 0; b < R; ++b) s[b] = wave_sum_fast(acc[j][b]);
            if (lane == 0 && r < N) epi(r, s);
        }
    }
}

// nn.LayerNorm(E, eps) with affine gamma / beta over every row of x [R][E] -> xs (two-pass statistics, biased variance: the
// arithmetic of gemv_kernel's prologue 3).  One WAVE per row (R <= 4 = the waves of a workgroup): no barrier inside a row, and
// gamma / beta of the first 1024 columns are requested before the hand-off wait in front of the LayerNorm (ct_ln_issue) -- round 3's
// workgroup-wide version (4 barriers per row, gamma / beta read when needed) took 1.25 us per row, 13% of a layer
// (profiles/r04_codec_tr_phases.txt).
constexpr int CT_LN_J = 16;
struct CtLn { float g[CT_LN_J], b[CT_LN_J]; };

__device__ __forceinline__ void ct_ln_issue(CtLn& c, const float* gamma, const float* beta, int E, int lane) {
#pragma unroll
    for (int j = 0; j < CT_LN_J; ++j) {
        const int i = min(lane + 64 * j, E - 1);          // clamped: unconditional loads
        c.g[j] = gamma[i];
        c.b[j] = beta[i];
    }
}

template <int R>
__device__ __forceinline__ void ct_layernorm(const float* x, const CtLn& c, const float* gamma, const float* beta, float eps, int E, float* xs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < R; b += DF_WAVES) {
        const float* xr = x + b * E;
        float xv[CT_LN_J];          // the row's first 1024 columns stay in registers across the two passes
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CT_LN_J; ++j) {
            const int i = lane + 64 * j;
            xv[j] = i < E ? xr[i] : 0.f;
            s += xv[j];
        }
        for (int i = lane + 64 * CT_LN_J; i < E; i += 64) s += xr[i];
        const float mean = wave_sum_fast(s) / (float)E;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < CT_LN_J; ++j) {
            const float d = lane + 64 * j < E ? xv[j] - mean : 0.f;
            v = fmaf(d, d, v);
        }
        for (int i = lane + 64 * CT_LN_J; i < E; i += 64) { const float d = xr[i] - mean; v = fmaf(d, d, v); }
        const float rstd = 1.0f / sqrtf(wave_sum_fast(v) / (float)E + eps);
#pragma unroll
        for (int j = 0; j < CT_LN_J; ++j) {
            const int i = lane + 64 * j;
            if (i < E) xs[b * E + i] = (xv[j] - mean) * rstd * c.g[j] + c.b[j];
        }
        for (int i = lane + 64 * CT_LN_J; i < E; i += 64) xs[b * E + i] = (xr[i] - mean) * rstd * gamma[i] + beta[i];
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
    // LDS carve (floats): header | xs [R][max(E, F)] | xres [R][E] | qh [T][3][D] | sraw [R][capS] | pB [capS][R] | opart [1024 / D][R][D] | wred [2][4][R]
    // (capS = cap + 16384 / D: the ring rounded up to whole batches of 16 slots per slot class)
    float* xs = lds + DF_HDR_FLOATS;
    float* xres = xs + R * XW;
    float* qh = xres + R * E;
    const int capS = cap + 16384 / D;
    float* sraw = qh + T * 3 * D;                 // scores of query t against ring slot s: sraw[t * capS + s]
    float* pB = sraw + R * capS;                  // softmax numerators, slot-major: pB[s * R + t] (zero for masked / unused slots)
    float* opart = pB + R * capS;                 // partial outputs of the 1024 / D slot classes
    float* wred = opart + 1024 * R;               // per-wave row maxima | row sums
    u64* gX = p.gran + (SOLO ? (long)R * (5L * E + F) : 0L);
    u64* gQKV = gX + (long)R * E;
    u64* gATT = gQKV + (long)R * 3 * E;
    u64* gH = gATT + (long)R * E;
    if (tid == 0) sh.dead = 0;
#ifdef RST_ABLATION
    __shared__ unsigned long long ct_lds_stamps[CT_STAMPS_PER_LAYER * RST_CTR_MAX_L + 2];
    for (int i = tid; i < CT_STAMPS_PER_LAYER * RST_CTR_MAX_L + 2; i += DF_THREADS) ct_lds_stamps[i] = 0;     // (the barrier below orders it)
#endif
    CtPre<2, 1> pq;         // in-projection rows of this wave
    CtPre<1, 1> po;         // out-projection
    CtPre<2, 1> p1;         // linear1
    CtPre<1, 4> p2;         // linear2
    CtLn ln1, ln2;          // gamma / beta of the two LayerNorms of a layer
    ct_rows_issue<2, 1>(pq, p.in_proj[0], 3 * E, E, gw, W, lane);
    ct_ln_issue(ln1, p.n1g[0], p.n1b[0], E, lane);
    for (int i = tid; i < R * E; i += DF_THREADS) xres[i] = p.x[i];
    const long pos = *p.pos_dev;
    __syncthreads();
    unsigned eX = 0, eQKV = 0, eATT = 0, eH = 0;
    CT_STAMP(0);

    for (int l = 0; l < p.L; ++l) {
#ifdef RST_ABLATION
        const int sb = 1 + l * CT_STAMPS_PER_LAYER;
#endif
        // ---- in-projection of the normed rows
        ct_layernorm<R>(xres, ln1, p.n1g[l], p.n1b[l], p.eps, E, xs);
        ct_ln_issue(ln2, p.n2g[l], p.n2b[l], E, lane);
        CT_STAMP(sb + 0);
        ++eQKV;
        ct_rows<R, 2, 1>(pq, p.in_proj[l], 3 * E, E, xs, gw, W, lane, [&](int r, float (&s)[R]) {
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gQKV + (long)b * 3 * E + r, eQKV, s[b]);
        });
        CT_STAMP(sb + 1);
        ct_rows_issue<1, 1>(po, p.out_proj[l], E, E, gw, W, lane);
        // ---- attention of (stream, head) = workgroup index
        ++eATT;
        for (int bh = wg; bh < p.B * H; bh += G) {
            const int b = bh / H, h = bh - b * H;
            // q / k / v of the head for the T new steps of stream b: item i = (t, part, d)
            df_gather<2>(gQKV, T * 3 * D, eQKV, qh, [&](int i) { const int t = i / (3 * D), j = i - t * 3 * D, part = j / D;
                                                              return ((long)(b * T + t) * 3 + part) * E + h * D + (j - part * D); }, sh, p.status, 2u);
            CT_STAMP(sb + 2);
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
            // Thread (dq, cls): dims 4 dq .. 4 dq + 3 of the ring slots of class cls (slot % NC == cls, NC = 1024 / D classes) -- in the
            // ring append here AND in the K / V sweeps below, so a new step's row is read back by the very thread that stored it
            // (program order: no barrier, no wait for the store to land).
            const int GS = D >> 2, NC = DF_THREADS / GS, dq = tid % GS, cls = tid / GS;
            const int slot0 = (int)(pos % cap);            // new step t sits in slot (slot0 + t) % cap
            for (int t = 0; t < T; ++t) {
                int sl = slot0 + t;
                sl -= sl >= cap ? cap : 0;
                if ((sl & (NC - 1)) == cls) {
                    *reinterpret_cast<f32x4*>(kring + sl * D + 4 * dq) = *reinterpret_cast<const f32x4*>(qh + (t * 3 + 1) * D + 4 * dq);
                    *reinterpret_cast<f32x4*>(vring + sl * D + 4 * dq) = *reinterpret_cast<const f32x4*>(qh + (t * 3 + 2) * D + 4 * dq);
                }
            }
            CT_STAMP(sb + 3);
            // The T queries against the ring with ONE memory round trip (round 4): the K and V rows of a batch of 16 slots per class
            // are requested together (coalesced: a wave instruction covers whole rows), then (A) q . k per slot, summed over the D / 4
            // lanes of a row with DPP steps -> LDS; a two-pass softmax (the reference's: row maximum, numerators, row sum) with one
            // slot per thread; (B) o[t] += p[t][slot] v over the thread's slots; the NC partial outputs meet in LDS.  Masked and
            // never-written slots get p = 0, and a slot past the used part of the ring is not loaded at all (its V stays 0: a ring the
            // caller did not zero may hold Inf / NaN there).  Round 3 walked the ring once per query in 16-dim lane groups with an
            // online softmax and a 4-level shuffle merge: 9.2 us per query, 48% of a layer (profiles/r04_codec_tr_phases_*.txt).
            {
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
                    for (int j = 0; j < JB; ++j)
                        kreg[j] = *reinterpret_cast<const f32x4*>(kring + min(base + cls + NC * j, n_used - 1) * D + 4 * dq);
                };
                auto v_issue = [&](int base) {
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        const int sl = base + cls + NC * j;
                        vreg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (sl < n_used) vreg[j] = *reinterpret_cast<const f32x4*>(vring + sl * D + 4 * dq);
                    }
                };
                k_issue(0);
                v_issue(0);
                f32x4 q4[R];
#pragma unroll
                for (int t = 0; t < R; ++t) q4[t] = *reinterpret_cast<const f32x4*>(qh + ((t < T ? t : 0) * 3) * D + 4 * dq);
                CT_STAMP(sb + 13);
                // (A)
                for (int bt = 0; bt < nb; ++bt) {
                    if (bt) k_issue(bt * SPB);
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        float tot[R];
#pragma unroll
                        for (int t = 0; t < R; ++t) {
                            float a = kreg[j][0] * q4[t][0];
                            a = fmaf(kreg[j][1], q4[t][1], a); a = fmaf(kreg[j][2], q4[t][2], a); a = fmaf(kreg[j][3], q4[t][3], a);
                            tot[t] = group_sum(a, GS);
                        }
                        if (dq == 0) {
#pragma unroll
                            for (int t = 0; t < R; ++t) sraw[t * capS + bt * SPB + cls + NC * j] = tot[t];
                        }
                    }
                }
                CT_STAMP(sb + 14);
                __syncthreads();
                // softmax, one slot per thread and sweep
                float tmax[R], M[R], tsum[R];
#pragma unroll
                for (int t = 0; t < R; ++t) { tmax[t] = -INFINITY; tsum[t] = 0.f; }
                for (int sl = tid; sl < nb * SPB; sl += DF_THREADS) {
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        const bool ok = t < T && sl < n_used && ring_visible_at(sl, pos + t, cap, p.context, end_offset, end_index);
                        const float sv = ok ? sraw[t * capS + sl] * scale : -INFINITY;
                        sraw[t * capS + sl] = sv;
                        tmax[t] = fmaxf(tmax[t], sv);
                    }
                }
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const float m = wave_max_fast(tmax[t]);
                    if (lane == 0) wred[wave * R + t] = m;
                }
                __syncthreads();
                CT_STAMP(sb + 15);
#pragma unroll
                for (int t = 0; t < R; ++t) M[t] = fmaxf(fmaxf(wred[t], wred[R + t]), fmaxf(wred[2 * R + t], wred[3 * R + t]));
                for (int sl = tid; sl < nb * SPB; sl += DF_THREADS) {
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        const float sv = sraw[t * capS + sl];
                        const float pv = sv == -INFINITY ? 0.f : expf(sv - M[t]);
                        pB[sl * R + t] = pv;
                        tsum[t] += pv;
                    }
                }
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const float a = wave_sum_fast(tsum[t]);
                    if (lane == 0) wred[(DF_WAVES + wave) * R + t] = a;
                }
                __syncthreads();
                CT_STAMP(sb + 16);
                // (B)
                f32x4 o[R];
#pragma unroll
                for (int t = 0; t < R; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int bt = 0; bt < nb; ++bt) {
                    if (bt) v_issue(bt * SPB);
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        const float* pr = pB + (bt * SPB + cls + NC * j) * R;
#pragma unroll
                        for (int t = 0; t < R; ++t) {
                            const float w = pr[t];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[t][e] = fmaf(w, vreg[j][e], o[t][e]);
                        }
                    }
                }
                CT_STAMP(sb + 17);
#pragma unroll
                for (int t = 0; t < R; ++t) *reinterpret_cast<f32x4*>(opart + (cls * R + t) * D + 4 * dq) = o[t];
                CT_STAMP(sb + 18);
                __syncthreads();
                CT_STAMP(sb + 19);
                for (int i = tid; i < T * D; i += DF_THREADS) {
                    const int t = i / D, d = i - t * D;
                    float Oq = 0.f;
                    for (int c = 0; c < NC; ++c) Oq += opart[(c * R + t) * D + d];
                    const float* ws = wred + DF_WAVES * R + t;
                    const float Lq = (ws[0] + ws[R]) + (ws[2 * R] + ws[3 * R]);
                    df_publish(gATT + (long)(b * T + t) * E + h * D + d, eATT, Lq > 0.f ? Oq / Lq : 0.f);
                }
            }
        }
        CT_STAMP(sb + 4);
        // ---- out-projection, LayerScale, residual
        df_gather<4>(gATT, R * E, eATT, xs, [](int i) { return i; }, sh, p.status, 4u);
        CT_STAMP(sb + 5);
        ++eX;
        ct_rows<R, 1, 1>(po, p.out_proj[l], E, E, xs, gw, W, lane, [&](int r, float (&s)[R]) {
            const float sc1 = p.ls1[l] ? p.ls1[l][r] : 1.0f;
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gX + (long)b * E + r, eX, xres[b * E + r] + sc1 * s[b]);
        });
        CT_STAMP(sb + 6);
        ct_rows_issue<2, 1>(p1, p.lin1[l], F, E, gw, W, lane);
        df_gather<4>(gX, R * E, eX, xres, [](int i) { return i; }, sh, p.status, 8u);
        CT_STAMP(sb + 7);
        // ---- FFN: x + scale2 * W2 gelu(W1 LN(x))
        ct_layernorm<R>(xres, ln2, p.n2g[l], p.n2b[l], p.eps, E, xs);
        if (l + 1 < p.L) ct_ln_issue(ln1, p.n1g[l + 1], p.n1b[l + 1], E, lane);
        CT_STAMP(sb + 8);
        ++eH;
        ct_rows<R, 2, 1>(p1, p.lin1[l], F, E, xs, gw, W, lane, [&](int r, float (&s)[R]) {
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gH + (long)b * F + r, eH, rst_gelu(s[b]));
        });
        CT_STAMP(sb + 9);
        ct_rows_issue<1, 4>(p2, p.lin2[l], E, F, gw, W, lane);
        df_gather<8>(gH, R * F, eH, xs, [](int i) { return i; }, sh, p.status, 16u);
        CT_STAMP(sb + 10);
        ++eX;
        ct_rows<R, 1, 4>(p2, p.lin2[l], E, F, xs, gw, W, lane, [&](int r, float (&s)[R]) {
            const float sc2 = p.ls2[l] ? p.ls2[l][r] : 1.0f;
#pragma unroll
            for (int b = 0; b < R; ++b) df_publish(gX + (long)b * E + r, eX, xres[b * E + r] + sc2 * s[b]);
        });
        CT_STAMP(sb + 11);
        if (l + 1 < p.L) ct_rows_issue<2, 1>(pq, p.in_proj[l + 1], 3 * E, E, gw, W, lane);
        df_gather<4>(gX, R * E, eX, xres, [](int i) { return i; }, sh, p.status, 32u);
        CT_STAMP(sb + 12);
    }
    if (wg == 0)
        for (int i = tid; i < R * E; i += DF_THREADS) p.y[i] = xres[i];
#ifdef RST_ABLATION
    if (!SOLO && tid == 0 && (wg == 0 || wg == G - 1))
        for (int i = 0; i < CT_STAMPS_PER_LAYER * RST_CTR_MAX_L + 2; ++i) ct_stamps[wg ? 1 : 0][i] = ct_lds_stamps[i];
#endif
    if (SOLO) df_solo_done(p.status);
}

}  // namespace

#ifdef RST_ABLATION
// tools build only: the stamps of the last persistent launch (2 workgroups x (1 + 13 per layer)), 100 MHz ticks
extern "C" int rst_debug_codec_tr_stamps(unsigned long long* out, int n) {
    const int have = 2 * (CT_STAMPS_PER_LAYER * RST_CTR_MAX_L + 2);
    if (n > have) n = have;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ct_stamps), sizeof(unsigned long long) * n) == hipSuccess ? n : -1;
}
#endif

namespace {

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
    const size_t lds = ((size_t)DF_HDR_FLOATS + (size_t)R * XW + (size_t)R * E + (size_t)T * 3 * D + 2 * (size_t)R * (cap + 16384 / D) + (size_t)(1024 + 2 * DF_WAVES) * R) * sizeof(float);
    if (lds > 150 * 1024) return 0;
    int G = df_grid_for_rows(ct_cu_count(), min(3 * E, F));
    // tools build only -- RST_CTR_GRID=n (RST_DF_GRID for the depth frame): at most n workgroups (fewer participants per all-to-all hand-off against fewer CUs
    // streaming the weights: the measurement behind DESIGN.md 6)
    static const int gcap = rst_knob("RST_CTR_GRID", 0);
    if (gcap > 0 && G > gcap) G = gcap;
    if (B * H > G) return 0;
    static signed char fits_dev[RST_MAX_DEVICES];        // per device: 0 = not asked yet, 1 = fits, -1 = does not
    signed char uncached = 0;
    signed char& fits = rst_device_cell(fits_dev, 1, uncached);
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
    const size_t lds = ((size_t)DF_HDR_FLOATS + (size_t)R * XW + (size_t)R * p.E + (size_t)p.T * 3 * p.D + 2 * (size_t)R * (p.cap + 16384 / p.D) + (size_t)(1024 + 2 * DF_WAVES) * R) * sizeof(float);
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
