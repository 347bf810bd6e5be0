// The depth phase of one LM frame as ONE persistent launch: dep_q sequential steps x (L layers x {in-proj, attention, out-proj,
// gated FFN in / out} + head + sampler), batch <= 2 -- models/model.py:564-597 (LMGen.depformer_step: forward_depformer :392-428
// + sample_token per codebook) and the codecformer loop of infer_no_streaming.py:262-283 / llama_streaming.py:727-749.
//
// Why: as separate launches the phase is a chain of ~210 dependent kernels of 2-11 MB each; a graph-replayed GEMV of that size
// costs 4.0-6.4 us against 1.7 us for an empty kernel (tools/bench_depth.py), i.e. the chain is launch + fill / drain latency,
// not bandwidth (1.33 GB per frame = 0.25 ms at 5.3 TB/s, measured 1.5 ms).  Here every op boundary is an in-launch all-to-all
// hand-off instead (cdna_hip_programming.md Guideline 16, form R2): an op's output vector lives in a global array of 8-byte
// {epoch tag, fp32 value} granules, each written by ONE relaxed agent-scope (sc1, write-through) store; every consuming
// workgroup sweeps the granules it needs with relaxed agent-scope loads until all tags equal the op's epoch and stages the values
// in LDS.  No flag, no fence, no dispatch-order or placement assumption; the workspace is zeroed by a memset node in front of
// every launch (epochs count from 1 inside a launch), every spin is bounded and a timeout poisons nothing but this frame
// (status word != 0, the launch still terminates).
//
// Work split: G workgroups (one per CU, all resident) x 4 waves; weight row r of an op belongs to wave (r mod 4G) -- the same
// for every op, so a row's residual input is the value the workgroup already gathered.  Every workgroup keeps the whole
// residual stream x [B][E] in LDS and recomputes the RMSNorms itself.  Attention: head h is owned by workgroup h, which keeps
// that head's keys / values of the frame's earlier steps in its LDS (the depth transformer's KV ring never touches HBM here)
// and reproduces RingKVCache.complete's slot -> position map for a ring of dep_q slots (modules/transformer.py:254-278, incl. the
// `delta <= 0` slot that hides step 0 at the last step).  The sampler (lm_sample_impl.h) runs in workgroup 0.
// Arithmetic (per-lane k order, norm partials, gate, softmax) is that of the launch-per-op path (lm_step.hip, lm_attn.hip) up to the
// order of the additions inside a wave reduction (DPP steps here, a shuffle butterfly there).
#include "lm_common.h"
#include "lm_sample_impl.h"
#include "persist.h"

namespace {

// tools build only: wall-clock stamps (100 MHz) of workgroup 0 at the op boundaries of every (step, layer), kept in LDS while the
// launch runs (tools/probes/depth_frame_phases.py prints them; a stamp costs ~0.2 us itself)
#ifdef RST_ABLATION
#define DF_ST_LAYER 13
#define DF_ST_STEP (DF_ST_LAYER * RST_DEPTH_MAX_L + 4 + 8)
#define DF_ST_TOTAL (DF_ST_STEP * RST_DEPTH_MAX_Q + 2)
__device__ unsigned long long df_stamps[DF_ST_TOTAL];
#define DF_STAMP(i) do { if (!SOLO && tid == 0) df_lds_stamps[(i)] = wall_clock64(); } while (0)
#else
#define DF_STAMP(i) do {} while (0)
#endif

// xs[b][i] = x[b][i] * alpha[i] / sqrt(eps + mean(x[b]^2))   (modules/transformer.py:34-46; the summation order of gemv_kernel)
// Every wave computes the statistic of every row for itself (no exchange through LDS, one barrier at the end): the four partial
// sums are those of the four waves of gemv_kernel's prologue (element i belongs to partial (i / 64) % 4), added in the same order.
// Round 3's version exchanged the partials through LDS (3 barriers per row, shuffle butterflies): ~1 us per norm.
// alpha of the first DF_NORM_J * 256 columns rides in registers across the hand-off in front of the norm (df_norm_issue): read
// where it is used it cost a global round trip (~0.8 us) in every norm.
constexpr int DF_NORM_J = 8;
struct DfNorm { float a[DF_NORM_J]; };
__device__ __forceinline__ void df_norm_issue(DfNorm& c, const float* alpha, int E) {
#pragma unroll
    for (int j = 0; j < DF_NORM_J; ++j) c.a[j] = alpha[min((int)threadIdx.x + DF_THREADS * j, E - 1)];      // clamped: unconditional loads
}

template <int B>
__device__ __forceinline__ void df_rmsnorm(const float* x, const DfNorm& c, const float* alpha, float eps, int E, float* xs) {
    const int tid = threadIdx.x, lane = tid & 63;
    for (int b = 0; b < B; ++b) {
        float s[DF_WAVES] = {0.f, 0.f, 0.f, 0.f};
        for (int i = lane; i < E; i += DF_THREADS) {
#pragma unroll
            for (int w = 0; w < DF_WAVES; ++w) {
                const float v = i + 64 * w < E ? x[b * E + i + 64 * w] : 0.f;
                s[w] = fmaf(v, v, s[w]);
            }
        }
#pragma unroll
        for (int w = 0; w < DF_WAVES; ++w) s[w] = wave_sum_fast(s[w]);
        const float tot = s[0] + s[1] + s[2] + s[3];
        const float r = 1.0f / sqrtf(eps + tot / (float)E);
#pragma unroll
        for (int j = 0; j < DF_NORM_J; ++j) {
            const int i = tid + DF_THREADS * j;
            if (i < E) xs[b * E + i] = x[b * E + i] * (c.a[j] * r);
        }
        for (int i = tid + DF_THREADS * DF_NORM_J; i < E; i += DF_THREADS) xs[b * E + i] = x[b * E + i] * (alpha[i] * r);
    }
    __syncthreads();
}

// Rows r = gw, gw + W, ... of w [N][K] (bf16) against xs [B][K] (LDS), in blocks of RU rows x CU chunks of 512 k whose loads are
// all requested before the first is used.  PAIR: "row" q stands for the rows (q, N/2 + q) of a stacked gated layer.
// The FIRST block (rows gw + j W, k < CU * 512 -- the whole op at the depth transformer's real shape) can be requested ahead of
// time: df_rows_issue() before the hand-off that produces xs, df_rows() after it -- the weights do not depend on activations, so
// their HBM latency hides behind the wait (the "prefetch-credit" of MI355X_MICROARCH.md's price list).  epi(row, sums) on lane 0.
template <int RU, int CU, bool PAIR> struct DfPre { u32x4 wv[RU][PAIR ? 2 : 1][CU]; };

template <int RU, int CU, bool PAIR>
__device__ __forceinline__ void df_rows_load(DfPre<RU, CU, PAIR>& pre, const unsigned short* w, int rows, int K, int r0, int kb, int W, int lane) {
    constexpr int HV = PAIR ? 2 : 1;
#pragma unroll
    for (int j = 0; j < RU; ++j)
#pragma unroll
        for (int h = 0; h < HV; ++h)
#pragma unroll
            for (int c = 0; c < CU; ++c) {
                const int r = r0 + j * W, kk = kb + c * 512 + lane * 8;
                pre.wv[j][h][c] = u32x4{0u, 0u, 0u, 0u};
                if (r < rows && kk < K)
                    pre.wv[j][h][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + ((long)r + (long)h * rows) * K + kk));
            }
}

template <int RU, int CU, bool PAIR>
__device__ __forceinline__ void df_rows_issue(DfPre<RU, CU, PAIR>& pre, const unsigned short* w, int N, int K, int gw, int W, int lane) {
    df_rows_load<RU, CU, PAIR>(pre, w, PAIR ? N / 2 : N, K, gw, 0, W, lane);
}

template <int B, int RU, int CU, bool PAIR, typename Epi>
__device__ __forceinline__ void df_rows(DfPre<RU, CU, PAIR>& pre, const unsigned short* w, int N, int K, const float* xs, int gw, int W, int lane,
                                        Epi epi) {
    constexpr int HV = PAIR ? 2 : 1;
    const int rows = PAIR ? N / 2 : N;
    for (int r0 = gw; r0 < rows; r0 += RU * W) {
        float acc[RU][HV][B];
#pragma unroll
        for (int j = 0; j < RU; ++j)
#pragma unroll
            for (int h = 0; h < HV; ++h)
#pragma unroll
                for (int b = 0; b < B; ++b) acc[j][h][b] = 0.f;
        for (int kb = 0; kb < K; kb += CU * 512) {
            if (r0 != gw || kb != 0) df_rows_load<RU, CU, PAIR>(pre, w, rows, K, r0, kb, W, lane);     // the first block was issued ahead
#pragma unroll
            for (int c = 0; c < CU; ++c) {
                const int kk = kb + c * 512 + lane * 8;
                if (kk < K) {
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + b * K + kk);
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + b * K + kk + 4);
#pragma unroll
                        for (int j = 0; j < RU; ++j)
#pragma unroll
                            for (int h = 0; h < HV; ++h) {
                                const u32x4 v = pre.wv[j][h][c];
                                float a = acc[j][h][b];
                                a = fmaf(bf16_lo(v[0]), x0[0], a); a = fmaf(bf16_hi(v[0]), x0[1], a);
                                a = fmaf(bf16_lo(v[1]), x0[2], a); a = fmaf(bf16_hi(v[1]), x0[3], a);
                                a = fmaf(bf16_lo(v[2]), x1[0], a); a = fmaf(bf16_hi(v[2]), x1[1], a);
                                a = fmaf(bf16_lo(v[3]), x1[2], a); a = fmaf(bf16_hi(v[3]), x1[3], a);
                                acc[j][h][b] = a;
                            }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            const int r = r0 + j * W;
            float s[HV][B];
#pragma unroll
            for (int h = 0; h < HV; ++h)
#pragma unroll
                for (int b = 0; b < B; ++b) s[h][b] = wave_sum_fast(acc[j][h][b]);
            if (lane == 0 && r < rows) epi(r, s);
        }
    }
}

// The KV history of the frame's earlier steps: LDS of the head's owner; in the SOLO (repair) launch one workgroup owns every head and
// the history of all of them lives in a global scratch, read and written with agent-scope (L1-bypassing) accesses.
template <bool SOLO> struct DfHist {
    float* base;
    __device__ __forceinline__ float rd(long i) const { return SOLO ? __hip_atomic_load(base + i, DF_RLX) : base[i]; }
    __device__ __forceinline__ void wr(long i, float v) const { if (SOLO) __hip_atomic_store(base + i, v, DF_RLX); else base[i] = v; }
};

template <int B, bool SOLO>
__global__ __launch_bounds__(DF_THREADS) void depth_frame_kernel(const DepthFrameParams p) {
    // SOLO: the repair launch behind the persistent one (persist.h) -- a no-op unless a hand-off of that launch timed out
    if (SOLO && __hip_atomic_load(p.status, DF_RLX) == 0u) return;
    // all LDS is carved from the dynamic region (static objects in front of it would shift its 16-byte alignment: G17)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    DfShared& sh = *reinterpret_cast<DfShared*>(lds);
    SampleShared<DF_THREADS>& ssh = *reinterpret_cast<SampleShared<DF_THREADS>*>(lds + DF_HDR_FLOATS / 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x, G = gridDim.x;
    const int gw = wg * DF_WAVES + wave, W = G * DF_WAVES;
    const int E = p.E, Hd = p.Hd, D = p.D, card = p.card, dep_q = p.dep_q;
    const int XW = E > Hd ? E : Hd;
    // LDS carve (floats): header | xs [B][max(E, Hd)] | xres [B][E] | lg [B][card] | qh [B][3D] | hist [L][dep_q][B][2][D] | comp (u64)
    float* xs = lds + DF_HDR_FLOATS;
    float* xres = xs + B * XW;
    float* lg = xres + B * E;
    float* qh = lg + B * card;
    const long hist_head = (long)p.L * dep_q * B * 2 * D;       // history floats of one head
    const DfHist<SOLO> hist{SOLO ? p.hist_solo : qh + B * 3 * D};
    long off = DF_HDR_FLOATS + (long)B * XW + (long)B * E + (long)B * card + (long)B * 3 * D + (long)p.L * dep_q * B * 2 * D;
    off += off & 1;
    u64* comp = reinterpret_cast<u64*>(lds + off);
    // granule workspace (the repair launch has its own zeroed copy behind the persistent launch's)
    u64* gX = p.gran + (SOLO ? (long)B * (5L * E + Hd + card + 1) : 0L);
    u64* gQKV = gX + (long)B * E;
    u64* gATT = gQKV + (long)B * 3 * E;
    u64* gH = gATT + (long)B * E;
    u64* gLOG = gH + (long)B * Hd;
    u64* gTOK = gLOG + (long)B * card;
    if (tid == 0) sh.dead = 0;
#ifdef RST_ABLATION
    __shared__ __attribute__((aligned(16))) unsigned long long df_lds_stamps[DF_ST_TOTAL];
    for (int i = tid; i < DF_ST_TOTAL; i += DF_THREADS) df_lds_stamps[i] = 0;
#endif
    __syncthreads();
    DF_STAMP(0);
    unsigned eX = 0, eQKV = 0, eATT = 0, eH = 0, eLOG = 0, eTOK = 0;      // epochs (number of completed writes) per buffer
    const float att_div = sqrtf((float)D);

    DfPre<3, 2, false> pq;      // in-projection rows of this wave (requested one hand-off ahead)
    DfPre<1, 2, false> po;      // out-projection
    DfPre<3, 2, true> pi;       // gated FFN in (u, v row pairs)
    DfPre<1, 6, false> pf;      // gated FFN out
    DfPre<2, 2, false> ph;      // head
    DfNorm n1, n2;              // alpha of the two norms of a layer
    df_rows_issue<3, 2, false>(pq, p.in_proj[0], 3 * E, E, gw, W, lane);
    for (int k = 0; k < dep_q; ++k) {
        df_norm_issue(n1, p.norm1[0], E);
        // ---- input of the step: x = depformer_in[k](h) + emb_k[previous token]   (models/model.py:411-417)
        if (k == 0) {
            if (tid < B) sh.tok[tid] = p.tokens[(long)tid * p.tok_stride];
            __syncthreads();
        } else {
            df_gather<1>(gTOK, B, eTOK, sh.tokf, [](int i) { return i; }, sh, p.status, 1u);
            if (tid < B) sh.tok[tid] = (long)__float_as_int(sh.tokf[tid]);
            __syncthreads();
        }
        for (int b = 0; b < B; ++b) {
            long tok = sh.tok[b];
            const bool zero = tok == -1;
            tok = tok < 0 ? 0 : (tok >= p.emb_rows[k] ? p.emb_rows[k] - 1 : tok);
            const unsigned short* er = p.emb[k] + tok * (long)E;
            for (int i = tid; i < E; i += DF_THREADS) {
                float v = p.h_all[(long)b * p.ld_h + (long)k * E + i];
                if (!zero) v += __uint_as_float((unsigned)er[i] << 16);
                xres[b * E + i] = v;
            }
        }
        __syncthreads();
#ifdef RST_ABLATION
        const int sk = 1 + k * DF_ST_STEP;
#endif
        DF_STAMP(sk);

        for (int l = 0; l < p.L; ++l) {
#ifdef RST_ABLATION
            const int sl = sk + 1 + l * DF_ST_LAYER;
#endif
            // ---- in-projection: qkv = W_in[k] rmsnorm(x)   (its weights were requested before the hand-off that produced x)
            df_rmsnorm<B>(xres, n1, p.norm1[l], p.eps, E, xs);
            df_norm_issue(n2, p.norm2[l], E);
            DF_STAMP(sl + 0);
            ++eQKV;
            df_rows<B, 3, 2, false>(pq, p.in_proj[l] + (long)k * 3 * E * E, 3 * E, E, xs, gw, W, lane, [&](int r, float (&s)[1][B]) {
#pragma unroll
                for (int b = 0; b < B; ++b) df_publish(gQKV + (long)b * 3 * E + r, eQKV, s[0][b]);
            });
            DF_STAMP(sl + 1);
            df_rows_issue<1, 2, false>(po, p.out_proj[l] + (long)k * E * E, E, E, gw, W, lane);
            // ---- attention of head `wg` (modules/transformer.py:376-416 on a ring of ring_cap slots, no rope)
            ++eATT;
            for (int h = wg; h < p.H; h += G) {
                const long hb = SOLO ? h * hist_head : 0;
                df_gather<2>(gQKV, B * 3 * D, eQKV, qh, [&](int i) { const int b = i / (3 * D), j = i - b * 3 * D, part = j / D;
                                                                  return b * 3 * E + part * E + h * D + (j - part * D); }, sh, p.status, 2u);
                DF_STAMP(sl + 2);
                const long hk = hb + (((long)l * dep_q + k) * B) * 2 * D;       // [B][2][D] of this (layer, step)
                for (int i = tid; i < B * 2 * D; i += DF_THREADS) {
                    const int b = i / (2 * D), j = i - b * 2 * D;
                    hist.wr(hk + i, qh[b * 3 * D + D + j]);
                }
                if (SOLO) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the write-through stores have landed before the reads below
                __syncthreads();
                DF_STAMP(sl + 3);
                if (wave == 0) {
                    // steps unrolled to the table size (scores stay in registers; the dot products of the steps are independent)
                    const int end_index = (k + 1) % p.ring_cap;
                    for (int b = 0; b < B; ++b) {
                        float sc[RST_DEPTH_MAX_Q], m = -INFINITY;
#pragma unroll
                        for (int s = 0; s < RST_DEPTH_MAX_Q; ++s) {
                            sc[s] = -INFINITY;
                            if (s <= k) {
                                const long ks = hb + ((((long)l * dep_q + s) * B + b) * 2) * D;
                                float d = 0.f;
                                for (int dd = lane; dd < D; dd += 64) d = fmaf(hist.rd(ks + dd), qh[b * 3 * D + dd], d);
                                d = wave_sum_fast(d);
                                sc[s] = ring_visible_at(s, k, p.ring_cap, p.context, (long)k + 1, end_index) ? d / att_div : -INFINITY;
                                m = fmaxf(m, sc[s]);
                            }
                        }
                        float lsum = 0.f;
#pragma unroll
                        for (int s = 0; s < RST_DEPTH_MAX_Q; ++s) { sc[s] = sc[s] == -INFINITY ? 0.f : expf(sc[s] - m); lsum += sc[s]; }
                        for (int dd = lane; dd < D; dd += 64) {
                            float o = 0.f;
#pragma unroll
                            for (int s = 0; s < RST_DEPTH_MAX_Q; ++s)
                                if (s <= k) o = fmaf(sc[s], hist.rd(hb + ((((long)l * dep_q + s) * B + b) * 2 + 1) * D + dd), o);
                            df_publish(gATT + (long)b * E + h * D + dd, eATT, o / lsum);
                        }
                    }
                }
                DF_STAMP(sl + 4);
            }
            // ---- out-projection + residual
            df_gather<4>(gATT, B * E, eATT, xs, [](int i) { return i; }, sh, p.status, 4u);
            DF_STAMP(sl + 5);
            ++eX;
            df_rows<B, 1, 2, false>(po, p.out_proj[l] + (long)k * E * E, E, E, xs, gw, W, lane, [&](int r, float (&s)[1][B]) {
#pragma unroll
                for (int b = 0; b < B; ++b) df_publish(gX + (long)b * E + r, eX, xres[b * E + r] + s[0][b]);
            });
            DF_STAMP(sl + 6);
            df_rows_issue<3, 2, true>(pi, p.gate_in[l][k], 2 * Hd, E, gw, W, lane);
            df_gather<4>(gX, B * E, eX, xres, [](int i) { return i; }, sh, p.status, 8u);
            DF_STAMP(sl + 7);
            // ---- gated FFN: x + W_out (silu(u) * v), [u ; v] = W_in rmsnorm(x)   (modules/gating.py:12-51)
            df_rmsnorm<B>(xres, n2, p.norm2[l], p.eps, E, xs);
            if (l + 1 < p.L) df_norm_issue(n1, p.norm1[l + 1], E);
            DF_STAMP(sl + 8);
            ++eH;
            df_rows<B, 3, 2, true>(pi, p.gate_in[l][k], 2 * Hd, E, xs, gw, W, lane, [&](int q, float (&s)[2][B]) {
#pragma unroll
                for (int b = 0; b < B; ++b) df_publish(gH + (long)b * Hd + q, eH, silu(s[0][b]) * s[1][b]);
            });
            DF_STAMP(sl + 9);
            df_rows_issue<1, 6, false>(pf, p.gate_out[l][k], E, Hd, gw, W, lane);
            df_gather<6>(gH, B * Hd, eH, xs, [](int i) { return i; }, sh, p.status, 16u);
            DF_STAMP(sl + 10);
            ++eX;
            df_rows<B, 1, 6, false>(pf, p.gate_out[l][k], E, Hd, xs, gw, W, lane, [&](int r, float (&s)[1][B]) {
#pragma unroll
                for (int b = 0; b < B; ++b) df_publish(gX + (long)b * E + r, eX, xres[b * E + r] + s[0][b]);
            });
            DF_STAMP(sl + 11);
            // the next consumer of x: the next layer's in-projection, or the head of this step
            if (l + 1 < p.L) df_rows_issue<3, 2, false>(pq, p.in_proj[l + 1] + (long)k * 3 * E * E, 3 * E, E, gw, W, lane);
            else df_rows_issue<2, 2, false>(ph, p.heads[k], card, E, gw, W, lane);
            df_gather<4>(gX, B * E, eX, xres, [](int i) { return i; }, sh, p.status, 32u);
            DF_STAMP(sl + 12);
        }

        // ---- head: logits = linears[k](x)   (models/model.py:425-427; no norm in front)
        ++eLOG;
        df_rows<B, 2, 2, false>(ph, p.heads[k], card, E, xres, gw, W, lane, [&](int r, float (&s)[1][B]) {
            const float bias = p.head_bias[k] ? p.head_bias[k][r] : 0.f;
#pragma unroll
            for (int b = 0; b < B; ++b) df_publish(gLOG + (long)b * card + r, eLOG, s[0][b] + bias);
        });
        DF_STAMP(sk + 1 + DF_ST_LAYER * RST_DEPTH_MAX_L);
        if (k + 1 < dep_q) df_rows_issue<3, 2, false>(pq, p.in_proj[0] + (long)(k + 1) * 3 * E * E, 3 * E, E, gw, W, lane);
        // ---- sampler: utils/sampling.py:85-105.  Workgroup b draws the token of batch row b (round 5: with the rows one after the other
        // in workgroup 0 the second row's 14 us sat on every step's critical path at batch 2); the one-workgroup repair launch takes all rows
        ++eTOK;
        if (SOLO || wg < B) {
            const int b_lo = SOLO ? 0 : wg, b_hi = SOLO ? B : wg + 1;
            df_gather<8>(gLOG + (long)b_lo * card, (b_hi - b_lo) * card, eLOG, lg + b_lo * card, [](int i) { return i; }, sh, p.status, 64u);
            DF_STAMP(sk + 2 + DF_ST_LAYER * RST_DEPTH_MAX_L);
            for (int b = b_lo; b < b_hi; ++b) {
                const float* nz = p.noise ? p.noise + (long)b * p.noise_stride + (long)k * p.top_k : nullptr;
                const int limit = p.v_limit ? p.v_limit[k] : 0;
#ifdef RST_ABLATION
                unsigned long long* sdbg = SOLO || b ? nullptr : df_lds_stamps + sk + 4 + DF_ST_LAYER * RST_DEPTH_MAX_L;
#else
                unsigned long long* sdbg = nullptr;
#endif
                const int tok = card <= 8 * DF_THREADS
                    ? sample_row<DF_THREADS, 8>(lg + b * card, nz, card, p.top_k, p.use_sampling && p.temp > 0.f, p.temp, limit, comp, ssh, 0, nullptr, sdbg)
                    : sample_row<DF_THREADS, 16>(lg + b * card, nz, card, p.top_k, p.use_sampling && p.temp > 0.f, p.temp, limit, comp, ssh, 0, nullptr, sdbg);
                if (tid == 0) {
                    p.tokens[(long)b * p.tok_stride + k + 1] = tok;
                    df_publish(gTOK + b, eTOK, __int_as_float(tok));
                }
                __syncthreads();
            }
            DF_STAMP(sk + 3 + DF_ST_LAYER * RST_DEPTH_MAX_L);
        }
    }
#ifdef RST_ABLATION
    if (!SOLO && wg == 0) {
        __syncthreads();
        for (int i = tid; i < DF_ST_TOTAL; i += DF_THREADS) df_stamps[i] = df_lds_stamps[i];
    }
#endif
    if (SOLO) df_solo_done(p.status);
}

}  // namespace

#ifdef RST_ABLATION
// tools build only: the stamps of workgroup 0 of the last persistent launch, 100 MHz ticks; layout: [0] start, then per step
// [embed | 13 per layer x RST_DEPTH_MAX_L | head rows | sampler gather | sampler done]
extern "C" int rst_debug_depth_frame_stamps(unsigned long long* out, int n) {
    if (n > DF_ST_TOTAL) n = DF_ST_TOTAL;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(df_stamps), sizeof(unsigned long long) * n) == hipSuccess ? n : -1;
}
#endif

namespace {

int df_cu_count() {
    return rst_cu_count();       // per device (rst_common.h)
}

size_t df_lds_bytes(const DepthFrameParams& p) {
    const int XW = p.E > p.Hd ? p.E : p.Hd;
    static_assert(sizeof(DfShared) <= DF_HDR_FLOATS * 2 && sizeof(SampleShared<DF_THREADS>) <= DF_HDR_FLOATS * 2, "LDS header too small");
    size_t fl = DF_HDR_FLOATS + (size_t)p.B * XW + (size_t)p.B * p.E + (size_t)p.B * p.card + (size_t)p.B * 3 * p.D + (size_t)p.L * p.dep_q * p.B * 2 * p.D;
    fl += fl & 1;
    const int k = p.top_k > 0 && p.top_k < p.card ? p.top_k : p.card;
    return fl * sizeof(float) + (size_t)((k + 7) & ~7) * 8;
}

}  // namespace

long rst_depth_frame_workspace_granules(int B, int E, int Hd, int card) { return (long)B * (5L * E + Hd + card + 1); }

// workspace: granules of the persistent launch | granules of the repair launch | KV history of the repair launch (fp32, all heads)
long rst_depth_frame_workspace_bytes_impl(int B, int E, int Hd, int card) {
    return 2 * rst_depth_frame_workspace_granules(B, E, Hd, card) * 8 + (long)RST_DEPTH_MAX_L * RST_DEPTH_MAX_Q * B * 2 * E * 4;
}

// Workgroups of the persistent launch for a shape, 0 if it is not served: every workgroup must own a row of the in-projection
// (3E rows), the gated FFN-in (Hd row pairs) and the head (card rows) -- the all-to-all ops between two writes of a hand-off buffer
// (persist.h) -- every head needs a workgroup, and one workgroup of that LDS footprint must fit a CU (occupancy query).
int rst_depth_frame_grid(const DepthFrameParams& p) {
    if (!(p.B >= 1 && p.B <= 2 && p.E > 0 && p.E % 8 == 0 && p.Hd > 0 && p.Hd % 8 == 0 && p.H > 0 && p.D > 0 && p.H * p.D == p.E && p.card > 0 &&
          p.card <= 16 * DF_THREADS && p.dep_q >= 1 && p.dep_q <= RST_DEPTH_MAX_Q && p.L >= 1 && p.L <= RST_DEPTH_MAX_L))
        return 0;
    const size_t lds = df_lds_bytes(p);
    if (lds > 150 * 1024) return 0;
    const int rows = min(min(3 * p.E, p.Hd), p.card);
    int G = df_grid_for_rows(df_cu_count(), rows);
    static const int cap = rst_knob("RST_DF_GRID", 0);      // tools build only (see codec_tr.hip)
    if (cap > 0 && G > cap) G = cap;
    if (p.H > G) return 0;
    // residency: all G workgroups must run at once, one per CU -- ask the runtime whether a CU takes a workgroup of this footprint
    static signed char fits[RST_MAX_DEVICES][2][2];      // [device][B - 1][lds > 64 KB]: 0 = not asked yet, 1 = fits, -1 = does not; the
                                                         // answer does not change within a footprint class
    signed char uncached = 0;
    signed char& f = rst_device_cell(&fits[0][p.B - 1][lds > 64 * 1024], 4, uncached);
    if (f == 0) {
        const void* kern = p.B == 1 ? reinterpret_cast<const void*>(depth_frame_kernel<1, false>) : reinterpret_cast<const void*>(depth_frame_kernel<2, false>);
        (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        int nb = 0;
        const size_t probe = lds > 64 * 1024 ? 150 * 1024 : 64 * 1024;
        const hipError_t e = p.B == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, depth_frame_kernel<1, false>, DF_THREADS, probe)
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, depth_frame_kernel<2, false>, DF_THREADS, probe);
        (void)hipGetLastError();
        f = (e == hipSuccess && nb >= 1) ? 1 : -1;
    }
    return f > 0 ? G : 0;
}

int rst_launch_depth_frame(const DepthFrameParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.B <= 2, "depth_frame: batch %d (the persistent form serves 1 and 2)", p.B);
    RST_REQUIRE(p.h_all && p.tokens && p.gran && p.hist_solo && p.status && p.ld_h >= p.dep_q * p.E && p.tok_stride >= p.dep_q + 1, "depth_frame: null / short buffers");
    RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || (p.noise && p.noise_stride >= p.dep_q * p.top_k), "depth_frame: sampling needs dep_q * top_k noise values per row");
    const int G = rst_depth_frame_grid(p);
    RST_REQUIRE(G > 0, "depth_frame: unsupported shape (E=%d Hd=%d H=%d D=%d card=%d dep_q=%d L=%d) or no resident grid for it", p.E, p.Hd, p.H, p.D,
                p.card, p.dep_q, p.L);
    for (int l = 0; l < p.L; ++l) {
        RST_REQUIRE(p.in_proj[l] && p.out_proj[l] && p.norm1[l] && p.norm2[l], "depth_frame: layer %d pointers", l);
        for (int k = 0; k < p.dep_q; ++k) RST_REQUIRE(p.gate_in[l][k] && p.gate_out[l][k], "depth_frame: gating pointers of layer %d step %d", l, k);
    }
    for (int k = 0; k < p.dep_q; ++k) RST_REQUIRE(p.heads[k] && p.emb[k] && p.emb_rows[k] >= 1, "depth_frame: head / embedding of step %d", k);
    const size_t lds = df_lds_bytes(p);
    // every polled word (both granule sets) starts at zero in EVERY launch (a memset node when captured): epochs count from 1
    if (hipMemsetAsync(p.gran, 0, (size_t)rst_depth_frame_workspace_granules(p.B, p.E, p.Hd, p.card) * 16, stream) != hipSuccess) {
        rst_set_error("depth_frame: workspace memset failed");
        return RST_ERR_LAUNCH;
    }
    auto go = [&](auto kern, int grid) {
        static RstOncePerDevice attr_once;       // one flag per kernel instance (the lambda is instantiated per `kern` type)
        if (attr_once.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(DF_THREADS), lds, stream, p);
    };
    // the persistent launch, then its one-workgroup repair launch (a no-op unless a hand-off timed out: persist.h)
    if (p.B == 1) { go(depth_frame_kernel<1, false>, G); go(depth_frame_kernel<1, true>, 1); }
    else { go(depth_frame_kernel<2, false>, G); go(depth_frame_kernel<2, true>, 1); }
    return rst_check_launch("depth_frame");
}
