// Internal launcher interface between the C-ABI (capi.hip) and the kernel files.
#pragma once
#include "rst_common.h"

// ---- gemm_win.hip -----------------------------------------------------------------------------
struct GemmWinParams {
    const float* x;      // [B][T_in][C] channels-last activations
    const float* hist;   // [B][P][C] rows preceding x (streaming state) or nullptr
    const float* w;      // [N][K] packed weights, K contiguous, K index = tap*C + c
    const float* bias;   // [N] or nullptr
    const float* res;    // residual with the layout of y, or nullptr
    const float* scale;  // [N] LayerScale (y = res + scale*acc) or nullptr
    float* y;            // [B*T_out][ldy]
    int B, T_in, T_out;
    int C;               // floats per input time step
    int K, N;
    int S;               // input time steps advanced per output row
    int P;               // left padding in time steps
    int pad_mode;        // 0: zeros, 1: replicate first / last step (left side ignored when hist != nullptr)
    long x_bstride;      // floats between consecutive batches of x
    int ldy;             // floats between consecutive output rows (>= N)
    int act_in;          // 0: none, 1: ELU applied to A on load
    int act_out;         // 0: none, 1: exact GELU (before the residual), 2: ELU (applied last, after the residual)
    int split_k;         // > 1: K split over gridDim.y workgroups (M <= 4096), partials in ws, one counter per tile
    float* ws;           // [split_k][M][N]
    unsigned* counters;  // [rst_gemm_split_tiles_impl(M, N)], zero before the first launch (self re-arming)
    const short* w3;     // optional: w as three bf16 planes in staging order (rst_launch_gemm_win_b3_pack) -- large-M launches then run
                         // on the bf16 matrix instruction (six products per fp32 product, fp32 accuracy); nullptr: f32 instruction
};
int rst_launch_gemm_win(const GemmWinParams& p, hipStream_t stream);
bool rst_gemm_win_b3_shape_ok(int B, int T_in, int T_out, int C, int K, int N, int pad_mode, long x_bstride, bool has_hist);
long rst_gemm_win_b3_weight_elems_impl(int N, int K);
int rst_launch_gemm_win_b3_pack(const float* w, unsigned short* w3, int N, int K, hipStream_t stream);
int rst_gemm_split_plan_impl(long M, int N, int K);
int rst_gemm_split_tiles_impl(long M, int N);

// ---- skinny_f32.hip: few-row (M <= 128) fp32 GEMM of the codec streaming steps ---------------------------------------
struct SkinnyF32PackParams {      // the A operand of GemmWinParams, gathered + packed
    const float* x;
    const float* hist;
    float* xp;                    // [ceil(M/32)][Kp/8][64][4]
    int B, T_in, T_out, C, K, Kp, S, P, pad_mode, act_in;
    long x_bstride;
};
int rst_launch_skinny_f32_pack_ln(const float* x, const float* gamma, const float* beta, float* xp, int M, int K, float eps, hipStream_t stream);
struct SkinnyF32Params {
    const float* xp;              // packed activation windows (nullptr: the rows come row-major, below)
    const float* xr;              // row-major rows x [M][ldx] of a plain linear (Kp = K, K % 8 == 0): no packing launch in front of the GEMM
    int ldx;
    const float* wp;              // packed weights [ceil(N/32)][Kp/8][64][4]
    const float* bias;            // [N] or nullptr
    const float* res;             // [M][ldy] or nullptr
    const float* scale;           // [N] or nullptr
    float* y;                     // [M][ldy]
    int M, N, Kp, ldy, act_out;   // act_out as GemmWinParams
    int Np_out;                   // != 0 (= N, N % 8 == 0): y is written in the packed operand order of the NEXT few-row GEMM ([ceil(M/32)*32][N])
    int split_k;                  // > 1: K split over gridDim.y workgroups (rst_skinny_f32_split_plan_impl), partials in ws
    float* ws;                    // [split_k][M][N]
    unsigned* counters;           // [ceil(N/32)], zero before the first launch (self re-arming)
};
int rst_skinny_f32_split_plan_impl(int M, int N, int K);
int rst_launch_skinny_f32_pack_weight(const float* w, float* wp, int N, int K, hipStream_t stream);
int rst_launch_skinny_f32_pack_win(const SkinnyF32PackParams& p, hipStream_t stream);
int rst_launch_gemm_skinny_f32(const SkinnyF32Params& p, hipStream_t stream);

// ---- codec_attn.hip -------------------------------------------------------------------------------
struct AttnStepParams {
    const float* qkv;             // [B][T][3][H][D]: the in-projection's output of the T new steps
    float* k;                     // rings [B][H][cap][D]
    float* v;
    float* out;                   // [B][T][H*D], or (out_rows) [out_rows][H*D] in the packed operand order of the few-row GEMM
    const long* pos_dev;          // position of the first new step
    int B, T, H, D, cap, context, rope;
    int out_rows;                 // 0: row-major result; else B*T rounded up to 32 (64 above 32, 128 above 64: rst_skinny_f32_pack_win's rows)
    float rope_coef;
};
int rst_attn_step_supported_impl(int T, int D, int cap);
int rst_launch_attn_step(const AttnStepParams& p, hipStream_t stream);

// ---- resblock.hip ---------------------------------------------------------------------------------
struct ResblockParams {
    const float* x;     // [B][T][C] block input, or (pre) the mono audio [B][T]
    const float* hist;  // [B][Kw-1][C] streaming history of x (plain variant only) or nullptr
    const float* w0;    // pre : first conv weight [C][K0], bias b0 [C]
    const float* b0;
    const float* w1;    // [H][Kw*C] (tap-major K), b1 [H]
    const float* b1;
    const float* w2;    // [C][H], b2 [C]
    const float* b2;
    const float* wf;    // post: last conv weight [Kf][C], bias bf [1]
    const float* bf;
    float* y;           // [B][T][C], or (post) the mono waveform [B][T]
    int B, T, C, H, Kw, K0, Kf;
    int pre, post;
    int elu_out;        // apply ELU to y before the store (the only consumer is an ELU -> conv), plain / pre variants
};
bool rst_resblock_supported(int C, int H, int Kw, int pre, int post, int K0, int Kf);
int rst_launch_resblock(const ResblockParams& p, hipStream_t stream);

// ---- resblock_b3.hip: the same block on the bf16 matrix instruction (three-plane operands), batched encode / decode only
struct ResblockB3Params {
    const float* x;             // [B][T][C] block input, or (pre) the mono audio [B][T]
    const unsigned short* wp;   // packed weight planes (rst_launch_resblock_b3_pack)
    const float* b0;            // pre: bias of the first convolution [C]
    const float* b1;            // [H]
    const float* b2;            // [C]
    const float* wf;            // post: last conv weight [Kf][C] (fp32), bias bf [1]
    const float* bf;
    float* y;                   // [B][T][C], or (post) the mono waveform [B][T]
    int B, T, C, H, Kw, K0, Kf;
    int pre, post, elu_out;
};
bool rst_resblock_b3_supported(int C, int H, int Kw, int pre, int post, int K0, int Kf);
long rst_resblock_b3_weight_elems(int C);
int rst_launch_resblock_b3_pack(const float* w0, const float* w1, const float* w2, unsigned short* out, int C, int H, int Kw, int K0,
                                hipStream_t stream);
int rst_launch_resblock_b3(const ResblockB3Params& p, hipStream_t stream);

// ---- norm_elt.hip -----------------------------------------------------------------------------
int rst_launch_layernorm(const float* x, const float* gamma, const float* beta, float* y, long rows, int D,
                         float eps, hipStream_t stream);
int rst_launch_transpose(const float* x, float* y, int B, int R, int Ccols, hipStream_t stream);  // [B][R][C]->[B][C][R]
int rst_launch_convtr_depthwise(const float* x, const float* hist, const float* w, float* y, int B, int T_in, int C,
                                int K, int S, hipStream_t stream);
// hist_out [P_out rows] = last P_out rows of concat(hist_in [P_in rows], x [T_in rows]); no aliasing
int rst_launch_hist_update(const float* x, const float* hist_in, float* hist_out, int B, int T_in, int P_in, int P_out,
                           int C, hipStream_t stream);

#define RST_HIST_BATCH_MAX 32
struct HistBatchParams {          // in-place rolls: hist[i] [B][P][C] <- last P rows of concat(hist[i], x[i] [B][T_in][C])
    const float* x[RST_HIST_BATCH_MAX];
    float* hist[RST_HIST_BATCH_MAX];
    int T_in[RST_HIST_BATCH_MAX], P[RST_HIST_BATCH_MAX], C[RST_HIST_BATCH_MAX];
    int n, B;
};
int rst_launch_hist_update_batch(const HistBatchParams& p, hipStream_t stream);

int rst_launch_act(const float* x, float* y, long n, int act, hipStream_t stream);  // 1: ELU, 2: GELU
int rst_launch_mask_tail(float* x, const int* lengths, int B, int T, int C, int mode, hipStream_t stream);

// ---- attention.hip ----------------------------------------------------------------------------
struct RopeSplitParams {
    const float* qkv;   // [B][T][3][H][D]
    float* q;           // [B][H][T][D]
    float* k;           // [B][H][cap][D]
    float* v;           // [B][H][cap][D]
    const long* pos_dev;  // optional device scalar: position of the first new step (overrides pos0)
    long pos0;          // position of the first new step
    int B, T, H, D, cap;
    int ring;           // 0: k/v slot = t (cap == T), 1: slot = (pos0 + t) % cap
    int rope;           // 0: none, 1: interleaved pairs (modules/rope.py)
    float rope_coef;    // -ln(max_period) * 2 / D
};
int rst_launch_rope_split(const RopeSplitParams& p, hipStream_t stream);

struct AttentionParams {
    const float* q;     // [B][H][T][D]
    const float* k;     // [B][H][cap][D]
    const float* v;     // [B][H][cap][D]
    float* out;         // [B][T][H*D]
    const long* pos_dev;  // optional device scalar, as above
    long pos0;          // position of query 0
    int B, T, H, D, cap;
    int ring;           // 0: slot s holds position s; 1: ring cache, end_offset = pos0 + T (RingKVCache.complete)
    int context;        // <= 0: unlimited
    // fused-QKV form (whole-utterance pass, ring == 0, pos0 == 0, cap == T): q / k / v are read in place from the in-projection's output
    // [B][T][3][H][D] (q = qkv, k = qkv + H*D, v = qkv + 2*H*D; row stride 3*H*D) and rotated on their way in by `rope_tab`
    // ([T][D]: (cos, sin) of pair i of position t at [t][2i], [t][2i+1]; nullptr = no rotation) -- no rope_split launch, no q/k/v copies
    int row_stride;     // floats between consecutive positions of q / k / v (D for the split layout)
    const float* rope_tab;
};
int rst_launch_attention(const AttentionParams& p, hipStream_t stream);
int rst_launch_rope_table(float* tab, int T, int D, float rope_coef, long pos0, hipStream_t stream);

// ---- rvq.hip ----------------------------------------------------------------------------------
// packed codebook for one level: [D/8][n_codes][2][4] floats followed by nothing; e2: [n_codes]
int rst_launch_rvq_pack(const float* emb, float* packed, float* e2, int n_codes, int D, hipStream_t stream);
struct RvqSearchParams {
    const float* x;        // [M][ldx], group g reads columns [g*D, (g+1)*D)
    const float* emb;      // [L][n_codes][D] plain codebooks (residual update)
    const float* packed;   // [L][D/8][n_codes][2][4]
    const float* e2;       // [L][n_codes]
    long* codes;           // [B][L][F] int64, M = B*F
    float* dist;           // optional [L][M] winning score (debug / tests) or nullptr
    int M, F, ldx, D, n_codes, L;
    int n_groups;          // 1 or 2
    int group_begin[2];    // first level of each group
    int group_count[2];
};
int rst_launch_rvq_search(const RvqSearchParams& p, hipStream_t stream);
// few-frame variant: codes spread over workgroups, one launch per level; keys [L][M] uint64, all-ones before the first call
int rst_launch_rvq_search_small(const RvqSearchParams& p, unsigned long long* keys, hipStream_t stream);
int rst_rvq_chain_slices(int n_codes);
int rst_rvq_chain_supported_impl(int M, int n_codes, int L, int D, int n_groups);
int rst_launch_rvq_search_chain(const RvqSearchParams& p, unsigned long long* slots, unsigned* status, hipStream_t stream);
struct RvqGatherParams {
    const long* codes;     // [B][L][F]
    const float* emb;      // [L][n_codes][D]
    float* out;            // [M][n_groups*D]
    int M, F, D, n_codes, L, n_groups;
    int group_begin[2];
    int group_count[2];
};
int rst_launch_rvq_gather(const RvqGatherParams& p, hipStream_t stream);

// ---- lm_step.hip / lm_attn.hip / lm_sample.hip / lm_skinny.hip ------------------------------------------------------------------------------
struct GemvParams {
    const float* x;             // [B][ldx] fp32 activations (prologue 2: [B][2K] = [u ; v])
    const float* alpha;         // prologue 1: RMSNorm gain [K]; prologue 3: LayerNorm gamma [K]
    const float* beta;          // prologue 3: LayerNorm beta [K]
    const void* w;              // [N][K] bf16 (w_f32 == 0) or fp32 (w_f32 == 1)
    const float* res;           // optional [B][ldy]
    const float* bias;          // optional [N]
    const float* scale;         // optional [N]: LayerScale applied before the residual
    float* y;                   // [B][ldy]
    int B, N, K, ldx, ldy;
    int prologue;               // 0 none, 1 RMSNorm, 2 SiLU gate, 3 LayerNorm
    int act_out;                // 0 none, 1 exact GELU (after the bias)
    int w_f32;
    int gate_out;               // bf16, N even, no res / scale / act: y[b][q] = silu(row q) * (row N/2 + q), q < N/2 (stacked gated layer)
    float eps;
    // prologue 4 -- attention over a short un-rotated ring (the depth transformer): x = the qkv row [B][ldx] = [q | k | v] of
    // at_H heads of at_D dims (K = at_H * at_D); P(x) = softmax(q.K^T / sqrt(D)) V over the ring slots visible at *at_pos plus
    // the new step itself; workgroup 0 appends the new k / v to slot *at_pos % at_cap of at_k / at_v [B][H][cap][D].
    float* at_k;
    float* at_v;
    const long* at_pos;
    int at_H, at_D, at_cap, at_context;
    // prologue 5 -- embedding + RMSNorm (first layer of a depth step): x_in = x[b] (fp32, e.g. depformer_in[k](h)) +
    // em_table[em_tokens[b * em_tok_stride + em_col]] (bf16 [em_rows][K]; id -1 = zero row, other ids clamped into the
    // table); P(x) = RMSNorm(x_in) with alpha / eps; workgroup 0 stores x_in to em_x_out [B][K] (the layer's residual).
    const unsigned short* em_table;
    const long* em_tokens;
    float* em_x_out;
    int em_tok_stride, em_col, em_rows;
};
int rst_launch_gemv(const GemvParams& p, hipStream_t stream);

#define RST_MAX_TABLES 24
struct EmbedSumParams {
    const long* tokens;                          // [B][tok_stride]
    const unsigned short* tables[RST_MAX_TABLES];  // bf16 [rows][D]
    int tok_index[RST_MAX_TABLES];               // token column feeding table i
    int rows[RST_MAX_TABLES];                    // rows of table i (ids are clamped into it; 0 = unknown: not checked)
    const float* add;                            // optional fp32 [B][add_stride >= D] added first
    float* out;                                  // [B][D]
    int B, D, n_tables, tok_stride, add_stride;
};
int rst_launch_embed_sum(const EmbedSumParams& p, hipStream_t stream);
int rst_launch_rmsnorm(const float* x, const float* alpha, float* y, long rows, int D, float eps, hipStream_t stream);

struct LmRopeAppendParams {
    const float* qkv;     // [B][T][ldqkv], T new steps: [q (H*D) | k (G*D) | v (G*D)]
    float* q;             // [B][H][T][D] rotated queries
    float* k;             // [B][G][cap][D]
    float* v;
    const long* pos_dev;  // position of the first new step (device scalar; == steps already in the ring)
    int B, T, H, G, D, cap, ldqkv, rope, rope_dims;
    float rope_coef;
};
int rst_launch_lm_rope_append(const LmRopeAppendParams& p, hipStream_t stream);

struct LmAttnParams {
    const float* qkv;     // [B][ldqkv]: [q (H*D) | k (G*D) | v (G*D)] of the new step (un-rotated); or nullptr with q_pre
    const float* q_pre;   // optional [B][H][T][D]: rotated queries of T new steps whose keys are already in the ring
    int T;
    void* k;              // [B][G][cap][D] ring (the new step is appended): fp32, or bf16 when kv_bf16
    void* v;
    int kv_bf16;
    float* ws;            // [B][H][splits][D+2] workspace: (m, l, o[D]) per split (splits > 1)
    unsigned* counters;   // [B][H] arrival counters, zero before the first launch (re-armed by the kernel)
    float* out;           // [B][H*D]
    unsigned short* out_packed;   // optional instead of out: bf16 hi / lo planes in skinny-GEMM operand order, K = H*D (T = 1 only)
    long out_plane;       // elements per plane of out_packed (= ceil(B/32)*32 * H*D); pad rows are never written
    const long* pos_dev;  // position of the new step
    int B, H, D, cap, context, splits, ldqkv, rope;
    float rope_coef;
    int G;                // key/value heads (grouped-query attention: query head h reads kv head h / (H/G)); G == H for MHA
    int rope_dims;        // leading head dims that rotate (pairs (2i, 2i+1), i < rope_dims/2); D = all
    const float* rope_cs; // optional [D/2][2] (cos, sin) of this step's rotation (rst_launch_lm_rope_table): replaces the in-kernel trig
};
int rst_launch_lm_attn(const LmAttnParams& p, hipStream_t stream);
// (cos, sin) of angle_i = exp(i * rope_coef) * pos for i < D/2 (identity beyond rope_dims/2): the rotation of ONE step, computed once
// per frame with the arithmetic of the attention kernels and read by every layer's launch
int rst_launch_lm_rope_table(const long* pos_dev, float* out, int D, int rope_dims, float rope_coef, hipStream_t stream);

struct LmSampleParams {
    const float* logits;  // [B][ld]
    const float* noise;   // [B][noise_stride] Exp(1) draws (sampling only)
    long* tokens;         // tokens[b * tok_stride]
    int B, V, ld, top_k, noise_stride, tok_stride, use_sampling;
    float temp;
    int v_limit;               // sampling only: ids >= v_limit are never drawn (probabilities blanked AFTER the softmax); 0 = V
    const int* v_limit_dev;    // optional device scalar overriding v_limit (lets one captured graph serve changing limits)
    float top_p;               // > 0: nucleus sampling (sample_top_p) instead of top-k; noise then holds V values per row
    void* ws;                  // scratch of rst_lm_sample_workspace_bytes_impl bytes: chunk records + candidates of the two-level form
    long ws_bytes;             // for V > 32768, the sort buffer of top_p; NULL: one-level kernels only (top_p then refused)
};
long rst_lm_sample_workspace_bytes_impl(int B, int V, int top_k, int top_p_mode);
int rst_launch_lm_sample(const LmSampleParams& p, hipStream_t stream);

// ---- lm_depth.hip: the depth phase of a frame as one persistent launch
#define RST_DEPTH_MAX_L 8
#define RST_DEPTH_MAX_Q 8
struct DepthFrameParams {
    const unsigned short* in_proj[RST_DEPTH_MAX_L];     // per layer bf16 [dep_q * 3E][E] (step-major slices, multi_linear)
    const unsigned short* out_proj[RST_DEPTH_MAX_L];    // per layer bf16 [dep_q * E][E]
    const float* norm1[RST_DEPTH_MAX_L];                // fp32 RMSNorm gains [E]
    const float* norm2[RST_DEPTH_MAX_L];
    const unsigned short* gate_in[RST_DEPTH_MAX_L][RST_DEPTH_MAX_Q];    // bf16 [2 * Hd][E]  (rows u then v)
    const unsigned short* gate_out[RST_DEPTH_MAX_L][RST_DEPTH_MAX_Q];   // bf16 [E][Hd]
    const unsigned short* heads[RST_DEPTH_MAX_Q];       // bf16 [card][E]
    const float* head_bias[RST_DEPTH_MAX_Q];            // optional fp32 [card]
    const unsigned short* emb[RST_DEPTH_MAX_Q];         // embedding table of step k's input token, bf16 [emb_rows[k]][E]
    int emb_rows[RST_DEPTH_MAX_Q];
    const float* h_all;         // fp32 [B][ld_h]: columns [k * E, (k + 1) * E) = depformer_in[k](transformer_out)
    long* tokens;               // int64 [B][tok_stride]: column 0 = the text token (input), column k + 1 = token sampled at step k
    const float* noise;         // fp32 [B][noise_stride] Exp(1): step k uses columns [k * top_k, (k + 1) * top_k)   (sampling only)
    const int* v_limit;         // optional device int [dep_q]: ids >= v_limit[k] are never drawn at step k
    unsigned long long* gran;   // granule workspace: 2 x rst_depth_frame_workspace_granules(B, E, Hd, card) 8-byte words (persistent + repair launch)
    float* hist_solo;           // KV history of the repair launch: RST_DEPTH_MAX_L * RST_DEPTH_MAX_Q * B * 2 * E floats
    unsigned* status;           // 4 device words: [0] time-out codes of the frame in flight (cleared by the repair launch), [1] frames
                                // repaired so far, [2] OR of the codes of every repaired frame, [3] reserved
    int B, E, H, D, Hd, card, dep_q, L, ld_h, tok_stride, noise_stride, top_k, use_sampling, context;
    int ring_cap;               // capacity of the (virtual) KV ring: dep_q for LMGen, dep_q + 1 where the caller sized it so (no hidden slot)
    float eps, temp;
};
long rst_depth_frame_workspace_granules(int B, int E, int Hd, int card);
long rst_depth_frame_workspace_bytes_impl(int B, int E, int Hd, int card);
int rst_depth_frame_grid(const DepthFrameParams& p);       // workgroups of the persistent launch, 0 = shape not served
int rst_launch_depth_frame(const DepthFrameParams& p, hipStream_t stream);

// ---- lm_temporal.hip: the temporal transformer of one batch-1 LM step (all layers) as one persistent launch
#define RST_TEMPORAL_MAX_L 40
struct TemporalFrameParams {
    // device table of device pointers [8][L]: row 0 in_proj (bf16 [3E][E]), 1 out_proj (bf16 [E][E]), 2 gate_in (bf16 [2 Hd][E], rows u then
    // v), 3 gate_out (bf16 [E][Hd]), 4 norm1, 5 norm2 (fp32 RMSNorm gains [E]), 6 / 7 the K / V rings [1][H][cap][D] (bf16 when kv_bf16, else
    // fp32; the new step is appended).  In device memory so that a layer's pointers are one indexed scalar load (kernel-argument
    // arrays indexed at run time were compiled into per-op address arithmetic and branches in front of every block of the weight stream)
    const unsigned long long* tab;
    const float* x;                 // [E] input of the first layer
    float* y;                       // [E] output of the last layer
    const long* pos_dev;            // position of the new step (device scalar)
    const float* rope_cs;           // [D/2][2] (cos, sin) of the step's rotation (rst_launch_lm_rope_table), nullptr: no rotation
    unsigned long long* gran;       // 2 x rst_temporal_frame_workspace_granules(E, Hd, H, D) 8-byte words (persistent + repair launch)
    unsigned* status;               // 4 device words, as DepthFrameParams::status
    int E, H, D, Hd, L, cap, context, kv_bf16;
    int thin;                       // one block per weight wave in flight across a hand-off instead of two (set by the launcher)
    float eps;
};
long rst_temporal_frame_workspace_granules(int E, int Hd, int H, int D);
int rst_temporal_frame_grid(const TemporalFrameParams& p);      // workgroups of the persistent launch, 0 = shape not served
int rst_launch_temporal_frame(const TemporalFrameParams& p, hipStream_t stream);

// ---- codec_tr.hip: one streaming step of a Mimi transformer (all layers) as one persistent launch
#define RST_CTR_MAX_L 8
struct CodecTrParams {
    const float* in_proj[RST_CTR_MAX_L];    // fp32 [3E][E]
    const float* out_proj[RST_CTR_MAX_L];   // fp32 [E][E]
    const float* lin1[RST_CTR_MAX_L];       // fp32 [F][E]
    const float* lin2[RST_CTR_MAX_L];       // fp32 [E][F]
    const float* n1g[RST_CTR_MAX_L];        // LayerNorm gamma / beta [E]
    const float* n1b[RST_CTR_MAX_L];
    const float* n2g[RST_CTR_MAX_L];
    const float* n2b[RST_CTR_MAX_L];
    const float* ls1[RST_CTR_MAX_L];        // LayerScale [E] or nullptr
    const float* ls2[RST_CTR_MAX_L];
    float* kc[RST_CTR_MAX_L];               // KV rings [B][H][cap][D] (the new steps are appended)
    float* vc[RST_CTR_MAX_L];
    const float* x;                         // [B][T][E]
    float* y;                               // [B][T][E]
    const long* pos_dev;                    // position of the first new step (device scalar)
    unsigned long long* gran;               // 2 x rst_codec_tr_workspace_granules(B * T, E, F) 8-byte words (persistent + repair launch)
    unsigned* status;                       // 4 device words, as DepthFrameParams::status
    int B, T, E, H, D, F, L, cap, context, rope;
    float rope_coef, eps;
};
long rst_codec_tr_workspace_granules(int R, int E, int F);
int rst_codec_tr_grid(int B, int T, int E, int H, int F, int L, int cap);      // workgroups of the persistent launch, 0 = shape not served
int rst_launch_codec_tr(const CodecTrParams& p, hipStream_t stream);

// ---- lm_ring.hip: LMGen's token ring / delay pattern
struct LmRingParams {
    long* cache;                // [B][K][CT] int64
    const long* user;           // begin: [B][Ki] tokens of the user streams (codebooks first_user .. first_user + Ki - 1)
    const long* initial;        // begin: [K] initial token per codebook
    const int* delays;          // [K] device
    long* offset_dev;           // int64 scalar: frames stepped so far (commit increments it)
    long* input_out;            // begin: [B][K] the model input of this step
    const long* tokens;         // commit: [B][n_out] generated (text, audio_0 .. audio_{dep_q-1})
    long* out;                  // commit: [B][n_out] delay-aligned output
    int B, K, CT, Ki, first_user, n_out, max_delay;
};
int rst_launch_lm_ring_begin(const LmRingParams& p, hipStream_t stream);
int rst_launch_lm_ring_commit(const LmRingParams& p, hipStream_t stream);

struct SkinnyParams {
    const unsigned short* xp;   // packed activations [2][ceil(B/32)][K/16][64][8] bf16 (hi plane, lo plane)
    const unsigned short* w;    // packed weights [ceil(N/32)][K/16][64][8] bf16
    const float* res;           // optional [B][ldy]
    const float* bias;          // optional [N]
    float* y;                   // [B][ldy]
    int B, N, K, ldy;
    unsigned short* gate_out;   // optional instead of y (weights packed with interleave): silu(u) * v as packed hi / lo planes, K = N/2
    long gate_plane;            // elements per plane of gate_out
    int split_k;                // > 1: K also split over gridDim.y workgroups (rst_skinny_bf16_split_plan_impl), partials in ws
    float* ws;                  // [split_k][ceil(B/32)*32][N]
    unsigned* counters;         // [ceil(N/32)], zero before the first launch (self re-arming)
    // fp32-input form (xp == nullptr): the launch forms its own operand from x -- no activation-packing launch in front of it
    const float* xf;            // [B][ldx] fp32
    const float* alpha;         // xmode 1: RMSNorm gains [K]
    int ldx, xmode;             // 0: x as is; 1: RMSNorm (x * alpha in the operand, 1 / rms applied to the accumulators)
    float eps;
};
int rst_launch_gemm_skinny(const SkinnyParams& p, hipStream_t stream);
int rst_skinny_bf16_split_plan_impl(int B, int N, int K);
struct SkinnyFp8Params {
    const unsigned char* xp;    // packed fp8 activations [ceil(B/32)][K/32][64][16]
    const float* xscale;        // [ceil(B/32)*32] per-row scales of the activations
    const unsigned char* wp;    // packed fp8 weights [ceil(N/32)][K/32][64][16]
    const float* wscale;        // [ceil(N/32)*32] per-row scales of the weights
    const float* res;           // optional [B][ldy]
    const float* bias;          // optional [N]
    float* y;                   // [B][ldy]
    int B, N, K, ldy;
};
int rst_launch_gemm_skinny_fp8(const SkinnyFp8Params& p, hipStream_t stream);
int rst_launch_skinny_pack_weight_fp8(const unsigned short* w, unsigned char* wp, float* scale, int N, int K, hipStream_t stream);
int rst_launch_skinny_pack_act_fp8(const float* x, const float* alpha, unsigned char* xp, float* xscale, int B, int K, int ldx, int mode,
                                   float eps, hipStream_t stream);
int rst_launch_skinny_pack_weight(const unsigned short* w, unsigned short* wp, int N, int K, int interleave, hipStream_t stream);
int rst_launch_skinny_pack_act(const float* x, const float* alpha, unsigned short* xp, int B, int K, int ldx, int mode, float eps,
                               hipStream_t stream);
