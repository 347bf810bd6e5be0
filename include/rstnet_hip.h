/*
 * librstnet_hip.so -- C ABI of the MI355X (gfx950) kernels behind RSTnet's real-time generation hot path.
 *
 * The reference (yangdongchao/RSTnet) has no FFI layer: its "operators" are the PyTorch ATen calls issued by the
 * Kyutai streaming modules.  Each entry point below replaces one of those call sites; the file:line given is the
 * reference interface it stands in for (paths relative to MLLM_v2/tools/tokenizer/MimiCodec/model/ unless noted).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer; the library never allocates, frees or synchronises (hipGraph-capturable);
 *   - the library is stateless: weights, activations, streaming history and KV rings are caller-owned;
 *   - activations are fp32, CHANNELS-LAST: [B][T][C] with C contiguous (the reference's [B][C][T] tensors are
 *     converted at the model boundary with rst_transpose_f32);  codes are int64 [B][K][F] as in the reference;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and re-entrant;
 *   - return value: 0 ok, <0 error (RST_ERR_*); rst_last_error() gives the message of the calling thread.
 */
#ifndef RSTNET_HIP_H
#define RSTNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RST_OK 0
#define RST_ERR_INVALID_ARG (-1)
#define RST_ERR_UNSUPPORTED (-2)
#define RST_ERR_LAUNCH (-3)

#define RST_ACT_NONE 0
#define RST_ACT_ELU 1   /* act_in : ELU(alpha=1) applied to the input on load (nn.ELU before every SEANet conv) */
#define RST_ACT_GELU 1  /* act_out: exact erf GELU (F.gelu, modules/transformer.py:551-569), applied before the residual */
#define RST_ACT_ELU_OUT 2 /* act_out: ELU applied last (after the residual): for outputs whose only consumer is ELU -> conv,
                             so that the consumer needs no act_in (one ELU per element instead of one per window tap) */
#define RST_PAD_ZERO 0
#define RST_PAD_REPLICATE 1

typedef void* rst_stream_t;

int rst_version(void);
const char* rst_last_error(void);
/* Build id: the first 16 hex digits of the SHA-256 of every source file the library was built from (csrc/Makefile), written to out
 * (n >= 32 bytes, NUL-terminated); returns its length.  Measurement files under profiles/ carry the id of the build they were taken
 * with (tools/profile_meta.py); bench.py quotes a committed kernel trace only for the library that produced it. */
int rst_build_id(char* out, int n);

/* Generic windowed GEMM:
 *   y[b*T_out + t][n] = epi( sum_{k<K} act_in(A(b,t,k)) * w[n][k] + bias[n] ),  A(b,t,k) = xflat_b[(t*S - P)*C + k]
 *   epi(v) = res ? res + (scale ? scale[n] : 1) * act_out(v) : act_out(v)
 * Elements before the start of a batch item come from hist ([B][P][C]) if given, else are zero / replicated;
 * elements past the end are zero.  All three named wrappers below are thin fronts for it.
 * split_k > 1 (B*T_out <= 4096 rows -- the per-frame streaming steps, whose few tiles leave the chip idle and walk K as a chain
 * of exposed load latencies): K is split over split_k workgroups per tile; ws [split_k][B*T_out][N] floats and counters
 * [rst_gemm_win_split_tiles(M, N)] uint32 (zeroed once by the caller, self re-arming) carry the deterministic in-launch
 * reduction.  rst_gemm_win_split_plan(M, N, K) returns the recommended split_k (1 = no split). */
int rst_gemm_win_split_plan(int64_t M, int N, int K);
int rst_gemm_win_split_tiles(int64_t M, int N);
int rst_gemm_win_f32(const float* x, const float* hist, const float* w, const float* bias, const float* res,
                     const float* scale, float* y, int B, int T_in, int T_out, int C, int K, int N, int S, int P,
                     int pad_mode, int64_t x_bstride, int ldy, int act_in, int act_out, int split_k, float* ws,
                     uint32_t* counters, rst_stream_t stream);

/* The same contraction on the bf16 matrix instruction at fp32 accuracy, for the large launches: every fp32 operand is the exact sum of
 * three bf16 numbers (hi + mid + lo, each rounded to nearest even from the exact remainder), and six of the nine cross products are
 * accumulated in fp32 (the three dropped are below 2^-23 |x||w|, one fp32 rounding of the product) -- 192 matrix-pipe cycles per
 * 32x32x16 block instead of the f32 instruction's 512.
 * Shapes served -- rst_gemm_win_b3_supported(...) != 0: B*T_out > 4096 rows, N > 64, K % 64 == 0, C % 16 == 0, zero padding
 * (pad_mode RST_PAD_ZERO), no history buffer, x_bstride % 4 == 0, activations and split weights each below 4 GB; all pointers 16-byte
 * aligned.  rst_gemm_win_b3_f32 FAILS (RST_ERR_INVALID_ARG, rst_last_error) on any other call -- it never falls back to the f32
 * instruction silently; callers route those launches to rst_gemm_win_f32 themselves.  Rows whose window reaches into the zero
 * padding or past the utterance, and ragged last tiles, are masked inside the kernel; w (fp32) is not read.
 * The caller splits the weights once: w3 = rst_gemm_win_b3_weight_elems(N, K) uint16, filled by rst_gemm_win_b3_pack_weight (K % 16
 * == 0 suffices for the packing; layout [8*ceil(N/256)][K/16][3][64][8] -- matrix-instruction operand order: blocks of 32 rows, lane
 * (n % 32) + 32 * (k % 16 / 8) of a (block, k-tile, plane) holds k % 8 .. + 7 of row n, so a wave-level load is one contiguous KB; rows past
 * N zero); activations are split inside the launch.
 * Domain of the fp32-accuracy claim: finite operands with |x|, |w| in {0} u [2^-110, 3.38e38].  Below 2^-110 the lo (then mid) plane
 * leaves bf16's normal range: the result stays within 2^-126 sum_k |w_k| (resp. |x_k|) absolute of the exact one -- flush-to-zero
 * class.  +-Inf / NaN operands (and finite ones beyond the largest bf16, 3.3895e38, whose hi plane rounds to Inf) make every output
 * they touch NaN, where the f32 instruction gives +-Inf or NaN: non-finite either way, not the same non-finite value.
 * Replaces the conv / linear bodies of AudioCodec/MimiCodec/modules/conv.py:178-252 for batched (non-streaming) encode / decode. */
int rst_gemm_win_b3_supported(int B, int T_in, int T_out, int C, int K, int N, int S, int P, int pad_mode, int64_t x_bstride, int has_hist);
int rst_gemm_win_b3_weight_elems(int N, int K);   /* -1: bad sizes */
int rst_gemm_win_b3_pack_weight(const float* w, uint16_t* w3, int N, int K, rst_stream_t stream);
int rst_gemm_win_b3_f32(const float* x, const float* hist, const float* w, const uint16_t* w3, const float* bias, const float* res,
                        const float* scale, float* y, int B, int T_in, int T_out, int C, int K, int N, int S, int P,
                        int pad_mode, int64_t x_bstride, int ldy, int act_in, int act_out, rst_stream_t stream);

/* The few-row form of rst_gemm_win_f32 (M = B*T_out <= 128: one streaming frame for up to 64 streams), three entry points:
 *   rst_skinny_f32_pack_weight: w [N][K] fp32 -> wp [ceil(N/32)*32][Kp], Kp = K rounded up to 8, in MFMA operand order
 *     ([tile of 32 rows][Kp/8][64 lanes = 32*(k%2) + row%32][4 floats: k = 8q + 2e + k%2]); once per weight.
 *   rst_skinny_f32_pack_win: gathers the activation windows A(b,t,k) of rst_gemm_win_f32 (history / zero / replicate padding,
 *     ELU on load) into xp [32 | 64 | 128 rows: M rounded up to one of these][Kp] in the same order (rows past M zero).
 *   rst_gemm_skinny_f32: y = epi(A w^T + bias) with the epilogue of rst_gemm_win_f32 (act_out, res, scale); one workgroup per
 *     32 output columns whose 8 waves split K and meet in LDS in a fixed order (deterministic, no cross-workgroup reduction);
 *     v_mfma_f32_32x32x2_f32 with k ascending per wave.  Every Conv1d / ConvTranspose1d / Linear of a streaming step
 *     (modules/streaming.py:216-303, modules/transformer.py:395-562) is weight-bandwidth bound and goes through here.
 *     split_k > 1 (rst_skinny_f32_split_plan(M, N, K); 1 = none): K is also split over split_k workgroups per column tile --
 *     N / 32 workgroups alone stream a 6-33 MB layer at well under 1 TB/s; ws [split_k][M][N] floats and counters
 *     [ceil(M/32)][ceil(N/32)] uint32 (zeroed once by the caller, self re-arming) carry the deterministic in-launch reduction.
 *     More than 32 rows: every 32-row tile of the batch is a workgroup of its own per column tile (same arithmetic per element, so
 *     the same bits as the one-workgroup form of rounds 2 - 5; the weights cross L2 once per row tile).
 *   rst_skinny_f32_pack_ln: the plain-linear case of the packing (rows x [M][K], no window) with nn.LayerNorm(K) applied on the way
 *     (gamma, beta, eps; rst_layernorm_f32's arithmetic, so the operand equals LayerNorm followed by the pack bit for bit): the
 *     norm1 / norm2 in front of in_proj / linear1 of a streamed transformer layer (modules/transformer.py:595-650) costs no launch. */
int rst_skinny_f32_pack_weight(const float* w, float* wp, int N, int K, rst_stream_t stream);
int rst_skinny_f32_pack_ln(const float* x, const float* gamma, const float* beta, float eps, float* xp, int M, int K, rst_stream_t stream);
int rst_skinny_f32_pack_win(const float* x, const float* hist, float* xp, int B, int T_in, int T_out, int C, int K, int S, int P,
                            int pad_mode, int64_t x_bstride, int act_in, rst_stream_t stream);
int rst_skinny_f32_split_plan(int M, int N, int K);
/*     y_packed != 0 (N % 8 == 0, no residual): y [ceil(M/32)*32][N] is written in the packed order of rst_skinny_f32_pack_win, i.e. as the
 *     operand `xp` of the NEXT few-row GEMM (linear1 -> GELU -> linear2 of a streamed layer: no packing launch between the two). */
int rst_gemm_skinny_f32(const float* xp, const float* wp, const float* bias, const float* res, const float* scale, float* y, int M,
                        int N, int K, int ldy, int act_out, int split_k, float* ws, uint32_t* counters, int y_packed, rst_stream_t stream);
/*   rst_linear_few_rows_f32 (round 6): the plain-linear case of the few-row GEMM WITHOUT a packing launch -- the rows x [M][ldx]
 *     (K % 8 == 0, ldx % 4 == 0, 16-byte aligned) are read row-major inside the GEMM, a lane picking its four k of every 8-k chunk:
 *     the result equals rst_skinny_f32_pack_win followed by rst_gemm_skinny_f32 bit for bit.  (A LayerNorm in front of the linear stays
 *     with rst_skinny_f32_pack_ln: applied inside the GEMM it measured slower than the launch it saves.)
 *     wp, bias, res, scale, act_out, split_k, ws, counters, y_packed as rst_gemm_skinny_f32. */
int rst_linear_few_rows_f32(const float* x, int ldx, const float* wp, const float* bias, const float* res, const float* scale, float* y,
                            int M, int N, int K, int ldy, int act_out, int split_k, float* ws, uint32_t* counters, int y_packed,
                            rst_stream_t stream);

/* Causal Conv1d.  Replaces F.conv1d in RawStreamingConv1d.forward (modules/streaming.py:216-244) together with the
 * padding logic of StreamingConv1d.forward (modules/conv.py:232-254).
 *   x [B][T_in][Cin];  w_packed [Cout][Kw_eff*Cin] with w_packed[co][tap*Cin + ci] = weight[co][ci][tap] (dilated taps
 *   zero-filled);  y [B][T_out][Cout];  left padding P = Kw_eff - stride steps (zeros / replicate / hist [B][P][Cin]).
 *   res (optional, layout of y) is added to the result (SEANetResnetBlock skip, modules/seanet.py:92-94). */
int rst_conv1d_causal_f32(const float* x, const float* hist, const float* w_packed, const float* bias,
                          const float* res, float* y, int B, int T_in, int T_out, int Cin, int Cout, int Kw_eff,
                          int stride, int pad_mode, int act_in, int act_out, rst_stream_t stream);

/* Causal ConvTranspose1d, right-trimmed (trim_right_ratio = 1).  Replaces F.conv_transpose1d in
 * RawStreamingConvTranspose1d.forward (modules/streaming.py:271-303) + the trim of StreamingConvTranspose1d.forward
 * (modules/conv.py:305-329).  With q = ceil(Kw/stride):
 *   w_packed [stride*Cout][q*Cin],  w_packed[j*Cout + co][i*Cin + ci] = weight[ci][co][j + (q-1-i)*stride] (0 if >= Kw)
 *   bias_tiled [stride*Cout] (bias repeated `stride` times) or NULL;   x [B][T_in][Cin] -> y [B][T_in*stride][Cout].
 *   hist [B][q-1][Cin] = the q-1 input steps preceding x (streaming) or NULL (zeros). */
int rst_convtr1d_causal_f32(const float* x, const float* hist, const float* w_packed, const float* bias_tiled,
                            float* y, int B, int T_in, int Cin, int Cout, int Kw, int stride, int act_in, int act_out,
                            rst_stream_t stream);

/* Fused SEANetResnetBlock.forward (modules/seanet.py:92-94): y = x + conv_k1(ELU(conv_kKw(ELU(x)))), dilation 1, hidden
 * H = C/2, one launch, hidden activation kept in LDS.   w1 [H][Kw*C] (tap-major, as rst_conv1d_causal_f32), w2 [C][H].
 * Optional end fusions:  w0 != NULL ("pre"):  x is the mono audio [B][T] and the block input is computed on the fly as
 *   the first encoder conv, Conv1d(1, C, K0) (encoder.model.0, modules/seanet.py:184-193), w0 [C][K0];
 * wf != NULL ("post"): the block output goes through ELU + the last decoder conv, Conv1d(C, 1, Kf)
 *   (decoder.model.14, modules/seanet.py:368-379), wf [Kf][C], bf [1], and y is the mono waveform [B][T].
 * elu_out != 0: y = ELU(y) (plain / pre variants) for a block whose only consumer is ELU -> conv.
 * hist [B][Kw-1][C] (plain variant only) = streaming history of x.  rst_seanet_resblock_supported() tells whether a
 * shape is covered (otherwise the caller composes the block from rst_conv1d_causal_f32 calls). */
int rst_seanet_resblock_supported(int C, int H, int Kw, int pre, int post, int K0, int Kf);
int rst_seanet_resblock_f32(const float* x, const float* hist, const float* w0, const float* b0, const float* w1,
                            const float* b1, const float* w2, const float* b2, const float* wf, const float* bf,
                            float* y, int B, int T, int C, int H, int Kw, int K0, int Kf, int elu_out,
                            rst_stream_t stream);

/* The same block on the bf16 matrix instruction at fp32 accuracy (three-plane operands, as rst_gemm_win_b3_f32) for the batched
 * (whole-utterance) encode / decode: C = 64 (plain, "pre" or "post") and C = 128 (plain), H = C / 2, Kw = 3, no streaming history.
 * Every activation is split into its planes once (input tile at staging, hidden activation in the first GEMM's epilogue); the first
 * convolution of "pre" runs on the matrix pipe as well.  rst_seanet_resblock_b3_supported(...) != 0 tells whether a call is served
 * (an utterance must span less than 4 GB); rst_seanet_resblock_b3_f32 fails on any other call -- no silent change of kernel.
 * The caller packs the weights once: wp = rst_seanet_resblock_b3_weight_elems(C) uint16, filled by rst_seanet_resblock_b3_pack from
 * w0 [C][K0] (or NULL), w1 [H][Kw*C] (tap-major, as rst_conv1d_causal_f32) and w2 [C][H]: the planes of every matrix in matrix-instruction
 * operand order (W2's hidden index permuted to the first GEMM's accumulator order for C = 64).  "pre" is selected by b0 != NULL (x is
 * then the mono audio and wp must have been packed with w0), "post" by wf != NULL (wf [Kf][C] and bf [1] stay fp32; y is the waveform).
 * Same numerics contract as rst_gemm_win_b3_f32.  Replaces modules/seanet.py:21-94 (+ :184-193 / :368-379 for pre / post). */
int rst_seanet_resblock_b3_supported(int B, int T, int C, int H, int Kw, int pre, int post, int K0, int Kf);
int rst_seanet_resblock_b3_weight_elems(int C);   /* -1: C not served */
int rst_seanet_resblock_b3_pack(const float* w0, const float* w1, const float* w2, uint16_t* wp, int C, int H, int Kw, int K0,
                                rst_stream_t stream);
int rst_seanet_resblock_b3_f32(const float* x, const uint16_t* wp, const float* b0, const float* b1, const float* b2, const float* wf,
                               const float* bf, float* y, int B, int T, int C, int H, int Kw, int K0, int Kf, int elu_out,
                               rst_stream_t stream);

/* y[M][N] = epi(x[M][K] * w[N][K]^T + bias): F.linear call sites of modules/transformer.py:395,421,562 and the 1x1
 * Conv1d projections of quantization/vq.py:88-96. */
int rst_linear_f32(const float* x, const float* w, const float* bias, const float* res, const float* scale, float* y,
                   int64_t M, int K, int N, int act_out, rst_stream_t stream);

/* rst_linear_f32 for M = B <= 4 rows (the streaming steps of the Mimi transformers: one or two 25 Hz positions per 80 ms
 * frame): a weight-streaming GEMV over the fp32 [N][K] matrix -- every CU pulls rows, no split-K hand-off.  Same epilogue
 * order as rst_linear_f32: y = res + scale[n] * act(bias[n] + LN(x) . w[n]), act_out 0 / RST_ACT_GELU.  K % 8 == 0.
 * ln_gamma / ln_beta [K] (both or neither): the nn.LayerNorm(K, ln_eps) that precedes the linear in a transformer layer
 * (norm1 -> in_proj, norm2 -> linear1, modules/transformer.py:540-569) runs as the prologue of this launch. */
int rst_gemv_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* w, const float* bias,
                 const float* res, const float* scale, float* y, int B, int N, int K, int act_out, rst_stream_t stream);

/* nn.LayerNorm(D, eps) over the last dim (modules/transformer.py:113-114). */
int rst_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int D, float eps,
                      rst_stream_t stream);

/* "b t (p h d) -> p b h t d" split + interleaved RoPE (modules/rope.py:11-68) + KV placement
 * (RingKVCache.complete index_copy_, modules/transformer.py:255-262).  qkv [B][T][3*H*D]; q [B][H][T][D];
 * k, v [B][H][cap][D].  ring = 0: slot = t;  ring = 1: slot = (pos + t) % cap.  pos = *pos_dev if given else pos0. */
int rst_rope_split_f32(const float* qkv, float* q, float* k, float* v, const int64_t* pos_dev, int64_t pos0, int B,
                       int T, int H, int D, int cap, int ring, int rope, float rope_coef, rst_stream_t stream);

/* Masked attention = F.scaled_dot_product_attention(q, k, v, attn_bias) of modules/transformer.py:404-416.
 * Mask: key position >= 0, 0 <= pos_q - pos_k (< context if context > 0).  ring = 1 reproduces the slot->position map of
 * RingKVCache.complete including its `delta <= 0` behaviour.  out [B][T][H*D]. */
int rst_attention_f32(const float* q, const float* k, const float* v, float* out, const int64_t* pos_dev, int64_t pos0,
                      int B, int T, int H, int D, int cap, int ring, int context, rst_stream_t stream);

/* The whole-utterance pass of the same attention WITHOUT the split launch: q / k / v are read in place from the in-projection's output
 * qkv [B][T][3][H][D] ("b t (p h d)", modules/transformer.py:376-388) and rotated on their way in by `rope_table` [T][D] -- (cos, sin) of
 * pair i of position t at [t][2i], [t][2i+1], filled once per (T, D, max_period) by rst_rope_table_f32 with the arithmetic of
 * modules/rope.py:37-62 (angle = exp(i * rope_coef) * (pos0 + t)); NULL = no rotation.  Positions 0 .. T-1, causal + `context` mask,
 * no ring (the streaming steps keep rst_rope_split_f32 + rst_attention_f32 / rst_attn_decode_multi_f32).  out [B][T][H*D]. */
int rst_rope_table_f32(float* table, int T, int D, float rope_coef, int64_t pos0, rst_stream_t stream);
int rst_attention_qkv_f32(const float* qkv, const float* rope_table, float* out, int B, int T, int H, int D, int context,
                          rst_stream_t stream);

/* Codebook preparation for rst_rvq_search_f32: packed [D/8][n_codes][2][4], e2[n_codes] = |e|^2 (k-ordered fmaf). */
int rst_rvq_pack_f32(const float* emb, float* packed, float* e2, int n_codes, int D, rst_stream_t stream);

/* Residual VQ nearest-codeword search, all levels fused: EuclideanCodebook._quantize + the residual loop of
 * ResidualVectorQuantization.encode (quantization/core_vq.py:179-185, 365-376).  x [M][ldx] holds the projected
 * latents of group g in columns [g*D, (g+1)*D); group g runs levels [group_begin[g], +group_count[g]).
 * codes [B][L][F] int64 with M = B*F.  dist (optional) [L][M] = winning score |e|^2 - 2 x.e.
 * keys == NULL: one launch, a workgroup per 32 frames runs all levels (thousands of frames).  keys != NULL ([L][M] uint64,
 * all-ones before the first call, re-armed by the call): the few-frame streaming form -- codes are spread over workgroups,
 * one launch per level, winners published with a 64-bit atomicMin; decisions are bit-identical to the fused form. */
int rst_rvq_search_f32(const float* x, const float* emb, const float* packed, const float* e2, int64_t* codes,
                       float* dist, uint64_t* keys, int M, int F, int ldx, int D, int n_codes, int L, int n_groups,
                       const int* group_begin, const int* group_count, rst_stream_t stream);

/* The few-frame streaming form as ONE launch for all levels (+ a one-workgroup finish launch): the workgroups of a (group, 32-frame
 * tile) -- one per 128 codes, all resident -- hand every level's decision over in-kernel (each publishes its best (score, index) key
 * per frame into its slot of `slots`, sweeps all slices' slots and takes the minimum; the residual stays in LDS).  Same k-ordered
 * scores, same tie rule: codes and `dist` bit-identical to rst_rvq_search_f32.  slots: rst_rvq_chain_slot_elems(M, n_codes, L) uint64,
 * all-ones before the first call, re-armed by every call.  status uint32[4] (zeroed once): [0] time-out code of a workgroup whose
 * peers were not resident (waits are bounded by the wall clock), [1] calls the finish launch had to recompute on its own (outputs are
 * right either way), [2] OR of their codes -- the convention of rst_depth_decode_frame.  Needs n_codes / 128 * n_groups * ceil(M / 32)
 * <= CUs.  Replaces the residual loop of SplitResidualVectorQuantizer.encode per streamed frame (moshi/models/compression.py:368-389). */
int rst_rvq_chain_slot_elems(int M, int n_codes, int L);   /* -1: bad sizes */
int rst_rvq_chain_supported(int M, int n_codes, int L, int D, int n_groups);   /* 1: the chain serves the shape on this device, 0: use rst_rvq_search_f32 */
int rst_rvq_search_chain_f32(const float* x, const float* emb, const float* packed, const float* e2, int64_t* codes, float* dist,
                             uint64_t* slots, uint32_t* status, int M, int F, int ldx, int D, int n_codes, int L, int n_groups,
                             const int* group_begin, const int* group_count, rst_stream_t stream);

/* Sum of codebook rows per group = ResidualVectorQuantization.decode (core_vq.py:378-384): out [M][n_groups*D]. */
int rst_rvq_gather_f32(const int64_t* codes, const float* emb, float* out, int M, int F, int D, int n_codes, int L,
                       int n_groups, const int* group_begin, const int* group_count, rst_stream_t stream);

/* Depth-wise causal ConvTranspose1d (ConvTrUpsample1d channel_wise, modules/resample.py:109-119): w [C][Kw]. */
int rst_convtr_depthwise_f32(const float* x, const float* hist, const float* w, float* y, int B, int T_in, int C,
                             int Kw, int stride, rst_stream_t stream);

/* Stand-alone elementwise activation (nn.ELU / F.gelu module calls that are not fused into a GEMM): act 1 = ELU, 2 = GELU. */
int rst_act_f32(const float* x, float* y, int64_t n, int act, rst_stream_t stream);

/* Layout adapter [B][R][C] -> [B][C][R] (reference [B,C,T] <-> channels-last). */
int rst_transpose_f32(const float* x, float* y, int B, int R, int C, rst_stream_t stream);

/* Ragged batches: rows t >= lengths[b] of x [B][T][C] (in place) become zero (mode 0) or copies of row lengths[b] - 1 (mode 1).
 * A zero-padded batch then reproduces what the reference computes for each utterance alone: its convolutions pad the END of
 * every layer's input themselves -- zeros in pad_for_conv1d (modules/conv.py:82-100), replicate in ConvDownsample1d
 * (modules/resample.py:14-65) -- so the rows past an utterance's length must hold exactly that when a strided layer reads them. */
int rst_mask_tail_f32(float* x, const int32_t* lengths, int B, int T, int C, int mode, rst_stream_t stream);

/* Streaming history roll: hist_out [P_out steps] = last P_out steps of concat(hist_in [P_in steps], x [T_in steps])
 * (the `previous` buffer of RawStreamingConv1d, modules/streaming.py:224-236; conv-transpose keeps its input history the
 * same way instead of the reference's `partial` output buffer).  hist_out == hist_in (in-place roll, what a graph-replayed
 * streaming step wants) is allowed when P_in == P_out and P_out * C <= 16384; otherwise the buffers must not overlap. */
int rst_hist_update_f32(const float* x, const float* hist_in, float* hist_out, int B, int T_in, int P_in, int P_out,
                        int C, rst_stream_t stream);

/* The in-place rolls of up to 32 histories in ONE launch: entry i rolls hist[i] [B][P[i]][C[i]] over x[i] [B][T_in[i]][C[i]]
 * (P[i] * C[i] <= 16384).  A streaming codec step ends with one of these per codec half instead of a roll launch behind every
 * convolution (`previous` of every RawStreamingConv1d / the input history of every RawStreamingConvTranspose1d,
 * modules/streaming.py:224-236,279-303).  x, hist, T_in, P, C are HOST arrays of length n. */
int rst_hist_update_batch_f32(const float* const* x, float* const* hist, const int* T_in, const int* P, const int* C, int n, int B,
                              rst_stream_t stream);

/* rst_codec_transformer_frame -- ONE streaming step of a Mimi transformer (ProjectedTransformer / StreamingTransformer with all its
 * StreamingTransformerLayers, modules/transformer.py:434-750, as MimiModel calls it per 80 ms frame: compression.py:368-419) as one
 * persistent launch, for B streams x T new positions with B * T <= 4 rows: per layer LayerNorm(eps) -> in_proj [3E][E] ->
 * interleaved RoPE on q, k at positions *pos_dev + t (modules/rope.py) -> append to the ring k_cache / v_cache [B][H][cap][D]
 * (slot (pos + t) % cap) -> attention of every new query over the ring with RingKVCache.complete's slot -> position map and the
 * causal / context mask (:254-278, 404-414) -> out_proj [E][E] -> x + scale1 * . -> LayerNorm -> linear1 [F][E] -> exact GELU ->
 * linear2 [E][F] -> x + scale2 * .   Tables are HOST arrays of L device pointers (scale1 / scale2 may be NULL: no LayerScale).
 * x, y fp32 [B][T][E]; workspace of rst_codec_transformer_workspace_bytes(B * T, E, F) bytes (zeroed by the call on `stream`);
 * status: 4 device words with the protocol of rst_depth_decode_frame (a timed-out hand-off is repaired in-stream by the
 * one-workgroup launch the call enqueues behind the persistent one; status[1] counts repaired steps).
 * rst_codec_transformer_supported: > 0 (the grid of the persistent launch) when the shape is served and one workgroup of its
 * footprint fits a CU (occupancy query), else 0 -- callers then run the launch-per-op layer loop. */
int rst_codec_transformer_workspace_bytes(int rows, int E, int F);
int rst_codec_transformer_supported(int B, int T, int E, int H, int F, int L, int cap);
int rst_codec_transformer_frame(const float* const* in_proj, const float* const* out_proj, const float* const* linear1,
                                const float* const* linear2, const float* const* norm1_w, const float* const* norm1_b,
                                const float* const* norm2_w, const float* const* norm2_b, const float* const* scale1,
                                const float* const* scale2, float* const* k_cache, float* const* v_cache, const float* x, float* y,
                                const int64_t* pos_dev, void* workspace, uint32_t* status, int B, int T, int E, int H, int F, int L,
                                int cap, int context, int rope, float rope_coef, float eps, rst_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * RQ-Transformer decode step (T = 1 per call, small batch).  bf16 weights, fp32 activations / accumulation.
 * Reference files below are relative to MLLM_v2/.
 * ------------------------------------------------------------------------------------------------------------------ */

/* y[b][n] = (res ? res[b][n] : 0) + sum_k P(x)[b][k] * w[n][k]  -- the F.linear call sites of one decode step:
 * in_proj / out_proj (modules/transformer.py:391-395,418-421, incl. the per-step slices of multi_linear :155-179),
 * gating linear_in / linear_out (modules/gating.py:12-22), depformer_in, text_linear, linears[k] (models/model.py:384,
 * 411-425), and of the litgpt backbone: LoRAQKVLinear / LoRALinear after merge (models/llama_streaming.py:113-143,368-371),
 * LLaMAMLP fc_1|fc_2 / proj (models/lit_model.py:399-403), lm_head.  w bf16 [N][K] row-major, K % 8 == 0, 1 <= B <= 4;
 * bias optional fp32 [N] (config.bias / lm_head_bias).  prologue P: 0 identity; 1 RMSNorm x*alpha*rsqrt(eps+mean(x^2))
 * (modules/transformer.py:34-46 eps 1e-8; lit_model.py:693-717 with weight as alpha); 2 SiLU gate: x is [B][2K] = [u ; v],
 * P(x) = silu(u)*v.  gate_out != 0 (w = [W_u ; W_v] stacked, N even, no res): the gate moves into the PRODUCER instead --
 * y[b][q] = silu(row q) * (row N/2 + q) for q < N/2 (each wave owns a (u, v) row pair), so the consumer reads N/2 values with
 * prologue 0.  Batch-1 RMSNorm layers with N*K >= 2^24, K <= 4096 run a schedule that issues the x loads before the weight
 * stream (same arithmetic). */
int rst_gemv_bf16_f32(const float* x, const float* alpha, const uint16_t* w, const float* res, const float* bias, float* y,
                      int B, int N, int K, int ldx, int ldy, int prologue, float eps, int gate_out, rst_stream_t stream);

/* Two fused forms of the batch <= 2 GEMV for the depth transformer ("depformer", models/model.py:392-428,564-597;
 * "codecformer", models/llama_streaming.py:727-749), whose 8 steps x 6 layers are a chain of ~5 us launches:
 *
 * rst_gemv_attn_bf16_f32 -- out_proj of a layer with the attention itself as the prologue (modules/transformer.py:376-423 for
 *   a ring of cap <= 8 slots and no rotary embedding, the depth transformer's setup): qkv [B][ldqkv] = [q | k | v] (H heads of
 *   D dims each, D a power of two), k_cache / v_cache [B][H][cap][D] fp32 rings, *pos_dev = position of the new step.  Every
 *   workgroup recomputes softmax(q K^T / sqrt(D)) V over the slots RingKVCache.complete makes visible (slot -> position map
 *   :254-278 incl. the `delta <= 0` slot) plus the new step, taken straight from the qkv row; workgroup 0 appends the new
 *   k / v at slot pos % cap.  y[b][n] = (res +) (bias +) sum_k attn[b][k] w[n][k], w bf16 [N][H*D].
 *
 * rst_gemv_embed_bf16_f32 -- in_proj of the FIRST layer of a depth step with the step's input as the prologue:
 *   x_in[b] = add[b] (fp32 row, ld_add apart: depformer_in[k](transformer_out), e.g. a column block of one up-front
 *   [B][dep_q * K] product) + table[tokens[b * tok_stride + tok_col]] (ScaledEmbedding, models/model.py:67-91: id -1 gives the
 *   zero row; other ids are clamped into [0, table_rows) -- the reference raises on an out-of-range id);
 *   y = (bias +) RMSNorm(x_in; alpha, eps) w^T, and x_in is stored to x_out [B][K] (the residual of the layer). */
int rst_gemv_attn_bf16_f32(const float* qkv, float* k_cache, float* v_cache, const int64_t* pos_dev, const uint16_t* w,
                           const float* res, const float* bias, float* y, int B, int N, int H, int D, int cap, int context, int ldqkv,
                           int ldy, rst_stream_t stream);
int rst_gemv_embed_bf16_f32(const float* add, const uint16_t* table, const int64_t* tokens, float* x_out, const float* alpha,
                            const uint16_t* w, const float* bias, float* y, int B, int N, int K, int ld_add, int ldy, int tok_stride,
                            int tok_col, int table_rows, float eps, rst_stream_t stream);

/* rst_depth_decode_frame -- the WHOLE depth phase of a frame as ONE persistent launch, batch 1 or 2:
 * LMGen.depformer_step (models/model.py:564-597: per codebook k forward_depformer :392-428 = depformer_in[k](h) + embedding of
 * the previous token -> the L-layer depth transformer in streaming mode with per-step weights (modules/transformer.py:155-179,
 * 376-423, 551-592; modules/gating.py:12-51) -> linears[k] -> sample_token (utils/sampling.py:85-105)), which the reference
 * wraps in one CUDA graph (:486); equally the codecformer loop of models/llama_streaming.py:727-749.  Tables are HOST arrays of
 * device pointers: in_proj[l] bf16 [dep_q * 3E][E] and out_proj[l] bf16 [dep_q * E][E] (step-major slices), norm1[l] / norm2[l]
 * fp32 [E], gate_in[l * dep_q + k] bf16 [2 * Hd][E], gate_out[l * dep_q + k] bf16 [E][Hd], heads[k] bf16 [card][E] (+ optional
 * head_bias[k] fp32 [card]), emb[k] bf16 [emb_rows[k]][E] = the table of step k's input token (text table for k = 0).
 * h_all fp32 [B][ld_h]: columns [k * E, (k + 1) * E) hold depformer_in[k](transformer_out) (one up-front GEMV).  tokens int64
 * [B][tok_stride]: column 0 is the text token (input), column k + 1 receives the token of step k.  noise fp32 [B][noise_stride]:
 * Exp(1) draws, step k reads columns [k * top_k, (k + 1) * top_k).  v_limit_dev: optional device int [dep_q], ids >= v_limit[k]
 * are never drawn at step k (the id blanking of sample_token_audio / _2048).  context <= 0: none (the depth transformer's).
 * The KV ring of the depth transformer (ring_cap >= dep_q slots: dep_q for LMGen -- RingKVCache.complete's positions then hide
 * step 0 at the last step, the `delta <= 0` slot -- or dep_q + 1 for a caller that sized its ring so) lives in the LDS of the launch; ops hand their output vectors over in `workspace` (rst_depth_frame_workspace_bytes bytes, 8-byte
 * {epoch, value} granules, zeroed by the call itself on `stream`).
 * Residency and what happens without it: the persistent launch is a grid of up to one workgroup per CU (sized so that every
 * workgroup owns rows in the all-to-all ops; refused -- rst_depth_frame_supported == 0 -- when the occupancy query says a CU
 * cannot take one), all of which must be resident at once.  A device shared with other work may not grant that: every spin is
 * bounded (0.1 s), a timed-out hand-off ORs a code into status[0] and the launch drains.  The call enqueues, right behind it, the
 * same kernel as ONE workgroup: it returns at once while status[0] == 0 and otherwise recomputes the frame alone (it depends on no
 * other workgroup, so it cannot time out), overwrites the tokens, increments status[1] (frames repaired), ORs the codes into
 * status[2] and clears status[0] -- as with the reference's depformer_step (models/model.py:564-597) wrong tokens never leave the
 * stream.  `status`: 4 device words, zero before the first call; hosts read status[1] now and then and move a device that keeps
 * repairing to the launch-per-op chain (a repaired frame costs the time-outs + ~30 ms).
 * rst_depth_frame_supported: the grid (> 0) when the shape is served, else 0. */
int rst_depth_frame_workspace_bytes(int B, int E, int Hd, int card);
int rst_depth_frame_supported(int B, int E, int H, int Hd, int card, int dep_q, int L, int top_k);
int rst_depth_decode_frame(const uint16_t* const* in_proj, const uint16_t* const* out_proj, const float* const* norm1,
                           const float* const* norm2, const uint16_t* const* gate_in, const uint16_t* const* gate_out,
                           const uint16_t* const* heads, const float* const* head_bias, const uint16_t* const* emb, const int* emb_rows,
                           const float* h_all, int64_t* tokens, const float* noise, const int* v_limit_dev, void* workspace,
                           uint32_t* status, int B, int E, int H, int Hd, int card, int dep_q, int L, int ld_h, int tok_stride,
                           int noise_stride, int top_k, int use_sampling, float temp, float eps, int context, int ring_cap,
                           rst_stream_t stream);

/* rst_temporal_decode_frame -- ALL layers of the temporal transformer of one batch-1 LM step as ONE persistent launch:
 * LMModel.forward_text's `self.transformer(input_)` at T = 1 in streaming mode (models/model.py:364-389), i.e. L x
 * StreamingTransformerLayer (modules/transformer.py:551-592: rms_norm_f32 -> StreamingMultiheadAttention :376-423 with RoPE
 * (modules/rope.py:37-62) and the RingKVCache of :211-278 -> + x; rms_norm_f32 -> ActivationGating (modules/gating.py:12-51) -> + x).
 * dev_tables: a DEVICE array [8][L] of device pointers (uint64), row 0 in_proj[l] bf16 [3E][E], 1 out_proj[l] bf16 [E][E], 2 gate_in[l]
 * bf16 [2 Hd][E] (rows u then v), 3 gate_out[l] bf16 [E][Hd], 4 norm1[l] / 5 norm2[l] fp32 [E], 6 k_cache[l] / 7 v_cache[l] the rings
 * [1][H][cap][E / H] (bf16 when kv_bf16, else fp32; the new step is appended at slot *pos_dev % cap) -- built once per session by the caller
 * (a layer's pointers are then one indexed scalar load inside the launch).  x fp32 [E] -> y fp32 [E] (y != x).  rope_cs: fp32 [D / 2][2]
 * (cos, sin) of the step's rotation (rst_lm_rope_table_f32) or NULL (no rotation).  context <= 0: none.  workspace:
 * rst_temporal_frame_workspace_bytes(E, Hd, H) bytes of 8-byte {epoch, value} granules, zeroed by the call on `stream`.
 * Weights stream continuously across op and layer boundaries (each weight wave keeps 32 KB in flight in registers, requested before the
 * hand-off that produces the op's input); the op boundaries are in-launch all-to-all hand-offs.  Residency, bounded waits, the
 * one-workgroup repair launch behind it and `status` follow rst_depth_decode_frame's protocol.
 * rst_temporal_frame_supported: the grid (> 0) when the shape is served (E <= 4096, head dim 64 / 128, <= 40 layers, the LDS footprint,
 * the occupancy query), else 0 -- callers then run the launch-per-op chain. */
int rst_temporal_frame_workspace_bytes(int E, int Hd, int H);
int rst_temporal_frame_supported(int E, int H, int Hd, int L, int cap, int kv_bf16);
int rst_temporal_decode_frame(const uint64_t* dev_tables, const float* x, float* y, const int64_t* pos_dev, const float* rope_cs, void* workspace,
                              uint32_t* status, int E, int H, int Hd, int L, int cap, int context, int kv_bf16, float eps, rst_stream_t stream);

/* The same contraction for 4 < B <= 64 on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), in three entry points.
 * Both operands are kept in the order the MFMA consumes them -- [tile of 32 rows][K/16 steps][64 lanes][8 bf16], lane =
 * 32 * ((k / 8) % 2) + row % 32 -- so that every wave-level load is one contiguous kilobyte:
 *   rst_skinny_pack_weight_bf16: w [N][K] row-major -> wp [ceil(N/32)*32][K] in that order (pad rows zero); once per weight.
 *     interleave_halves (gated layers, w = [W_u ; W_v], N % 32 == 0): tile t holds rows 16t.. of W_u then the same rows of W_v.
 *   rst_skinny_pack_act_f32: P(x) of one decode step, split into bf16 hi + lo planes (x = hi + lo to 2^-17: fp32-class
 *     accuracy against the fp32 oracle) -> xp [2][ceil(B/32)*32][K]; mode 0 identity, 1 RMSNorm (alpha, eps), 2 SiLU gate
 *     (x rows = [u ; v] of length 2K) -- the same prologues as rst_gemv_bf16_f32.  K % 16 == 0, ldx % 4 == 0.
 *   rst_gemm_skinny_bf16_f32: y[b][n] = (res +) (bias +) sum_k P(x)[b][k] w[n][k]: weights streamed from HBM exactly once,
 *     one workgroup per 32 (64, 128 for large N) weight rows whose 8 waves split K and meet in LDS in a fixed order
 *     (deterministic, no cross-workgroup reduction).  gate_out (optional, then y may be NULL; wp packed with
 *     interleave_halves, no residual): the epilogue applies silu(u) * v (gating_forward_kernel / LLaMAMLP) and writes the
 *     result as the packed operand [2][ceil(B/32)*32][N/2] of the next GEMM -- the gated activation never exists in fp32. */
int rst_skinny_pack_weight_bf16(const uint16_t* w, uint16_t* wp, int N, int K, int interleave_halves, rst_stream_t stream);
int rst_skinny_pack_act_f32(const float* x, const float* alpha, uint16_t* xp, int B, int K, int ldx, int mode, float eps,
                            rst_stream_t stream);
int rst_gemm_skinny_bf16_f32(const uint16_t* xp, const uint16_t* wp, const float* res, const float* bias, float* y, int B, int N,
                             int K, int ldy, uint16_t* gate_out, int split_k, float* ws, uint32_t* counters, rst_stream_t stream);
/* The same GEMM taking the fp32 activations x [B][ldx] directly (no rst_skinny_pack_act_f32 launch in front of it): every lane forms
 * the hi / lo operand of its (batch row, 8 k) piece in registers; mode 1 (RMSNorm) puts x * alpha into the operand and applies
 * 1 / sqrt(eps + mean(x^2)) to the accumulators (the factor commutes with the contraction); the rows are read coalesced and transposed
 * into operand order through a wave-private LDS tile.  No K split, K % 256 == 0, ldx % 4 == 0; meant for K <= 2048 (the depth
 * transformer and the GPT blocks, whose launches are latency chains: one launch less per linear). */
int rst_gemm_skinny_x32_bf16_f32(const float* x, const float* alpha, float eps, int mode, int ldx, const uint16_t* wp, const float* res,
                                 const float* bias, float* y, int B, int N, int K, int ldy, uint16_t* gate_out, rst_stream_t stream);
/* split_k = rst_skinny_bf16_split_plan(B, N, K) (> 1 only for K >= 2048): K is also split over workgroups, each taking four (two
 * above 32 rows) adjacent column tiles, so that the packed activations are pulled through a CU's load path once per four weight
 * tiles and ~256-384 workgroups stream; ws [split_k][ceil(B/32)*32][N] fp32 and counters [ceil(N/32)] (zero before the first
 * launch, self re-arming) are caller-owned scratch; the last workgroup of a column group sums the partials in split order
 * (deterministic).  split_k <= 1 (ws / counters may be NULL): one workgroup per column tile(s) over all of K, as before. */
int rst_skinny_bf16_split_plan(int B, int N, int K);

/* Opt-in fp8 form of the three entry points above (BASELINE.json configs[4]: fp8 MFMA GEMMs on the temporal blocks at batch
 * 32): OCP e4m3 operands on v_mfma_f32_32x32x16_fp8_fp8, weights with one scale per row (amax / 448, quantised once),
 * activations with one dynamic scale per batch row (quantised by the prologue launch), y = scale_x[b] * scale_w[n] * acc.
 * Layout [tile of 32 rows][K/32][64 lanes][16 B = two 8-k steps]; wp N32*K bytes, wscale N32 floats, xp B32*K bytes, xscale
 * B32 floats (N32 / B32 = rounded up to 32).  K % 32 == 0, K <= 16384.  Half the streamed bytes of the bf16 path at fp8
 * accuracy (~1e-2 relative on logits): never selected implicitly. */
int rst_skinny_pack_weight_fp8(const uint16_t* w, uint8_t* wp, float* wscale, int N, int K, rst_stream_t stream);
int rst_skinny_pack_act_fp8(const float* x, const float* alpha, uint8_t* xp, float* xscale, int B, int K, int ldx, int mode, float eps,
                            rst_stream_t stream);
int rst_gemm_skinny_fp8_f32(const uint8_t* xp, const float* xscale, const uint8_t* wp, const float* wscale, const float* res,
                            const float* bias, float* y, int B, int N, int K, int ldy, rst_stream_t stream);

/* out[b] = (add ? add[b] : 0) + sum_i table_i[tokens[b][tok_index[i]]]: the ScaledEmbedding sums of
 * LMModel.forward_text / forward_depformer (models/model.py:67-91, 372-380, 413-419): id -1 -> zero row; tables bf16
 * [rows][D], summed in table order in fp32.  Ids outside a table are clamped into it (other negative ids -> row 0, ids >=
 * table_rows[i] -> the last row; the reference's F.embedding raises -- use LMGen(check=True) for that behaviour; table_rows
 * may be NULL: no upper check).  `tables` / `tok_index` / `table_rows` are HOST arrays (n_tables <= 24).  `add_stride`: floats between
 * the rows of `add` (>= D: a column block of a wider matrix, e.g. step k's slice of the stacked depformer_in product, is read in place). */
int rst_embed_sum_bf16(const int64_t* tokens, const uint16_t* const* tables, const int* tok_index, const int* table_rows, int n_tables,
                       const float* add, float* out, int B, int D, int tok_stride, int add_stride, rst_stream_t stream);

/* RMSNorm rows (rms_norm_f32, modules/transformer.py:34-46): out_norm of LMModel.forward_text. */
int rst_rmsnorm_f32(const float* x, const float* alpha, float* y, int64_t rows, int D, float eps, rst_stream_t stream);

/* T new steps (prompt prefill): split qkv [B][T][ldqkv] = [q (H*D) | k (G*D) | v (G*D)], rotate q and k (interleaved RoPE on
 * the leading rope_dims head dims, modules/rope.py; rotate-half checkpoints -- lit_model.py:560-573 -- are served by
 * permuting the q / k weight rows at load time) at positions *pos_dev + t, write q [B][H][T][D] and append k, v to ring
 * slots (*pos_dev + t) % cap of [B][G][cap][D] (RingKVCache.complete, transformer.py:255-262 / lit_model.py RingKVCache).
 * kv_heads = G (0 -> H), rope_dims 0 -> D.  Follow with rst_attn_decode_multi_f32. */
int rst_lm_rope_append_f32(const float* qkv, float* q, float* k, float* v, const int64_t* pos_dev, int B, int T, int H,
                           int kv_heads, int D, int cap, int ldqkv, int rope, float rope_coef, int rope_dims,
                           rst_stream_t stream);

/* Single-query attention over the ring, straight from the qkv row of the new step: interleaved RoPE on q and on the new key
 * (modules/rope.py), append of k / v to ring slot *pos_dev % cap (RingKVCache.complete, transformer.py:255-262), masked
 * softmax(q k^T / sqrt(D)) v with the mask of transformer.py:404-414 and the slot->position map incl. SURVEY Q1.
 * out [B][H*D].  cap <= 64 with splits == 1 (the depth transformer): one wave per (b, h), no workspace.  Otherwise slots
 * are split over `splits` workgroups per (b, h): ws [B][H][splits][D+2] floats, counters [B][H] uint32 zero-initialised
 * once by the caller (the last-arriving workgroup combines and re-arms its counter; agent-scope release / acquire).
 * Grouped-query attention (CausalSelfAttention, models/llama_streaming.py:935-998): qkv = [q (H*D) | k (G*D) | v (G*D)],
 * ring [B][G][cap][D] with G = kv_heads (0 -> H) -- the reference caches keys expanded to H heads (:965-971), the grouped
 * ring holds the same values once.  rope_dims: leading head dims that rotate (config.rope_n_elem; 0 -> D).
 * out_packed (optional, then out may be NULL; D = 64 / 128): the result is written as the packed bf16 hi / lo operand
 * [2][ceil(B/32)*32][H*D] of rst_gemm_skinny_bf16_f32 (the out-projection that follows) instead of fp32 -- one launch less per
 * layer; rows past B of the buffer are not touched (keep them zero).
 * kv_bf16 != 0 (long-ring form only): k / v are bf16 rings -- the reference's own cache precision (RingKVCache dtype,
 * modules/transformer.py:228 with the model in bf16, moshi/models/loaders.py:144): appended keys / values are rounded to
 * nearest-even, the new step attends to its own key / value at that precision, reads are widened to fp32.
 * rope_table (optional, long-ring form): the step's rotation as [D/2][2] (cos, sin) pairs from rst_lm_rope_table_f32 -- the same
 * values the launch would compute itself (angle_i = exp(i * rope_coef) * pos, identity beyond rope_dims / 2), computed ONCE per
 * frame instead of by every lane of every layer's launch (24 libm calls per lane: most of the launch at short context). */
int rst_lm_rope_table_f32(const int64_t* pos_dev, float* table, int D, int rope_dims, float rope_coef, rst_stream_t stream);
int rst_lm_attn_decode_f32(const float* qkv, void* k, void* v, float* ws, uint32_t* counters, float* out,
                           const int64_t* pos_dev, int B, int H, int D, int cap, int context, int splits, int ldqkv, int rope,
                           float rope_coef, int kv_heads, int rope_dims, uint16_t* out_packed, int kv_bf16, const float* rope_table,
                           rst_stream_t stream);

/* The few-query form of rst_attention_f32(ring = 1) for streaming steps of the codec transformers (T <= a few new steps per
 * call): q [B][H][T][D] already rotated and k / v already appended by rst_rope_split_f32; every (b, t, h) query is split
 * over the occupied ring slots like rst_lm_attn_decode_f32 (same mask / slot map with end_offset = *pos_dev + T).
 * out [B][T][H*D].  ws [B*T][H][splits][D+2], counters [B*T][H] (needed when splits > 1).  k / v [B][G][cap][D] with
 * G = kv_heads (0 -> H). */
int rst_attn_decode_multi_f32(const float* q, const float* k, const float* v, float* ws, uint32_t* counters, float* out,
                              const int64_t* pos_dev, int B, int T, int H, int D, int cap, int context, int splits,
                              int kv_heads, rst_stream_t stream);

/* One streaming step of a codec transformer layer's attention as ONE launch (round 6; rst_rope_split_f32(ring = 1) +
 * rst_attn_decode_multi_f32 in one workgroup per (stream, head)): qkv [B][T][3][H][D] is the in-projection's output of the T new steps
 * ("b t (p h d)", modules/transformer.py:376-388); q and k are rotated (interleaved RoPE at positions *pos_dev + t, modules/rope.py:37-62;
 * rope = 0: none), k / v appended to ring slots (*pos_dev + t) % cap of [B][H][cap][D] (RingKVCache.complete, transformer.py:255-262), and
 * the T queries run against the ring with the mask / slot map of rst_attention_f32(ring = 1), end_offset = *pos_dev + T.  out [B][T][H*D].
 * out_packed_rows != 0 (= B*T rounded up to 32 / 64 / 128, the rows of rst_skinny_f32_pack_win's buffer; H*D % 8 == 0): out is written as the
 * packed operand of the out-projection's few-row GEMM instead, rows past B*T zero (rst_gemm_skinny_f32 consumes it: no packing launch).
 * Served shapes (rst_attention_step_supported): 1 <= T <= 4, D in {32, 64, 128}, the ring's scores in LDS (cap up to ~4000 at T = 4). */
int rst_attention_step_supported(int T, int D, int cap);
int rst_attention_step_f32(const float* qkv, float* k, float* v, float* out, const int64_t* pos_dev, int B, int T, int H, int D, int cap,
                           int context, int rope, float rope_coef, int out_packed_rows, rst_stream_t stream);

/* sample_token (utils/sampling.py:85-105): greedy argmax, or softmax(logits/temp) -> top-k (sorted descending) ->
 * argmax_j p_j / noise_j with caller-provided Exp(1) noise [B][noise_stride] (the reference draws it with
 * Tensor.exponential_, :44-46).  tokens[b * tok_stride] = result.  v_limit (0 = V; v_limit_dev, when given, is a device
 * int32 that overrides it): ids >= limit are never drawn -- the probability blanking of sample_token_audio (2049) and
 * sample_token_audio_2048 (2048), utils/sampling.py:107-158, applied after the softmax like there; greedy ignores it.
 * top_p > 0: nucleus sampling instead of top-k, as sample_token prefers it (:96-99) -- sample_top_p (:66-82): probabilities
 * sorted descending (ties: lowest id first), entries kept while the exclusive prefix sum is <= top_p, survivors renormalised,
 * token = idx[argmax_j q_j / noise_j] with ONE noise value per sorted position: noise then needs V values per row
 * (noise_stride >= V).  Ids >= limit are left out of the nucleus (the reference's own combination of blanking and top_p yields NaN).
 * workspace: rst_lm_sample_workspace_bytes(B, V, top_k, top_p > 0) bytes (0 when none is needed): the sort buffer of top_p, and for
 * V > 32768 the chunk records of the two-level form (one 256-thread workgroup per 10 240 ids selects the chunk's exact top-k, one
 * workgroup per row merges them: same tokens as the one-level kernel, which runs when workspace is NULL). */
int rst_lm_sample_workspace_bytes(int B, int V, int top_k, int top_p_mode);
int rst_lm_sample_f32(const float* logits, const float* noise, int64_t* tokens, int B, int V, int ld, int top_k,
                      int noise_stride, int tok_stride, int use_sampling, float temp, int v_limit, const int32_t* v_limit_dev,
                      float top_p, void* workspace, int64_t workspace_bytes, rst_stream_t stream);

/* LMGen.step's token ring and delay pattern (models/model.py:506-562) on the device.  cache int64 [B][K][CT] (CT = max_delay
 * + 2), delays int32 [K] and the frame counter offset_dev (int64 scalar) stay in HBM, so that a frame is one captured graph.
 *   begin (:506-521): cache[b][k][(offset + delay_k) % CT] = user_tokens[b][k - first_user_k] for the Ki user codebooks;
 *     cache[b][k][offset % CT] = initial[k] while offset <= delay_k; input_out[b][k] = cache[b][k][offset % CT].
 *   commit (:545-562): offset += 1; cache[b][k][offset % CT] = tokens[b][k] for k < n_out (text, then the dep_q audio
 *     tokens); out[b][k] = cache[b][k][(offset - max_delay + delay_k) % CT] (meaningful once offset > max_delay). */
int rst_lm_ring_begin_i64(int64_t* cache, const int64_t* user_tokens, const int64_t* initial, const int32_t* delays,
                          int64_t* offset_dev, int64_t* input_out, int B, int K, int CT, int Ki, int first_user_k, rst_stream_t stream);
int rst_lm_ring_commit_i64(int64_t* cache, const int64_t* tokens, const int32_t* delays, int64_t* offset_dev, int64_t* out, int B, int K,
                           int CT, int n_out, int max_delay, rst_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RSTNET_HIP_H */
