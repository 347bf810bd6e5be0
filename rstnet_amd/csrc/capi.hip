// extern "C" boundary of librstnet_hip.so -- argument checking and parameter marshalling only.
#include "../../include/rstnet_hip.h"
#include "rst_kernels.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void rst_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int rst_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rst_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return RST_ERR_LAUNCH;
    }
    return RST_OK;
}

extern "C" {

int rst_version(void) { return 121; }      // 121: + rst_linear_few_rows_f32, rst_attention_step_f32 / _supported; 120 = round 6: + rst_temporal_decode_frame / _supported / _workspace_bytes, rst_build_id, rst_rvq_chain_supported; 110 = round 5: three-plane weights in operand order (re-pack!), + rst_attention_qkv_f32 / rst_rope_table_f32,
                                           // rst_rvq_search_chain_f32, rst_embed_sum_bf16(add_stride); 105: round 4
const char* rst_last_error(void) { return g_err; }

#ifndef RST_SRC_SHA
#define RST_SRC_SHA "unknown"
#endif
int rst_build_id(char* out, int n) {
    static const char id[] = RST_SRC_SHA;
    if (!out || n < (int)sizeof(id)) return RST_ERR_INVALID_ARG;
    for (int i = 0; i < (int)sizeof(id); ++i) out[i] = id[i];
    return (int)sizeof(id) - 1;
}

static int gemm_win_common(const float* x, const float* hist, const float* w, const uint16_t* w3, const float* bias, const float* res,
                           const float* scale, float* y, int B, int T_in, int T_out, int C, int K, int N, int S, int P,
                           int pad_mode, int64_t x_bstride, int ldy, int act_in, int act_out, int split_k, float* ws,
                           uint32_t* counters, rst_stream_t stream) {
    RST_REQUIRE(ldy >= N, "gemm_win: ldy (%d) < N (%d)", ldy, N);
    RST_REQUIRE(act_in == 0 || act_in == 1, "gemm_win: unknown act_in %d", act_in);
    RST_REQUIRE(act_out >= 0 && act_out <= 2, "gemm_win: unknown act_out %d", act_out);
    RST_REQUIRE(pad_mode == 0 || pad_mode == 1, "gemm_win: unknown pad_mode %d", pad_mode);
    GemmWinParams p;
    p.x = x; p.hist = hist; p.w = w; p.bias = bias; p.res = res; p.scale = scale; p.y = y;
    p.B = B; p.T_in = T_in; p.T_out = T_out; p.C = C; p.K = K; p.N = N; p.S = S; p.P = P;
    p.pad_mode = pad_mode; p.x_bstride = x_bstride; p.ldy = ldy; p.act_in = act_in; p.act_out = act_out;
    p.split_k = split_k; p.ws = ws; p.counters = counters;
    p.w3 = reinterpret_cast<const short*>(w3);
    RST_REQUIRE(split_k <= 1 || ldy == N, "gemm_win: split-K needs ldy == N");
    return rst_launch_gemm_win(p, (hipStream_t)stream);
}

int rst_gemm_win_f32(const float* x, const float* hist, const float* w, const float* bias, const float* res,
                     const float* scale, float* y, int B, int T_in, int T_out, int C, int K, int N, int S, int P,
                     int pad_mode, int64_t x_bstride, int ldy, int act_in, int act_out, int split_k, float* ws,
                     uint32_t* counters, rst_stream_t stream) {
    return gemm_win_common(x, hist, w, nullptr, bias, res, scale, y, B, T_in, T_out, C, K, N, S, P, pad_mode, x_bstride, ldy, act_in,
                           act_out, split_k, ws, counters, stream);
}

int rst_gemm_win_b3_f32(const float* x, const float* hist, const float* w, const uint16_t* w3, const float* bias, const float* res,
                        const float* scale, float* y, int B, int T_in, int T_out, int C, int K, int N, int S, int P,
                        int pad_mode, int64_t x_bstride, int ldy, int act_in, int act_out, rst_stream_t stream) {
    RST_REQUIRE(w3, "gemm_win_b3: the split weights are required (rst_gemm_win_b3_pack_weight)");
    return gemm_win_common(x, hist, w, w3, bias, res, scale, y, B, T_in, T_out, C, K, N, S, P, pad_mode, x_bstride, ldy, act_in,
                           act_out, 1, nullptr, nullptr, stream);
}

int rst_gemm_win_b3_supported(int B, int T_in, int T_out, int C, int K, int N, int S, int P, int pad_mode, int64_t x_bstride, int has_hist) {
    if (S <= 0 || P < 0) return 0;
    return rst_gemm_win_b3_shape_ok(B, T_in, T_out, C, K, N, pad_mode, (long)x_bstride, has_hist != 0) ? 1 : 0;
}

int rst_gemm_win_b3_weight_elems(int N, int K) {
    const long n = (N > 0 && K > 0) ? rst_gemm_win_b3_weight_elems_impl(N, K) : -1;
    return n > 0 && n < 0x7fffffffL ? (int)n : -1;
}

int rst_gemm_win_b3_pack_weight(const float* w, uint16_t* w3, int N, int K, rst_stream_t stream) {
    return rst_launch_gemm_win_b3_pack(w, w3, N, K, (hipStream_t)stream);
}

int rst_gemm_win_split_plan(int64_t M, int N, int K) { return rst_gemm_split_plan_impl((long)M, N, K); }
int rst_gemm_win_split_tiles(int64_t M, int N) { return rst_gemm_split_tiles_impl((long)M, N); }

int rst_conv1d_causal_f32(const float* x, const float* hist, const float* w_packed, const float* bias,
                          const float* res, float* y, int B, int T_in, int T_out, int Cin, int Cout, int Kw_eff,
                          int stride, int pad_mode, int act_in, int act_out, rst_stream_t stream) {
    RST_REQUIRE(Kw_eff >= stride && stride > 0, "conv1d: kernel (%d) must be >= stride (%d)", Kw_eff, stride);
    return rst_gemm_win_f32(x, hist, w_packed, bias, res, nullptr, y, B, T_in, T_out, Cin, Kw_eff * Cin, Cout, stride,
                            Kw_eff - stride, pad_mode, (int64_t)T_in * Cin, Cout, act_in, act_out, 1, nullptr, nullptr, stream);
}

int rst_convtr1d_causal_f32(const float* x, const float* hist, const float* w_packed, const float* bias_tiled,
                            float* y, int B, int T_in, int Cin, int Cout, int Kw, int stride, int act_in, int act_out,
                            rst_stream_t stream) {
    RST_REQUIRE(Kw >= stride && stride > 0, "convtr1d: kernel (%d) must be >= stride (%d)", Kw, stride);
    const int q = (Kw + stride - 1) / stride;
    return rst_gemm_win_f32(x, hist, w_packed, bias_tiled, nullptr, nullptr, y, B, T_in, T_in, Cin, q * Cin,
                            stride * Cout, 1, q - 1, 0, (int64_t)T_in * Cin, stride * Cout, act_in, act_out, 1, nullptr, nullptr,
                            stream);
}

int rst_linear_f32(const float* x, const float* w, const float* bias, const float* res, const float* scale, float* y,
                   int64_t M, int K, int N, int act_out, rst_stream_t stream) {
    RST_REQUIRE(M >= 0 && M < 0x7fffffffLL, "linear: M out of range");
    return rst_gemm_win_f32(x, nullptr, w, bias, res, scale, y, 1, (int)M, (int)M, K, K, N, 1, 0, 0, M * K, N, 0,
                            act_out, 1, nullptr, nullptr, stream);
}

int rst_seanet_resblock_supported(int C, int H, int Kw, int pre, int post, int K0, int Kf) {
    return rst_resblock_supported(C, H, Kw, pre, post, K0, Kf) ? 1 : 0;
}

int rst_seanet_resblock_f32(const float* x, const float* hist, const float* w0, const float* b0, const float* w1,
                            const float* b1, const float* w2, const float* b2, const float* wf, const float* bf,
                            float* y, int B, int T, int C, int H, int Kw, int K0, int Kf, int elu_out,
                            rst_stream_t stream) {
    ResblockParams p;
    p.x = x; p.hist = hist; p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.wf = wf; p.bf = bf; p.y = y;
    p.B = B; p.T = T; p.C = C; p.H = H; p.Kw = Kw; p.K0 = K0; p.Kf = Kf;
    p.pre = w0 != nullptr; p.post = wf != nullptr; p.elu_out = elu_out;
    RST_REQUIRE(!(p.post && elu_out), "resblock: elu_out has no meaning with the fused last conv");
    return rst_launch_resblock(p, (hipStream_t)stream);
}

int rst_seanet_resblock_b3_supported(int B, int T, int C, int H, int Kw, int pre, int post, int K0, int Kf) {
    if (B <= 0 || T <= 0 || (long)T * C * 4 >= 0xfffff000L) return 0;       // (32-bit buffer offsets inside an utterance)
    return rst_resblock_b3_supported(C, H, Kw, pre, post, K0, Kf) ? 1 : 0;
}

int rst_seanet_resblock_b3_weight_elems(int C) { return (int)rst_resblock_b3_weight_elems(C); }

int rst_seanet_resblock_b3_pack(const float* w0, const float* w1, const float* w2, uint16_t* wp, int C, int H, int Kw, int K0,
                                rst_stream_t stream) {
    return rst_launch_resblock_b3_pack(w0, w1, w2, wp, C, H, Kw, K0, (hipStream_t)stream);
}

int rst_seanet_resblock_b3_f32(const float* x, const uint16_t* wp, const float* b0, const float* b1, const float* b2, const float* wf,
                               const float* bf, float* y, int B, int T, int C, int H, int Kw, int K0, int Kf, int elu_out,
                               rst_stream_t stream) {
    ResblockB3Params p;
    p.x = x; p.wp = wp; p.b0 = b0; p.b1 = b1; p.b2 = b2; p.wf = wf; p.bf = bf; p.y = y;
    p.B = B; p.T = T; p.C = C; p.H = H; p.Kw = Kw; p.K0 = K0; p.Kf = Kf;
    p.pre = b0 != nullptr; p.post = wf != nullptr; p.elu_out = elu_out;
    RST_REQUIRE(!(p.post && elu_out), "resblock_b3: elu_out has no meaning with the fused last conv");
    RST_REQUIRE(B <= 0 || T <= 0 || rst_seanet_resblock_b3_supported(B, T, C, H, Kw, p.pre, p.post, K0, Kf),
                "resblock_b3: unsupported shape (B=%d T=%d C=%d H=%d Kw=%d pre=%d post=%d K0=%d Kf=%d); ask rst_seanet_resblock_b3_supported", B, T, C, H,
                Kw, p.pre, p.post, K0, Kf);
    return rst_launch_resblock_b3(p, (hipStream_t)stream);
}

int rst_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int D, float eps,
                      rst_stream_t stream) {
    return rst_launch_layernorm(x, gamma, beta, y, rows, D, eps, (hipStream_t)stream);
}

int rst_rope_split_f32(const float* qkv, float* q, float* k, float* v, const int64_t* pos_dev, int64_t pos0, int B,
                       int T, int H, int D, int cap, int ring, int rope, float rope_coef, rst_stream_t stream) {
    RopeSplitParams p;
    p.qkv = qkv; p.q = q; p.k = k; p.v = v; p.pos_dev = (const long*)pos_dev; p.pos0 = pos0;
    p.B = B; p.T = T; p.H = H; p.D = D; p.cap = cap; p.ring = ring; p.rope = rope; p.rope_coef = rope_coef;
    return rst_launch_rope_split(p, (hipStream_t)stream);
}

int rst_attention_f32(const float* q, const float* k, const float* v, float* out, const int64_t* pos_dev, int64_t pos0,
                      int B, int T, int H, int D, int cap, int ring, int context, rst_stream_t stream) {
    AttentionParams p;
    p.q = q; p.k = k; p.v = v; p.out = out; p.pos_dev = (const long*)pos_dev; p.pos0 = pos0;
    p.B = B; p.T = T; p.H = H; p.D = D; p.cap = cap; p.ring = ring; p.context = context;
    p.row_stride = 0; p.rope_tab = nullptr;
    return rst_launch_attention(p, (hipStream_t)stream);
}

int rst_rope_table_f32(float* table, int T, int D, float rope_coef, int64_t pos0, rst_stream_t stream) {
    return rst_launch_rope_table(table, T, D, rope_coef, pos0, (hipStream_t)stream);
}

int rst_attention_qkv_f32(const float* qkv, const float* rope_table, float* out, int B, int T, int H, int D, int context,
                          rst_stream_t stream) {
    RST_REQUIRE(qkv && out && B >= 0 && T >= 0 && H > 0 && D > 0, "attention_qkv: bad arguments");
    RST_REQUIRE((uintptr_t)qkv % 16 == 0 && (!rope_table || (uintptr_t)rope_table % 16 == 0), "attention_qkv: pointers must be 16-byte aligned");
    AttentionParams p;
    p.q = qkv; p.k = qkv + (long)H * D; p.v = qkv + 2L * H * D; p.out = out; p.pos_dev = nullptr; p.pos0 = 0;
    p.B = B; p.T = T; p.H = H; p.D = D; p.cap = T; p.ring = 0; p.context = context;
    p.row_stride = 3 * H * D; p.rope_tab = rope_table;
    return rst_launch_attention(p, (hipStream_t)stream);
}

int rst_rvq_pack_f32(const float* emb, float* packed, float* e2, int n_codes, int D, rst_stream_t stream) {
    return rst_launch_rvq_pack(emb, packed, e2, n_codes, D, (hipStream_t)stream);
}

int rst_rvq_search_f32(const float* x, const float* emb, const float* packed, const float* e2, int64_t* codes,
                       float* dist, uint64_t* keys, int M, int F, int ldx, int D, int n_codes, int L, int n_groups,
                       const int* group_begin, const int* group_count, rst_stream_t stream) {
    RST_REQUIRE(n_groups >= 1 && n_groups <= 2 && group_begin && group_count, "rvq_search: bad groups");
    RvqSearchParams p;
    p.x = x; p.emb = emb; p.packed = packed; p.e2 = e2; p.codes = (long*)codes; p.dist = dist;
    p.M = M; p.F = F; p.ldx = ldx; p.D = D; p.n_codes = n_codes; p.L = L; p.n_groups = n_groups;
    for (int g = 0; g < 2; ++g) {
        p.group_begin[g] = g < n_groups ? group_begin[g] : 0;
        p.group_count[g] = g < n_groups ? group_count[g] : 0;
        RST_REQUIRE(p.group_begin[g] >= 0 && p.group_count[g] >= 0 && p.group_begin[g] + p.group_count[g] <= L,
                    "rvq_search: group %d out of range", g);
    }
    if (keys) return rst_launch_rvq_search_small(p, (unsigned long long*)keys, (hipStream_t)stream);
    return rst_launch_rvq_search(p, (hipStream_t)stream);
}

int rst_rvq_chain_supported(int M, int n_codes, int L, int D, int n_groups) { return rst_rvq_chain_supported_impl(M, n_codes, L, D, n_groups); }

int rst_rvq_chain_slot_elems(int M, int n_codes, int L) {
    if (M <= 0 || n_codes <= 0 || L <= 0 || (long)L * ((M + 31) / 32 * 32) * rst_rvq_chain_slices(n_codes) > 0x7fffffffL) return -1;
    return L * ((M + 31) / 32 * 32) * rst_rvq_chain_slices(n_codes);
}

int rst_rvq_search_chain_f32(const float* x, const float* emb, const float* packed, const float* e2, int64_t* codes, float* dist,
                             uint64_t* slots, uint32_t* status, int M, int F, int ldx, int D, int n_codes, int L, int n_groups,
                             const int* group_begin, const int* group_count, rst_stream_t stream) {
    RST_REQUIRE(n_groups >= 1 && n_groups <= 2 && group_begin && group_count, "rvq_search_chain: bad groups");
    RvqSearchParams p;
    p.x = x; p.emb = emb; p.packed = packed; p.e2 = e2; p.codes = (long*)codes; p.dist = dist;
    p.M = M; p.F = F; p.ldx = ldx; p.D = D; p.n_codes = n_codes; p.L = L; p.n_groups = n_groups;
    for (int g = 0; g < 2; ++g) {
        p.group_begin[g] = g < n_groups ? group_begin[g] : 0;
        p.group_count[g] = g < n_groups ? group_count[g] : 0;
        RST_REQUIRE(p.group_begin[g] >= 0 && p.group_count[g] >= 0 && p.group_begin[g] + p.group_count[g] <= L,
                    "rvq_search_chain: group %d out of range", g);
    }
    return rst_launch_rvq_search_chain(p, (unsigned long long*)slots, (unsigned*)status, (hipStream_t)stream);
}

int rst_rvq_gather_f32(const int64_t* codes, const float* emb, float* out, int M, int F, int D, int n_codes, int L,
                       int n_groups, const int* group_begin, const int* group_count, rst_stream_t stream) {
    RST_REQUIRE(n_groups >= 1 && n_groups <= 2 && group_begin && group_count, "rvq_gather: bad groups");
    RvqGatherParams p;
    p.codes = (const long*)codes; p.emb = emb; p.out = out;
    p.M = M; p.F = F; p.D = D; p.n_codes = n_codes; p.L = L; p.n_groups = n_groups;
    for (int g = 0; g < 2; ++g) {
        p.group_begin[g] = g < n_groups ? group_begin[g] : 0;
        p.group_count[g] = g < n_groups ? group_count[g] : 0;
        RST_REQUIRE(p.group_begin[g] >= 0 && p.group_count[g] >= 0 && p.group_begin[g] + p.group_count[g] <= L,
                    "rvq_gather: group %d out of range", g);
    }
    return rst_launch_rvq_gather(p, (hipStream_t)stream);
}

int rst_convtr_depthwise_f32(const float* x, const float* hist, const float* w, float* y, int B, int T_in, int C,
                             int Kw, int stride, rst_stream_t stream) {
    return rst_launch_convtr_depthwise(x, hist, w, y, B, T_in, C, Kw, stride, (hipStream_t)stream);
}

int rst_act_f32(const float* x, float* y, int64_t n, int act, rst_stream_t stream) {
    return rst_launch_act(x, y, n, act, (hipStream_t)stream);
}

int rst_transpose_f32(const float* x, float* y, int B, int R, int C, rst_stream_t stream) {
    return rst_launch_transpose(x, y, B, R, C, (hipStream_t)stream);
}

int rst_skinny_f32_pack_weight(const float* w, float* wp, int N, int K, rst_stream_t stream) {
    return rst_launch_skinny_f32_pack_weight(w, wp, N, K, (hipStream_t)stream);
}

int rst_skinny_f32_pack_win(const float* x, const float* hist, float* xp, int B, int T_in, int T_out, int C, int K, int S, int P,
                            int pad_mode, int64_t x_bstride, int act_in, rst_stream_t stream) {
    SkinnyF32PackParams p;
    p.x = x; p.hist = hist; p.xp = xp; p.B = B; p.T_in = T_in; p.T_out = T_out; p.C = C; p.K = K; p.Kp = (K + 7) / 8 * 8; p.S = S; p.P = P;
    p.pad_mode = pad_mode; p.act_in = act_in; p.x_bstride = x_bstride;
    return rst_launch_skinny_f32_pack_win(p, (hipStream_t)stream);
}

int rst_skinny_f32_pack_ln(const float* x, const float* gamma, const float* beta, float eps, float* xp, int M, int K, rst_stream_t stream) {
    return rst_launch_skinny_f32_pack_ln(x, gamma, beta, xp, M, K, eps, (hipStream_t)stream);
}

int rst_skinny_f32_split_plan(int M, int N, int K) { return rst_skinny_f32_split_plan_impl(M, N, K); }

int rst_gemm_skinny_f32(const float* xp, const float* wp, const float* bias, const float* res, const float* scale, float* y, int M,
                        int N, int K, int ldy, int act_out, int split_k, float* ws, uint32_t* counters, int y_packed, rst_stream_t stream) {
    SkinnyF32Params p = {};
    p.xp = xp; p.wp = wp; p.bias = bias; p.res = res; p.scale = scale; p.y = y; p.M = M; p.N = N; p.Kp = (K + 7) / 8 * 8; p.ldy = ldy;
    p.act_out = act_out; p.split_k = split_k; p.ws = ws; p.counters = counters; p.Np_out = y_packed ? N : 0;
    return rst_launch_gemm_skinny_f32(p, (hipStream_t)stream);
}

int rst_linear_few_rows_f32(const float* x, int ldx, const float* wp, const float* bias, const float* res, const float* scale, float* y, int M,
                            int N, int K, int ldy, int act_out, int split_k, float* ws, uint32_t* counters, int y_packed, rst_stream_t stream) {
    RST_REQUIRE(x && K > 0 && K % 8 == 0, "linear_few_rows: K %% 8 == 0 required (K=%d)", K);
    SkinnyF32Params p = {};
    p.xr = x; p.ldx = ldx;
    p.wp = wp; p.bias = bias; p.res = res; p.scale = scale; p.y = y; p.M = M; p.N = N; p.Kp = K; p.ldy = ldy;
    p.act_out = act_out; p.split_k = split_k; p.ws = ws; p.counters = counters; p.Np_out = y_packed ? N : 0;
    return rst_launch_gemm_skinny_f32(p, (hipStream_t)stream);
}

int rst_attention_step_supported(int T, int D, int cap) { return rst_attn_step_supported_impl(T, D, cap); }

int rst_attention_step_f32(const float* qkv, float* k, float* v, float* out, const int64_t* pos_dev, int B, int T, int H, int D, int cap,
                           int context, int rope, float rope_coef, int out_packed_rows, rst_stream_t stream) {
    AttnStepParams p = {};
    p.out_rows = out_packed_rows;
    p.qkv = qkv; p.k = k; p.v = v; p.out = out; p.pos_dev = reinterpret_cast<const long*>(pos_dev);
    p.B = B; p.T = T; p.H = H; p.D = D; p.cap = cap; p.context = context; p.rope = rope; p.rope_coef = rope_coef;
    return rst_launch_attn_step(p, (hipStream_t)stream);
}

int rst_hist_update_batch_f32(const float* const* x, float* const* hist, const int* T_in, const int* P, const int* C, int n, int B,
                              rst_stream_t stream) {
    RST_REQUIRE(n >= 0 && n <= RST_HIST_BATCH_MAX && (n == 0 || (x && hist && T_in && P && C)), "hist_update_batch: bad table (n=%d)", n);
    HistBatchParams p = {};
    for (int i = 0; i < n; ++i) { p.x[i] = x[i]; p.hist[i] = hist[i]; p.T_in[i] = T_in[i]; p.P[i] = P[i]; p.C[i] = C[i]; }
    p.n = n; p.B = B;
    return rst_launch_hist_update_batch(p, (hipStream_t)stream);
}

int rst_mask_tail_f32(float* x, const int32_t* lengths, int B, int T, int C, int mode, rst_stream_t stream) {
    return rst_launch_mask_tail(x, lengths, B, T, C, mode, (hipStream_t)stream);
}

int rst_hist_update_f32(const float* x, const float* hist_in, float* hist_out, int B, int T_in, int P_in, int P_out,
                        int C, rst_stream_t stream) {
    return rst_launch_hist_update(x, hist_in, hist_out, B, T_in, P_in, P_out, C, (hipStream_t)stream);
}

// ---- LM decode step -------------------------------------------------------------------------------------------------
static void gemv_defaults(GemvParams& p) {
    p.beta = nullptr; p.scale = nullptr; p.act_out = 0; p.w_f32 = 0; p.gate_out = 0;
    p.at_k = p.at_v = nullptr; p.at_pos = nullptr; p.at_H = p.at_D = p.at_cap = p.at_context = 0;
    p.em_table = nullptr; p.em_tokens = nullptr; p.em_x_out = nullptr; p.em_tok_stride = p.em_col = p.em_rows = 0;
}

int rst_gemv_bf16_f32(const float* x, const float* alpha, const uint16_t* w, const float* res, const float* bias, float* y,
                      int B, int N, int K, int ldx, int ldy, int prologue, float eps, int gate_out, rst_stream_t stream) {
    RST_REQUIRE(prologue >= 0 && prologue <= 2, "gemv_bf16: prologue must be 0 (none), 1 (RMSNorm) or 2 (SiLU gate)");
    GemvParams p;
    gemv_defaults(p);
    p.x = x; p.alpha = alpha; p.w = w; p.res = res; p.bias = bias; p.y = y; p.B = B; p.N = N;
    p.K = K; p.ldx = ldx; p.ldy = ldy; p.prologue = prologue; p.gate_out = gate_out; p.eps = eps;
    return rst_launch_gemv(p, (hipStream_t)stream);
}

int rst_gemv_attn_bf16_f32(const float* qkv, float* k_cache, float* v_cache, const int64_t* pos_dev, const uint16_t* w,
                           const float* res, const float* bias, float* y, int B, int N, int H, int D, int cap, int context, int ldqkv,
                           int ldy, rst_stream_t stream) {
    GemvParams p;
    gemv_defaults(p);
    p.x = qkv; p.alpha = nullptr; p.w = w; p.res = res; p.bias = bias; p.y = y; p.B = B; p.N = N; p.K = H * D; p.ldx = ldqkv; p.ldy = ldy;
    p.prologue = 4; p.eps = 0.f;
    p.at_k = k_cache; p.at_v = v_cache; p.at_pos = reinterpret_cast<const long*>(pos_dev); p.at_H = H; p.at_D = D; p.at_cap = cap;
    p.at_context = context;
    return rst_launch_gemv(p, (hipStream_t)stream);
}

int rst_gemv_embed_bf16_f32(const float* add, const uint16_t* table, const int64_t* tokens, float* x_out, const float* alpha,
                            const uint16_t* w, const float* bias, float* y, int B, int N, int K, int ld_add, int ldy, int tok_stride,
                            int tok_col, int table_rows, float eps, rst_stream_t stream) {
    GemvParams p;
    gemv_defaults(p);
    p.x = add; p.alpha = alpha; p.w = w; p.res = nullptr; p.bias = bias; p.y = y; p.B = B; p.N = N; p.K = K; p.ldx = ld_add; p.ldy = ldy;
    p.prologue = 5; p.eps = eps;
    p.em_table = table; p.em_tokens = reinterpret_cast<const long*>(tokens); p.em_x_out = x_out; p.em_tok_stride = tok_stride;
    p.em_col = tok_col; p.em_rows = table_rows;
    return rst_launch_gemv(p, (hipStream_t)stream);
}

int rst_depth_frame_workspace_bytes(int B, int E, int Hd, int card) { return (int)rst_depth_frame_workspace_bytes_impl(B, E, Hd, card); }

int rst_depth_frame_supported(int B, int E, int H, int Hd, int card, int dep_q, int L, int top_k) {
    if (H <= 0 || E % H) return 0;
    DepthFrameParams p = {};
    p.B = B; p.E = E; p.H = H; p.D = E / H; p.Hd = Hd; p.card = card; p.dep_q = dep_q; p.L = L; p.top_k = top_k;
    return rst_depth_frame_grid(p);
}

int rst_depth_decode_frame(const uint16_t* const* in_proj, const uint16_t* const* out_proj, const float* const* norm1,
                           const float* const* norm2, const uint16_t* const* gate_in, const uint16_t* const* gate_out,
                           const uint16_t* const* heads, const float* const* head_bias, const uint16_t* const* emb, const int* emb_rows,
                           const float* h_all, int64_t* tokens, const float* noise, const int* v_limit_dev, void* workspace,
                           uint32_t* status, int B, int E, int H, int Hd, int card, int dep_q, int L, int ld_h, int tok_stride,
                           int noise_stride, int top_k, int use_sampling, float temp, float eps, int context, int ring_cap,
                           rst_stream_t stream) {
    RST_REQUIRE(ring_cap >= dep_q, "depth_decode_frame: a ring of %d slots cannot hold the %d steps of a frame", ring_cap, dep_q);
    RST_REQUIRE(in_proj && out_proj && norm1 && norm2 && gate_in && gate_out && heads && emb && emb_rows, "depth_decode_frame: null table");
    RST_REQUIRE(L >= 1 && L <= RST_DEPTH_MAX_L && dep_q >= 1 && dep_q <= RST_DEPTH_MAX_Q, "depth_decode_frame: L=%d (<= %d), dep_q=%d (<= %d)", L,
                RST_DEPTH_MAX_L, dep_q, RST_DEPTH_MAX_Q);
    RST_REQUIRE(H > 0 && E % H == 0, "depth_decode_frame: %d heads do not divide E=%d", H, E);
    DepthFrameParams p = {};
    for (int l = 0; l < L; ++l) {
        p.in_proj[l] = in_proj[l]; p.out_proj[l] = out_proj[l]; p.norm1[l] = norm1[l]; p.norm2[l] = norm2[l];
        for (int k = 0; k < dep_q; ++k) { p.gate_in[l][k] = gate_in[l * dep_q + k]; p.gate_out[l][k] = gate_out[l * dep_q + k]; }
    }
    for (int k = 0; k < dep_q; ++k) {
        p.heads[k] = heads[k]; p.head_bias[k] = head_bias ? head_bias[k] : nullptr; p.emb[k] = emb[k]; p.emb_rows[k] = emb_rows[k];
    }
    p.h_all = h_all; p.tokens = reinterpret_cast<long*>(tokens); p.noise = noise; p.v_limit = v_limit_dev;
    p.gran = static_cast<unsigned long long*>(workspace); p.status = status;
    p.hist_solo = workspace ? reinterpret_cast<float*>(p.gran + 2 * rst_depth_frame_workspace_granules(B, E, Hd, card)) : nullptr;
    p.B = B; p.E = E; p.H = H; p.D = E / H; p.Hd = Hd; p.card = card; p.dep_q = dep_q; p.L = L; p.ld_h = ld_h; p.tok_stride = tok_stride;
    p.noise_stride = noise_stride; p.top_k = top_k; p.use_sampling = use_sampling; p.context = context; p.ring_cap = ring_cap; p.eps = eps; p.temp = temp;
    return rst_launch_depth_frame(p, (hipStream_t)stream);
}

int rst_temporal_frame_workspace_bytes(int E, int Hd, int H) {
    if (H <= 0 || E % H) return -1;
    return (int)(2 * rst_temporal_frame_workspace_granules(E, Hd, H, E / H) * 8);
}

int rst_temporal_frame_supported(int E, int H, int Hd, int L, int cap, int kv_bf16) {
    if (H <= 0 || E % H) return 0;
    TemporalFrameParams p = {};
    p.E = E; p.H = H; p.D = E / H; p.Hd = Hd; p.L = L; p.cap = cap; p.kv_bf16 = kv_bf16;
    return rst_temporal_frame_grid(p);
}

int rst_temporal_decode_frame(const uint64_t* dev_tables, const float* x, float* y, const int64_t* pos_dev, const float* rope_cs, void* workspace,
                              uint32_t* status, int E, int H, int Hd, int L, int cap, int context, int kv_bf16, float eps, rst_stream_t stream) {
    RST_REQUIRE(dev_tables, "temporal_decode_frame: null pointer table");
    RST_REQUIRE(L >= 1 && L <= RST_TEMPORAL_MAX_L && H > 0 && E % H == 0, "temporal_decode_frame: L=%d (<= %d), H=%d, E=%d", L, RST_TEMPORAL_MAX_L, H, E);
    RST_REQUIRE(x != y, "temporal_decode_frame: the repair launch re-reads x: y must be another buffer");
    TemporalFrameParams p = {};
    p.tab = reinterpret_cast<const unsigned long long*>(dev_tables);
    p.x = x; p.y = y; p.pos_dev = reinterpret_cast<const long*>(pos_dev); p.rope_cs = rope_cs;
    p.gran = static_cast<unsigned long long*>(workspace); p.status = status;
    p.E = E; p.H = H; p.D = E / H; p.Hd = Hd; p.L = L; p.cap = cap; p.context = context; p.kv_bf16 = kv_bf16; p.eps = eps;
    return rst_launch_temporal_frame(p, (hipStream_t)stream);
}

int rst_codec_transformer_workspace_bytes(int rows, int E, int F) { return (int)(rst_codec_tr_workspace_granules(rows, E, F) * 16); }

int rst_codec_transformer_supported(int B, int T, int E, int H, int F, int L, int cap) { return rst_codec_tr_grid(B, T, E, H, F, L, cap); }

int rst_codec_transformer_frame(const float* const* in_proj, const float* const* out_proj, const float* const* linear1,
                                const float* const* linear2, const float* const* norm1_w, const float* const* norm1_b,
                                const float* const* norm2_w, const float* const* norm2_b, const float* const* scale1,
                                const float* const* scale2, float* const* k_cache, float* const* v_cache, const float* x, float* y,
                                const int64_t* pos_dev, void* workspace, uint32_t* status, int B, int T, int E, int H, int F, int L,
                                int cap, int context, int rope, float rope_coef, float eps, rst_stream_t stream) {
    RST_REQUIRE(in_proj && out_proj && linear1 && linear2 && norm1_w && norm1_b && norm2_w && norm2_b && k_cache && v_cache,
                "codec_transformer_frame: null table");
    RST_REQUIRE(L >= 1 && L <= RST_CTR_MAX_L && H > 0 && E % H == 0, "codec_transformer_frame: L=%d (<= %d), H=%d, E=%d", L, RST_CTR_MAX_L, H, E);
    CodecTrParams p = {};
    for (int l = 0; l < L; ++l) {
        p.in_proj[l] = in_proj[l]; p.out_proj[l] = out_proj[l]; p.lin1[l] = linear1[l]; p.lin2[l] = linear2[l];
        p.n1g[l] = norm1_w[l]; p.n1b[l] = norm1_b[l]; p.n2g[l] = norm2_w[l]; p.n2b[l] = norm2_b[l];
        p.ls1[l] = scale1 ? scale1[l] : nullptr; p.ls2[l] = scale2 ? scale2[l] : nullptr;
        p.kc[l] = k_cache[l]; p.vc[l] = v_cache[l];
    }
    p.x = x; p.y = y; p.pos_dev = reinterpret_cast<const long*>(pos_dev); p.gran = static_cast<unsigned long long*>(workspace); p.status = status;
    p.B = B; p.T = T; p.E = E; p.H = H; p.D = E / H; p.F = F; p.L = L; p.cap = cap; p.context = context; p.rope = rope;
    p.rope_coef = rope_coef; p.eps = eps;
    return rst_launch_codec_tr(p, (hipStream_t)stream);
}

int rst_gemv_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* w, const float* bias,
                 const float* res, const float* scale, float* y, int B, int N, int K, int act_out, rst_stream_t stream) {
    RST_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "gemv_f32: LayerNorm needs both gamma and beta");
    GemvParams p;
    gemv_defaults(p);
    p.x = x; p.alpha = ln_gamma; p.beta = ln_beta; p.w = w; p.res = res; p.bias = bias; p.scale = scale; p.y = y; p.B = B; p.N = N;
    p.K = K; p.ldx = K; p.ldy = N; p.prologue = ln_gamma ? 3 : 0; p.act_out = act_out; p.w_f32 = 1; p.gate_out = 0; p.eps = ln_eps;
    return rst_launch_gemv(p, (hipStream_t)stream);
}

int rst_skinny_pack_weight_bf16(const uint16_t* w, uint16_t* wp, int N, int K, int interleave_halves, rst_stream_t stream) {
    return rst_launch_skinny_pack_weight(w, wp, N, K, interleave_halves, (hipStream_t)stream);
}

int rst_skinny_pack_act_f32(const float* x, const float* alpha, uint16_t* xp, int B, int K, int ldx, int mode, float eps,
                            rst_stream_t stream) {
    return rst_launch_skinny_pack_act(x, alpha, xp, B, K, ldx, mode, eps, (hipStream_t)stream);
}

int rst_skinny_bf16_split_plan(int B, int N, int K) { return rst_skinny_bf16_split_plan_impl(B, N, K); }

int rst_gemm_skinny_bf16_f32(const uint16_t* xp, const uint16_t* wp, const float* res, const float* bias, float* y, int B, int N,
                             int K, int ldy, uint16_t* gate_out, int split_k, float* ws, uint32_t* counters, rst_stream_t stream) {
    RST_REQUIRE(xp, "gemm_skinny_bf16: null packed activations");
    SkinnyParams p = {};
    p.xp = xp; p.w = wp; p.res = res; p.bias = bias; p.y = y; p.B = B; p.N = N; p.K = K; p.ldy = ldy;
    p.gate_out = gate_out; p.gate_plane = (long)((B + 31) / 32 * 32) * (N / 2);
    p.split_k = split_k; p.ws = ws; p.counters = counters;
    return rst_launch_gemm_skinny(p, (hipStream_t)stream);
}

int rst_gemm_skinny_x32_bf16_f32(const float* x, const float* alpha, float eps, int mode, int ldx, const uint16_t* wp, const float* res,
                                 const float* bias, float* y, int B, int N, int K, int ldy, uint16_t* gate_out, rst_stream_t stream) {
    RST_REQUIRE(x, "gemm_skinny_x32: null activations");
    SkinnyParams p = {};
    p.xf = x; p.alpha = alpha; p.eps = eps; p.xmode = mode; p.ldx = ldx;
    p.w = wp; p.res = res; p.bias = bias; p.y = y; p.B = B; p.N = N; p.K = K; p.ldy = ldy;
    p.gate_out = gate_out; p.gate_plane = (long)((B + 31) / 32 * 32) * (N / 2);
    p.split_k = 1;
    return rst_launch_gemm_skinny(p, (hipStream_t)stream);
}

int rst_skinny_pack_weight_fp8(const uint16_t* w, uint8_t* wp, float* wscale, int N, int K, rst_stream_t stream) {
    return rst_launch_skinny_pack_weight_fp8(w, wp, wscale, N, K, (hipStream_t)stream);
}

int rst_skinny_pack_act_fp8(const float* x, const float* alpha, uint8_t* xp, float* xscale, int B, int K, int ldx, int mode, float eps,
                            rst_stream_t stream) {
    return rst_launch_skinny_pack_act_fp8(x, alpha, xp, xscale, B, K, ldx, mode, eps, (hipStream_t)stream);
}

int rst_gemm_skinny_fp8_f32(const uint8_t* xp, const float* xscale, const uint8_t* wp, const float* wscale, const float* res,
                            const float* bias, float* y, int B, int N, int K, int ldy, rst_stream_t stream) {
    SkinnyFp8Params p;
    p.xp = xp; p.xscale = xscale; p.wp = wp; p.wscale = wscale; p.res = res; p.bias = bias; p.y = y; p.B = B; p.N = N; p.K = K; p.ldy = ldy;
    return rst_launch_gemm_skinny_fp8(p, (hipStream_t)stream);
}

int rst_embed_sum_bf16(const int64_t* tokens, const uint16_t* const* tables, const int* tok_index, const int* table_rows, int n_tables,
                       const float* add, float* out, int B, int D, int tok_stride, int add_stride, rst_stream_t stream) {
    RST_REQUIRE(n_tables >= 0 && n_tables <= RST_MAX_TABLES && (n_tables == 0 || (tables && tok_index)), "embed_sum: bad tables");
    RST_REQUIRE(!add || add_stride >= D, "embed_sum: add_stride %d < D %d", add_stride, D);
    EmbedSumParams p;
    p.tokens = (const long*)tokens; p.add = add; p.out = out; p.B = B; p.D = D; p.n_tables = n_tables; p.tok_stride = tok_stride;
    p.add_stride = add ? add_stride : D;
    for (int i = 0; i < n_tables; ++i) { p.tables[i] = tables[i]; p.tok_index[i] = tok_index[i]; p.rows[i] = table_rows ? table_rows[i] : 0; }
    return rst_launch_embed_sum(p, (hipStream_t)stream);
}

int rst_rmsnorm_f32(const float* x, const float* alpha, float* y, int64_t rows, int D, float eps, rst_stream_t stream) {
    return rst_launch_rmsnorm(x, alpha, y, rows, D, eps, (hipStream_t)stream);
}

int rst_lm_rope_append_f32(const float* qkv, float* q, float* k, float* v, const int64_t* pos_dev, int B, int T, int H,
                           int kv_heads, int D, int cap, int ldqkv, int rope, float rope_coef, int rope_dims,
                           rst_stream_t stream) {
    LmRopeAppendParams p;
    p.qkv = qkv; p.q = q; p.k = k; p.v = v; p.pos_dev = (const long*)pos_dev; p.B = B; p.T = T; p.H = H;
    p.G = kv_heads > 0 ? kv_heads : H; p.D = D; p.cap = cap;
    p.ldqkv = ldqkv; p.rope = rope; p.rope_coef = rope_coef; p.rope_dims = rope_dims > 0 ? rope_dims : D;
    return rst_launch_lm_rope_append(p, (hipStream_t)stream);
}

int rst_lm_attn_decode_f32(const float* qkv, void* k, void* v, float* ws, uint32_t* counters, float* out,
                           const int64_t* pos_dev, int B, int H, int D, int cap, int context, int splits, int ldqkv, int rope,
                           float rope_coef, int kv_heads, int rope_dims, uint16_t* out_packed, int kv_bf16, const float* rope_table,
                           rst_stream_t stream) {
    LmAttnParams p;
    p.kv_bf16 = kv_bf16; p.rope_cs = rope && !(cap <= 64 && splits == 1) ? rope_table : nullptr;
    p.q_pre = nullptr; p.T = 1; p.G = kv_heads > 0 ? kv_heads : H; p.rope_dims = rope_dims > 0 ? rope_dims : D;
    p.out_packed = out_packed; p.out_plane = (long)((B + 31) / 32 * 32) * H * D;
    p.qkv = qkv; p.k = k; p.v = v; p.ws = ws; p.counters = counters; p.out = out; p.pos_dev = (const long*)pos_dev;
    p.B = B; p.H = H; p.D = D; p.cap = cap; p.context = context; p.splits = splits; p.ldqkv = ldqkv; p.rope = rope;
    p.rope_coef = rope_coef;
    return rst_launch_lm_attn(p, (hipStream_t)stream);
}

int rst_lm_rope_table_f32(const int64_t* pos_dev, float* table, int D, int rope_dims, float rope_coef, rst_stream_t stream) {
    return rst_launch_lm_rope_table((const long*)pos_dev, table, D, rope_dims > 0 ? rope_dims : D, rope_coef, (hipStream_t)stream);
}

int rst_attn_decode_multi_f32(const float* q, const float* k, const float* v, float* ws, uint32_t* counters, float* out,
                              const int64_t* pos_dev, int B, int T, int H, int D, int cap, int context, int splits,
                              int kv_heads, rst_stream_t stream) {
    LmAttnParams p;
    p.kv_bf16 = 0; p.rope_cs = nullptr;
    p.G = kv_heads > 0 ? kv_heads : H; p.rope_dims = D; p.out_packed = nullptr; p.out_plane = 0;
    p.qkv = nullptr; p.q_pre = q; p.T = T; p.k = const_cast<float*>(k); p.v = const_cast<float*>(v); p.ws = ws; p.counters = counters;
    p.out = out; p.pos_dev = (const long*)pos_dev; p.B = B; p.H = H; p.D = D; p.cap = cap; p.context = context; p.splits = splits;
    p.ldqkv = 0; p.rope = 0; p.rope_coef = 0.f;
    return rst_launch_lm_attn(p, (hipStream_t)stream);
}

int rst_lm_sample_workspace_bytes(int B, int V, int top_k, int top_p_mode) {
    const long n = rst_lm_sample_workspace_bytes_impl(B, V, top_k, top_p_mode);
    return n > 0x7fffffffL ? -1 : (int)n;
}

int rst_lm_sample_f32(const float* logits, const float* noise, int64_t* tokens, int B, int V, int ld, int top_k,
                      int noise_stride, int tok_stride, int use_sampling, float temp, int v_limit, const int32_t* v_limit_dev,
                      float top_p, void* workspace, int64_t workspace_bytes, rst_stream_t stream) {
    LmSampleParams p;
    p.v_limit = v_limit; p.v_limit_dev = v_limit_dev; p.top_p = top_p; p.ws = workspace; p.ws_bytes = workspace ? workspace_bytes : 0;
    p.logits = logits; p.noise = noise; p.tokens = (long*)tokens; p.B = B; p.V = V; p.ld = ld; p.top_k = top_k;
    p.noise_stride = noise_stride; p.tok_stride = tok_stride; p.use_sampling = use_sampling; p.temp = temp;
    return rst_launch_lm_sample(p, (hipStream_t)stream);
}

int rst_lm_ring_begin_i64(int64_t* cache, const int64_t* user_tokens, const int64_t* initial, const int32_t* delays,
                          int64_t* offset_dev, int64_t* input_out, int B, int K, int CT, int Ki, int first_user_k, rst_stream_t stream) {
    LmRingParams p{};
    p.cache = (long*)cache; p.user = (const long*)user_tokens; p.initial = (const long*)initial; p.delays = delays;
    p.offset_dev = (long*)offset_dev; p.input_out = (long*)input_out; p.B = B; p.K = K; p.CT = CT; p.Ki = Ki; p.first_user = first_user_k;
    return rst_launch_lm_ring_begin(p, (hipStream_t)stream);
}

int rst_lm_ring_commit_i64(int64_t* cache, const int64_t* tokens, const int32_t* delays, int64_t* offset_dev, int64_t* out, int B, int K,
                           int CT, int n_out, int max_delay, rst_stream_t stream) {
    LmRingParams p{};
    p.cache = (long*)cache; p.tokens = (const long*)tokens; p.delays = delays; p.offset_dev = (long*)offset_dev; p.out = (long*)out;
    p.B = B; p.K = K; p.CT = CT; p.n_out = n_out; p.max_delay = max_delay;
    return rst_launch_lm_ring_commit(p, (hipStream_t)stream);
}

}  // extern "C"
