// Row-wise LayerNorm, batched transpose (layout adapters), depth-wise transposed convolution (the 25 Hz -> 12.5 Hz
// ConvTrUpsample1d) and the streaming history roll.  All HBM-bound: one pass, 16-byte accesses where the shape allows.
#include "rst_common.h"
#include "rst_kernels.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wave per row; two passes over the row (second one hits L1/L2), fp32 statistics like ATen's CPU LayerNorm
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float* yr = y + row * D;
    float s = 0.f;
    if ((D & 3) == 0) {
        for (int i = lane * 4; i < D; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        for (int i = lane; i < D; i += 64) s += xr[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    if ((D & 3) == 0) {
        for (int i = lane * 4; i < D; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
        }
    } else {
        for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; q = fmaf(d, d, q); }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    if ((D & 3) == 0) {
        for (int i = lane * 4; i < D; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * g[e] + b[e];
            *reinterpret_cast<f32x4*>(yr + i) = o;
        }
    } else {
        for (int i = lane; i < D; i += 64) yr[i] = (xr[i] - mean) * rstd * gamma[i] + beta[i];
    }
}

// [B][R][C] -> [B][C][R] through a padded 32x32 LDS tile
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C) {
    __shared__ float tile[32][33];
    const long b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* xb = x + b * (long)R * C;
    float* yb = y + b * (long)R * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        if (r < R && c < C) tile[ty + 8 * j][tx] = xb[(long)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (r < R && c < C) yb[(long)c * R + r] = tile[tx][ty + 8 * j];
    }
}

// depth-wise ConvTranspose1d, kernel K = q*S, stride S, causal (right side trimmed):
//   y[b, t*S + j, c] = sum_{i<q} x[b, t-i, c] * w[c][j + i*S]       (x[t<0] from hist [B][q-1][C] or 0)
__global__ __launch_bounds__(256) void convtr_depthwise_kernel(const float* __restrict__ x, const float* __restrict__ hist,
                                                              const float* __restrict__ w, float* __restrict__ y,
                                                              int B, int T_in, int C, int K, int S) {
    const int q = (K + S - 1) / S;
    const long total = (long)B * T_in * S * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const long to = idx / C;                 // b*T_in*S + t*S + j
        const int j = (int)(to % S);
        const long bt = to / S;
        const int t = (int)(bt % T_in);
        const long b = bt / T_in;
        float acc = 0.f;
        for (int i = 0; i < q; ++i) {
            const int kk = j + i * S;
            if (kk >= K) break;
            const int ti = t - i;
            float xv = 0.f;
            if (ti >= 0) xv = x[(b * T_in + ti) * C + c];
            else if (hist) xv = hist[(b * (q - 1) + (q - 1 + ti)) * C + c];
            acc = fmaf(xv, w[(long)c * K + kk], acc);
        }
        y[idx] = acc;
    }
}

__global__ __launch_bounds__(256) void hist_update_kernel(const float* __restrict__ x, const float* __restrict__ hin,
                                                         float* __restrict__ hout, int B, int T_in, int P_in, int P_out,
                                                         int C) {
    const long total = (long)B * P_out * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int pr = (int)((idx / C) % P_out);
        const long b = idx / ((long)C * P_out);
        const int src = pr + T_in - P_out;  // row index in x; negative -> old history row (P_in + src)
        hout[idx] = src >= 0 ? x[(b * T_in + src) * C + c] : hin[(b * P_in + (P_in + src)) * C + c];
    }
}

// In-place form (hout == hin, P_in == P_out): one workgroup per batch row reads every element of the new history into
// registers, then -- after a barrier -- writes them; rows are disjoint buffers, so no other hazard exists.
constexpr int HIST_INPLACE_MAX = 16;   // elements per thread at 1024 threads -> P * C <= 16384
__global__ __launch_bounds__(1024) void hist_update_inplace_kernel(const float* __restrict__ x, float* h, int T_in, int P, int C) {
    const long b = blockIdx.x;
    const int E = P * C;
    float v[HIST_INPLACE_MAX];
#pragma unroll
    for (int j = 0; j < HIST_INPLACE_MAX; ++j) {
        const int e = threadIdx.x + j * 1024;
        v[j] = 0.f;
        if (e < E) {
            const int c = e % C, pr = e / C;
            const int src = pr + T_in - P;
            v[j] = src >= 0 ? x[(b * T_in + src) * C + c] : h[(b * P + (P + src)) * C + c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < HIST_INPLACE_MAX; ++j) {
        const int e = threadIdx.x + j * 1024;
        if (e < E) h[b * (long)E + e] = v[j];
    }
}

// The in-place roll of up to RST_HIST_BATCH_MAX histories in ONE launch (blockIdx.y = entry): a streaming codec step rolls the
// input history of every convolution; as separate launches those are ~20 dependent 4-5 us kernels per frame.
__global__ __launch_bounds__(1024) void hist_update_batch_kernel(const HistBatchParams p) {
    const int en = blockIdx.y;
    const float* __restrict__ x = p.x[en];
    float* h = p.hist[en];
    const int T_in = p.T_in[en], P = p.P[en], C = p.C[en];
    const long b = blockIdx.x;
    const int E = P * C;
    float v[HIST_INPLACE_MAX];
#pragma unroll
    for (int j = 0; j < HIST_INPLACE_MAX; ++j) {
        const int e = threadIdx.x + j * 1024;
        v[j] = 0.f;
        if (e < E) {
            const int c = e % C, pr = e / C;
            const int src = pr + T_in - P;
            v[j] = src >= 0 ? x[(b * T_in + src) * C + c] : h[(b * P + (P + src)) * C + c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < HIST_INPLACE_MAX; ++j) {
        const int e = threadIdx.x + j * 1024;
        if (e < E) h[b * (long)E + e] = v[j];
    }
}

// Rows t >= len[b] of x [B][T][C] become zero (mode 0) or a copy of row len[b] - 1 (mode 1).
__global__ __launch_bounds__(256) void mask_tail_kernel(float* __restrict__ x, const int* __restrict__ len, int B, int T, int C, int mode) {
    const long total = (long)B * T * (C / 4);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % (C / 4)) * 4;
        const int t = (int)((idx / (C / 4)) % T);
        const long b = idx / ((long)(C / 4) * T);
        const int L = len[b];
        if (t < L) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (mode == 1 && L > 0) v = *reinterpret_cast<const f32x4*>(x + (b * T + L - 1) * C + c4);
        *reinterpret_cast<f32x4*>(x + (b * T + t) * C + c4) = v;
    }
}

__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[i] = act == 1 ? rst_elu(x[i]) : (act == 2 ? rst_gelu(x[i]) : x[i]);
}

inline unsigned grid_for(long total) {
    long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

int rst_launch_layernorm(const float* x, const float* gamma, const float* beta, float* y, long rows, int D, float eps,
                         hipStream_t stream) {
    RST_REQUIRE(D > 0 && rows >= 0, "layernorm: bad sizes");
    if (rows == 0) return RST_OK;
    RST_REQUIRE(x && gamma && beta && y, "layernorm: null pointer");
    RST_REQUIRE((rows + 3) / 4 < 0x7fffffffL, "layernorm: too many rows");
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, gamma, beta, y, rows, D, eps);
    return rst_check_launch("layernorm");
}

int rst_launch_transpose(const float* x, float* y, int B, int R, int C, hipStream_t stream) {
    RST_REQUIRE(B >= 0 && R >= 0 && C >= 0, "transpose: bad sizes");
    if (B == 0 || R == 0 || C == 0) return RST_OK;
    RST_REQUIRE(x && y, "transpose: null pointer");
    RST_REQUIRE(B <= 65535 && (R + 31) / 32 <= 65535, "transpose: grid too large (B=%d R=%d)", B, R);
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, B), dim3(256), 0, stream, x, y, R, C);
    return rst_check_launch("transpose");
}

int rst_launch_convtr_depthwise(const float* x, const float* hist, const float* w, float* y, int B, int T_in, int C,
                                int K, int S, hipStream_t stream) {
    RST_REQUIRE(B >= 0 && T_in >= 0 && C > 0 && K > 0 && S > 0 && S <= K, "convtr_depthwise: bad sizes");
    const long total = (long)B * T_in * S * C;
    if (total == 0) return RST_OK;
    RST_REQUIRE(x && w && y, "convtr_depthwise: null pointer");
    hipLaunchKernelGGL(convtr_depthwise_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, hist, w, y, B, T_in, C, K, S);
    return rst_check_launch("convtr_depthwise");
}

int rst_launch_hist_update(const float* x, const float* hist_in, float* hist_out, int B, int T_in, int P_in, int P_out,
                           int C, hipStream_t stream) {
    RST_REQUIRE(B >= 0 && T_in >= 0 && P_in >= 0 && P_out >= 0 && C > 0 && P_out <= P_in + T_in,
                "hist_update: bad sizes (T_in=%d P_in=%d P_out=%d)", T_in, P_in, P_out);
    const long total = (long)B * P_out * C;
    if (total == 0) return RST_OK;
    RST_REQUIRE(hist_out && (x || T_in == 0) && (hist_in || P_in == 0), "hist_update: null pointer");
    if (hist_out == hist_in) {
        RST_REQUIRE(P_in == P_out && (long)P_out * C <= 1024L * HIST_INPLACE_MAX,
                    "hist_update: in-place update needs P_in == P_out and P * C <= %d (P=%d C=%d)", 1024 * HIST_INPLACE_MAX, P_out, C);
        hipLaunchKernelGGL(hist_update_inplace_kernel, dim3(B), dim3(1024), 0, stream, x, hist_out, T_in, P_out, C);
        return rst_check_launch("hist_update_inplace");
    }
    hipLaunchKernelGGL(hist_update_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, hist_in, hist_out, B, T_in, P_in,
                       P_out, C);
    return rst_check_launch("hist_update");
}

int rst_launch_hist_update_batch(const HistBatchParams& p, hipStream_t stream) {
    RST_REQUIRE(p.n >= 0 && p.n <= RST_HIST_BATCH_MAX && p.B >= 0, "hist_update_batch: %d entries (max %d)", p.n, RST_HIST_BATCH_MAX);
    if (p.n == 0 || p.B == 0) return RST_OK;
    for (int i = 0; i < p.n; ++i)
        RST_REQUIRE(p.x[i] && p.hist[i] && p.T_in[i] >= 0 && p.P[i] > 0 && p.C[i] > 0 && (long)p.P[i] * p.C[i] <= 1024L * HIST_INPLACE_MAX,
                    "hist_update_batch: entry %d (T_in=%d P=%d C=%d; P * C <= %d)", i, p.T_in[i], p.P[i], p.C[i], 1024 * HIST_INPLACE_MAX);
    hipLaunchKernelGGL(hist_update_batch_kernel, dim3(p.B, p.n), dim3(1024), 0, stream, p);
    return rst_check_launch("hist_update_batch");
}

int rst_launch_mask_tail(float* x, const int* lengths, int B, int T, int C, int mode, hipStream_t stream) {
    RST_REQUIRE(B >= 0 && T >= 0 && C > 0 && C % 4 == 0 && (mode == 0 || mode == 1), "mask_tail: bad arguments (C %% 4 == 0 required, C=%d)", C);
    const long total = (long)B * T * (C / 4);
    if (total == 0) return RST_OK;
    RST_REQUIRE(x && lengths, "mask_tail: null pointer");
    hipLaunchKernelGGL(mask_tail_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, lengths, B, T, C, mode);
    return rst_check_launch("mask_tail");
}

int rst_launch_act(const float* x, float* y, long n, int act, hipStream_t stream) {
    RST_REQUIRE(n >= 0 && act >= 0 && act <= 2, "act: bad arguments");
    if (n == 0) return RST_OK;
    RST_REQUIRE(x && y, "act: null pointer");
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n, act);
    return rst_check_launch("act");
}
