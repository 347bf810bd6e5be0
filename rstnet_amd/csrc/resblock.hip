// Fused SEANet residual block for the HBM-bound high-rate levels (C = 64 @ 24 kHz, C = 128 @ 6 kHz):
//
//     y[t] = x[t] + b2 + W2 * ELU( b1 + W1 * [ELU(x[t-Kw+1]) .. ELU(x[t])] )        (modules/seanet.py:21-94)
//
// in ONE launch: the ELU'd input tile (BM + Kw - 1 rows, all C channels) is staged once in LDS, the hidden
// activation (C/2 channels) never leaves LDS, and HBM sees exactly one read of x and one write of y.
// Optional fusions at the two ends of the codec (template flags):
//   PRE : x is not read but computed on the fly as conv0(audio) (Conv1d 1 -> C, kernel K0: encoder.model.0), so
//         the 24 kHz C-channel tensor is written once and never read back by this level;
//   POST: y is not written; the tile goes through ELU and the final Conv1d C -> 1 (kernel Kf: decoder.model.14)
//         and only the mono waveform is written (tiles overlap by Kf-1 recomputed rows).
// Both GEMMs run on v_mfma_f32_32x32x2_f32 with the b128 k-permutation of gemm_win.hip; W1 streams through a
// double-buffered LDS ring, W2 is prefetched into registers during GEMM1 and parked in the LDS bytes of the
// (then dead) input tile.
#include "rst_common.h"
#include "rst_kernels.h"

namespace {

constexpr int BK = 32;
constexpr int WLD = BK + 4;

// LDS carve-up (floats), shared by kernel and launcher.  Region R0 holds the ELU'd input tile during GEMM1 and is
// recycled afterwards (W2, and the hidden tile when it fits behind W2; POST: the output tile).
template <int C, int BM>
struct Carve {
    static constexpr int H = C / 2, XLD = C + 4, HLD = H + 4;
    int r0, hs_off, w1_off, as_off, w0_off, wf_off, total;
    __host__ __device__ Carve(int Kw, bool pre, bool post) {
        const int xt = (BM + Kw - 1) * XLD, w2 = C * HLD, hs = BM * HLD;
        const bool hs_in_r0 = xt >= w2 + hs;
        r0 = xt > w2 ? xt : w2;
        const int w1 = 2 * H * WLD;        // W1 streams through a double-buffered ring of BK-wide k-tiles
        w1_off = r0;
        // the hidden tile lives behind W2 inside R0 if there is room, else over the (dead) W1 ring, else on its own
        hs_off = hs_in_r0 ? w2 : (w1 >= hs ? w1_off : r0 + w1);
        const int end = (hs_in_r0 || w1 >= hs) ? r0 + w1 : r0 + w1 + hs;
        as_off = end;
        w0_off = as_off + (pre ? BM + 24 : 0);
        wf_off = w0_off + (pre ? C * 9 : 0);
        total = wf_off + (post ? 4 * C : 0);
    }
};

template <int C, int BM, int WM, int WN, bool PRE, bool POST>
__global__ __launch_bounds__(256) void resblock_kernel(const ResblockParams p) {
    constexpr int H = C / 2;
    constexpr int XLD = C + 4, HLD = H + 4;
    constexpr int NT1 = H / 32 / WN;      // GEMM1 column tiles per wave
    constexpr int NT2 = C / 32 / WN;      // GEMM2 column tiles per wave
    constexpr int W1CH = H * 8 / 256;     // float4 chunks of a W1 k-tile per thread
    constexpr int W2CH = C * H / 4 / 256; // float4 chunks of W2 per thread
    constexpr int MAXK0 = 8;
    static_assert(BM == 32 * WM && WM * WN == 4 && NT1 >= 1 && W1CH >= 1 && W2CH >= 1, "tile config");
    static_assert(!POST || BM == 128, "the fused last conv maps two lanes to each of the BM output rows");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Carve<C, BM> cv(p.Kw, PRE, POST);
    float* Xs = smem;                    // [(BM+Kw-1)][XLD] ELU(x); later W2s [C][HLD] (+ Hs); later (POST) Ys [BM][XLD]
    float* Hs = smem + cv.hs_off;        // [BM][HLD]
    float* W1s = smem + cv.w1_off;       // ring [2][H][WLD]
    float* As = smem + cv.as_off;        // PRE: audio tile [BM + Kw-1 + K0-1]
    float* W0s = smem + cv.w0_off;       // PRE: [C][MAXK0+1]
    float* Wfs = smem + cv.wf_off;       // POST: [Kf][C]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int Kw = p.Kw, T = p.T;
    const int halo = POST ? p.Kf - 1 : 0;
    const int BMo = BM - halo;                         // output rows produced per tile
    const int tiles = (T + BMo - 1) / BMo;
    const long b = blockIdx.x / tiles;
    const int t0 = (blockIdx.x % tiles) * BMo - halo;  // time of tile row 0
    const int XR = BM + Kw - 1;

    // ---- W2 prefetch (lands during GEMM1)
    f32x4 w2r[W2CH];
#pragma unroll
    for (int i = 0; i < W2CH; ++i) w2r[i] = *reinterpret_cast<const f32x4*>(p.w2 + (size_t)(tid + 256 * i) * 4);

    if (POST)
        for (int i = tid; i < p.Kf * C; i += 256) Wfs[i] = p.wf[i];

    // ---- phase 0: stage ELU(x) tile
    if (PRE) {
        const int K0 = p.K0;
        const int tA0 = t0 - (Kw - 1) - (K0 - 1);
        for (int i = tid; i < BM + 24; i += 256) {   // the whole carve (incl. the tail read with zero weights) is defined
            const int t = tA0 + i;
            As[i] = (i < XR + K0 - 1 && t >= 0 && t < T) ? p.x[b * T + t] : 0.f;
        }
        for (int i = tid; i < C * K0; i += 256) W0s[(i / K0) * (MAXK0 + 1) + i % K0] = p.w0[i];
        __syncthreads();
        {   // conv0 + ELU: a thread owns one channel (weights in registers) and walks groups of 4 consecutive rows, so the
            // 4 x K0 products need only K0+3 (wave-uniform, broadcast) LDS reads of the audio tile
            constexpr int G = 256 / C;
            const int c = tid & (C - 1), g = tid / C;
            float w0r[MAXK0];
#pragma unroll
            for (int k = 0; k < MAXK0; ++k) w0r[k] = k < K0 ? W0s[c * (MAXK0 + 1) + k] : 0.f;
            const float b0r = p.b0[c];
            for (int rx0 = g * 4; rx0 < XR; rx0 += 4 * G) {
                float av[MAXK0 + 3];
#pragma unroll
                for (int k = 0; k < MAXK0 + 3; ++k) av[k] = As[rx0 + k];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rx = rx0 + q;
                    const int t = t0 - (Kw - 1) + rx;
                    float v = b0r;
#pragma unroll
                    for (int k = 0; k < MAXK0; ++k) v = fmaf(w0r[k], av[q + k], v);
                    if (rx < XR) Xs[rx * XLD + c] = (t >= 0 && t < T) ? rst_elu(v) : 0.f;
                }
            }
        }
    } else {
        // all global loads of the tile are issued before the first one is consumed (one exposed HBM latency, not nine)
        constexpr int XCH = ((BM + 3) * (C / 4) + 255) / 256;
        f32x4 xv[XCH];
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int idx = tid + 256 * i;
            const int rx = idx / (C / 4), c4 = (idx - rx * (C / 4)) * 4;
            const int t = t0 - (Kw - 1) + rx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (rx < XR) {
                if (t >= 0) {
                    if (t < T) v = *reinterpret_cast<const f32x4*>(p.x + (b * T + t) * C + c4);
                } else if (p.hist && t >= -(Kw - 1)) {
                    v = *reinterpret_cast<const f32x4*>(p.hist + (b * (Kw - 1) + (Kw - 1) + t) * C + c4);
                }
            }
            xv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int idx = tid + 256 * i;
            const int rx = idx / (C / 4), c4 = (idx - rx * (C / 4)) * 4;
            if (rx < XR) {
                f32x4 v = xv[i];
                v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]);
                *reinterpret_cast<f32x4*>(Xs + rx * XLD + c4) = v;
            }
        }
    }

    // ---- phase 1: GEMM1  acc1[BM x H] = Xwin[BM x Kw*C] * W1^T
    const int nk = Kw * C / BK;
    f32x16 acc1[NT1];
#pragma unroll
    for (int j = 0; j < NT1; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    {
        f32x4 w1r[W1CH];
        auto load_w1 = [&](int kt) {
#pragma unroll
            for (int i = 0; i < W1CH; ++i) {
                const int ch = tid + 256 * i;   // row = ch / 8, k4 = ch % 8
                w1r[i] = *reinterpret_cast<const f32x4*>(p.w1 + (size_t)(ch >> 3) * (Kw * C) + kt * BK + (ch & 7) * 4);
            }
        };
        auto store_w1 = [&](int buf) {
#pragma unroll
            for (int i = 0; i < W1CH; ++i) {
                const int ch = tid + 256 * i;
                *reinterpret_cast<f32x4*>(W1s + buf * H * WLD + (ch >> 3) * WLD + (ch & 7) * 4) = w1r[i];
            }
        };
        load_w1(0);
        store_w1(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_w1(kt + 1);
            const int tap = (kt * BK) / C, ci0 = (kt * BK) % C;
            const float* a = Xs + (wm * 32 + frow + tap) * XLD + ci0 + fk;
            const float* bw = W1s + (kt & 1) * H * WLD + (wn * NT1 * 32 + frow) * WLD + fk;
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                const f32x4 fa = *reinterpret_cast<const f32x4*>(a + ks * 8);
                f32x4 fb[NT1];
#pragma unroll
                for (int j = 0; j < NT1; ++j) fb[j] = *reinterpret_cast<const f32x4*>(bw + j * 32 * WLD + ks * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < NT1; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc1[j], 0, 0, 0);
            }
            if (kt + 1 < nk) store_w1((kt + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue 1: Hs = ELU(acc1 + b1);  W2 registers -> LDS over the dead X tile
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
        const int col = (wn * NT1 + j) * 32 + (lane & 31);
        const float bias = p.b1[col];
#pragma unroll
        for (int e = 0; e < 16; ++e) Hs[(wm * 32 + rst_mfma32_row(e, lane)) * HLD + col] = rst_elu(acc1[j][e] + bias);
    }
    float* W2s = Xs;
#pragma unroll
    for (int i = 0; i < W2CH; ++i) {
        const int ch = tid + 256 * i;           // row = ch / (H/4), k4 = ch % (H/4)
        *reinterpret_cast<f32x4*>(W2s + (ch / (H / 4)) * HLD + (ch % (H / 4)) * 4) = w2r[i];
    }
    __syncthreads();

    // skip-connection operand: issue the global loads now, they land under GEMM2's MFMAs
    float xres[NT2][16];
    if (!PRE) {
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int col = (wn * NT2 + j) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = t0 + wm * 32 + rst_mfma32_row(e, lane);
                xres[j][e] = (t >= 0 && t < T) ? p.x[(b * T + t) * C + col] : 0.f;
            }
        }
    }

    // ---- phase 2: GEMM2  acc2[BM x C] = Hs[BM x H] * W2^T
    f32x16 acc2[NT2];
#pragma unroll
    for (int j = 0; j < NT2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
    {
        const float* a = Hs + (wm * 32 + frow) * HLD + fk;
        const float* bw = W2s + (wn * NT2 * 32 + frow) * HLD + fk;
#pragma unroll
        for (int ks = 0; ks < H / 8; ++ks) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(a + ks * 8);
            f32x4 fb[NT2];
#pragma unroll
            for (int j = 0; j < NT2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(bw + j * 32 * HLD + ks * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT2; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc2[j], 0, 0, 0);
        }
    }

    // ---- epilogue 2: y = x + acc2 + b2
    if (POST) __syncthreads();  // every wave is done with W2s before the tile is overwritten with Ys
#pragma unroll
    for (int j = 0; j < NT2; ++j) {
        const int col = (wn * NT2 + j) * 32 + (lane & 31);
        const float bias = p.b2[col];
        float w0c[MAXK0], awin[MAXK0 + 3];
        float b0c = 0.f;
        if (PRE) {
            b0c = p.b0[col];
#pragma unroll
            for (int k = 0; k < MAXK0; ++k) w0c[k] = k < p.K0 ? W0s[col * (MAXK0 + 1) + k] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = wm * 32 + rst_mfma32_row(e, lane);
            const int t = t0 + r;
            const bool valid = t >= 0 && t < T;
            float xr = 0.f;
            if (PRE) {
                if ((e & 3) == 0) {   // rows e..e+3 of this lane are consecutive: one K0+3 window serves all four
#pragma unroll
                    for (int k = 0; k < MAXK0 + 3; ++k) awin[k] = As[r + (Kw - 1) + k];
                }
                xr = b0c;
#pragma unroll
                for (int k = 0; k < MAXK0; ++k) xr = fmaf(w0c[k], awin[(e & 3) + k], xr);
                if (!valid) xr = 0.f;
            } else {
                xr = xres[j][e];
            }
            const float yv = xr + (acc2[j][e] + bias);
            if (POST) {
                Xs[r * XLD + col] = valid ? rst_elu(yv) : 0.f;
            } else if (valid) {
                p.y[(b * T + t) * C + col] = p.elu_out ? rst_elu(yv) : yv;
            }
        }
    }
    if (POST) {
        // final Conv1d C -> 1, kernel Kf: out[t] = bf + sum_{k,c} wf[k][c] * ELU(y[t-Kf+1+k][c]).
        // Four lanes per output row, each reducing C/4 channels with 16-byte LDS reads (conflict-free: 4 rows x 4 quarters
        // of a 16-lane group fall into 16 distinct slots), two passes of 64 rows.
        __syncthreads();
        const int Kf = p.Kf;
        const int q = tid & 3;
#pragma unroll
        for (int pass = 0; pass < BM / 64; ++pass) {
            const int r = halo + pass * 64 + (tid >> 2);
            float s = 0.f;
            if (r < BM) {
                for (int k = 0; k < Kf; ++k) {
                    const float* yr = Xs + (r - (Kf - 1) + k) * XLD + q * (C / 4);
                    const float* wk = Wfs + k * C + q * (C / 4);
#pragma unroll
                    for (int c4 = 0; c4 < C / 16; ++c4) {
                        const f32x4 yv = *reinterpret_cast<const f32x4*>(yr + 4 * c4);
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wk + 4 * c4);
                        s = fmaf(wv[0], yv[0], s); s = fmaf(wv[1], yv[1], s); s = fmaf(wv[2], yv[2], s); s = fmaf(wv[3], yv[3], s);
                    }
                }
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            const int t = t0 + r;
            if (q == 0 && r < BM && t >= 0 && t < T) p.y[b * T + t] = s + p.bf[0];
        }
    }
}

template <int C, int BM, int WM, int WN, bool PRE, bool POST>
int launch(const ResblockParams& p, hipStream_t stream) {
    const Carve<C, BM> cv(p.Kw, PRE, POST);
    const size_t lds = (size_t)cv.total * sizeof(float);
    const int halo = POST ? p.Kf - 1 : 0;
    const long tiles = (long)p.B * ((p.T + (BM - halo) - 1) / (BM - halo));
    if (tiles > 0x7fffffffL) { rst_set_error("resblock: grid too large"); return RST_ERR_UNSUPPORTED; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_kernel<C, BM, WM, WN, PRE, POST>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((resblock_kernel<C, BM, WM, WN, PRE, POST>), dim3((unsigned)tiles), dim3(256), lds, stream, p);
    return rst_check_launch("resblock");
}

}  // namespace

bool rst_resblock_supported(int C, int H, int Kw, int pre, int post, int K0, int Kf) {
    if (H * 2 != C || Kw < 1 || Kw > 4) return false;
    if (pre && (K0 < 1 || K0 > 8)) return false;
    if (post && (Kf < 1 || Kf > 4)) return false;
    if (C == 64) return true;
    if (C == 128) return !pre && !post;
    return false;
}

int rst_launch_resblock(const ResblockParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 0 && p.T >= 0, "resblock: bad sizes");
    if (p.B == 0 || p.T == 0) return RST_OK;
    RST_REQUIRE(p.x && p.w1 && p.b1 && p.w2 && p.b2 && p.y, "resblock: null pointer");
    RST_REQUIRE(rst_resblock_supported(p.C, p.H, p.Kw, p.pre, p.post, p.K0, p.Kf),
                "resblock: unsupported shape C=%d H=%d Kw=%d pre=%d post=%d", p.C, p.H, p.Kw, p.pre, p.post);
    RST_REQUIRE(!p.pre || (p.w0 && p.b0), "resblock: PRE needs w0/b0");
    RST_REQUIRE(!p.post || (p.wf && p.bf), "resblock: POST needs wf/bf");
    RST_REQUIRE(!(p.hist && (p.pre || p.post)), "resblock: streaming history is only supported by the plain variant");
    if (p.C == 64) {
        if (p.pre && p.post) return launch<64, 128, 4, 1, true, true>(p, stream);
        if (p.pre) return launch<64, 128, 4, 1, true, false>(p, stream);
        if (p.post) return launch<64, 128, 4, 1, false, true>(p, stream);
        return launch<64, 128, 4, 1, false, false>(p, stream);
    }
    return launch<128, 64, 2, 2, false, false>(p, stream);
}
