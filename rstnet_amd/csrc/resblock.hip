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
#include <cstdlib>
#include <type_traits>
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

// One tile.  FULL: every row of the X tile and of the output tile lies inside the utterance (all but the first / last tile of an
// utterance) -- no per-lane conditions, so no load or store sits in a branch of its own.  (A load under a per-lane condition is
// branched around and waited for on the spot, a conditional store whose value needs a load waits for every outstanding memory
// operation: the edge form of this kernel serialises ~9 HBM and ~32 L2 round trips plus its own stores per tile.)
template <int C, int BM, int WM, int WN, bool PRE, bool POST, bool FULL>
__device__ __forceinline__ void resblock_tile(const ResblockParams& p, float* smem, const long b, const int t0) {
    constexpr int H = C / 2;
    constexpr int XLD = C + 4, HLD = H + 4;
    constexpr int NT1 = H / 32 / WN;      // GEMM1 column tiles per wave
    constexpr int NT2 = C / 32 / WN;      // GEMM2 column tiles per wave
    constexpr int W1CH = H * 8 / 256;     // float4 chunks of a W1 k-tile per thread
    constexpr int W2CH = C * H / 4 / 256; // float4 chunks of W2 per thread
    constexpr int MAXK0 = 8;
    static_assert(BM == 32 * WM && WM * WN == 4 && NT1 >= 1 && W1CH >= 1 && W2CH >= 1, "tile config");
    static_assert(!POST || BM == 128, "the fused last conv maps two lanes to each of the BM output rows");

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
            const float v = p.x[b * T + min(max(t, 0), T - 1)];
            As[i] = (i < XR + K0 - 1 && (FULL || (t >= 0 && t < T))) ? v : 0.f;
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
                    if (rx < XR) Xs[rx * XLD + c] = (FULL || (t >= 0 && t < T)) ? rst_elu(v) : 0.f;
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
            f32x4 v;
            if (FULL) {
                v = *reinterpret_cast<const f32x4*>(p.x + (b * T + min(t, t0 + BM - 1)) * C + c4);     // rows past the tile: unused
            } else {
                // clamped address (x, or the streaming history for t < 0), zeroed afterwards where the row does not exist
                const bool from_hist = p.hist && t < 0 && t >= -(Kw - 1);
                const float* src = from_hist ? p.hist + (b * (Kw - 1) + (Kw - 1) + t) * C + c4
                                             : p.x + (b * T + min(max(t, 0), T - 1)) * C + c4;
                v = *reinterpret_cast<const f32x4*>(src);
                if (!(rx < XR && ((t >= 0 && t < T) || from_hist))) v = f32x4{0.f, 0.f, 0.f, 0.f};
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
    // Two accumulator chains per column tile (even / odd k-steps, summed in epilogue 1): with one, every MFMA waits for its
    // predecessor's 64-cycle result, and any instruction the wave issues in between (the fragment reads) adds to that.
    const int nk = Kw * C / BK;
    f32x16 acc1[NT1], acc1b[NT1];
#pragma unroll
    for (int j = 0; j < NT1; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc1[j][e] = 0.f; acc1b[j][e] = 0.f; }
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
                    for (int j = 0; j < NT1; ++j) {
                        if (e & 1) acc1b[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc1b[j], 0, 0, 0);
                        else acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc1[j], 0, 0, 0);
                    }
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
        for (int e = 0; e < 16; ++e) Hs[(wm * 32 + rst_mfma32_row(e, lane)) * HLD + col] = rst_elu((acc1[j][e] + acc1b[j][e]) + bias);
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
                xres[j][e] = p.x[(b * T + (FULL ? t : min(max(t, 0), T - 1))) * C + col];   // used only where the row exists
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
            const bool valid = FULL || (t >= 0 && t < T);
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
            if (q == 0 && r < BM && (FULL || (t >= 0 && t < T))) p.y[b * T + t] = s + p.bf[0];
        }
    }
}

template <int C, int BM, int WM, int WN, bool PRE, bool POST>
__global__ __launch_bounds__(256) void resblock_kernel(const ResblockParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int halo = POST ? p.Kf - 1 : 0;
    const int BMo = BM - halo;                         // output rows produced per tile
    const int tiles = (p.T + BMo - 1) / BMo;
    const long b = blockIdx.x / tiles;
    const int t0 = (blockIdx.x % tiles) * BMo - halo;  // time of tile row 0
    const bool full = t0 - (p.Kw - 1) - (PRE ? p.K0 - 1 : 0) >= 0 && t0 + BM <= p.T;
    if (full) resblock_tile<C, BM, WM, WN, PRE, POST, true>(p, smem, b, t0);
    else resblock_tile<C, BM, WM, WN, PRE, POST, false>(p, smem, b, t0);
}

template <int C, int BM, int WM, int WN, bool PRE, bool POST>
int launch(const ResblockParams& p, hipStream_t stream) {
    const Carve<C, BM> cv(p.Kw, PRE, POST);
    const size_t lds = (size_t)cv.total * sizeof(float);
    const int halo = POST ? p.Kf - 1 : 0;
    const long tiles = (long)p.B * ((p.T + (BM - halo) - 1) / (BM - halo));
    if (tiles > 0x7fffffffL) { rst_set_error("resblock: grid too large"); return RST_ERR_UNSUPPORTED; }
    static RstOncePerDevice attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_kernel<C, BM, WM, WN, PRE, POST>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL((resblock_kernel<C, BM, WM, WN, PRE, POST>), dim3((unsigned)tiles), dim3(256), lds, stream, p);
    return rst_check_launch("resblock");
}

// ---- C = 64 (24 kHz level), Kw = 3: resident workgroups, weights in registers -----------------------------------------------
// A tile of this level is only 128 MFMAs per wave (3.5 us of matrix pipe) between one HBM round trip for x, two weight stagings
// and ~10 barriers, so the launch-per-tile form above spends most of its time in exposed latencies.  Here a workgroup is
// resident (two per CU) and walks its tiles:
//   * W1 (32 x 192) is loaded ONCE into registers as MFMA B fragments (96 VGPRs) and W2 (64 x 32) once into LDS: no weight
//     traffic, no k-tile barriers -- GEMM1 is 24 A-fragment ds_reads + 96 MFMAs, GEMM2 12 reads + 32 MFMAs;
//   * every load of a tile is unconditional (interior tiles: no per-lane conditions at all; edge tiles: clamped addresses) so
//     they are all in flight at once -- a load under a per-lane condition is branched around and waited for on the spot;
//   * a wave owns 32 rows through both GEMMs (WN = 1), so the hidden tile is wave-private: two workgroup barriers per tile
//     (X tile staged / X tile free; the fused last conv parks ELU(y) in the freed X tile and adds one for its taps).
constexpr int RS_C = 64, RS_H = 32, RS_BM = 128, RS_KW = 3, RS_LD = 68, RS_HLD = 36, RS_XR = RS_BM + RS_KW - 1, RS_MAXK0 = 8, RS_MAXKF = 4;
constexpr int RS_XS = RS_XR * RS_LD;                 // ELU(x) tile; POST: ELU(y) tile after GEMM1
constexpr int RS_HS = RS_BM * RS_HLD;                // hidden tile
constexpr int RS_W2 = RS_C * RS_HLD;                 // W2 [C][H]
constexpr int RS_AS = RS_BM + 24;                    // PRE: audio samples of a tile
constexpr int RS_LDS_FLOATS = RS_XS + RS_HS + RS_W2 + RS_AS + RS_C * (RS_MAXK0 + 1) + RS_MAXKF * RS_C + RS_MAXKF * RS_BM;

template <bool PRE, bool POST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void resblock64_stream_kernel(const ResblockParams p, const int tiles_u, const int total) {
    constexpr int C = RS_C, LD = RS_LD, HLD = RS_HLD, BM = RS_BM, XR = RS_XR, KW = RS_KW;
    constexpr int XCH = (XR * (C / 4) + 255) / 256;     // float4 chunks of the x tile per thread (9)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Hs = Xs + RS_XS;
    float* W2s = Hs + RS_HS;
    float* As = W2s + RS_W2;
    float* W0s = As + RS_AS;                            // [C][MAXK0 + 1]
    float* Wfs = W0s + C * (RS_MAXK0 + 1);              // [Kf][C]
    float* Ds = Wfs + RS_MAXKF * C;                     // [Kf][BM] per-row tap dots of the fused last conv

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const int T = p.T;
    const int halo = POST ? p.Kf - 1 : 0;
    const int BMo = BM - halo;
    const int K0 = p.K0;

    // ---- weights for the lifetime of the workgroup: W1 as B fragments in registers, W2 in LDS
    f32x4 fb1[KW * C / 8];
#pragma unroll
    for (int ks = 0; ks < KW * C / 8; ++ks) fb1[ks] = *reinterpret_cast<const f32x4*>(p.w1 + frow * (KW * C) + ks * 8 + fk);
    for (int i = tid; i < C * RS_H / 4; i += 256)
        *reinterpret_cast<f32x4*>(W2s + (i >> 3) * HLD + (i & 7) * 4) = *reinterpret_cast<const f32x4*>(p.w2 + (size_t)i * 4);
    const float bias1 = p.b1[frow];
    const float bias2[2] = {p.b2[frow], p.b2[32 + frow]};
    if (PRE)
        for (int i = tid; i < C * K0; i += 256) W0s[(i / K0) * (RS_MAXK0 + 1) + i % K0] = p.w0[i];
    if (POST)
        for (int i = tid; i < p.Kf * C; i += 256) Wfs[i] = p.wf[i];

    // ---- input requests of a tile (registers): x rows t0-2 .. t0+127, or the audio samples the fused first conv needs
    f32x4 xv[PRE ? 1 : XCH];
    auto request = [&](int idx) {
        const long b = idx / tiles_u;
        const int t0 = (idx - (int)b * tiles_u) * BMo - halo;
        // Raw, unconditional loads from addresses clamped into the utterance; rows that do not exist are zeroed where the tile is
        // staged (after a barrier, so the compiler cannot sink the load back under the condition: a load under a per-lane condition
        // is waited for on the spot -- one exposed round trip per load).
        if (PRE) {
            const int t = t0 - (KW - 1) - (K0 - 1) + tid;
            xv[0][0] = p.x[b * T + min(max(t, 0), T - 1)];
        } else {
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int ch = tid + 256 * i;
                const int rx = ch >> 4, c4 = (ch & 15) * 4;
                const int t = t0 - (KW - 1) + rx;
                xv[i] = *reinterpret_cast<const f32x4*>(p.x + (b * T + min(max(t, 0), T - 1)) * C + c4);
            }
        }
    };

    // One tile.  FULL: every row of the X tile and of the output tile lies inside the utterance (all but the first and last tile
    // of an utterance) -- no per-lane conditions anywhere, so no branches around loads / stores and counted waits only.
    auto tile_body = [&](auto full_tag, const int idx, const long b, const int t0) {
        constexpr bool FULL = decltype(full_tag)::value;
        request(idx);       // all of the tile's loads in flight at once (the other resident workgroup covers the round trip)

        // ---- stage ELU(x) of this tile from the registers, then request the next tile's input
        if (PRE) {
            {
                const int t = t0 - (KW - 1) - (K0 - 1) + tid;
                if (tid < RS_AS) As[tid] = (tid < XR + K0 - 1 && (FULL || (t >= 0 && t < T))) ? xv[0][0] : 0.f;
            }
            __syncthreads();
            // conv0 + ELU: a thread owns one channel (weights in registers) and walks groups of 4 consecutive rows
            constexpr int G = 256 / C;
            const int c = tid & (C - 1), g = tid / C;
            float w0r[RS_MAXK0];
#pragma unroll
            for (int k = 0; k < RS_MAXK0; ++k) w0r[k] = k < K0 ? W0s[c * (RS_MAXK0 + 1) + k] : 0.f;
            const float b0r = p.b0[c];
            for (int rx0 = g * 4; rx0 < XR; rx0 += 4 * G) {
                float av[RS_MAXK0 + 3];
#pragma unroll
                for (int k = 0; k < RS_MAXK0 + 3; ++k) av[k] = As[rx0 + k];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rx = rx0 + q;
                    const int t = t0 - (KW - 1) + rx;
                    float v = b0r;
#pragma unroll
                    for (int k = 0; k < RS_MAXK0; ++k) v = fmaf(w0r[k], av[q + k], v);
                    if (rx < XR) Xs[rx * LD + c] = (FULL || (t >= 0 && t < T)) ? rst_elu(v) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int ch = tid + 256 * i;
                const int rx = ch >> 4, c4 = (ch & 15) * 4;
                const int t = t0 - (KW - 1) + rx;
                if (rx < XR) {
                    f32x4 v = xv[i];
                    v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]);
                    if (!FULL && (t < 0 || t >= T)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(Xs + rx * LD + c4) = v;
                }
            }
        }
        __syncthreads();                                 // X tile staged

        // ---- GEMM1: acc1[32 x 32] = Xwin[32 x 192] * W1^T
        f32x16 acc1, acc1b;                              // two chains (even / odd k-steps): see resblock_tile
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc1[e] = 0.f; acc1b[e] = 0.f; }
        {
            const float* a = Xs + (wave * 32 + frow) * LD + fk;
#pragma unroll
            for (int ks = 0; ks < KW * C / 8; ++ks) {
                const f32x4 fa = *reinterpret_cast<const f32x4*>(a + (ks / 8) * LD + (ks % 8) * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e & 1) acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb1[ks][e], acc1b, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb1[ks][e], acc1, 0, 0, 0);
                }
            }
        }
        // ---- epilogue 1: hidden rows of this wave (wave-private: no workgroup barrier)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            Hs[(wave * 32 + rst_mfma32_row(e, lane)) * HLD + frow] = rst_elu((acc1[e] + acc1b[e]) + bias1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- skip-connection operand in the row-major layout of the output pass below: lane -> (row it*8 + lane/8, 4 channels),
        // 16-byte loads requested now, landing under GEMM2.  (The accumulator layout gives a lane one column of 16 rows: 32 + 32
        // dword loads / stores per lane, and the address path of the CU -- not HBM -- becomes what the tile waits for.)
        const int orow = lane >> 3, oc4 = (lane & 7) * 4;
        f32x4 xres[2][4];
        if (!PRE) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int t = t0 + wave * 32 + it * 8 + orow;
                    xres[j][it] = *reinterpret_cast<const f32x4*>(p.x + (b * T + (FULL ? t : min(max(t, 0), T - 1))) * C + j * 32 + oc4);
                }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- GEMM2: acc2[32 x 64] = H[32 x 32] * W2^T
        f32x16 acc2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
        {
            const float* a = Hs + (wave * 32 + frow) * HLD + fk;
            const float* bw = W2s + frow * HLD + fk;
#pragma unroll
            for (int ks = 0; ks < RS_H / 8; ++ks) {
                const f32x4 fa = *reinterpret_cast<const f32x4*>(a + ks * 8);
                f32x4 fb[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(bw + j * 32 * HLD + ks * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc2[j], 0, 0, 0);
            }
        }

        // ---- epilogue 2: y = x + acc2 + b2, 32 columns at a time through this wave's (now dead) hidden rows: accumulator layout in,
        // row-major 16-byte pieces out (wave-private: no workgroup barrier)
        float dk[4][RS_MAXKF];                  // POST: per-row tap dots of the fused last conv, 8 lanes per row
        if (POST) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int k = 0; k < RS_MAXKF; ++k) dk[it][k] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + frow;
            float w0c[RS_MAXK0], awin[RS_MAXK0 + 3];
            float b0c = 0.f;
            if (PRE) {
                b0c = p.b0[col];
#pragma unroll
                for (int k = 0; k < RS_MAXK0; ++k) w0c[k] = k < K0 ? W0s[col * (RS_MAXK0 + 1) + k] : 0.f;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // GEMM2 reads / the previous pass's reads of these rows are done
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wave * 32 + rst_mfma32_row(e, lane);
                float v = acc2[j][e] + bias2[j];
                if (PRE) {                  // the skip operand is conv0(audio), recomputed from the audio tile
                    if ((e & 3) == 0) {     // rows e..e+3 of this lane are consecutive: one K0+3 window serves all four
#pragma unroll
                        for (int k = 0; k < RS_MAXK0 + 3; ++k) awin[k] = As[r + (KW - 1) + k];
                    }
                    float xr = b0c;
#pragma unroll
                    for (int k = 0; k < RS_MAXK0; ++k) xr = fmaf(w0c[k], awin[(e & 3) + k], xr);
                    v = xr + v;
                }
                Hs[r * HLD + frow] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = wave * 32 + it * 8 + orow;
                const int t = t0 + r;
                const bool valid = FULL || (t >= 0 && t < T);
                f32x4 v = *reinterpret_cast<const f32x4*>(Hs + r * HLD + oc4);
                if (!PRE) { v[0] = xres[j][it][0] + v[0]; v[1] = xres[j][it][1] + v[1]; v[2] = xres[j][it][2] + v[2]; v[3] = xres[j][it][3] + v[3]; }
                if (POST || p.elu_out) { v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]); }
                if (POST) {
                    if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < RS_MAXKF; ++k) {
                        if (k < p.Kf) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(Wfs + k * C + j * 32 + oc4);
                            dk[it][k] = fmaf(wv[0], v[0], dk[it][k]); dk[it][k] = fmaf(wv[1], v[1], dk[it][k]);
                            dk[it][k] = fmaf(wv[2], v[2], dk[it][k]); dk[it][k] = fmaf(wv[3], v[3], dk[it][k]);
                        }
                    }
                } else if (valid) {
                    *reinterpret_cast<f32x4*>(p.y + (b * T + t) * C + j * 32 + oc4) = v;
                }
            }
        }
        if (POST) {
            // final Conv1d C -> 1, kernel Kf: out[t] = bf + sum_k d_k[t - Kf + 1 + k],  d_k[r] = sum_c wf[k][c] * ELU(y[r][c]):
            // the 8 lanes of a row meet, then the taps meet across rows / waves through Ds
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int k = 0; k < RS_MAXKF; ++k) {
                    float d = dk[it][k];
                    d += __shfl_xor(d, 1);
                    d += __shfl_xor(d, 2);
                    d += __shfl_xor(d, 4);
                    if ((lane & 7) == 0 && k < p.Kf) Ds[k * BM + wave * 32 + it * 8 + orow] = d;
                }
            __syncthreads();                             // taps of all rows present; every wave is also past GEMM1: X tile free
            if (tid < BM && tid >= halo) {
                const int t = t0 + tid;
                float s = p.bf[0];
                for (int k = 0; k < p.Kf; ++k) s += Ds[k * BM + tid - halo + k];
                if (FULL || (t >= 0 && t < T)) p.y[b * T + t] = s;
            }
        } else {
            __syncthreads();                             // every wave is past GEMM1: the X tile may be overwritten
        }
    };

    for (int idx = blockIdx.x; idx < total; idx += gridDim.x) {
        const long b = idx / tiles_u;
        const int t0 = (idx - (int)b * tiles_u) * BMo - halo;
        const bool full = t0 - (KW - 1) - (PRE ? K0 - 1 : 0) >= 0 && t0 + BM <= T;
        if (full) tile_body(std::true_type{}, idx, b, t0);
        else tile_body(std::false_type{}, idx, b, t0);
    }
}

template <bool PRE, bool POST>
int launch64_stream(const ResblockParams& p, hipStream_t stream) {
    const int halo = POST ? p.Kf - 1 : 0;
    const long tiles_u = (p.T + (RS_BM - halo) - 1) / (RS_BM - halo);
    const long total = (long)p.B * tiles_u;
    if (total > 0x7fffffffL) { rst_set_error("resblock: grid too large"); return RST_ERR_UNSUPPORTED; }
    const int cus = rst_cu_count();
    static RstOncePerDevice attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock64_stream_kernel<PRE, POST>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const long resident = 2L * cus;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    hipLaunchKernelGGL((resblock64_stream_kernel<PRE, POST>), dim3(grid), dim3(256), RS_LDS_FLOATS * sizeof(float), stream, p,
                       (int)tiles_u, (int)total);
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
    // tools build only -- RST_RESBLOCK_STREAM=0: the launch-per-tile form for every shape (A/B measurements)
    static const bool stream_off = rst_knob("RST_RESBLOCK_STREAM", 1) == 0;
    // the fused last conv measured faster in the launch-per-tile form (0.92 vs 1.00 ms at 16 x 10 s): three workgroups per CU
    // against two, and its output is one float per row either way; RST_RESBLOCK_STREAM=2 forces the resident form for it too
    static const bool stream_post = rst_knob("RST_RESBLOCK_STREAM", 1) == 2;
    if (p.C == 64 && p.Kw == RS_KW && !p.hist && !stream_off && !(p.pre && p.post) && (!p.post || stream_post)) {
        if (p.pre) return launch64_stream<true, false>(p, stream);
        if (p.post) return launch64_stream<false, true>(p, stream);
        return launch64_stream<false, false>(p, stream);
    }
    if (p.C == 64) {
        if (p.pre && p.post) return launch<64, 128, 4, 1, true, true>(p, stream);
        if (p.pre) return launch<64, 128, 4, 1, true, false>(p, stream);
        if (p.post) return launch<64, 128, 4, 1, false, true>(p, stream);
        return launch<64, 128, 4, 1, false, false>(p, stream);
    }
    return launch<128, 64, 2, 2, false, false>(p, stream);
}
