// MEASUREMENT PROBE (round 4) -- not part of librstnet_hip.so.  The design below (activations pre-split into three bf16 planes in HBM, both
// operands HBM -> LDS by DMA, no split / staging work in the K loop) was built, passed its fp64 parity tests on the GPU, and measured
// SLOWER than gemm_win_b3 on every shape of the headline step (profiles/r04_p3_vs_b3.txt: 13 shapes, 19.2 vs 17.8 ms; with the plane
// epilogue the next GEMM would need, 23.1 ms) -- see DESIGN.md section 6.  Kept for the record with tools/probes/bench_p3.py; to rebuild it, add
// it to csrc/Makefile and restore the C-ABI entries from git history (commit "p3 probe").
//
// Windowed GEMM on PRE-SPLIT operands ("p3"): activations arrive as three bf16 planes in HBM, written once by their producer, and both
// operands go HBM -> LDS by DMA (buffer_load ... lds).  The K loop of the large Conv1d / ConvTranspose1d / Linear launches then holds
// no VALU work, no staging registers and no LDS-write instructions at all: per 16-wide k-tile and wave it issues 6 DMA pieces, 12
// ds_read_b128 and 24 matrix instructions.
//
// Why: gemm_win_b3 (activations fp32 in HBM, split on their way into LDS) re-splits every activation element once per column tile
// and per window tap that touches it -- 2 .. 8 times -- and its ablations price that work (4.5 VALU per element + the LDS writes,
// issued between the matrix instructions of the same wave) at 193 -> 276 TFLOP/s fp32-equivalent (profiles/r03_b3_ablation.txt).
// Splitting ONCE, in the producer's epilogue, removes it; what is left of the loop is what that ablation measured.
//
// Planes in HBM: uint16 [3][B][Tp][C], Tp = T + 2 * pad rows per utterance, channels-last like the fp32 activations, so a convolution
// is still a GEMM over overlapping row windows of each plane; the `pad` rows in front of and behind every utterance are ZERO (written
// once when the buffer is made, never by a producer): a window that reaches into the causal zero padding or past the end reads zeros by
// itself -- no masks, no edge tiles.  (hi, mid, lo) = b3_common.h's exact three-way split.
//
// LDS image of a k-tile (16 k): per operand and plane 4 pieces of 1 KB = 32 rows x 32 bytes; a DMA instruction writes wave-uniform
// base + lane * 16, so the image is linear and the bank-conflict swizzle sits in the SOURCE address: chunk c = 2 r + (h ^ ((r >> 3) & 1))
// of a piece holds k-half h of its row r (the 16 rows of a ds_read_b128 group then cover 16 distinct slots).  The weights are
// pre-packed in exactly that image (rst_gemm_p3_pack_weight), so a k-tile of W is 3 x 4 linear KB.
// Pipeline: three LDS buffers of 24 KB (two workgroups per CU); the DMA stream runs two k-tiles ahead of the multiplies and straight
// across tile boundaries; per stage `s_waitcnt vmcnt(6)` (this wave's pieces of the current k-tile have landed: the six younger ones
// may still fly), ONE raw s_barrier, the next DMA burst, fragments, 24 matrix instructions.
//
// Both GEMM operands are swapped with respect to gemm_win (weights = A, activations = B): an accumulator lane then owns one output row
// and 4 x 4 consecutive columns, so the epilogue moves 16-byte fp32 pieces -- or, for an output that the next GEMM consumes, splits
// them and writes 8-byte plane pieces (`yp`), optionally both (a residual-block input: fp32 for the skip, ELU'd planes for the conv).
#include "b3_common.h"
#include "rst_kernels.h"

namespace {

constexpr int P3_BM = 128, P3_BN = 128, P3_KB = 16;
constexpr int P3_OPER = 3 * 4 * 1024;            // bytes of one operand's k-tile: 3 planes x 4 pieces
constexpr int P3_STAGE = 2 * P3_OPER;            // 24 KB
constexpr int P3_NBUF = 3;
constexpr int P3_LDS = P3_NBUF * P3_STAGE;       // 72 KB: two workgroups per CU

__device__ __forceinline__ int p3_xcd_first(int tiles, int x) {
    const int q = tiles >> 3, r = tiles & 7;
    return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}
// chunk (16 bytes) of a 1 KB piece that holds k-half h of row r (r < 32)
__host__ __device__ __forceinline__ constexpr int p3_chunk(int r, int h) { return 2 * r + (h ^ ((r >> 3) & 1)); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_p3_kernel(const GemmP3Params p, const int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m = lane & 31, h = lane >> 5;
    constexpr int QA_[6] = B3_QA, QB_[6] = B3_QB;

    const int M = p.B * p.T_out;
    const int tiles_n = (p.N + P3_BN - 1) / P3_BN;
    const int nk = p.K / P3_KB;

    const int xcd = blockIdx.x & 7;
    const int first = p3_xcd_first(tiles, xcd);
    const int count = p3_xcd_first(tiles, xcd + 1) - first;
    const int stride = ((int)gridDim.x + 7 - xcd) >> 3;
    int l = blockIdx.x >> 3;
    if (l >= count) return;

    // ---- DMA side.  Wave w moves piece w (rows 32 w .. 32 w + 31) of every plane of both operands: 6 instructions per k-tile.
    __amdgpu_buffer_rsrc_t rs_x[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
        rs_x[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.xp + q * p.x_plane), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w3), 0, 0xffffffff, 0x00020000);
    const int d_r = lane >> 1;                                   // row of the piece this lane fetches
    const int d_h = (lane & 1) ^ ((d_r >> 3) & 1);               // ... and its k-half: LDS chunk `lane` = p3_chunk(d_r, d_h)
    unsigned a_vo = 0;                                           // byte offset of this lane's row window (+ k-half) inside a plane
    unsigned w_so = 0;                                           // byte offset of the tile's first k-tile inside the packed weights (uniform)
    int d_l = l, d_kt = 0;                                       // DMA cursor: tile ordinal, k-tile
    auto dma_setup = [&](const int tile) {
        const int m0 = (tile / tiles_n) * P3_BM, nt = tile % tiles_n;
        const int row = min(m0 + 32 * wave + d_r, M - 1);
        const int b = row / p.T_out;
        const int t = row - b * p.T_out;
        a_vo = (unsigned)(b * p.Tp + p.pad + t * p.S - p.P) * (unsigned)p.C * 2u + (unsigned)d_h * 16u;
        w_so = (unsigned)nt * (unsigned)nk * (unsigned)P3_OPER;  // (n tile, k-tile): 3 planes x 4 KB
    };
    auto dma_issue = [&](const int buf) {
        unsigned char* dst = smem_raw + buf * P3_STAGE + wave * 1024;
        const unsigned so_a = (unsigned)d_kt * (P3_KB * 2);
        const unsigned so_w = w_so + (unsigned)d_kt * P3_OPER + (unsigned)wave * 1024u;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x[q], (lds_ptr_t)(dst + q * 4096), 16, a_vo, so_a, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + P3_OPER + q * 4096), 16, lane * 16, so_w + q * 4096, 0, 0);
        }
    };
    auto dma_advance = [&]() {
        if (++d_kt == nk) {
            d_kt = 0;
            if (d_l + stride < count) d_l += stride;             // (past the end of the run: harmless re-loads of the last tile)
            dma_setup(first + d_l);
        }
    };

    // ---- multiply side: per-lane fragment offsets inside a buffer (weights = A operand, activations = B operand)
    const int c16 = p3_chunk(m, h) * 16;
    const int a_frag = (2 * wm) * 1024 + c16;                    // activation rows 64 wm + 32 i + m: piece 2 wm + i
    const int w_frag = P3_OPER + (2 * wn) * 1024 + c16;          // weight rows 64 wn + 32 j + m: piece 2 wn + j

    f32x16 acc[2][2];                                            // [jn][im]: lane = row 32 im + m, registers = columns 32 jn + 8 g + 4 h + j

    dma_setup(first + l);
    dma_issue(0);
    dma_advance();
    dma_issue(1);
    dma_advance();
    int cur = 0, nxt = 2;                                        // buffer being multiplied, buffer the next DMA burst fills

    for (;;) {
        const int tile = first + l;
        const int m0 = (tile / tiles_n) * P3_BM, n0 = (tile % tiles_n) * P3_BN;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[jn][im][e] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
            // this wave's pieces of the current k-tile have landed (the six of the next one may still be in flight); after the barrier so
            // have everyone's, and every wave is done reading the buffer the next burst overwrites
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char* rd = smem_raw + cur * P3_STAGE;
            bf16x8 fa[2][3], fw[2][3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i][q] = *reinterpret_cast<const bf16x8*>(rd + a_frag + q * 4096 + i * 1024);
                    fw[i][q] = *reinterpret_cast<const bf16x8*>(rd + w_frag + q * 4096 + i * 1024);
                }
            __builtin_amdgcn_sched_barrier(0);
            // the next DMA burst is issued BETWEEN the matrix instructions (one piece behind every fourth): a piece costs ~60 - 100 issue
            // cycles, which then run under the matrix pipe instead of in front of it
            unsigned char* dst = smem_raw + nxt * P3_STAGE + wave * 1024;
            const unsigned so_a = (unsigned)d_kt * (P3_KB * 2);
            const unsigned so_w = w_so + (unsigned)d_kt * P3_OPER + (unsigned)wave * 1024u;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int im = 0; im < 2; ++im)
                        acc[jn][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[jn][QA_[t]], fa[im][QB_[t]], acc[jn][im], 0, 0, 0);
                if (t < 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x[t], (lds_ptr_t)(dst + t * 4096), 16, a_vo, so_a, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + P3_OPER + (t - 3) * 4096), 16, lane * 16, so_w + (t - 3) * 4096, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            dma_advance();
            cur = cur == P3_NBUF - 1 ? 0 : cur + 1;
            nxt = nxt == P3_NBUF - 1 ? 0 : nxt + 1;
        }

        // ---- epilogue: bias -> GELU? -> (residual + scale *) -> ELU? -> fp32 pieces and / or plane pieces
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const int row = m0 + 64 * wm + 32 * im + m;
            const bool row_ok = row < M;
            const int rowc = row_ok ? row : M - 1;
            long yp_row = 0;
            if (p.yp) {
                const int b = rowc / p.T_out;
                // (a transposed convolution's row holds y_S consecutive output steps of N / y_S channels: contiguous in the plane)
                yp_row = ((long)b * p.y_Tp + p.y_pad + (long)(rowc - b * p.T_out) * p.y_S) * (p.N / p.y_S);
            }
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + 64 * wn + 32 * jn + 8 * g + 4 * h;
                    const bool ok = row_ok && n < p.N;
                    const int nc = n < p.N ? n : 0;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[jn][im][4 * g + j];
                    if (p.bias) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + nc);
                        v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
                    }
                    if (p.act_out == 1) { v[0] = rst_gelu(v[0]); v[1] = rst_gelu(v[1]); v[2] = rst_gelu(v[2]); v[3] = rst_gelu(v[3]); }
                    if (p.res) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + (long)rowc * p.ldy + nc);
                        f32x4 sv = {1.f, 1.f, 1.f, 1.f};
                        if (p.scale) sv = *reinterpret_cast<const f32x4*>(p.scale + nc);
                        v[0] = rv[0] + sv[0] * v[0]; v[1] = rv[1] + sv[1] * v[1]; v[2] = rv[2] + sv[2] * v[2]; v[3] = rv[3] + sv[3] * v[3];
                    }
                    if (p.act_out == 2) { v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]); }
                    if (p.y && ok) *reinterpret_cast<f32x4*>(p.y + (long)row * p.ldy + n) = v;
                    if (p.yp && ok) {
                        if (p.yp_elu) { v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]); }
                        f32x2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            u32x2 w;
                            w[0] = b3_peel(p0);
                            w[1] = b3_peel(p1);
                            *reinterpret_cast<u32x2*>(p.yp + q * p.y_plane + yp_row + n) = w;
                        }
                    }
                }
        }
        if (l + stride >= count) break;
        l += stride;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the stream's last bursts: nobody reads them, but they target this LDS)
}

// w fp32 [N][K] -> the planes in the LDS image order of the kernel: [n tile of 128][k-tile][plane][piece of 32 rows][chunk][8 bf16],
// rows past N zero.  One thread per (row, k-half of a k-tile).
__global__ __launch_bounds__(256) void gemm_p3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ w3, int N, int K, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int halves = K / 8;
    const int n = (int)(i / halves);
    const int kh = (int)(i % halves);
    const int kt = kh >> 1, hh = kh & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = n < N ? w[(long)n * K + kt * 16 + hh * 8 + e] : 0.f;
    f32x2 pr[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
    const int nt = n / 128, r = n % 128;
    unsigned short* dst = w3 + ((long)nt * (K / 16) + kt) * (P3_OPER / 2) + (r >> 5) * 512 + p3_chunk(r & 31, hh) * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = b3_peel(pr[d]);
        *reinterpret_cast<u32x4*>(dst + q * 2048) = o;
    }
}

// fp32 [B][T][C] -> planes [3][B][Tp][C] (rows pad .. pad + T - 1 of every utterance; the pad rows are not touched), optional ELU first.
// One thread per 8 channels.
__global__ __launch_bounds__(256) void p3_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ xp, long plane, int B, int T, int Tp,
                                                       int pad, int C, int elu, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = C / 8;
    const long row = i / c8;
    const int c = (int)(i % c8) * 8;
    const int b = (int)(row / T);
    const int t = (int)(row - (long)b * T);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + row * C + c);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + row * C + c + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (elu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rst_elu(v[e]);
    }
    f32x2 pr[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
    unsigned short* dst = xp + ((long)b * Tp + pad + t) * C + c;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = b3_peel(pr[d]);
        *reinterpret_cast<u32x4*>(dst + q * plane) = o;
    }
}

// zero the pad rows in front of and behind every utterance of a plane buffer: one thread per 16 bytes
__global__ __launch_bounds__(256) void p3_zero_pads_kernel(unsigned short* __restrict__ xp, long plane, int B, int T, int Tp, int pad, int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int per_row = C / 8;
    const int rows = Tp - T;                                     // pad rows per utterance: `pad` in front, the rest behind
    const long r = i / per_row;
    const int c = (int)(i % per_row) * 8;
    const int q = (int)(r / ((long)B * rows));
    const long rr = r - (long)q * B * rows;
    const int b = (int)(rr / rows);
    const int k = (int)(rr - (long)b * rows);
    const int t = k < pad ? k : T + k;                           // row inside the utterance's Tp rows
    *reinterpret_cast<u32x4*>(xp + q * plane + ((long)b * Tp + t) * C + c) = u32x4{0u, 0u, 0u, 0u};
}

}  // namespace

bool rst_gemm_p3_shape_ok(int B, int T_out, int Tp, int pad, int C, int K, int N, int S, int P) {
    if (B <= 0 || T_out <= 0 || Tp <= 0 || C <= 0 || K <= 0 || N <= 0 || S <= 0 || P < 0 || pad < P) return false;
    if (K % P3_KB != 0 || C % 16 != 0 || N % 8 != 0) return false;
    if ((long)B * Tp * C * 2 >= 0xfffff000L) return false;                                   // a plane: 32-bit byte offsets
    if ((long)((N + 127) / 128) * 128 * K * 6 >= 0xfffff000L) return false;                  // the packed weights
    // the last window must end inside its utterance's Tp rows
    if ((long)(pad + (long)(T_out - 1) * S - P) * C + K > (long)Tp * C) return false;
    return (long)B * T_out < 0x7fffff00L;
}

long rst_gemm_p3_weight_elems_impl(int N, int K) { return (long)((N + 127) / 128) * 128 * K * 3; }

int rst_launch_gemm_p3_pack(const float* w, unsigned short* w3, int N, int K, hipStream_t stream) {
    RST_REQUIRE(w && w3 && N > 0 && K > 0 && K % 16 == 0, "gemm_p3_pack_weight: bad arguments (K %% 16 == 0 required, N=%d K=%d)", N, K);
    RST_REQUIRE((uintptr_t)w3 % 16 == 0, "gemm_p3_pack_weight: the packed buffer must be 16-byte aligned");
    const long total = (long)((N + 127) / 128) * 128 * (K / 8);
    hipLaunchKernelGGL(gemm_p3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, w3, N, K, total);
    return rst_check_launch("gemm_p3_pack_weight");
}

int rst_launch_p3_split(const float* x, unsigned short* xp, long plane, int B, int T, int Tp, int pad, int C, int elu, hipStream_t stream) {
    RST_REQUIRE(x && xp && B > 0 && T > 0 && C % 8 == 0 && Tp >= T + pad && pad >= 0, "p3_split: bad arguments");
    RST_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)xp % 16 == 0 && plane % 8 == 0, "p3_split: pointers / plane stride must be 16-byte aligned");
    const long total = (long)B * T * (C / 8);
    hipLaunchKernelGGL(p3_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, xp, plane, B, T, Tp, pad, C, elu, total);
    return rst_check_launch("p3_split");
}

int rst_launch_p3_zero_pads(unsigned short* xp, long plane, int B, int T, int Tp, int pad, int C, hipStream_t stream) {
    RST_REQUIRE(xp && B > 0 && T >= 0 && C % 8 == 0 && Tp >= T + pad && pad >= 0, "p3_zero_pads: bad arguments");
    const long total = 3L * B * (Tp - T) * (C / 8);
    if (total == 0) return RST_OK;
    hipLaunchKernelGGL(p3_zero_pads_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, xp, plane, B, T, Tp, pad, C, total);
    return rst_check_launch("p3_zero_pads");
}

int rst_launch_gemm_p3(const GemmP3Params& p, hipStream_t stream) {
    RST_REQUIRE(p.xp && p.w3 && (p.y || p.yp), "gemm_p3: null pointer");
    RST_REQUIRE(rst_gemm_p3_shape_ok(p.B, p.T_out, p.Tp, p.pad, p.C, p.K, p.N, p.S, p.P), "gemm_p3: shape not served (B=%d T_out=%d Tp=%d pad=%d C=%d K=%d N=%d S=%d P=%d)",
                p.B, p.T_out, p.Tp, p.pad, p.C, p.K, p.N, p.S, p.P);
    RST_REQUIRE((uintptr_t)p.xp % 16 == 0 && (uintptr_t)p.w3 % 16 == 0 && p.x_plane % 8 == 0, "gemm_p3: operands must be 16-byte aligned");
    RST_REQUIRE(!p.y || ((uintptr_t)p.y % 16 == 0 && p.ldy % 4 == 0 && p.ldy >= p.N), "gemm_p3: y must be 16-byte aligned with ldy %% 4 == 0");
    RST_REQUIRE(!p.res || (p.y && (uintptr_t)p.res % 16 == 0) || (!p.y && (uintptr_t)p.res % 16 == 0 && p.ldy % 4 == 0 && p.ldy >= p.N), "gemm_p3: res layout");
    RST_REQUIRE(!p.yp || ((uintptr_t)p.yp % 8 == 0 && p.y_plane % 4 == 0 && p.y_S >= 1 && p.N % p.y_S == 0 && (p.N / p.y_S) % 4 == 0 &&
                          p.y_Tp >= p.T_out * p.y_S + p.y_pad), "gemm_p3: plane output layout");
    RST_REQUIRE((!p.bias || (uintptr_t)p.bias % 16 == 0) && (!p.scale || (uintptr_t)p.scale % 16 == 0), "gemm_p3: bias / scale must be 16-byte aligned");
    const long M = (long)p.B * p.T_out;
    const long tiles = ((M + P3_BM - 1) / P3_BM) * ((p.N + P3_BN - 1) / P3_BN);
    static RstOncePerDevice attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipGetLastError();
    }
    const long resident = 2L * rst_cu_count();
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    hipLaunchKernelGGL(gemm_p3_kernel, dim3(grid), dim3(256), P3_LDS, stream, p, (int)tiles);
    return rst_check_launch("gemm_p3");
}
