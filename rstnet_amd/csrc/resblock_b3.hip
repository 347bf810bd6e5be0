// Fused SEANet residual block on the bf16 matrix instruction at fp32 accuracy (three-plane operands, b3_common.h) for the batched
// (whole-utterance) encode / decode:
//
//     y[t] = x[t] + b2 + W2 * ELU( b1 + W1 * [ELU(x[t-2]), ELU(x[t-1]), ELU(x[t])] )        (modules/seanet.py:21-94)
//
// The f32-instruction kernels of resblock.hip run 0.52 of the 157 TFLOP/s that instruction can deliver; at the six-product bf16 rate
// the same blocks are bound by HBM (one read of x, one write of y) and by the VALU work of ELU + split, not by the matrix pipe.
//
// Every activation element is split into its three planes exactly ONCE: the ELU'd input tile is staged in LDS as planes, the hidden
// activation goes from GEMM1's accumulators through bias + ELU + split straight into GEMM2's operand (registers for C = 64, LDS for
// C = 128) -- it never exists in fp32 outside the accumulators.  Both GEMMs are computed TRANSPOSED (weights are the A operand,
// activations the B operand): an accumulator lane then owns one time step and 4 x 4 consecutive channels, so
//   * GEMM1's accumulator is, up to a fixed permutation of the hidden index that is folded into the packing of W2, GEMM2's B operand;
//   * the output (and the skip operand) moves as 16-byte row pieces instead of 16 dword accesses per accumulator block.
//
// C = 64 (24 kHz level; H = 32, Kw = 3): `resblock64_b3_kernel`.  Eight waves per CU, and EVERY WAVE WALKS ITS OWN 32-ROW TILES: W1
// (36 KB as planes) and W2 (12 KB) sit in LDS once per workgroup, a wave's ELU'd input tile (34 rows x 64 channels x 3 planes = 13 KB)
// is private to it, so the tile loop contains no workgroup barrier at all -- the eight waves drift apart and one wave's staging / ELU /
// epilogue VALU work runs under its SIMD partner's matrix instructions.  The next tile's rows are requested while the current one is
// multiplied.  PRE: the block input is conv0(audio) (encoder.model.0, Conv1d 1 -> 64, K0 <= 8), evaluated on the matrix pipe as a
// K = 16 GEMM of audio windows, once for the input tile and once more, accumulated straight into GEMM2's accumulators, for the skip.
// POST: ELU + the last decoder convolution (64 -> 1, Kf <= 4) on the output tile; only the waveform is written (tiles overlap by
// Kf - 1 recomputed rows).
//
// C = 128 (6 kHz level; H = 64, Kw = 3): `resblock128_b3_kernel`.  W1 alone is 147 KB as planes, so the weights stream: one workgroup
// of eight waves per CU owns a 128-row tile whose ELU'd input planes stay resident in LDS (100 KB) while W1 / W2 pass through a
// double-buffered ring of 24 KB stages, already in operand order (one barrier per stage of 24 matrix instructions per wave, as in
// gemm_win_b3); the hidden tile overwrites the dead input planes.
#include <type_traits>
#include "b3_common.h"
#include "rst_kernels.h"

namespace {

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// A per-lane value the optimizer cannot see through.  Used INSIDE a tile body on the lane's coordinates: everything derived from it (LDS
// offsets of 9 staging pieces, 12 k-tiles, ...) is recomputed per tile in a couple of VALU instructions instead of being hoisted out of
// the tile loop into ~40 long-lived registers, which spilled -- and a spill RELOAD is a memory operation: waiting for it (in order) drains
// the next tile's requests that are meant to stay in flight.
__device__ __forceinline__ int opq(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// 8 fp32 values (consecutive k of one operand row) -> the three bf16x8 planes
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[3]) {
    f32x2 pr[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x4 w;
#pragma unroll
        for (int d = 0; d < 4; ++d) w[d] = b3_peel(pr[d]);
        out[q] = __builtin_bit_cast(bf16x8, w);
    }
}

// ---------------------------------------------------------------------------------------------------------------- C = 64
constexpr int R6_C = 64, R6_H = 32, R6_KW = 3, R6_M = 32, R6_XR = R6_M + R6_KW - 1;      // 34 input rows per wave tile
constexpr int R6_WAVES = 8;
constexpr int R6_W1F = 12 * 3 * 1024;            // bytes: [k-tile 12][plane 3][lane 64][8 bf16]
constexpr int R6_W2F = 2 * 2 * 3 * 1024;         // [c-block 2][k-tile 2][plane 3][lane][8]
constexpr int R6_W0F = 2 * 3 * 1024;             // [c-block 2][plane 3][lane][8] (PRE; lives in registers)
constexpr int R6_CST = 32 + 64 + 64 + 4 * 64;    // floats: b1 | b2 | b0 | wf [4][64]
constexpr int R6_XPLANE = R6_XR * 128;           // bytes per plane of a wave's input tile (64 bf16 per row)
constexpr int R6_AUD = 64;                       // PRE: audio samples of a wave tile, AUD[i] = a[t0 - 9 + i]
constexpr int R6_PRE_RO = R6_M - (R6_KW - 1);    // PRE: a tile stages 32 input rows (ONE block of the first convolution on the matrix pipe)
                                                 // and yields 30 output rows; staging the two halo rows as a second, 94 % empty block cost
                                                 // 12 matrix instructions and ~300 VALU per tile
constexpr int R6_MAXKF = 4, R6_MAXK0 = 8;
constexpr int R6_YLD = R6_C * 4 + 16;            // bytes per row of the output tile while it is transposed through LDS (32 rows: 8.5 KB of
                                                 // the wave's 13 KB input tile, dead by then)

template <bool PRE, bool POST>
constexpr int r6_per_wave() { return 3 * R6_XPLANE + (PRE ? R6_AUD * 4 : 0) + (POST ? R6_MAXKF * R6_M * 4 : 0); }
template <bool PRE, bool POST>
constexpr int r6_lds_bytes() { return R6_W1F + R6_W2F + R6_CST * 4 + R6_WAVES * r6_per_wave<PRE, POST>(); }

// byte offset of the 16-byte chunk `chunk` (8 channels) of row `rx` inside a plane: rows are 128 bytes, the chunk index is XORed with
// bits 1..3 of the row so that the 16 lanes of a ds_read_b128 group (rows distinct mod 16, one chunk index) cover 16 distinct slots
__device__ __forceinline__ int r6_xoff(int rx, int chunk) { return rx * 128 + ((chunk ^ ((rx >> 1) & 7)) << 4); }

template <bool PRE, bool POST>
__global__ __launch_bounds__(64 * R6_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock64_b3_kernel(const ResblockB3Params p, const int tiles_u, const int total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const W1F = smem_raw;
    unsigned char* const W2F = W1F + R6_W1F;
    float* const CST = reinterpret_cast<float*>(W2F + R6_W2F);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // (scalar: the tile bookkeeping below stays on the scalar unit)
    const int m = lane & 31, h = lane >> 5;
    unsigned char* const XW = reinterpret_cast<unsigned char*>(CST + R6_CST) + wave * r6_per_wave<PRE, POST>();
    float* const AUD = reinterpret_cast<float*>(XW + 3 * R6_XPLANE);
    float* const DS = reinterpret_cast<float*>(XW + 3 * R6_XPLANE + (PRE ? R6_AUD * 4 : 0));
    const int T = p.T;
    const int halo = POST ? p.Kf - 1 : 0;
    const int RO = PRE ? R6_PRE_RO : R6_M - halo;    // output rows a tile contributes

    // ---- constants of the workgroup: W1 / W2 planes and the small vectors into LDS, once
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(p.wp);
        for (int i = tid; i < (R6_W1F + R6_W2F) / 16; i += 64 * R6_WAVES) reinterpret_cast<u32x4*>(W1F)[i] = src[i];
        if (tid < 32) CST[tid] = p.b1[tid];
        if (tid < 64) {
            CST[32 + tid] = p.b2[tid];
            CST[96 + tid] = PRE ? p.b0[tid] : 0.f;
        }
        if (POST && tid < R6_MAXKF * 64) CST[160 + tid] = tid < p.Kf * 64 ? p.wf[tid] : 0.f;
    }
    bf16x8 w0f[2][3];
    if (PRE) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                w0f[cb][q] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(p.wp) + R6_W1F + R6_W2F +
                                                              ((cb * 3 + q) * 64 + lane) * 16);
    }
    __syncthreads();

    const int stride = (int)gridDim.x * R6_WAVES;
    int idx = (int)blockIdx.x * R6_WAVES + wave;
    if (idx >= total) return;

    // Global traffic goes through buffer instructions: the utterance's base in scalar registers, a 32-bit byte offset per lane (the
    // launcher checks that an utterance spans less than 4 GB).  Per-lane parts of the offsets, once:
    constexpr int XCB = PRE ? 4 : R6_C * 4;                      // bytes per time step of x
    const unsigned ld_lane = PRE ? 0u : (unsigned)((lane & 15) * 16);
    const unsigned io_lane = (unsigned)(4 * h) * 4u;             // skip loads / stores: 16 bytes at channel 32 cb + 8 g + 4 h
    auto rsrc_of = [](const float* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xffffffff, 0x00020000); };

    // ---- input requests of a tile: rows t0 - 2 .. t0 + 31 (lane -> row 4 i + lane / 16, 4 channels), or the audio window of PRE;
    // always from addresses clamped into the utterance (zeroed at staging where the row does not exist)
    f32x4 xv[PRE ? 1 : 9];
    float av[1];
    auto request = [&](const int b, const int tu) {
        const int t0 = tu * RO - halo;
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(p.x + (long)b * T * (PRE ? 1 : R6_C));
        if (PRE) {
            const int t = min(max(t0 - 9 + lane, 0), T - 1);
            av[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)t * 4u, 0, 0));
        } else {
            const int r0 = opq(lane >> 4);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int rx = min(4 * i + r0, R6_XR - 1);
                const int t = min(max(t0 - (R6_KW - 1) + rx, 0), T - 1);
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)t * XCB + ld_lane, 0, 0));
            }
        }
    };

    // conv0 of 32 rows on the matrix pipe: D[c][row] = acc + sum_s W0F[c][s] AUD[first + row + s], s < 16, where slot s = k + 8 - K0
    // holds tap k of w0 (the other slots are zero): x0[t][c] = b0[c] + sum_k w0[c][k] a[t - (K0 - 1) + k] with AUD[i] = a[t0 - 9 + i] and
    // first = (row 0's time) - t0 + 2.  A = the W0 planes (registers), B = the audio windows, split here.
    // (a matrix instruction waits for the previous one into the SAME accumulator: four independent chains -- even / odd products of the
    // two channel blocks -- keep three instructions between dependent ones; with two the pipe idled half the time)
    auto conv0 = [&](const int first, f32x16 (&acc)[2], f32x16 (&accx)[2]) {
        constexpr int QA_[6] = B3_QA, QB_[6] = B3_QB;      // weight-side / activation-side plane of product t
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[e] = AUD[first + m + 8 * h + e];
        bf16x8 af[3];
        split8(a8, af);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (t & 1) accx[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[cb][QA_[t]], af[QB_[t]], accx[cb], 0, 0, 0);
                else acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[cb][QA_[t]], af[QB_[t]], acc[cb], 0, 0, 0);
            }
    };
    // four consecutive channels of a lane's accumulator group -> three 8-byte plane pieces of input-tile row rx
    auto put_x = [&](const int rx, const int c0, f32x4 v) {
        f32x2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
        const int off = r6_xoff(rx, c0 >> 3) + ((c0 >> 2) & 1) * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x2 w;
            w[0] = b3_peel(p0);
            w[1] = b3_peel(p1);
            *reinterpret_cast<u32x2*>(XW + q * R6_XPLANE + off) = w;
        }
    };
    // operand fragments of GEMM1's k-tile kt (tap kt / 4, channels 16 (kt % 4) ..): W1 planes from the workgroup's copy, X planes from
    // the wave's tile
    auto frags1 = [&](const int m, const int h, const int kt, bf16x8 (&xb)[3], bf16x8 (&wa)[3]) {
        const int xo = r6_xoff(m + (kt >> 2), 2 * (kt & 3) + h);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            xb[q] = lds_frag(XW + q * R6_XPLANE + xo);
            wa[q] = lds_frag(W1F + ((kt * 3 + q) * 64 + lane) * 16);
        }
    };
    auto frags2 = [&](const int kt2, bf16x8 (&w2)[2][3]) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 3; ++q) w2[cb][q] = lds_frag(W2F + (((cb * 2 + kt2) * 3 + q) * 64 + lane) * 16);
    };

    auto tile_body = [&](auto full_tag, const int b, const int tu, const int b_next, const int tu_next) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int QA_[6] = B3_QA, QB_[6] = B3_QB;
        const int m = opq(lane & 31), h = opq(lane >> 5);    // (opaque copies: see opq)
        const int t0 = tu * RO - halo;                       // time of output row 0 of the tile
        const int t_out = t0 + m;
        const bool valid = (!PRE || m < R6_PRE_RO) && (FULL || (t_out >= 0 && t_out < T));      // (PRE: rows 30, 31 belong to the next tile)
        const unsigned io_off = (unsigned)(FULL ? t_out : min(max(t_out, 0), T - 1)) * (R6_C * 4) + io_lane;

        // ---- stage the ELU'd input tile as planes (rows rx = 0 .. 33 <-> t = t0 - 2 + rx)
        if (PRE) {
            AUD[lane] = (FULL || t0 - 9 + lane >= 0) ? av[0] : 0.f;            // the first convolution's own zero padding
            {                                        // input rows 0 .. 31 (rows 32, 33 of the LDS tile are never written: they only reach
                                                     // the discarded output rows 30, 31 -- a lane's column of the product sees its own rows)
                constexpr int rb = 0;
                f32x16 acc[2], accx[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(CST + 96 + 32 * cb + 8 * g + 4 * h);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { acc[cb][4 * g + j] = bv[j]; accx[cb][4 * g + j] = 0.f; }
                    }
                conv0(32 * rb, acc, accx);
                const int rx = 32 * rb + m;
                const int t = t0 - (R6_KW - 1) + rx;
                {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = (FULL || t >= 0) ? rst_elu(acc[cb][4 * g + j] + accx[cb][4 * g + j]) : 0.f;
                            put_x(rx, 32 * cb + 8 * g + 4 * h, v);
                        }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                // (i = 8: the lane groups past row 33 loaded row 33 again and store it again -- same values, no branch: a divergent
                // branch here makes the compiler wait for EVERY outstanding load at its join, the next tile's included)
                const int rx = min(4 * i + opq(lane >> 4), R6_XR - 1);
                const int t = t0 - (R6_KW - 1) + rx;
                f32x4 v = xv[i];
                v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]);
                if (!FULL && t < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
                put_x(rx, (lane & 15) * 4, v);
                if (i & 1) __builtin_amdgcn_sched_barrier(0);    // (two row pieces in flight at a time: all nine interleaved spill)
            }
        }

        // ---- the next tile's input: in flight under this tile's matrix instructions
        request(b_next, tu_next);

        // ---- GEMM1 (transposed): acc1[n][row] = sum_k W1[n][k] X[row + tap][c], k = tap * 64 + c; two accumulator chains.  The
        // fragments of k-tile kt + 1 are requested before the matrix instructions of k-tile kt (two register sets, pinned: left alone the
        // scheduler serialises read -> wait -> multiply through one set)
        f32x16 acc1[4];                                          // four independent chains (see conv0), summed in the hidden epilogue
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[c][e] = 0.f;
        bf16x8 xb[2][3], wa[2][3], w2a[2][3], w2b[2][3];
        frags1(m, h, 0, xb[0], wa[0]);
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) {
            if (kt + 1 < 12) frags1(m, h, kt + 1, xb[(kt + 1) & 1], wa[(kt + 1) & 1]);
            else frags2(0, w2a);                                 // (under the last k-tile: GEMM2's first weight fragments)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int c = (kt * 6 + t) & 3;
                acc1[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kt & 1][QA_[t]], xb[kt & 1][QB_[t]], acc1[c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- skip operand: x rows of the output tile as the 16-byte pieces of the accumulator layout (lane = row, 4 x 4 channels per
        // 32-channel block; L2 hits), landing under the hidden epilogue and GEMM2.  (Memory operations retire in order, so the wait for
        // these also waits for the next tile's rows requested above -- a whole GEMM1 ago.)  PRE: the skip is recomputed on the matrix pipe.
        // Not POST: the skip rows are fetched ROW-MAJOR (16 lanes = one row's 256 bytes, as the staging loads) and meet the output after
        // its transposition through LDS (below): in the accumulator layout a load / store instruction touches 32 rows x 32 bytes, and
        // with the output written that way the PRE form ran at 1.5 TB/s of stores, whatever its instruction count.
        f32x4 skip[PRE ? 1 : 8];
        const int rm_r = opq(lane >> 4), rm_c = opq((lane & 15) * 16);     // row-major coordinates: row 4 i + rm_r, byte rm_c of the row
        if (!PRE) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(p.x + (long)b * T * R6_C);
            if (POST) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        skip[cb * 4 + g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, io_off, (32 * cb + 8 * g) * 4, 0));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = t0 + 4 * i + rm_r;
                    skip[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs, (unsigned)(FULL ? t : min(max(t, 0), T - 1)) * (R6_C * 4) + rm_c, 0, 0));
                }
            }
        }

        // ---- hidden activation: bias + ELU + split, accumulator -> GEMM2's B operand in registers.  Lane (row, h) holds hidden
        // channels n = 4 h + (r & 3) + 8 (r >> 2); slot e of k-tile kt2 is r = 8 kt2 + e (the packing of W2 uses the same order).
        bf16x8 hop[2][3];
        {
            float hv[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(CST + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    hv[4 * g + j] = rst_elu(((acc1[0][4 * g + j] + acc1[1][4 * g + j]) + (acc1[2][4 * g + j] + acc1[3][4 * g + j])) + bv[j]);
            }
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2) {
                float v8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = hv[8 * kt2 + e];
                split8(v8, hop[kt2]);
            }
        }

        // ---- GEMM2 (transposed): acc2[c][row] = sum_n W2[c][n] H[row][n]  (+ PRE: conv0 of the output rows, the skip)
        f32x16 acc2[2], acc2x[2];                                // even / odd products: four chains
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = *reinterpret_cast<const f32x4*>(CST + 32 + 32 * cb + 8 * g + 4 * h);
                if (PRE) {
                    const f32x4 b0v = *reinterpret_cast<const f32x4*>(CST + 96 + 32 * cb + 8 * g + 4 * h);
                    bv[0] += b0v[0]; bv[1] += b0v[1]; bv[2] += b0v[2]; bv[3] += b0v[3];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc2[cb][4 * g + j] = bv[j]; acc2x[cb][4 * g + j] = 0.f; }
            }
        frags2(1, w2b);
        if (PRE) conv0(R6_KW - 1, acc2, acc2x);      // rows t0 .. t0 + 31: windows start two samples later than the input tile's
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (t & 1) acc2x[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2a[cb][QA_[t]], hop[0][QB_[t]], acc2x[cb], 0, 0, 0);
                else acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2a[cb][QA_[t]], hop[0][QB_[t]], acc2[cb], 0, 0, 0);
            }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (t & 1) acc2x[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2b[cb][QA_[t]], hop[1][QB_[t]], acc2x[cb], 0, 0, 0);
                else acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2b[cb][QA_[t]], hop[1][QB_[t]], acc2[cb], 0, 0, 0);
            }

        // ---- epilogue: y = x + acc2 (biases are in the accumulators)
        float dk[R6_MAXKF];
        if (POST) {
#pragma unroll
            for (int k = 0; k < R6_MAXKF; ++k) dk[k] = 0.f;
        }
        const __amdgpu_buffer_rsrc_t rs_y = rsrc_of(p.y + (long)b * T * (POST ? 1 : R6_C));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc2[cb][4 * g + j] + acc2x[cb][4 * g + j];
                if (!POST) {
                    // accumulator layout -> the wave's (dead) input-tile bytes as fp32 rows of 256 + 16 bytes (conflict-free both ways)
                    *reinterpret_cast<f32x4*>(XW + m * R6_YLD + (32 * cb + 8 * g + 4 * h) * 4) = v;
                    continue;
                }
                v[0] += skip[cb * 4 + g][0]; v[1] += skip[cb * 4 + g][1]; v[2] += skip[cb * 4 + g][2]; v[3] += skip[cb * 4 + g][3];
                v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]);
                if (POST) {
                    if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < R6_MAXKF; ++k) {
                        if (k >= p.Kf) break;                    // (uniform)
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(CST + 160 + k * 64 + 32 * cb + 8 * g + 4 * h);
                        dk[k] = fmaf(wv[0], v[0], dk[k]); dk[k] = fmaf(wv[1], v[1], dk[k]);
                        dk[k] = fmaf(wv[2], v[2], dk[k]); dk[k] = fmaf(wv[3], v[3], dk[k]);
                    }
                }
            }
        if (!POST) {
            // row-major: + skip, ELU?, whole 256-byte rows per 16 lanes
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 4 * i + rm_r;
                const int t = t0 + r;
                f32x4 v = *reinterpret_cast<const f32x4*>(XW + r * R6_YLD + rm_c);
                if (!PRE) { v[0] += skip[i][0]; v[1] += skip[i][1]; v[2] += skip[i][2]; v[3] += skip[i][3]; }
                if (p.elu_out) { v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]); }
                if ((!PRE || r < R6_PRE_RO) && (FULL || (t >= 0 && t < T)))
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, (unsigned)t * (R6_C * 4) + rm_c, 0, 0);
            }
        }
        if (POST) {
            // last convolution 64 -> 1: out[t] = bf + sum_k d_k[t - Kf + 1 + k], d_k[row] = sum_c wf[k][c] ELU(y[row][c]); the two
            // lanes of a row meet, the taps meet across rows through the wave's own LDS strip
#pragma unroll
            for (int k = 0; k < R6_MAXKF; ++k) {
                const float d = dk[k] + __shfl_xor(dk[k], 32);
                if (h == 0) DS[k * R6_M + m] = d;
            }
            if (lane >= halo && lane < R6_M) {
                float s = p.bf[0];
                for (int k = 0; k < p.Kf; ++k) s += DS[k * R6_M + lane - halo + k];
                const int t = t0 + lane;
                if (FULL || (t >= 0 && t < T)) p.y[(long)b * T + t] = s;
            }
        }
    };

    // (utterance, tile inside it) of the wave's current tile, advanced by `stride` tiles per step without a division in the loop
    const int dq = stride / tiles_u, dr = stride - dq * tiles_u;
    int b = idx / tiles_u, tu = idx - b * tiles_u;
    request(b, tu);
    for (; idx < total; idx += stride) {
        int b_next = b + dq, tu_next = tu + dr;
        if (tu_next >= tiles_u) { tu_next -= tiles_u; ++b_next; }
        if (idx + stride >= total) { b_next = b; tu_next = tu; }          // (at the end of the run: a harmless re-load of this tile)
        const int t0 = tu * RO - halo;
        const bool full = t0 - (R6_KW - 1) - (PRE ? R6_MAXK0 : 0) >= 0 && t0 + R6_M <= T;
        if (full) tile_body(std::true_type{}, b, tu, b_next, tu_next);
        else tile_body(std::false_type{}, b, tu, b_next, tu_next);
        b = b_next;
        tu = tu_next;
    }
}

// ---------------------------------------------------------------------------------------------------------------- C = 128
constexpr int R8_C = 128, R8_H = 64, R8_KW = 3, R8_BM = 128, R8_XR = R8_BM + R8_KW - 1;     // 130 input rows per workgroup tile
constexpr int R8_THREADS = 512;
constexpr int R8_XLD = 256 + 16, R8_HLD = 128 + 16;      // bytes per row of the input / hidden planes: 16 bytes of padding make the row stride
                                                         // odd in 16-byte slots (17 / 9), so the 16 rows of a ds_read_b128 group (distinct
                                                         // mod 16) hit 16 distinct slots, and every fragment address is ONE per-lane base
                                                         // plus a compile-time offset (an XOR swizzle costs a register per fragment: the
                                                         // loop-invariant addresses of 24 k-tiles spilled)
constexpr int R8_XPLANE = R8_XR * R8_XLD;        // bytes per plane of the input tile (128 bf16 per row)
constexpr int R8_HPLANE = R8_BM * R8_HLD;        // bytes per plane of the hidden tile (64 bf16 per row), laid over the dead input planes
constexpr int R8_STAGE = 24 * 1024;              // bytes of weights per stage: W1 4 k-tiles x 3 planes x 2 n-blocks, W2 2 k-tiles x 3 x 4 c-blocks
constexpr int R8_NSTAGE = 8;                     // per tile: stages 0 .. 5 = W1 (24 k-tiles), 6 .. 7 = W2 (4 k-tiles)
constexpr int R8_CST = 64 + 128;                 // floats: b1 | b2
constexpr int R8_LDS = 3 * R8_XPLANE + 2 * R8_STAGE + R8_CST * 4;


__global__ __launch_bounds__(R8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock128_b3_kernel(const ResblockB3Params p, const int tiles_u, const int total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* const RING = smem_raw;                        // two stages of weights
    float* const CST = reinterpret_cast<float*>(RING + 2 * R8_STAGE);
    // (the input planes, later the hidden planes, follow at byte XP0 = 2 * R8_STAGE + R8_CST * 4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                     // GEMM1: row block, hidden block; GEMM2: row block, pair of channel blocks
    const int T = p.T;
    constexpr int QA_[6] = B3_QA, QB_[6] = B3_QB;
    if ((int)blockIdx.x >= total) return;
    // Per-lane LDS bases, hidden from the optimizer: every access below is one of these plus a compile-time offset that fits the
    // instruction's 16-bit offset field.  (Left to itself the compiler materialises each of the ~100 loop-invariant addresses of the
    // unrolled tile body in a register of its own and spills them.)
    // (the opaque value is the byte OFFSET: a pointer through the asm would come back as a generic 64-bit address)
    auto opaque = [](int off) { asm volatile("" : "+v"(off)); return off; };
    constexpr int XP0 = 2 * R8_STAGE + R8_CST * 4;                                                 // byte offset of the planes
    const unsigned char* const ring_rd = smem_raw + opaque(wn * 1024 + lane * 16);                 // GEMM1 fragments (GEMM2: + wn * 1024 more)
    unsigned char* const ring_wr = smem_raw + opaque(tid * 16);
    const int row = 32 * wm + m;                                                                   // this lane's row of the tile in both GEMMs
    const unsigned char* x_rd[3];
    unsigned char* x_wr[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        x_rd[q] = smem_raw + opaque(XP0 + q * R8_XPLANE + row * R8_XLD + h * 16);
        x_wr[q] = smem_raw + opaque(XP0 + q * R8_XPLANE + (tid >> 5) * R8_XLD + (tid & 31) * 8);
    }
    const unsigned char* const h_rd = smem_raw + opaque(XP0 + row * R8_HLD + h * 16);
    unsigned char* const h_wr = smem_raw + opaque(XP0 + row * R8_HLD + (32 * wn + 4 * h) * 2);
    const float* const cst = reinterpret_cast<const float*>(smem_raw + opaque(2 * R8_STAGE + (32 * wn + 4 * h) * 4));

    if (tid < 64) CST[tid] = p.b1[tid];
    if (tid < 128) CST[64 + tid] = p.b2[tid];

    // ---- the weight stream: stage G (counted over the whole run) = stage G % 8 of the packed blob; three 16-byte pieces per thread
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wp), 0, 0xffffffff, 0x00020000);
    u32x4 wr[2][3];                                              // two register sets: the stages one and two ahead of the one in LDS
    auto w_load = [&](const int g, u32x4 (&dst)[3]) {
        const int so = (g & (R8_NSTAGE - 1)) * R8_STAGE;         // (scalar: the stage's byte offset; the lane's piece is tid * 16)
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid * 16, so + j * (R8_THREADS * 16), 0);
    };
    auto w_store = [&](const int buf, const u32x4 (&src)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(ring_wr + buf * R8_STAGE + j * R8_THREADS * 16) = src[j];
    };
    {
        u32x4 w0[3];
        w_load(0, w0);
        w_store(0, w0);
        w_load(1, wr[0]);
        w_load(2, wr[1]);
    }

    // ---- input rows of a tile: thread -> row 16 i + tid / 32, 4 channels.  Buffer loads: the utterance's base in scalar registers, a
    // 32-bit byte offset per lane (the launcher checks that an utterance spans less than 4 GB)
    f32x4 xv[9];
    auto request = [&](const int b, const int tu) {
        const int t0 = tu * R8_BM;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (long)b * T * R8_C), 0, 0xffffffff, 0x00020000);
        const int r0 = opq(tid >> 5), c0 = opq((tid & 31) * 16);        // (opaque: recomputed per tile, not hoisted into 18 registers)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int rx = min(16 * i + r0, R8_XR - 1);
            const int t = min(max(t0 - (R8_KW - 1) + rx, 0), T - 1);
            xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)t * (R8_C * 4) + c0, 0, 0));
        }
    };
    // one stage of the weight stream behind the matrix instructions of stage s: the set that holds stage s + 1 goes to the other ring
    // buffer, then requests stage s + 3
    auto advance = [&](const int gbase, const int s) {
        w_store((s + 1) & 1, wr[s & 1]);
        w_load(gbase + s + 3, wr[s & 1]);
    };

    auto tile_body = [&](auto full_tag, const int b, const int tu, const int b_next, const int tu_next, const int gbase) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int t0 = tu * R8_BM;

        // ---- stage the ELU'd input tile as planes (every wave is past the previous tile's last read of the hidden planes: barrier below)
        const int r0s = opq(tid >> 5);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int rx = 16 * i + r0s;
            const int t = t0 - (R8_KW - 1) + rx;
            f32x4 v = xv[i];
            v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]);
            if (!FULL && t < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < 8 || r0s < R8_XR - 128) {
                f32x2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x2 w;
                    w[0] = b3_peel(p0);
                    w[1] = b3_peel(p1);
                    *reinterpret_cast<u32x2*>(x_wr[q] + i * 16 * R8_XLD) = w;
                }
            }
        }
        request(b_next, tu_next);
        __syncthreads();                                         // input planes (and stage 0 of the weights) visible

        // ---- GEMM1 (transposed): acc1[n][row] = sum_k W1[n][k] X[row + tap][c], k = tap * 128 + c; six stages of four k-tiles
        f32x16 acc1[4];                                          // four independent accumulator chains (a matrix instruction waits for the
                                                                 // previous one into the same accumulator), summed in the hidden epilogue
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[c][e] = 0.f;
        // (fragments of k-tile kl + 1 are requested before the matrix instructions of k-tile kl: two register sets, pinned)
        bf16x8 xb[2][3], wa[2][3];
        auto frags1 = [&](const int s, const int kl, bf16x8 (&x)[3], bf16x8 (&w)[3]) {
            const int kt = 4 * s + kl;
            const int tap = kt >> 3, cq = kt & 7;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                x[q] = lds_frag(x_rd[q] + tap * R8_XLD + cq * 32);
                w[q] = lds_frag(ring_rd + (s & 1) * R8_STAGE + ((kl * 3 + q) * 2) * 1024);
            }
        };
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            frags1(s, 0, xb[0], wa[0]);
#pragma unroll
            for (int kl = 0; kl < 4; ++kl) {
                if (kl + 1 < 4) frags1(s, kl + 1, xb[(kl + 1) & 1], wa[(kl + 1) & 1]);
                if (kl == 1) advance(gbase, s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int c = ((4 * s + kl) * 6 + t) & 3;
                    acc1[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kl & 1][QA_[t]], xb[kl & 1][QB_[t]], acc1[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }

        // ---- skip operand of the output rows (accumulator layout of GEMM2: lane = row, 4 x 4 channels per 32-channel block; L2 hits).
        // (The next tile's input was requested six stages ago and has landed: the in-order wait for these costs nothing extra.)
        const int t_out = t0 + row;
        f32x4 skip[8];
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (long)b * T * R8_C), 0, 0xffffffff, 0x00020000);
            const unsigned vo = (unsigned)((FULL ? t_out : min(max(t_out, 0), T - 1)) * R8_C + 64 * wn + 4 * h) * 4u;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    skip[ci * 4 + g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (32 * ci + 8 * g) * 4, 0));
        }

        // ---- hidden activation: bias + ELU + split -> planes over the dead input tile (every wave is past GEMM1: barrier above).  Lane
        // (row, h) of wave (wm, wn) holds hidden channels n = 32 wn + 8 g + 4 h + j.
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(cst + 8 * g);
            auto hsum = [&](const int e) { return (acc1[0][e] + acc1[1][e]) + (acc1[2][e] + acc1[3][e]); };
            f32x2 p0 = {rst_elu(hsum(4 * g) + bv[0]), rst_elu(hsum(4 * g + 1) + bv[1])};
            f32x2 p1 = {rst_elu(hsum(4 * g + 2) + bv[2]), rst_elu(hsum(4 * g + 3) + bv[3])};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                u32x2 w;
                w[0] = b3_peel(p0);
                w[1] = b3_peel(p1);
                *reinterpret_cast<u32x2*>(h_wr + q * R8_HPLANE + g * 16) = w;
            }
        }
        __syncthreads();                                         // hidden planes visible

        // ---- GEMM2 (transposed): acc2[c][row] = sum_n W2[c][n] H[row][n]; wave (wm, wn) owns channel blocks 2 wn, 2 wn + 1
        f32x16 acc2[2], acc2x[2];                                // even / odd products: four chains
        // the accumulators START at b2 + x (round 5): the skip operand is consumed here instead of riding through GEMM2 in 32 registers --
        // with them the phase below needed more than the 256 a wave has, the compiler parked 5 of the 9 prefetched input vectors of the
        // NEXT tile in scratch right behind their loads (a wait for HBM in front of GEMM1, and the tile's input through memory twice)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(cst + 64 + 32 * wn + 32 * ci + 8 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc2[ci][4 * g + j] = bv[j] + skip[ci * 4 + g][j]; acc2x[ci][4 * g + j] = 0.f; }
            }
        bf16x8 hb[3], w2[2][3];                                  // (one register set: a second one spills, and GEMM2 is four k-tiles)
        auto frags2 = [&](const int s, const int kl, bf16x8 (&hx)[3], bf16x8 (&w)[2][3]) {
            const int kt2 = 2 * (s - 6) + kl;
            const unsigned char* rd = ring_rd + wn * 1024 + (s & 1) * R8_STAGE;      // (channel blocks 2 wn, 2 wn + 1)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                hx[q] = lds_frag(h_rd + q * R8_HPLANE + kt2 * 32);
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) w[ci][q] = lds_frag(rd + ((kl * 3 + q) * 4 + ci) * 1024);
            }
        };
#pragma unroll
        for (int s = 6; s < 8; ++s) {
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                frags2(s, kl, hb, w2);
                if (kl == 0) advance(gbase, s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) {
                        if (t & 1) acc2x[ci] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[ci][QA_[t]], hb[QB_[t]], acc2x[ci], 0, 0, 0);
                        else acc2[ci] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[ci][QA_[t]], hb[QB_[t]], acc2[ci], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                     // (s = 7: every wave is done with the hidden planes -> the next tile may stage)
        }

        // ---- epilogue: y = acc2 (x and b2 are in the accumulators)
        const bool valid = FULL || (t_out >= 0 && t_out < T);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc2[ci][4 * g + j] + acc2x[ci][4 * g + j];
                if (p.elu_out) { v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]); }
                if (valid) *reinterpret_cast<f32x4*>(p.y + ((long)b * T + t_out) * R8_C + 64 * wn + 32 * ci + 8 * g + 4 * h) = v;
            }
    };

    const int stride = gridDim.x;
    const int dq = stride / tiles_u, dr = stride - dq * tiles_u;
    int idx = blockIdx.x;
    int b = idx / tiles_u, tu = idx - b * tiles_u;
    request(b, tu);
    for (int it = 0; idx < total; idx += stride, ++it) {
        int b_next = b + dq, tu_next = tu + dr;
        if (tu_next >= tiles_u) { tu_next -= tiles_u; ++b_next; }
        if (idx + stride >= total) { b_next = b; tu_next = tu; }
        const int t0 = tu * R8_BM;
        const bool full = t0 - (R8_KW - 1) >= 0 && t0 + R8_BM <= T;
        if (full) tile_body(std::true_type{}, b, tu, b_next, tu_next, it * R8_NSTAGE);
        else tile_body(std::false_type{}, b, tu, b_next, tu_next, it * R8_NSTAGE);
        b = b_next;
        tu = tu_next;
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight packing
// One thread per 8-element operand piece: row `i` of the matrix, the eight k indices of a lane's slot, all three planes.
struct PackSrc {
    const float* w;
    int rows, ld;        // w [rows][ld]
};
__device__ __forceinline__ void pack_piece(const PackSrc s, const int i, const int (&k)[8], unsigned short* dst, const long plane_stride) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (i < s.rows && k[e] >= 0 && k[e] < s.ld) ? s.w[(long)i * s.ld + k[e]] : 0.f;
    bf16x8 pl[3];
    split8(v, pl);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(dst + q * plane_stride) = pl[q];
}

// C = 64 blob: W1F [kt 12][q][lane][8] | W2F [cb 2][kt2 2][q][lane][8] | W0F [cb 2][q][lane][8]
__global__ __launch_bounds__(64) void resblock64_b3_pack_kernel(const float* w0, const float* w1, const float* w2, unsigned short* out, int K0) {
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int piece = blockIdx.x;
    int k[8];
    if (piece < 12) {                                    // W1 [32][192]: k = 16 kt + 8 h + e
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = 16 * piece + 8 * h + e;
        pack_piece({w1, R6_H, R6_KW * R6_C}, m, k, out + (long)(piece * 3) * 512 + lane * 8, 512);
    } else if (piece < 16) {                             // W2 [64][32]: slot (kt2, h, e) holds hidden channel 16 kt2 + 8 (e >> 2) + 4 h + (e & 3)
        const int cb = (piece - 12) >> 1, kt2 = (piece - 12) & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = 16 * kt2 + 8 * (e >> 2) + 4 * h + (e & 3);
        pack_piece({w2, R6_C, R6_H}, 32 * cb + m, k, out + R6_W1F / 2 + (long)((cb * 2 + kt2) * 3) * 512 + lane * 8, 512);
    } else {                                             // W0 [64][K0]: slot s = 8 h + e holds tap s - (8 - K0) (zero elsewhere)
        const int cb = piece - 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = w0 ? 8 * h + e - (8 - K0) : -1;
        pack_piece({w0, w0 ? R6_C : 0, K0}, 32 * cb + m, k, out + (R6_W1F + R6_W2F) / 2 + (long)(cb * 3) * 512 + lane * 8, 512);
    }
}

// C = 128 blob: stages 0 .. 5: W1 [kl 4][q][nb 2][lane][8] (k-tile 4 s + kl) | stages 6, 7: W2 [kl 2][q][cb 4][lane][8] (k-tile 2 (s - 6) + kl)
__global__ __launch_bounds__(64) void resblock128_b3_pack_kernel(const float* w1, const float* w2, unsigned short* out) {
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int piece = blockIdx.x;
    int k[8];
    if (piece < 48) {                                    // W1 [64][384]: piece = kt * 2 + nb
        const int kt = piece >> 1, nb = piece & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = 16 * kt + 8 * h + e;
        const int s = kt >> 2, kl = kt & 3;
        pack_piece({w1, R8_H, R8_KW * R8_C}, 32 * nb + m, k, out + (long)s * (R8_STAGE / 2) + (long)((kl * 3) * 2 + nb) * 512 + lane * 8, 2 * 512);
    } else {                                             // W2 [128][64]: piece - 48 = kt2 * 4 + cb, natural hidden order
        const int kt2 = (piece - 48) >> 2, cb = (piece - 48) & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = 16 * kt2 + 8 * h + e;
        const int s = 6 + (kt2 >> 1), kl = kt2 & 1;
        pack_piece({w2, R8_C, R8_H}, 32 * cb + m, k, out + (long)s * (R8_STAGE / 2) + (long)((kl * 3) * 4 + cb) * 512 + lane * 8, 4 * 512);
    }
}

template <bool PRE, bool POST>
int launch64(const ResblockB3Params& p, hipStream_t stream) {
    const int halo = POST ? p.Kf - 1 : 0;
    const int RO = PRE ? R6_PRE_RO : R6_M - halo;
    const long tiles_u = (p.T + RO - 1) / RO;
    const long total = (long)p.B * tiles_u;
    if (total > 0x7ffffff0L - 8L * rst_cu_count()) { rst_set_error("resblock_b3: too many tiles (%ld)", total); return RST_ERR_UNSUPPORTED; }
    static RstOncePerDevice attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock64_b3_kernel<PRE, POST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long wgs = (total + R6_WAVES - 1) / R6_WAVES;
    const long resident = rst_cu_count();
    const unsigned grid = (unsigned)(wgs < resident ? wgs : resident);
    constexpr int lds = r6_lds_bytes<PRE, POST>();
    hipLaunchKernelGGL((resblock64_b3_kernel<PRE, POST>), dim3(grid), dim3(64 * R6_WAVES), lds, stream, p, (int)tiles_u, (int)total);
    return rst_check_launch("resblock_b3");
}

int launch128(const ResblockB3Params& p, hipStream_t stream) {
    const long tiles_u = (p.T + R8_BM - 1) / R8_BM;
    const long total = (long)p.B * tiles_u;
    if (total > 0x7fffffffL) { rst_set_error("resblock_b3: grid too large"); return RST_ERR_UNSUPPORTED; }
    static RstOncePerDevice attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock128_b3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long resident = rst_cu_count();
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    hipLaunchKernelGGL(resblock128_b3_kernel, dim3(grid), dim3(R8_THREADS), R8_LDS, stream, p, (int)tiles_u, (int)total);
    return rst_check_launch("resblock_b3");
}

}  // namespace

// Shapes the three-plane residual-block kernels serve: the two high-rate levels of Mimi (C = 64 with the first / last convolution folded
// in or not, C = 128 plain), kernel 3, hidden C / 2, no streaming history.
bool rst_resblock_b3_supported(int C, int H, int Kw, int pre, int post, int K0, int Kf) {
    if (H * 2 != C || Kw != 3) return false;
    if (pre && post) return false;
    if (pre && (K0 < 1 || K0 > R6_MAXK0)) return false;
    if (post && (Kf < 1 || Kf > R6_MAXKF)) return false;
    if (C == 64) return true;
    return C == 128 && !pre && !post;
}

long rst_resblock_b3_weight_elems(int C) { return C == 64 ? (R6_W1F + R6_W2F + R6_W0F) / 2 : C == 128 ? (long)R8_NSTAGE * R8_STAGE / 2 : -1; }

int rst_launch_resblock_b3_pack(const float* w0, const float* w1, const float* w2, unsigned short* out, int C, int H, int Kw, int K0,
                                hipStream_t stream) {
    RST_REQUIRE(w1 && w2 && out && rst_resblock_b3_supported(C, H, Kw, w0 != nullptr, 0, K0, 0), "resblock_b3_pack: unsupported shape C=%d H=%d Kw=%d K0=%d",
                C, H, Kw, K0);
    RST_REQUIRE((uintptr_t)out % 16 == 0, "resblock_b3_pack: the packed buffer must be 16-byte aligned");
    if (C == 64) hipLaunchKernelGGL(resblock64_b3_pack_kernel, dim3(18), dim3(64), 0, stream, w0, w1, w2, out, K0);
    else hipLaunchKernelGGL(resblock128_b3_pack_kernel, dim3(64), dim3(64), 0, stream, w1, w2, out);
    return rst_check_launch("resblock_b3_pack");
}

int rst_launch_resblock_b3(const ResblockB3Params& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 0 && p.T >= 0, "resblock_b3: bad sizes");
    if (p.B == 0 || p.T == 0) return RST_OK;
    RST_REQUIRE(p.x && p.wp && p.b1 && p.b2 && p.y, "resblock_b3: null pointer");
    RST_REQUIRE(rst_resblock_b3_supported(p.C, p.H, p.Kw, p.pre, p.post, p.K0, p.Kf), "resblock_b3: unsupported shape C=%d H=%d Kw=%d pre=%d post=%d",
                p.C, p.H, p.Kw, p.pre, p.post);
    RST_REQUIRE(!p.pre || p.b0, "resblock_b3: PRE needs b0");
    RST_REQUIRE(!p.post || (p.wf && p.bf), "resblock_b3: POST needs wf / bf");
    RST_REQUIRE((uintptr_t)p.wp % 16 == 0 && (p.pre || (uintptr_t)p.x % 16 == 0) && (p.post || (uintptr_t)p.y % 16 == 0),
                "resblock_b3: pointers must be 16-byte aligned");
    if (p.C == 64) {
        if (p.pre) return launch64<true, false>(p, stream);
        if (p.post) return launch64<false, true>(p, stream);
        return launch64<false, false>(p, stream);
    }
    return launch128(p, stream);
}
