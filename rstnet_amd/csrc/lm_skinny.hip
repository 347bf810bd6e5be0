#include <algorithm>

#include "lm_common.h"

// ======================================================================================================================
// Skinny GEMM for 4 < batch <= 64: y[b][n] = (res +) (bias +) sum_k x[b][k] * W[n][k] on the bf16 matrix cores.
// The weight matrix is streamed from HBM exactly once (the step stays bandwidth-bound up to batch ~64); the contraction runs
// on v_mfma_f32_32x32x16_bf16 with the fp32 activations split into bf16 hi + lo parts (two MFMAs per step): products carry
// ~17 mantissa bits of x, i.e. fp32-class accuracy against the fp32 oracle (the reference itself rounds activations to bf16).
// ======================================================================================================================
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

// ---- operand layouts of the skinny GEMM ------------------------------------------------------------------------------
// Both operands are stored in the order the 32x32x16 MFMA consumes them, so that every wave-level load is ONE contiguous
// kilobyte: [tile of 32 rows][step of 16 k][lane = 32 * (k / 8 % 2) + row % 32][8 bf16].
//   weights  Wp : [ceil(N/32)][K/16][64][8]   (rows beyond N zero)        -- packed once per weight (rst_skinny_pack_weight_bf16)
//   activations Xp: [2 = hi, lo][ceil(B/32)][K/16][64][8] (rows beyond B zero) -- packed per call by the (fused) prologue kernel
// interleave (gated layers, w = [W_u ; W_v] stacked): packed tile t holds rows 16t..16t+15 of W_u followed by the same rows of W_v,
// so that one output tile of the GEMM carries matching (u, v) pairs and its epilogue can apply silu(u) * v.
__global__ __launch_bounds__(256) void skinny_pack_weight_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wp,
                                                                int N, int K, int interleave) {
    const long total = (long)((N + 31) / 32) * 32 * (K / 8);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / (K / 8)), k = (int)(idx % (K / 8)) * 8;
        int src = row;
        if (interleave) {
            const int t = row >> 5, r = row & 31;
            src = (r < 16 ? 0 : N / 2) + t * 16 + (r & 15);
        }
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < N) v = *reinterpret_cast<const u32x4*>(w + (long)src * K + k);
        *reinterpret_cast<u32x4*>(wp + packed_index(row, k, K)) = v;
    }
}

// Prologue + hi/lo split + packing of one activation row per workgroup (pad rows of the last batch tile are zero-filled).
// hi = fp32 truncated to bf16 (an exact prefix, so x - hi is exact), lo = the residual rounded half-up: x = hi + lo to 2^-17.
// mode 0: identity; 1: RMSNorm x * alpha * rsqrt(eps + mean(x^2)); 2: SiLU gate, x row = [u ; v] -> silu(u) * v.
__global__ __launch_bounds__(256) void skinny_pack_act_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                             unsigned short* __restrict__ xp, int B, int K, int ldx, int mode, float eps) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long half = (long)((B + 31) / 32) * 32 * K;          // elements of the hi (and of the lo) plane
    float scale = 1.f;
    if (mode == 1 && b < B) {
        float s = 0.f;
        for (int k = tid * 4; k < K; k += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k);
            s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
        }
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        scale = 1.0f / sqrtf(eps + (red[0] + red[1] + red[2] + red[3]) / (float)K);
    }
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (b < B) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a0[j]; v[4 + j] = a1[j]; }
            if (mode == 1) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(alpha + k), g1 = *reinterpret_cast<const f32x4*>(alpha + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = v[j] * (g0[j] * scale); v[4 + j] = v[4 + j] * (g1[j] * scale); }
            } else if (mode == 2) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = silu(v[j]) * g0[j]; v[4 + j] = silu(v[4 + j]) * g1[j]; }
            }
        }
        store_packed8(xp, half, b, k, K, v);
    }
}

constexpr int SKINNY_WAVES = 8;

// One workgroup = CT adjacent tiles of 32 weight rows; its 8 waves each take an eighth of K.  Per MFMA step a wave loads
// 1 KB of activations (hi), 1 KB (lo) -- L2 hits -- and CT x 1 KB of weights from HBM, all contiguous; no LDS stage and no
// barrier in the main loop.  The activations are the MFMA "A" side, so an accumulator is C[b = row(e, lane)][n = lane & 31]
// and the global stores are 128-byte coalesced.  The 8 partial tiles meet in LDS and are summed in wave order
// (deterministic); there is no cross-workgroup reduction.
//
// What bounds a launch is the CU's load path, not HBM: a wave-step pulls 1 KB of weights AND 2 KB of packed activations (L2 hits)
// through the L1, which sustains ~60 GB/s per CU -- measured: time = 4-6 us + 3.2 ns x K per workgroup, whatever N (33.5 MB as
// 1024 x 16384 takes 57 us, as 16384 x 1024 10 us).  Hence (rst_skinny_bf16_split_plan_impl / rst_launch_gemm_skinny): where enough
// workgroups remain a workgroup takes CT = 4 adjacent column tiles (the activations of a step feed four weight tiles: 1.5 bytes through the L1 per
// weight byte instead of 3) and, for the longest K, only a slice of it (gridDim.y splits, so that ~256-384 workgroups exist); the partial tiles go to
// `ws` with write-through stores, one arrival counter per column group, and the LAST workgroup to arrive sums them in split order
// (deterministic) and runs the epilogue (the protocol of the fp32 few-row GEMM, csrc/skinny_f32.hip).
template <int NB, int CT>   // batch tiles of 32, weight-row tiles per workgroup
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_skinny_kernel(const SkinnyParams p) {
    __shared__ float red[SKINNY_WAVES][NB * 32][33];
    __shared__ float gtile[NB * 32 * 16];      // gated epilogue: silu(u) * v of a column tile
    __shared__ int sm_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = (p.N + 31) / 32;
    const int tile0 = blockIdx.x * CT;
    const int steps = p.K / 16;
    const int nsplit = gridDim.y;
    const int per_split = (steps + nsplit - 1) / nsplit;
    const int st_lo = blockIdx.y * per_split, st_hi = min(steps, st_lo + per_split);
    const int per = (max(st_hi - st_lo, 0) + SKINNY_WAVES - 1) / SKINNY_WAVES;
    const int s0 = st_lo + wave * per, s1 = min(st_hi, s0 + per);
    const long xplane = (long)NB * 32 * p.K;                        // elements of the hi plane
    const unsigned short* xh = p.xp + (long)lane * 8;
    const unsigned short* xl = xh + xplane;
    const unsigned short* wt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wt[c] = p.w + ((long)min(tile0 + c, tiles - 1) * steps * 64 + lane) * 8;
    f32x16 acc[NB][CT];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;
    // residual / bias of this thread's outputs: requested before the K loop (clamped addresses), used after it -- in the epilogue
    // each would be one more exposed round trip of a launch that lasts a handful of them
    constexpr int EP = NB * 32 * 32 / (64 * SKINNY_WAVES);          // output elements per thread and column tile
    float rpre[CT][EP], bpre[CT][EP];
    if (!p.gate_out) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < EP; ++q) {
                const int idx = tid + q * 64 * SKINNY_WAVES;
                const int b = min(idx >> 5, p.B - 1), n = min((tile0 + c) * 32 + (idx & 31), p.N - 1);
                rpre[c][q] = p.res ? p.res[(long)b * p.ldy + n] : 0.f;
                bpre[c][q] = p.bias ? p.bias[n] : 0.f;
            }
    }
    // MFMA steps whose loads are all requested before the first is consumed.  The loop body is loads -> wait -> MFMAs, i.e. one exposed
    // memory round trip (2-3 us for HBM weights under load) per iteration: with one batch tile and one column tile the registers
    // hold 8 steps, so the 1024-wide layers (8 steps per wave) are ONE iteration.
    constexpr int UN = (NB == 1 && CT == 1) ? 8 : ((NB * 2 + CT) <= 4 ? 4 : 2);
    // (Measured and dropped, round 3: requesting the weights of 32 / 16 steps up front with the activations following 8 steps at a time --
    // one HBM round trip per 32 steps instead of per 8 -- is SLOWER: 7.32 vs 7.10 ms per frame at the Moshi-7B shape, batch 32, 3.64 vs
    // 3.62 ms for the GPT frame, A/B inside one session.)
    for (int s = s0; s < s1; s += UN) {
        bf16x8 a[UN][CT], bh[UN][NB], bl[UN][NB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = s + u < s1;
            const long so = (long)(s + u) * 512;
#pragma unroll
            for (int c = 0; c < CT; ++c)
                a[u][c] = ok ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wt[c] + so)) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                bh[u][t] = ok ? *reinterpret_cast<const bf16x8*>(xh + (long)t * steps * 512 + so) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                bl[u][t] = ok ? *reinterpret_cast<const bf16x8*>(xl + (long)t * steps * 512 + so) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[u][t], a[u][c], acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[u][t], a[u][c], acc[t][c], 0, 0, 0);
                }
    }
    const int i = lane & 31;
    bool single = false;            // red[slot] alone holds the tile (the summed splits) instead of one partial per wave
    int slot = 0;
    auto tsum = [&](int b, int col) {
        if (single) return red[slot][b][col];
        float v = red[0][b][col];
#pragma unroll
        for (int w = 1; w < SKINNY_WAVES; ++w) v += red[w][b][col];
        return v;
    };
    auto emit = [&](int c) {        // epilogue of column tile c from the tile in `red`
        const int n0 = (tile0 + c) * 32;
        if (p.gate_out) {
            // interleaved gated layer: columns 0..15 of the tile are u, 16..31 the matching v; silu(u) * v is formed by one thread per
            // (row, pair) into LDS, then two threads per row pack 8 consecutive k each into the hi / lo operand of the next GEMM
            // (K_out = N / 2, zeros for the pad rows of the batch tile).  (Round 2: the packing threads summed the eight partial tiles for
            // their 16 values themselves -- 64 busy threads, 448 idle, ~3 us per column tile.)
            const int half = p.N / 2;
            __syncthreads();                 // the previous column tile's packing threads are done with gtile
            for (int idx = tid; idx < NB * 32 * 16; idx += 64 * SKINNY_WAVES) {
                const int b = idx >> 4, jj = idx & 15;
                const int kout = (tile0 + c) * 16 + jj;
                float u = tsum(b, jj), g = tsum(b, 16 + jj);
                if (p.bias && kout < half) { u += p.bias[kout]; g += p.bias[half + kout]; }
                gtile[idx] = b < p.B ? silu(u) * g : 0.f;
            }
            __syncthreads();
            for (int idx = tid; idx < NB * 32 * 2; idx += 64 * SKINNY_WAVES) {
                const int b = idx >> 1, j8 = (idx & 1) * 8;
                const int kout = (tile0 + c) * 16 + j8;
                if (kout < half) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = gtile[b * 16 + j8 + j];
                    store_packed8(p.gate_out, p.gate_plane, b, kout, half, v);
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < EP; ++q) {
            const int idx = tid + q * 64 * SKINNY_WAVES;
            const int b = idx >> 5, nl = idx & 31;
            const int n = n0 + nl;
            if (b < p.B && n < p.N) {
                float s = tsum(b, nl);
                const long o = (long)b * p.ldy + n;
                if (p.bias) s += bpre[c][q];
                p.y[o] = p.res ? rpre[c][q] + s : s;
            }
        }
    };
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t * 32 + rst_mfma32_row(e, lane)][i] = acc[t][c][e];
        __syncthreads();
        if (nsplit == 1) {
            emit(c);
        } else {
            // this split's partial tile (the 8 waves summed in wave order) -> ws, write-through.  16-byte pieces: a 4-byte `sc1`
            // store is a fabric write of its own (MI355X_MICROARCH.md: ~6x the time per byte of a 16-byte one), and a tile is 1024
            // of them per workgroup
            const int n0 = (tile0 + c) * 32;
            if ((p.N & 3) == 0) {
                if (tid < NB * 32 * 8) {
                    const int b = tid >> 3, q4 = (tid & 7) * 4;
                    if (b < p.B && n0 + q4 < p.N) {
                        const f32x4 v = {tsum(b, q4), tsum(b, q4 + 1), tsum(b, q4 + 2), tsum(b, q4 + 3)};
                        float* dst = p.ws + ((long)blockIdx.y * (NB * 32) + b) * p.N + n0 + q4;
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < EP; ++q) {
                    const int idx = tid + q * 64 * SKINNY_WAVES;
                    const int b = idx >> 5, n = n0 + (idx & 31);
                    if (b < p.B && n < p.N)
                        __hip_atomic_store(p.ws + ((long)blockIdx.y * (NB * 32) + b) * p.N + n, tsum(b, idx & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if (nsplit == 1) return;
    // every storing wave drains its write-through stores, ONE lane bumps the group's counter; the last arriver combines
    // (publish form: `sc1` stores, asm wait, relaxed agent-scope counter and loads -- MI355X_MICROARCH.md handoff-flag / splitk-seam)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(p.counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm_last = prev == (unsigned)nsplit - 1;
        if (sm_last) __hip_atomic_store(p.counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!sm_last) return;
    // the partials of ALL CT tiles are requested together (a round trip to memory each way: the writers' stores left their L2) and
    // summed into red[c] -- the eight per-wave slots double as per-tile slots here -- then the epilogues run
    single = true;
    static_assert(CT <= SKINNY_WAVES, "one LDS slot per column tile");
    {
        float v[CT][EP];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < EP; ++q) v[c][q] = 0.f;
        for (int ks = 0; ks < nsplit; ks += 4) {          // split order, CT * EP * 4 partials in flight
            float t[CT][EP][4];
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int q = 0; q < EP; ++q) {
                    const int idx = tid + q * 64 * SKINNY_WAVES;
                    const int bc = min(idx >> 5, p.B - 1), nc = min((tile0 + c) * 32 + (idx & 31), p.N - 1);   // pad: clamped, unused
                    rst_load_partials<4>(p.ws + (long)bc * p.N + nc, (long)(NB * 32) * p.N, ks, nsplit, t[c][q]);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ks + u < nsplit) {
#pragma unroll
                    for (int c = 0; c < CT; ++c)
#pragma unroll
                        for (int q = 0; q < EP; ++q) v[c][q] += t[c][q][u];
                }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < EP; ++q) {
                const int idx = tid + q * 64 * SKINNY_WAVES;
                red[c][idx >> 5][idx & 31] = v[c][q];
            }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        slot = c;
        emit(c);
    }
}

// ---- fp32-input form (round 3): no activation-packing launch in front of the GEMM ---------------------------------------------------
// x [B][ldx] fp32 goes straight into the launch; mode 1 (RMSNorm) puts x * alpha into the operand and applies 1 / rms to the
// accumulators (the row factor commutes with the contraction; the lanes sum the squares of what they convert and the waves' partial
// sums meet in LDS next to the partial tiles).  The MFMA wants lane l = (batch row l % 32, k-octet l / 32) -- a gather of 16-byte
// pieces from 32 rows 4 KB apart, which the CU's address path serves one cache line at a time: a first version that loaded the
// operand that way measured 9.6-12.9 us per launch against ~6 us for the packed GEMM (and so cost what the packing launch it
// removed had cost).  Here a wave loads its slice COALESCED (8 lanes cover one row's 128 bytes of a 32-k round: 8 full lines per
// instruction), all rounds of a chunk up front, and transposes through a wave-private LDS tile ([rows][32 + 4] floats: conflict-free
// ds_read_b128 in operand order); the RMSNorm gains of the chunk ride along in LDS.  K % 256 == 0 (every wave owns whole 32-k rounds).
template <int NB, int CT>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_skinny_x32_kernel(const SkinnyParams p) {
    constexpr int R = NB * 32;                       // batch rows (padded)
    constexpr int CH = NB == 1 ? 8 : 4;              // steps per chunk: the x values of a chunk sit in registers until their round
    constexpr int QI = R / 8;                        // load instructions per round (8 rows each)
    constexpr int ST = R * 36 + 16 * CH;             // floats of a wave's staging area: tile [R][36] | gains [16 * CH]
    constexpr int LDS_F = SKINNY_WAVES * ST > SKINNY_WAVES * R * 33 + SKINNY_WAVES * R ? SKINNY_WAVES * ST : SKINNY_WAVES * R * 33 + SKINNY_WAVES * R;
    __shared__ __attribute__((aligned(16))) float lds[LDS_F];
    float (*red)[R][33] = reinterpret_cast<float (*)[R][33]>(lds);                       // [waves][R][33] partial tiles (after the K loop)
    float (*ssq_red)[R] = reinterpret_cast<float (*)[R]>(lds + SKINNY_WAVES * R * 33);   // [waves][R]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stage = lds + wave * ST;
    float* gain = stage + R * 36;
    const int tiles = (p.N + 31) / 32;
    const int tile0 = blockIdx.x * CT;
    const int steps = p.K / 16;
    const int per = steps / SKINNY_WAVES;            // K % 256 == 0: whole, even
    const int s0 = wave * per, s1 = s0 + per;
    const unsigned short* wt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wt[c] = p.w + ((long)min(tile0 + c, tiles - 1) * steps * 64 + lane) * 8;
    f32x16 acc[NB][CT];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;
    constexpr int EP = R * 32 / (64 * SKINNY_WAVES);
    float rpre[CT][EP], bpre[CT][EP];
    if (!p.gate_out) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < EP; ++q) {
                const int idx = tid + q * 64 * SKINNY_WAVES;
                const int b = min(idx >> 5, p.B - 1), n = min((tile0 + c) * 32 + (idx & 31), p.N - 1);
                rpre[c][q] = p.res ? p.res[(long)b * p.ldy + n] : 0.f;
                bpre[c][q] = p.bias ? p.bias[n] : 0.f;
            }
    }
    // coalesced source of this lane: row q * 8 + lane / 8 of a round, floats (lane % 8) * 4 .. +3 of its 32
    const float* xsrc[QI];
#pragma unroll
    for (int q = 0; q < QI; ++q) xsrc[q] = p.xf + (long)min(q * 8 + (lane >> 3), p.B - 1) * p.ldx + (lane & 7) * 4;
    // operand side: batch row lane % 32 of every tile; rows past B are cleared through a mask
    unsigned xmask[NB];
    float ssq[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        xmask[t] = t * 32 + (lane & 31) < p.B ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(xmask[t]));
        ssq[t] = 0.f;
    }
    constexpr bool WAHEAD = CT <= 2;                 // weights of the whole chunk up front (else per round: the 4-tile head is HBM-bound anyway)
    for (int s = s0; s < s1; s += CH) {
        const int nst = min(CH, s1 - s);             // even
        f32x4 xr[CH / 2][QI];
#pragma unroll
        for (int r = 0; r < CH / 2; ++r) {
            const int rr = min(r, nst / 2 - 1);      // a round past the chunk's end re-reads its last one (never used)
#pragma unroll
            for (int q = 0; q < QI; ++q) xr[r][q] = *reinterpret_cast<const f32x4*>(xsrc[q] + (s + 2 * rr) * 16);
        }
        f32x4 ga = {1.f, 1.f, 1.f, 1.f};
        if (p.xmode == 1 && lane < 4 * CH) ga = *reinterpret_cast<const f32x4*>(p.alpha + s * 16 + min(lane * 4, nst * 16 - 4));
        bf16x8 a[WAHEAD ? CH : 2][CT];
        if (WAHEAD) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    a[u][c] = u < nst ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wt[c] + (long)(s + u) * 512)) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 4 * CH) *reinterpret_cast<f32x4*>(gain + lane * 4) = ga;
#pragma unroll
        for (int r = 0; r < CH / 2; ++r) {
            if (2 * r < nst) {                       // wave-uniform
                if (!WAHEAD) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            a[u][c] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wt[c] + (long)(s + 2 * r + u) * 512));
                }
                // transpose of the round through the wave's tile: written row-major as loaded, read in operand order
#pragma unroll
                for (int q = 0; q < QI; ++q) *reinterpret_cast<f32x4*>(stage + (q * 8 + (lane >> 3)) * 36 + (lane & 7) * 4) = xr[r][q];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gain + (2 * r + u) * 16 + (lane >> 5) * 8);
                    const f32x4 g1 = *reinterpret_cast<const f32x4*>(gain + (2 * r + u) * 16 + (lane >> 5) * 8 + 4);
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        const float* src = stage + (t * 32 + (lane & 31)) * 36 + u * 16 + (lane >> 5) * 8;
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float xj = __uint_as_float(__float_as_uint(j < 4 ? v0[j & 3] : v1[j & 3]) & xmask[t]);
                            ssq[t] = fmaf(xj, xj, ssq[t]);
                            v[j] = xj * (j < 4 ? g0[j & 3] : g1[j & 3]);
                        }
                        u32x4 hi, lo;
                        split_hi_lo8(v, hi, lo);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, hi), bl = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
                        for (int c = 0; c < CT; ++c) {
                            const bf16x8 aw = a[WAHEAD ? 2 * r + u : u][c];
                            acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, aw, acc[t][c], 0, 0, 0);
                            acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, aw, acc[t][c], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();     // the tile is rewritten by the next round
            }
        }
    }
    __syncthreads();                                 // every wave is done with its staging area: `red` / `ssq_red` take the space
    const int i = lane & 31;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const float sq = ssq[t] + __shfl_xor(ssq[t], 32);
        if (lane < 32) ssq_red[wave][t * 32 + lane] = sq;
    }
    // epilogue scratch behind the partial tiles: 1 / rms per row (computed once, not per output element) and the gated tile
    float* rinv_s = lds + SKINNY_WAVES * R * 33 + SKINNY_WAVES * R;      // [R]
    float* gtile = rinv_s + R;                                          // [R][16] silu(u) * v of a column tile
    static_assert(SKINNY_WAVES * R * 33 + SKINNY_WAVES * R + R + R * 16 <= LDS_F, "epilogue scratch does not fit behind the partial tiles");
    auto tsum = [&](int b, int col) {
        float v = red[0][b][col];
#pragma unroll
        for (int w = 1; w < SKINNY_WAVES; ++w) v += red[w][b][col];
        return v * rinv_s[b];
    };
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t * 32 + rst_mfma32_row(e, lane)][i] = acc[t][c][e];
        __syncthreads();
        if (c == 0) {               // RMSNorm: 1 / sqrt(eps + mean(x[b]^2)), the factor the operand left out (sums in wave order)
            if (tid < R) {
                float q = ssq_red[0][tid];
#pragma unroll
                for (int w = 1; w < SKINNY_WAVES; ++w) q += ssq_red[w][tid];
                rinv_s[tid] = p.xmode == 1 ? 1.0f / sqrtf(p.eps + q / (float)p.K) : 1.0f;
            }
            __syncthreads();
        }
        const int n0 = (tile0 + c) * 32;
        if (p.gate_out) {
            // interleaved gated layer: columns 0..15 of the tile are u, 16..31 the matching v.  silu(u) * v is formed by one thread per
            // (row, pair) -- R * 16 of them -- into LDS, then R * 2 threads pack 8 consecutive k each into the hi / lo operand of the
            // next GEMM (K_out = N / 2; zeros for the pad rows of the batch tile).  (Round 2's form had the R * 2 packing threads sum
            // the eight partial tiles for their 16 values themselves: 64 threads busy, 448 idle, ~3 us of a 10 us launch.)
            const int half = p.N / 2;
            for (int idx = tid; idx < R * 16; idx += 64 * SKINNY_WAVES) {
                const int b = idx >> 4, jj = idx & 15;
                const int kout = (tile0 + c) * 16 + jj;
                float u = tsum(b, jj), g = tsum(b, 16 + jj);
                if (p.bias && kout < half) { u += p.bias[kout]; g += p.bias[half + kout]; }
                gtile[idx] = b < p.B ? silu(u) * g : 0.f;
            }
            __syncthreads();
            for (int idx = tid; idx < R * 2; idx += 64 * SKINNY_WAVES) {
                const int b = idx >> 1, j8 = (idx & 1) * 8;
                const int kout = (tile0 + c) * 16 + j8;
                if (kout < half) {
                    float v[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) v[jj] = gtile[b * 16 + j8 + jj];
                    store_packed8(p.gate_out, p.gate_plane, b, kout, half, v);
                }
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < EP; ++q) {
            const int idx = tid + q * 64 * SKINNY_WAVES;
            const int b = idx >> 5, nl = idx & 31;
            const int n = n0 + nl;
            if (b < p.B && n < p.N) {
                float sv = tsum(b, nl);
                const long o = (long)b * p.ldy + n;
                if (p.bias) sv += bpre[c][q];
                p.y[o] = p.res ? rpre[c][q] + sv : sv;
            }
        }
    }
}

}  // namespace

int rst_launch_skinny_pack_weight(const unsigned short* w, unsigned short* wp, int N, int K, int interleave, hipStream_t stream) {
    RST_REQUIRE(w && wp && N > 0 && K > 0 && K % 16 == 0, "skinny_pack_weight: bad arguments (K %% 16 == 0 required, K=%d)", K);
    RST_REQUIRE(!interleave || N % 32 == 0, "skinny_pack_weight: interleaving the two halves needs N %% 32 == 0 (N=%d)", N);
    const long total = (long)((N + 31) / 32) * 32 * (K / 8);
    hipLaunchKernelGGL(skinny_pack_weight_kernel, dim3(cap_grid((total + 255) / 256, 8192)), dim3(256), 0, stream, w, wp, N, K, interleave);
    return rst_check_launch("skinny_pack_weight");
}

int rst_launch_skinny_pack_act(const float* x, const float* alpha, unsigned short* xp, int B, int K, int ldx, int mode, float eps,
                               hipStream_t stream) {
    RST_REQUIRE(x && xp && B >= 1 && B <= 64 && K > 0 && K % 16 == 0 && ldx % 4 == 0, "skinny_pack_act: bad arguments (B=%d K=%d)", B, K);
    RST_REQUIRE(mode == 0 || (mode == 1 && alpha) || mode == 2, "skinny_pack_act: mode 0 / 1 (needs alpha) / 2");
    hipLaunchKernelGGL(skinny_pack_act_kernel, dim3((B + 31) / 32 * 32), dim3(256), 0, stream, x, alpha, xp, B, K, ldx, mode, eps);
    return rst_check_launch("skinny_pack_act");
}

// Column tiles per workgroup.  A workgroup's bytes go through ONE CU's L1 (~60 GB/s sustained): the activations of a step (2 KB)
// are shared by CT weight tiles (CT KB), and two workgroups on a CU share that path.  Without a K split: the fewest tiles that
// leave at most one workgroup per CU, ceil(tiles / CUs) -- round 3's thresholds (one tile below 512, two below 2048) put 1.5
// workgroups per CU on the 12288-row qkv and 1.4 on the 22528-row gated layer of the Moshi-7B shape, and the CUs that held two took
// twice as long (31 / 47 us; `profiles/r04_skinny_ct.txt`: the batch-32 LM frame 7.02 -> 6.44 ms with this rule).  With a K split
// (long K, few tiles): as many as the kernel has, as before.
static int skinny_ct_round3(int B, int tiles, int K, int split) {
    const int max_ct = B <= 32 ? 4 : 2;
    if (K >= 2048 && (split > 1 || tiles >= 192 * max_ct)) return tiles >= max_ct ? max_ct : (tiles >= 2 ? 2 : 1);
    if (B <= 32) return tiles >= 2048 ? 4 : (tiles >= 512 ? 2 : 1);
    return tiles >= 512 ? 2 : 1;
}
static int skinny_ct(int B, int tiles, int K, int split) {
    // (two batch tiles: up to 3 column tiles without a K split -- 4 spill; RST_SKINNY_CT2MAX = 2, tools build, restores round 3's limit)
    static const int ct2max = rst_knob("RST_SKINNY_CT2MAX", 3) > 3 ? 3 : rst_knob("RST_SKINNY_CT2MAX", 3);
    const int max_ct = B <= 32 ? 4 : (split > 1 ? 2 : ct2max);
    int ct;
    if (split > 1) ct = tiles >= max_ct ? max_ct : (tiles >= 2 ? 2 : 1);
    else {
        const int cus = rst_cu_count();
        ct = (tiles + cus - 1) / cus;
        ct = ct < 1 ? 1 : (ct > max_ct ? max_ct : ct);
    }
    // tools build only -- RST_SKINNY_CT: 1..4 forces the column tiles per workgroup, 8 = round 3's thresholds
    static const int kct = rst_knob("RST_SKINNY_CT", 0);
    if (kct >= 1 && kct <= 4) ct = kct < max_ct ? kct : max_ct;
    if (kct == 8) ct = skinny_ct_round3(B, tiles, K, split);
    return ct > tiles ? tiles : ct;
}

// K splits of a launch (1: none).  The in-launch hand-off costs ~8 us (drained write-through stores, the counter round trip, the
// partials read back from memory, measured: 8 MB as 1024 x 4096 takes 14.7 us split 16 ways), so only long K pays: K >= 8192 for one
// batch tile (the 4096 x 11264 gated projection: 47 -> 33 us), K >= 4096 for two (64 x 4096 x 4096: 25 -> 20 us); at K = 4096 and
// <= 32 rows the split and the plain form measured equal (17 us), below that the split form loses.  Then: split until ~256-384
// workgroups exist, every wave keeping >= 2 MFMA steps.
int rst_skinny_bf16_split_plan_impl(int B, int N, int K) {
    if (B < 1 || B > 64 || N < 1 || K < (B <= 32 ? 8192 : 4096) || K % 16) return 1;
    const int tiles = (N + 31) / 32, steps = K / 16;
    const int max_ct = B <= 32 ? 4 : 2;
    const int ct = tiles >= max_ct ? max_ct : (tiles >= 2 ? 2 : 1);
    const int groups = (tiles + ct - 1) / ct;
    int s = 1;
    while (groups * s * 2 <= 384 && steps / (s * 2 * SKINNY_WAVES) >= 2 && s < 16) s *= 2;
    return s;
}

int rst_launch_gemm_skinny(const SkinnyParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.B <= 64 && p.N > 0 && p.K > 0 && p.K % 16 == 0, "gemm_skinny: need 1 <= B <= 64 and K %% 16 == 0 (B=%d K=%d)", p.B, p.K);
    RST_REQUIRE((p.xp || p.xf) && p.w && (p.y || p.gate_out), "gemm_skinny: null pointer");
    const bool xf = p.xp == nullptr;
    RST_REQUIRE(!xf || (p.K % 256 == 0 && p.ldx % 4 == 0 && p.ldx >= p.K && (uintptr_t)p.xf % 16 == 0 && (p.xmode == 0 || (p.xmode == 1 && p.alpha)) && p.split_k <= 1),
                "gemm_skinny: the fp32-input form needs K %% 256 == 0, 16-byte aligned rows (ldx %% 4 == 0), mode 0 / 1 (with alpha) and no K split");
    RST_REQUIRE(!p.gate_out || (p.N % 32 == 0 && !p.res), "gemm_skinny: the gated epilogue needs N %% 32 == 0 and takes no residual");
    const int tiles = (p.N + 31) / 32;
    const int threads = 64 * SKINNY_WAVES;
    const int split = p.split_k > 1 ? p.split_k : 1;
    RST_REQUIRE(split == 1 || (p.ws && p.counters && split <= 64), "gemm_skinny: split-K needs the scratch buffers (split <= 64)");
    const int ct = skinny_ct(p.B, tiles, p.K, split);
    const dim3 grid((tiles + ct - 1) / ct, split);
    if (xf) {
        if (p.B <= 32) {
            if (ct == 4) hipLaunchKernelGGL((gemm_skinny_x32_kernel<1, 4>), grid, dim3(threads), 0, stream, p);
            else if (ct == 3) hipLaunchKernelGGL((gemm_skinny_x32_kernel<1, 3>), grid, dim3(threads), 0, stream, p);
            else if (ct == 2) hipLaunchKernelGGL((gemm_skinny_x32_kernel<1, 2>), grid, dim3(threads), 0, stream, p);
            else hipLaunchKernelGGL((gemm_skinny_x32_kernel<1, 1>), grid, dim3(threads), 0, stream, p);
        } else {
            if (ct == 3) hipLaunchKernelGGL((gemm_skinny_x32_kernel<2, 3>), grid, dim3(threads), 0, stream, p);
            else if (ct == 2) hipLaunchKernelGGL((gemm_skinny_x32_kernel<2, 2>), grid, dim3(threads), 0, stream, p);
            else hipLaunchKernelGGL((gemm_skinny_x32_kernel<2, 1>), grid, dim3(threads), 0, stream, p);
        }
        return rst_check_launch("gemm_skinny_x32");
    }
    if (p.B <= 32) {
        if (ct == 4) hipLaunchKernelGGL((gemm_skinny_kernel<1, 4>), grid, dim3(threads), 0, stream, p);
        else if (ct == 3) hipLaunchKernelGGL((gemm_skinny_kernel<1, 3>), grid, dim3(threads), 0, stream, p);
        else if (ct == 2) hipLaunchKernelGGL((gemm_skinny_kernel<1, 2>), grid, dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_kernel<1, 1>), grid, dim3(threads), 0, stream, p);
    } else {
        if (ct == 3) hipLaunchKernelGGL((gemm_skinny_kernel<2, 3>), grid, dim3(threads), 0, stream, p);
        else if (ct == 2) hipLaunchKernelGGL((gemm_skinny_kernel<2, 2>), grid, dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_kernel<2, 1>), grid, dim3(threads), 0, stream, p);
    }
    return rst_check_launch("gemm_skinny");
}

// ======================================================================================================================
// fp8 (OCP e4m3) variant of the skinny GEMM -- BASELINE.json configs[4]: "fp8 MFMA GEMMs for temporal attention, batch 32".
// Weights: per-row scale (amax / 448), quantised once; activations: per-row dynamic scale, quantised by the prologue launch.
// v_mfma_f32_32x32x16_fp8_fp8 consumes 8 bytes per lane per step; two steps are packed per 16-byte lane load:
//   [tile of 32 rows][K/32][64 lanes = 32 * ((k / 8) % 2) + row % 32][16 B = step 2p (8 k) | step 2p+1 (8 k)].
// Half the streamed bytes and half the MFMA work of the bf16 hi/lo path, at fp8 accuracy (3 mantissa bits: ~1e-2 on logits);
// opt-in, never the default.
// ======================================================================================================================
namespace {

__device__ __forceinline__ long fp8_packed_index(int row, int k, int K) {     // byte offset
    return ((((long)(row >> 5) * (K >> 5) + (k >> 5)) * 64) + ((k >> 3) & 1) * 32 + (row & 31)) * 16 + ((k >> 4) & 1) * 8 + (k & 7);
}

// v / scale with a true division (bit-compatible with `(t / scale).to(float8_e4m3fn)`), round-to-nearest-even conversion
__device__ __forceinline__ uint2 quant8_fp8(const float (&v)[8], float scale) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] / scale, v[1] / scale, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] / scale, v[3] / scale, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] / scale, v[5] / scale, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] / scale, v[7] / scale, hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

__device__ __forceinline__ float block_max_256(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one workgroup per (padded) weight row
__global__ __launch_bounds__(256) void skinny_pack_weight_fp8_kernel(const unsigned short* __restrict__ w, unsigned char* __restrict__ wp,
                                                                    float* __restrict__ scale, int N, int K) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float amax = 0.f;
    if (row < N)
        for (int k = tid * 8; k < K; k += 2048) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(w + (long)row * K + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16_lo(v[j])), fabsf(bf16_hi(v[j]))));
        }
    amax = block_max_256(amax, red);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) scale[row] = sc;
    const float inv = sc;
    for (int k = tid * 8; k < K; k += 2048) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row < N) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(w + (long)row * K + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_lo(q[j]); v[2 * j + 1] = bf16_hi(q[j]); }
        }
        *reinterpret_cast<uint2*>(wp + fp8_packed_index(row, k, K)) = quant8_fp8(v, inv);
    }
}

// one workgroup per (padded) batch row: prologue (0 identity, 1 RMSNorm, 2 SiLU gate) -> per-row amax -> fp8
constexpr int FP8_ACT_CHUNKS = 8;      // K <= 8 * 2048
__global__ __launch_bounds__(256) void skinny_pack_act_fp8_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                                 unsigned char* __restrict__ xp, float* __restrict__ xscale, int B, int K,
                                                                 int ldx, int mode, float eps) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float vals[FP8_ACT_CHUNKS][8];
    float ssq = 0.f;
#pragma unroll
    for (int c = 0; c < FP8_ACT_CHUNKS; ++c) {
        const int k = tid * 8 + c * 2048;
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[c][j] = 0.f;
        if (b < B && k < K) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + k + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { vals[c][j] = a0[j]; vals[c][4 + j] = a1[j]; }
            if (mode == 2) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(x + (long)b * ldx + K + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vals[c][j] = silu(vals[c][j]) * g0[j]; vals[c][4 + j] = silu(vals[c][4 + j]) * g1[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ssq = fmaf(vals[c][j], vals[c][j], ssq);
        }
    }
    if (mode == 1) {
        float s = wave_sum(ssq);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const float r = 1.0f / sqrtf(eps + (red[0] + red[1] + red[2] + red[3]) / (float)K);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < FP8_ACT_CHUNKS; ++c) {
            const int k = tid * 8 + c * 2048;
            if (b < B && k < K) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(alpha + k), g1 = *reinterpret_cast<const f32x4*>(alpha + k + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vals[c][j] *= g0[j] * r; vals[c][4 + j] *= g1[j] * r; }
            }
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < FP8_ACT_CHUNKS; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(vals[c][j]));
    amax = block_max_256(amax, red);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) xscale[b] = sc;
    const float inv = sc;
#pragma unroll
    for (int c = 0; c < FP8_ACT_CHUNKS; ++c) {
        const int k = tid * 8 + c * 2048;
        if (k < K) *reinterpret_cast<uint2*>(xp + fp8_packed_index(b, k, K)) = quant8_fp8(vals[c], inv);
    }
}

template <int NB, int CT>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void gemm_skinny_fp8_kernel(const SkinnyFp8Params p) {
    __shared__ float red[SKINNY_WAVES][NB * 32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = (p.N + 31) / 32;
    const int tile0 = blockIdx.x * CT;
    const int pairs = p.K / 32;
    const int per = (pairs + SKINNY_WAVES - 1) / SKINNY_WAVES;
    const int s0 = wave * per, s1 = min(pairs, s0 + per);
    const unsigned char* xq = p.xp + (long)lane * 16;
    const unsigned char* wt[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) wt[c] = p.wp + ((long)min(tile0 + c, tiles - 1) * pairs * 64 + lane) * 16;
    f32x16 acc[NB][CT];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;
    typedef long i64x2 __attribute__((ext_vector_type(2)));
    constexpr int UN = 4;
    for (int s = s0; s < s1; s += UN) {
        i64x2 a[UN][CT], bx[UN][NB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool ok = s + u < s1;
            const long so = (long)(s + u) * 1024;
#pragma unroll
            for (int c = 0; c < CT; ++c) a[u][c] = ok ? __builtin_nontemporal_load(reinterpret_cast<const i64x2*>(wt[c] + so)) : i64x2{0, 0};
#pragma unroll
            for (int t = 0; t < NB; ++t) bx[u][t] = ok ? *reinterpret_cast<const i64x2*>(xq + (long)t * pairs * 1024 + so) : i64x2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bx[u][t][0], a[u][c][0], acc[t][c], 0, 0, 0);
                    acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bx[u][t][1], a[u][c][1], acc[t][c], 0, 0, 0);
                }
    }
    const int i = lane & 31;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][t * 32 + rst_mfma32_row(e, lane)][i] = acc[t][c][e];
        __syncthreads();
        const int n0 = (tile0 + c) * 32;
        for (int idx = tid; idx < NB * 32 * 32; idx += 64 * SKINNY_WAVES) {
            const int b = idx >> 5, nl = idx & 31;
            const int n = n0 + nl;
            if (b < p.B && n < p.N) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < SKINNY_WAVES; ++w) s += red[w][b][nl];
                s *= p.xscale[b] * p.wscale[n];
                const long o = (long)b * p.ldy + n;
                if (p.bias) s += p.bias[n];
                p.y[o] = p.res ? p.res[o] + s : s;
            }
        }
    }
}

}  // namespace

int rst_launch_skinny_pack_weight_fp8(const unsigned short* w, unsigned char* wp, float* scale, int N, int K, hipStream_t stream) {
    RST_REQUIRE(w && wp && scale && N > 0 && K > 0 && K % 32 == 0, "skinny_pack_weight_fp8: bad arguments (K %% 32 == 0 required, K=%d)", K);
    hipLaunchKernelGGL(skinny_pack_weight_fp8_kernel, dim3((N + 31) / 32 * 32), dim3(256), 0, stream, w, wp, scale, N, K);
    return rst_check_launch("skinny_pack_weight_fp8");
}

int rst_launch_skinny_pack_act_fp8(const float* x, const float* alpha, unsigned char* xp, float* xscale, int B, int K, int ldx, int mode,
                                   float eps, hipStream_t stream) {
    RST_REQUIRE(x && xp && xscale && B >= 1 && B <= 64 && K > 0 && K % 32 == 0 && K <= 2048 * FP8_ACT_CHUNKS && ldx % 4 == 0,
                "skinny_pack_act_fp8: bad arguments (B=%d K=%d; K %% 32 == 0, K <= %d)", B, K, 2048 * FP8_ACT_CHUNKS);
    RST_REQUIRE(mode == 0 || (mode == 1 && alpha) || mode == 2, "skinny_pack_act_fp8: mode 0 / 1 (needs alpha) / 2");
    hipLaunchKernelGGL(skinny_pack_act_fp8_kernel, dim3((B + 31) / 32 * 32), dim3(256), 0, stream, x, alpha, xp, xscale, B, K, ldx, mode, eps);
    return rst_check_launch("skinny_pack_act_fp8");
}

int rst_launch_gemm_skinny_fp8(const SkinnyFp8Params& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 1 && p.B <= 64 && p.N > 0 && p.K > 0 && p.K % 32 == 0, "gemm_skinny_fp8: need 1 <= B <= 64 and K %% 32 == 0 (B=%d K=%d)", p.B, p.K);
    RST_REQUIRE(p.xp && p.wp && p.xscale && p.wscale && p.y, "gemm_skinny_fp8: null pointer");
    const int tiles = (p.N + 31) / 32;
    const int threads = 64 * SKINNY_WAVES;
    if (p.B <= 32) {
        if (tiles >= 2048) hipLaunchKernelGGL((gemm_skinny_fp8_kernel<1, 4>), dim3((tiles + 3) / 4), dim3(threads), 0, stream, p);
        else if (tiles >= 512) hipLaunchKernelGGL((gemm_skinny_fp8_kernel<1, 2>), dim3((tiles + 1) / 2), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_fp8_kernel<1, 1>), dim3(tiles), dim3(threads), 0, stream, p);
    } else {
        if (tiles >= 512) hipLaunchKernelGGL((gemm_skinny_fp8_kernel<2, 2>), dim3((tiles + 1) / 2), dim3(threads), 0, stream, p);
        else hipLaunchKernelGGL((gemm_skinny_fp8_kernel<2, 1>), dim3(tiles), dim3(threads), 0, stream, p);
    }
    return rst_check_launch("gemm_skinny_fp8");
}
