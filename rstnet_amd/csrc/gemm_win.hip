// Windowed fp32 GEMM on the gfx950 f32 matrix cores -- the one contraction kernel behind every causal
// Conv1d / ConvTranspose1d / Linear of the MimiCodec path.
//
// Activations are kept channels-last ([B][T][C], C contiguous).  In that layout a causal convolution with
// kernel Kw, stride S over C input channels is a plain GEMM whose A rows are *overlapping windows* of the
// flat activation array:
//      y[b, t, n] = sum_{k < Kw*C} A(b,t,k) * W[n][k],      A(b,t,k) = xflat_b[(t*S - P)*C + k]
// (P = Kw_eff - S left padding; out-of-range elements are 0, or replicated, or come from a caller-owned
// history buffer in streaming mode).  A transposed convolution with kernel q*S, stride S is the same thing
// with window q steps (S_rows = 1, P = q-1) and N = S*Cout output columns, written as S consecutive output
// time steps; a Linear layer is the degenerate window (Kw=1).  See DESIGN.md section 3.
//
// Tile: 4 waves, each TM x TN tiles of v_mfma_f32_32x32x2_f32, k-chunks of KB = 32 (16 for the large-M 128 x 128 case),
// register-prefetched double-buffered LDS ([rows][KB + 4] floats: a 16-lane ds_read_b128 group covers 16 distinct 16-byte
// slots -> conflict free).
// K order inside a BK chunk is permuted (lane half h owns k = 8s + 4h .. +3) so that fragments are read with
// one ds_read_b128 per four MFMAs; both operands use the same permutation so the product is unchanged.
#include <algorithm>
#include "rst_common.h"
#include "rst_kernels.h"
#include "b3_common.h"

namespace {

constexpr int BK = 32;

// ---- epilogue of one tile: bias -> GELU? -> (residual + scale *) -> ELU? -> store; 32 consecutive columns per half wave.
// The residual loads of a 32 x 32 block are issued back to back (predicated, no branches) before any of them is consumed.
template <int TM, int TN, bool FULL = false>
__device__ __forceinline__ void gw_epilogue(const GemmWinParams& p, const f32x16 (&acc)[TM][TN], int m0w, int n0w, int M, int lane) {
    // FULL: every row / column of the tile exists (the tile-streaming kernel's interior tiles) -- no predicates, no branches
    const bool has_res = p.res != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0w + j * 32 + (lane & 31);
        const bool n_ok = FULL || n < p.N;
        const float bias = (p.bias && n_ok) ? p.bias[n] : 0.0f;
        const float scale = (p.scale && n_ok) ? p.scale[n] : 1.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0w + i * 32 + 4 * (lane >> 5);
            const long o = (long)mb * p.ldy + n;        // element e of the accumulator sits (e & 3) + 8 * (e >> 2) rows further down
            float r[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dm = (e & 3) + 8 * (e >> 2);
                r[e] = (has_res && n_ok && (FULL || mb + dm < M)) ? p.res[o + (long)dm * p.ldy] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dm = (e & 3) + 8 * (e >> 2);
                float v = acc[i][j][e] + bias;
                if (p.act_out == 1) v = rst_gelu(v);
                if (has_res) v = r[e] + scale * v;
                if (p.act_out == 2) v = rst_elu(v);
                if (n_ok && (FULL || mb + dm < M)) p.y[o + (long)dm * p.ldy] = v;
            }
        }
    }
}

// One output tile, start to finish (staging coordinates, K loop, split-K reduction, epilogue).  FAST compiles the one-basic-block
// K loop of interior tiles and the split-K protocol in; the tile-streaming kernel below calls it with FAST = false for the few
// tiles that touch an utterance edge.
template <int TM, int TN, int WM, int WN, bool VEC, bool ELU, int KB, bool FAST>
__device__ __forceinline__ void gw_tile(const GemmWinParams& p, const int tile, const int m0, const int n0, float* smem) {
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int LDS_LD = KB + 4;      // floats per LDS row
    constexpr int RP = 1024 / KB;       // rows staged per pass of the 256 threads (16 bytes each)
    constexpr int RA = BM / RP;         // A rows staged per thread
    constexpr int RB = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0, "tile smaller than one staging pass");
    float* As = smem;                     // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;   // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int M = p.B * p.T_out;
    const int TC = p.T_in * p.C;
    const int PC = p.P * p.C;

    // ---- per-thread staging coordinates
    const int lrow = tid / (KB / 4);         // 0 .. RP-1
    const int lk = (tid % (KB / 4)) * 4;     // 0, 4, .., KB-4
    long a_off[RA];                  // float offset of the batch inside x
    long h_off[RA];                  // float offset of the batch inside hist
    int a_f0[RA];                    // flat index of the window start inside the batch (may be < 0)
    int a_klo[RA], a_khi[RA];        // k range served by a plain load from x (everything else: padding / history)
    bool a_ok[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + lrow + RP * j;
        a_ok[j] = m < M;
        const int mm = a_ok[j] ? m : 0;
        const int b = mm / p.T_out;
        const int t = mm - b * p.T_out;
        a_off[j] = (long)b * p.x_bstride;
        h_off[j] = (long)b * PC;
        a_f0[j] = (t * p.S - p.P) * p.C;
        a_klo[j] = a_ok[j] ? max(0, -a_f0[j]) : 0x7fffffff;
        a_khi[j] = a_ok[j] ? min(p.K, TC - a_f0[j]) : 0;
    }

    f32x4 ra[RA], rb[RB];

    auto load_elem_a = [&](int j, int f) -> float {
        float v = 0.0f;
        if (f >= 0) {
            if (f < TC) v = p.x[a_off[j] + f];
            else if (p.pad_mode == 1) v = p.x[a_off[j] + TC - p.C + f % p.C];  // F.pad(mode="replicate") on the right
        } else if (p.hist) {
            v = p.hist[h_off[j] + PC + f];
        } else if (p.pad_mode == 1) {
            int c = f % p.C;
            if (c < 0) c += p.C;
            v = p.x[a_off[j] + c];
        }
        return v;
    };

    // 16-byte pieces of the general path: the source ADDRESS is chosen per lane (activations / history / replicated edge, or a dummy
    // when the piece is padding), the load itself is unconditional, and padding is cleared with a mask the compiler cannot see
    // through.  A load under a per-lane condition (or one feeding a select it can fold back into a condition) is branched around
    // and waited for on the spot, which turns a k-tile into RA + RB exposed round trips -- the streaming steps live on this path.
    // (Round 6: the mask is applied when the piece is WRITTEN to LDS, after the matrix instructions of the current k-tile -- applied right
    // behind the load it made every k-tile of an edge tile one exposed round trip: the 3072-row layers of a 32-stream frame, a third of
    // whose tiles reach into the history, took 53 us against 32 us for the same shape without edges.)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    int ma[RA], mb[RB];              // all ones: the piece is data; zero: padding
    auto masked_load = [&](const float* src, bool ok, int& m) -> f32x4 {
        m = ok ? -1 : 0;
        asm volatile("" : "+v"(m));
        return *reinterpret_cast<const f32x4*>(ok ? src : p.w);
    };
    auto apply_mask = [&](const f32x4 v, const int m) -> f32x4 {
        const i32x4 b = __builtin_bit_cast(i32x4, v) & m;
        return __builtin_bit_cast(f32x4, b);
    };
    auto load_tiles = [&](int kt) {
        const int k = kt * KB + lk;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            if (VEC) {
                const int f = a_f0[j] + k;
                bool ok = a_ok[j] && k < p.K;
                const float* src = p.x + a_off[j] + f;
                if (f < 0) {
                    if (p.hist) {
                        src = p.hist + h_off[j] + PC + f;
                    } else if (p.pad_mode == 1) {
                        int c = f % p.C;
                        if (c < 0) c += p.C;
                        src = p.x + a_off[j] + c;
                    } else {
                        ok = false;
                    }
                } else if (f >= TC) {
                    if (p.pad_mode == 1) src = p.x + a_off[j] + TC - p.C + f % p.C;   // F.pad(mode="replicate") also replicates the right extra padding
                    else ok = false;
                }
                ra[j] = masked_load(src, ok, ma[j]);
            } else {
                ma[j] = -1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (a_ok[j]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < p.K) v[e] = load_elem_a(j, a_f0[j] + k + e);
                }
                ra[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int n = n0 + lrow + RP * j;
            const float* wp = p.w + (long)n * p.K + k;
            if (VEC) {
                rb[j] = masked_load(wp, n < p.N && k < p.K, mb[j]);
            } else {
                mb[j] = -1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < p.N) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < p.K) v[e] = wp[e];
                }
                rb[j] = v;
            }
        }
    };

    auto store_tiles = [&](int buf, const bool masked = true) {      // (masked = false: the interior loop, whose loads need none)
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            f32x4 v = (VEC && masked) ? apply_mask(ra[j], ma[j]) : ra[j];
            if (ELU) {
                v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]);
            }
            *reinterpret_cast<f32x4*>(a + (lrow + RP * j) * LDS_LD + lk) = v;
        }
#pragma unroll
        for (int j = 0; j < RB; ++j)
            *reinterpret_cast<f32x4*>(b + (lrow + RP * j) * LDS_LD + lk) = (VEC && masked) ? apply_mask(rb[j], mb[j]) : rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // split-K (few-row streaming steps): workgroup blockIdx.y owns k-tiles [kt0, kt1)
    const int nk_all = (p.K + KB - 1) / KB;
    const int nsplit = FAST ? (int)gridDim.y : 1;
    const int per_split = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = FAST ? blockIdx.y * per_split : 0;
    const int nk = min(nk_all, kt0 + per_split);
    if (kt0 < nk) {
        load_tiles(kt0);
        store_tiles(kt0 & 1);
    }
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    auto mma_tile = [&](int buf, int ks0 = 0, int ks1 = KB / 8) {
        const float* a = As + buf * BM * LDS_LD + (wm * TM * 32 + frag_row) * LDS_LD + frag_k;
        const float* b = Bs + buf * BN * LDS_LD + (wn * TN * 32 + frag_row) * LDS_LD + frag_k;
#pragma unroll
        for (int ks = ks0; ks < ks1; ++ks) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
    };

    // Interior tiles (every window of the tile lies inside its utterance, full tile, K a multiple of BK -- all but the
    // first / last tile of an utterance): the K loop is ONE basic block of unconditional 16-byte loads, so the scheduler can
    // spread the global loads and the LDS writes of the next k-tile between the MFMAs of this one instead of running them
    // as separate phases with the matrix pipe idle.
    bool interior = FAST && VEC && gridDim.y == 1 && m0 + BM <= M && n0 + BN <= p.N && p.K % KB == 0;
#pragma unroll
    for (int j = 0; j < RA; ++j) interior = interior && a_klo[j] == 0 && a_khi[j] == p.K;
    if (FAST && __syncthreads_and(interior)) {
        const float* ap[RA];
        const float* bp[RB];
#pragma unroll
        for (int j = 0; j < RA; ++j) ap[j] = p.x + a_off[j] + a_f0[j] + lk;
#pragma unroll
        for (int j = 0; j < RB; ++j) bp[j] = p.w + (long)(n0 + lrow + RP * j) * p.K + lk;
        for (int kt = kt0; kt + 1 < nk; ++kt) {
#pragma unroll
            for (int j = 0; j < RA; ++j) ra[j] = *reinterpret_cast<const f32x4*>(ap[j] + (kt + 1) * KB);
#pragma unroll
            for (int j = 0; j < RB; ++j) rb[j] = *reinterpret_cast<const f32x4*>(bp[j] + (kt + 1) * KB);
            // the loads go out first and stay in flight under the MFMAs of this tile (left alone, the scheduler sinks them to
            // the end of the block to save registers and then waits out the full memory latency) ...
            __builtin_amdgcn_sched_barrier(0);
            mma_tile(kt & 1, 0, KB / 8 - 1);
            __builtin_amdgcn_sched_barrier(0);
            // ... and their LDS writes (other buffer) are spread between the last 16 MFMAs
            mma_tile(kt & 1, KB / 8 - 1, KB / 8);
            store_tiles((kt & 1) ^ 1, false);
#pragma unroll
            for (int g = 0; g < RA + RB; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, (4 * TM * TN) / (RA + RB) > 0 ? (4 * TM * TN) / (RA + RB) : 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                          // DS write
            }
            __syncthreads();
        }
        mma_tile((nk - 1) & 1);
    } else {
        for (int kt = kt0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tiles(kt + 1);
            __builtin_amdgcn_s_setprio(1);
            mma_tile(buf);
            __builtin_amdgcn_s_setprio(0);
            if (kt + 1 < nk) store_tiles(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- split-K reduction: write-through partials + arrival counter; the last workgroup of the tile sums them in a
    // fixed order (deterministic) and runs the epilogue (cdna_hip_programming.md G16, counter form)
    if (FAST && gridDim.y > 1) {
        __shared__ int sm_last;
        float* wsb = p.ws + (long)blockIdx.y * M * p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + (wm * TM + i) * 32 + rst_mfma32_row(e, lane);
                    if (n < p.N && m < M)
                        __hip_atomic_store(wsb + (long)m * p.N + n, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        // publish form: `sc1` write-through stores drained by every storing wave (asm wait), one relaxed agent-scope counter bump,
        // `sc1` loads on the reading side -- valid without a fence pair per MI355X_MICROARCH.md (handoff-flag / splitk-seam rows)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned prev = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sm_last = prev == gridDim.y - 1;
            if (sm_last) __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!sm_last) return;
        // the partials of a 32 x 32 block are read UB splits at a time (128 / 64 loads in flight per lane) and summed in split order
        constexpr int UB = TM * TN == 1 ? 8 : (TM * TN == 2 ? 4 : 1);    // the 2 x 2 configuration is never split (M <= 4096 takes 32-row tiles)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = min(n0 + (wn * TN + j) * 32 + (lane & 31), p.N - 1);       // rows / columns past the edge: clamped, never stored
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                for (int ks = 0; ks < (int)gridDim.y; ks += UB) {
                    float t[16][UB];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = min(m0 + (wm * TM + i) * 32 + rst_mfma32_row(e, lane), M - 1);
                        rst_load_partials<UB>(p.ws + (long)m * p.N + n, (long)M * p.N, ks, (int)gridDim.y, t[e]);
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        if (ks + u < (int)gridDim.y) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[i][j][e] += t[e][u];
                        }
                }
            }
        }
    }

    gw_epilogue<TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, lane);
}

// XCD-aware tile order: blocks that land on one XCD (bid % 8) walk consecutive tiles, n fastest, so the A rows they share stay
// in that XCD's L2: XCD x owns tiles [xcd_first(x), xcd_first(x + 1)).
__device__ __forceinline__ int gw_xcd_first(int tiles, int x) {
    const int q = tiles >> 3, r = tiles & 7;
    return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}

template <int TM, int TN, int WM, int WN, bool VEC, bool ELU, int KB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KB == 16 ? 3 : 1, KB == 16 ? 3 : 2))) void gemm_win_kernel(const GemmWinParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = gw_xcd_first(gridDim.x, blockIdx.x & 7) + (blockIdx.x >> 3);
    const int tiles_n = (p.N + 32 * TN * WN - 1) / (32 * TN * WN);
    gw_tile<TM, TN, WM, WN, VEC, ELU, KB, true>(p, tile, (tile / tiles_n) * (32 * TM * WM), (tile % tiles_n) * (32 * TN * WN), smem);
}

// ---- tile-streaming form of the 128 x 128 configuration -------------------------------------------------------------------
// The launches that carry the codec's FLOPs have 10^3 .. 10^5 tiles of 8 .. 192 k-tiles each; with one workgroup per tile the matrix
// pipe idles through every tile's first load round trip, its index arithmetic, its epilogue and the dispatch of its successor,
// and the workgroups of a CU (dispatched together, same K) go through those phases together.  Here a workgroup is resident
// (two / three per CU) and walks its share of the tiles as ONE stream of k-tiles: the loads of the next tile's first k-tile are
// issued under the MFMAs of this tile's last one and land in the free LDS buffer, so the only per-tile cost left is the
// epilogue itself.  Tiles that touch an utterance edge (padding / history / ragged rows) take the general routine.
template <bool ELU, int KB, int DBG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KB == 16 ? 3 : 1, KB == 16 ? 3 : 2))) void gemm_win_stream_kernel(const GemmWinParams p, const int tiles) {
    constexpr int TM = 2, TN = 2, WM = 2, WN = 2;
    constexpr int BM = 128, BN = 128;
    constexpr int LDS_LD = KB + 4;
    constexpr int RP = 1024 / KB;
    constexpr int RA = BM / RP, RB = BN / RP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid / (KB / 4);
    const int lk = (tid % (KB / 4)) * 4;
    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    const int M = p.B * p.T_out;
    const int TC = p.T_in * p.C;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nk = p.K / KB;                       // fast path only when K % KB == 0 (checked per tile)
    const bool k_ok = p.K % KB == 0;

    // this workgroup's tiles: XCD x owns a contiguous range, its resident workgroups stride through it together
    const int xcd = blockIdx.x & 7;
    const int first = gw_xcd_first(tiles, xcd);
    const int count = gw_xcd_first(tiles, xcd + 1) - first;
    const int stride = ((int)gridDim.x + 7 - xcd) >> 3;
    int l = blockIdx.x >> 3;
    if (l >= count) return;

    struct Ctx {                     // staging pointers of a tile (this thread's RA rows of A, RB rows of W)
        const float* ap[RA];
        unsigned bo[RB];             // float offset from p.w
    };
    // staging pointers of a tile + "every window inside its utterance, full tile" (all threads vote: one barrier)
    auto setup = [&](int m0, int n0, Ctx& c) -> bool {
        bool interior = k_ok && m0 + BM <= M && n0 + BN <= p.N;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int m = min(m0 + lrow + RP * j, M - 1);
            const int b = m / p.T_out;
            const int t = m - b * p.T_out;
            const int f0 = (t * p.S - p.P) * p.C;
            interior = interior && f0 >= 0 && f0 + p.K <= TC;
            c.ap[j] = p.x + (long)b * p.x_bstride + f0 + lk;
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) c.bo[j] = (unsigned)min(n0 + lrow + RP * j, p.N - 1) * (unsigned)p.K + lk;
        return __syncthreads_and(interior);
    };

    f32x4 ra[RA], rb[RB];
    auto issue = [&](const Ctx& c, int kt) {
#pragma unroll
        for (int j = 0; j < RA; ++j) ra[j] = *reinterpret_cast<const f32x4*>(c.ap[j] + kt * KB);
#pragma unroll
        for (int j = 0; j < RB; ++j) rb[j] = *reinterpret_cast<const f32x4*>(p.w + c.bo[j] + kt * KB);
    };
    auto store_tiles = [&](int buf) {
        float* a = As + buf * BM * LDS_LD;
        float* b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            f32x4 v = ra[j];
            if (ELU) {
                v[0] = rst_elu(v[0]); v[1] = rst_elu(v[1]); v[2] = rst_elu(v[2]); v[3] = rst_elu(v[3]);
            }
            *reinterpret_cast<f32x4*>(a + (lrow + RP * j) * LDS_LD + lk) = v;
        }
#pragma unroll
        for (int j = 0; j < RB; ++j)
            *reinterpret_cast<f32x4*>(b + (lrow + RP * j) * LDS_LD + lk) = rb[j];
    };

    f32x16 acc[TM][TN];
    auto clear = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    };
    auto mma_tile = [&](int buf, int ks0, int ks1) {
        const float* a = As + buf * BM * LDS_LD + (wm * TM * 32 + frag_row) * LDS_LD + frag_k;
        const float* b = Bs + buf * BN * LDS_LD + (wn * TN * 32 + frag_row) * LDS_LD + frag_k;
#pragma unroll
        for (int ks = ks0; ks < ks1; ++ks) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
    };
    // One k-tile of the stream.  Registers hold the k-tile that follows the one in LDS (requested a whole step ago, so a loaded-HBM
    // round trip of ~2 us is covered by a step's worth of the CU's MFMAs -- with the loads requested and consumed inside the same
    // step the matrix pipe measured 11 % idle on them): they go to the other LDS buffer first, the registers are refilled with the
    // k-tile after that (`c`, `kt`: possibly the next tile's), then the MFMAs of this k-tile run.
    auto step = [&](const Ctx& c, int kt, int buf) {
        if (DBG == 5) {      // MFMA only (ceiling measurement)
#pragma unroll
            for (int e = 0; e < 4 * (KB / 8); ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[0][e & 3], rb[0][e & 3], acc[i][j], 0, 0, 0);
            return;
        }
        store_tiles(buf ^ 1);
        if (DBG != 1) issue(c, kt);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(buf, 0, KB / 8);
        __syncthreads();
    };
    // start of a run of interior tiles: k-tile 0 to LDS, k-tile 1 to the registers
    auto prime = [&](const Ctx& c, int buf) {
        issue(c, 0);
        store_tiles(buf);
        issue(c, 1);
        __syncthreads();
    };

    Ctx c, cn;
    int tile = first + l;
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    bool fast = nk >= 2 && setup(m0, n0, c);
    int buf = 0;
    if (fast) prime(c, 0);
    for (;;) {
        const bool has_next = l + stride < count;
        const int tile_n = tile + stride;
        const int m0n = (tile_n / tiles_n) * BM, n0n = (tile_n % tiles_n) * BN;
        bool fast_n = false;
        if (fast) {
            clear();
            int lkt = 2;                 // k-tile the next request fetches
            for (int kt = 0; kt < nk; ++kt) {
                if (lkt == nk) {
                    // the request stream crosses into the next tile: its pointers take the place of this tile's (before an edge
                    // tile / at the end: harmless re-loads of this tile's first k-tiles)
                    lkt = 0;
                    if (has_next) {
                        fast_n = setup(m0n, n0n, cn);
                        if (fast_n) c = cn;
                    }
                }
                step(c, lkt, buf);
                ++lkt;
                buf ^= 1;
            }
            gw_epilogue<TM, TN, true>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, lane);
        } else {
            gw_tile<TM, TN, WM, WN, true, ELU, KB, false>(p, tile, m0, n0, smem);
            if (has_next) fast_n = nk >= 2 && setup(m0n, n0n, c);
        }
        if (!has_next) break;
        if (fast_n && !fast) prime(c, buf);     // after a general tile
        fast = fast_n;
        l += stride;
        tile = tile_n;
        m0 = m0n;
        n0 = n0n;
    }
}

// ---- three-plane bf16 form of the tile-streaming kernel ("b3") ---------------------------------------------------------------
// The f32 matrix instruction runs at the VECTOR rate (157 TFLOP/s, 64 cycles per 32x32x2); the bf16 one is sixteen times faster.
// An fp32 number is EXACTLY the sum of three bf16 numbers (hi = rne(x), mid = rne(x - hi), lo = rne(x - hi - mid): 8 + 8 + 8
// significand bits, the two subtractions are exact), so an fp32 product is the sum of nine bf16 products, each exact in the fp32
// accumulator.  Six of them are kept -- hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi -- and the three dropped (mid*lo, lo*mid, lo*lo)
// are bounded by 2^-23 |x||w| (2^-8 * 2^-16 twice), the size of ONE fp32 rounding of the product: the result carries fp32 accuracy
// (tests: <= 2e-6 of the fp64 value's scale, the same bound the f32-instruction path meets; RVQ codes as exact as with it) at
// 6 x 32 = 192 matrix-pipe cycles per 32x32x16 block instead of 8 x 64 = 512.
// Weights are split once on the host side of the ABI (rst_gemm_win_b3_pack_weight) into matrix-instruction operand order
// [block of 32 rows][K / 16][plane][64 lanes][8 k] (round 5, DIRW below: every wave requests the fragments of its own columns from
// L2 straight into registers; the tools build keeps the round-3 form, [n tile of 128][K / 16][plane][128 row slots][16 k] staged through
// LDS, for the A/B of DESIGN.md 3.1b); activations stay fp32 in HBM
// and are split on their way into LDS (v_cvt_pk_bf16_f32 + subtractions, ~4.5 VALU per element).  LDS rows are 16 bf16 + 8 pad
// = 48 bytes: the 16 lanes of a ds_read_b128 group hold rows that are distinct mod 16, and 3 * row mod 16 is a bijection, so every
// group covers 16 distinct 16-byte slots.  Two buffers of 128 activation rows x 3 planes: 36 KB (staged form: (128 + BN) rows, 72 /
// 108 KB).  Rows whose window leaves the utterance are masked inside the kernel (below); only
// launches with a history buffer / replicate padding / K % 64 != 0 run on the f32-instruction kernels above.

constexpr int B3_SETS_A = 4;                  // k-tiles of activations / of weights held in registers ahead of the one being multiplied
constexpr int B3_SETS_B = 2;
constexpr int B3_KB = 16;                     // k per stage (one bf16 matrix instruction deep)
constexpr int B3_RS = 24;                     // shorts per LDS row (16 + 8 pad)
constexpr int B3_WTILE = 3 * 128 * B3_KB;     // shorts per packed (n tile, k tile) piece of the weights

template <int V>
struct b3_int { static constexpr int value = V; };

// LDS rows are 48 bytes apart, so the rows whose 32-byte (activations: 4 lanes x 8 bytes; weights: 2 lanes x 16) pieces tile a
// 128-byte bank window are r, r + 2, r + 4, r + 6: a write group (16 / 8 consecutive lanes = four consecutive row slots) is given
// exactly those.  Slot g of a pass therefore stages row b3_row(g) (and the packed weights hold row b3_row(g) at slot g): with the
// identity map every write group ran 2-way conflicted, a third of the kernel's LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
__host__ __device__ __forceinline__ constexpr int b3_row(int g) { return (g & ~7) | (2 * (g & 3) + ((g >> 2) & 1)); }
__host__ __device__ __forceinline__ constexpr int b3_slot(int r) { return (r & ~7) | (((r & 7) >> 1) + 4 * (r & 1)); }

// The stream of k-tiles (tile after tile, as in gemm_win_stream_kernel) is a software pipeline in registers.  A stage (768 matrix-pipe
// cycles, ~0.35 us) is shorter than a loaded round trip to L2 / HBM, so the loads run several stages ahead: the fp32 activation rows
// of k-tile g + 4 and the weight planes (L2-resident) of k-tile g + 2 are requested at the top of stage g, four / two register sets
// rotate, and the rotation is written out (b3_int<0..3>; K is a multiple of 64, so every tile starts on set 0).  While k-tile g is
// multiplied out of one LDS buffer, k-tile g + 1 is split and written to the other buffer BETWEEN this stage's matrix instructions -- one 4-instruction peel
// or one LDS write behind every second one, pinned there with scheduling fences: the wave issues in order, and whatever sits in
// front of the first matrix instruction is time the pipe idles.
// MASK: the launch has rows whose window reaches into the zero padding in front of / behind an utterance, or ragged last tiles.
// Since C % 16 == 0, a window leaves its utterance on a k-tile boundary: every row carries the range [klo, khi) of k-tiles it really
// reads, a k-tile outside it is loaded from a valid address of the same row and cleared when it is consumed (a bit per row travels
// with the register set).  Launches with a history buffer or replicate padding stay on the f32-instruction kernels.
// NWN: waves along N.  2: 128 x 128 tile, 256 threads, two workgroups per CU.  4: 128 x 256 tile, 512 threads, one workgroup per CU --
// the same eight waves per CU, but every activation element is split for 256 columns instead of 128, which halves the split work
// (VALU + LDS writes) per matrix instruction: measured (tools build, RST_B3_DBG) that work is what the 128-wide form spends most on.
// DBG (tools build only, WRONG results): 1 = no split / LDS writes, 2 = no global loads, 3 = matrix instructions only, 4 = no barriers,
// 5 = LDS writes of unsplit bits (what activations handed over already split would cost)
// DIRW: the weight fragments never pass through LDS.  The packed weights are then in matrix-instruction operand order ([32-column
// block][k-tile][plane][64 lanes][8 bf16]: a wave-level load is one contiguous KB, as in lm_skinny.hip) and every wave requests the
// fragments of its own 64 columns straight into the registers the matrix instructions read -- two k-tiles ahead, each plane into
// the registers its last product of the current k-tile has just released.  LDS then holds the split activations only: a third of
// the write instructions and half of the fragment reads of a stage are gone (what the ablation table prices highest), at the cost
// of the weights crossing the L1 once per wave row (twice per workgroup) instead of once.
template <bool ELU, bool MASK, int NWN, int DBG = 0, bool BUFL = true, bool DIRW = false>
__global__ __launch_bounds__(128 * NWN) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_win_b3_stream_kernel(const GemmWinParams p, const int tiles) {
    constexpr int TM = 2, TN = 2;
    constexpr int NT = 128 * NWN;                  // threads
    constexpr int BM = 128, BN = 64 * NWN;
    constexpr int RA = 512 / NT;                   // activation row passes: NT / 4 row slots x 4 threads (16 bytes of fp32 each) per pass
    constexpr int A_PLANE = BM * B3_RS, W_PLANE = DIRW ? 0 : BN * B3_RS;          // shorts per plane in LDS
    constexpr int BUF = 3 * (A_PLANE + W_PLANE);                       // shorts per buffer: activation planes 0..2, weight planes 0..2
    constexpr int NSA = B3_SETS_A, NSB = B3_SETS_B;
    static_assert(NSA == 4 && NSB == 2, "the rotation below is written out for four activation sets and two weight sets");
    static_assert(NWN == 2 || NWN == 4, "two or four waves along N");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    short* const lds = reinterpret_cast<short*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int lrow = b3_row(tid >> 2);             // row (of the first pass) this thread stages
    const int lk = (tid & 3) * 4;
    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;

    const int M = p.B * p.T_out;
    const int TC = p.T_in * p.C;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nk = p.K / B3_KB;                    // a multiple of 4 (K % 64 == 0): checked by the launcher

    const int xcd = blockIdx.x & 7;
    const int first = gw_xcd_first(tiles, xcd);
    const int count = gw_xcd_first(tiles, xcd + 1) - first;
    const int stride = ((int)gridDim.x + 7 - xcd) >> 3;
    int l = blockIdx.x >> 3;
    if (l >= count) return;

    struct Ctx {
        unsigned ao[RA];             // byte offset of the row's window (+ this thread's 16 bytes) from p.x: the loads take the uniform
                                     // base + k-tile in scalar registers and this as the 32-bit vector offset
        int klo[RA], kn[RA];         // MASK: k-tiles klo .. klo + kn - 1 of the row's window lie inside its utterance
    };
    auto setup = [&](int m0, Ctx& c) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int m = m0 + lrow + (NT / 4) * j;
            const int mm = min(m, M - 1);
            const int b = mm / p.T_out;
            const int t = mm - b * p.T_out;
            const int f0 = (t * p.S - p.P) * p.C;
            long off = (long)b * p.x_bstride + f0 + lk;
            if (MASK) {
                int klo = max(0, -f0) / B3_KB, khi = min(p.K, TC - f0) / B3_KB;
                if (m >= M || klo >= khi) {          // no such row / a window entirely in the padding: all zeros, loads parked on x
                    klo = khi = 0;
                    off = lk;
                }
                c.klo[j] = klo;
                c.kn[j] = khi - klo;
                // (rows that start in the padding: the offset is taken at k-tile klo, which is where their loads start)
                off += (long)klo * B3_KB;
            }
            c.ao[j] = (unsigned)(off * 4);
        }
    };
    // this thread's 16 bytes of plane 0 of the tile's first k-tile: the packed weights come in 128-row tiles of [k-tile][plane][128][16]
    // DIRW: [32-column block][k-tile][plane][64 lanes][8]: 3 KB per (block, k-tile); the wave's first block is part of the lane offset
    auto w_tile = [&](int n0) {
        return DIRW ? (unsigned)(n0 / 32) * (unsigned)nk * 3072u : (unsigned)(n0 / 128) * (unsigned)nk * (unsigned)(B3_WTILE * 2);
    };      // (bytes, uniform; the thread's part is wo)
    const unsigned wo = DIRW ? (unsigned)(wn * TN) * (unsigned)nk * 3072u + (unsigned)lane * 16u
                             : (unsigned)(((long)(tid >> 8) * nk * B3_WTILE + (tid & 255) * 8) * 2);      // bytes
    const unsigned wo_blk = (unsigned)nk * 3072u;       // DIRW: from one 32-column block to the next

    f32x4 ra[NSA][RA];
    int rm[NSA][RA];                 // MASK: all ones where row j of the set is real data, zero where it is padding
    u32x4 rb[NSB][3];                // !DIRW: the thread's staging pieces of a k-tile's three weight planes
    u32x4 rw[NSB][TN][3];            // DIRW: the wave's weight fragments (32-column block j, plane q) of k-tiles g, g + 1 (sets g % 2)
    // LDS destinations of this thread's pieces (shorts from the start of a buffer)
    const int a_dst = lrow * B3_RS + lk;
    const int b_dst = 3 * A_PLANE + ((tid >> 8) * 128 + b3_row((tid & 255) >> 1)) * B3_RS + (tid & 1) * 8;
    const int a_frag = (wm * TM * 32 + frag_row) * B3_RS + frag_k;
    const int b_frag = 3 * A_PLANE + (wn * TN * 32 + frag_row) * B3_RS + frag_k;

    // Requests go out as buffer loads: resource (base pointer) and the k-tile's byte offset in scalar registers, the thread's own
    // offset a 32-bit vector register set up once per tile -- no address arithmetic on the vector unit in the K loop (the launcher
    // checks that the activations span less than 4 GB).  MM: with the masks; without, every row's whole window is real data.
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.w3), 0, 0xffffffff, 0x00020000);
    auto load_a = [&](auto MM, const Ctx& c, const int kt, f32x4 (&dst)[RA], int (&mask)[RA]) {
        constexpr bool MK = decltype(MM)::value != 0;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            if (MK) {
                const unsigned d = (unsigned)(kt - c.klo[j]);          // (ao is taken at k-tile klo)
                const bool ok = d < (unsigned)c.kn[j];
                mask[j] = ok ? -1 : 0;
                const unsigned vo = c.ao[j] + (ok ? d : 0u) * (B3_KB * 4);
                if (BUFL) dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, 0, 0));
                else dst[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.x) + vo);
            } else {
                if (BUFL) dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, c.ao[j], kt * (B3_KB * 4), 0));
                else dst[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.x) + (long)kt * (B3_KB * 4) + c.ao[j]);
            }
        }
    };
    // wb: byte offset of the tile's first k-tile inside the packed weights (uniform)
    auto load_w = [&](const unsigned wb, const int kt, u32x4 (&dst)[3]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const unsigned so = wb + (unsigned)kt * (B3_WTILE * 2) + q * (128 * B3_KB * 2);
            if (BUFL) dst[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wo, so, 0);
            else dst[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w3) + so + wo);
        }
    };
    // DIRW: plane q of the wave's TN blocks of k-tile kt (stream position relative to wb) into set U
    auto load_wq = [&](auto UU, auto QQ, const unsigned wb, const int kt) {
        constexpr int U = decltype(UU)::value, q = decltype(QQ)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned so = wb + (unsigned)kt * 3072u + q * 1024u;
            if (BUFL) rw[U][j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wo + j * wo_blk, so, 0);
            else rw[U][j][q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w3) + so + wo + j * wo_blk);
        }
    };
    // the fp32 values of a staged row piece as two pairs: padding cleared (MASK), ELU applied
    auto take_a = [&](auto MM, const f32x4 src, const int mask, f32x2 (&v)[2]) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        f32x4 x = src;
        if (decltype(MM)::value != 0) x = __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, x) & mask);
        if (ELU) {
            x[0] = rst_elu(x[0]); x[1] = rst_elu(x[1]); x[2] = rst_elu(x[2]); x[3] = rst_elu(x[3]);
        }
        v[0] = f32x2{x[0], x[1]};
        v[1] = f32x2{x[2], x[3]};
    };

    f32x16 acc[TM][TN];
    auto clear = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    };

    // one stage: the k-tile of LDS buffer `buf` into the accumulators; register sets S -- the k-tile after it -- into buffer buf ^ 1;
    // the requests of the stream positions ahead (activations: context c, k-tile kta; weights: bp, ktb) into the sets just freed
    auto stage = [&](auto SS, auto MM, const Ctx& c, const int kta, const unsigned bp, const int ktb, const int buf) {
        constexpr int S = decltype(SS)::value;
        constexpr int F = (S + 3) % 4, SB = S % 2, FB = (S + 1) % 2;
        if (DBG != 2 && DBG != 3) {
            // weights first: the wait for them counts requests in order, and must leave the younger activation requests in flight
            if (!DIRW) {
                load_w(bp, ktb, rb[FB]);
                __builtin_amdgcn_sched_barrier(0);
            }
            load_a(MM, c, kta, ra[F], rm[F]);
        }
        bf16x8 fa[TM][3], fb[TN][3];
        const short* rd = lds + buf * BUF;
        short* wr = lds + (buf ^ 1) * BUF;
        // fragments in the order the products below consume them
        // !DIRW: smallest terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi.  DIRW: the three products of weight plane 0 first,
        // then plane 2, then plane 1 -- a plane's registers are re-requested (k-tile g + 2) as soon as its last product has issued, which
        // gives every request >= 1.5 stages; the order inside a k-tile moves the result by roundings of the running sum only
        constexpr int QA[6] = {2, DIRW ? 1 : 0, DIRW ? 0 : 1, DIRW ? 0 : 1, DIRW ? 1 : 0, 0};
        constexpr int QB[6] = {0, DIRW ? 0 : 2, DIRW ? 0 : 1, DIRW ? 2 : 0, 1, DIRW ? 1 : 0};
        if (DIRW && DBG != 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][2] = *reinterpret_cast<const bf16x8*>(rd + a_frag + 2 * A_PLANE + i * 32 * B3_RS);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][1] = *reinterpret_cast<const bf16x8*>(rd + a_frag + A_PLANE + i * 32 * B3_RS);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][0] = *reinterpret_cast<const bf16x8*>(rd + a_frag + i * 32 * B3_RS);
        } else if (DBG != 3) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j][0] = *reinterpret_cast<const bf16x8*>(rd + b_frag + j * 32 * B3_RS);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][2] = *reinterpret_cast<const bf16x8*>(rd + a_frag + 2 * A_PLANE + i * 32 * B3_RS);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][0] = *reinterpret_cast<const bf16x8*>(rd + a_frag + i * 32 * B3_RS);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j][2] = *reinterpret_cast<const bf16x8*>(rd + b_frag + 2 * W_PLANE + j * 32 * B3_RS);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][1] = *reinterpret_cast<const bf16x8*>(rd + a_frag + A_PLANE + i * 32 * B3_RS);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j][1] = *reinterpret_cast<const bf16x8*>(rd + b_frag + W_PLANE + j * 32 * B3_RS);
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i][q] = __builtin_bit_cast(bf16x8, ra[S][0]);
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) fb[jj][q] = __builtin_bit_cast(bf16x8, rb[SB][q]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        f32x2 v[RA][2];
        u32x2 h[RA];
        constexpr bool SIDE = DBG != 1 && DBG != 3;
        // side work, one small piece per slot: 0..2 the weight planes' LDS writes; then per plane q: 2 * RA half-row peels (4 VALU
        // each) and the plane's LDS write
        constexpr int PB = 2 * RA + 1, WOPS = DIRW ? 0 : 3, NOPS = WOPS + 3 * PB;       // 18 (RA = 2: slots 1 .. 18) or 12 (RA = 1: the odd slots); DIRW: 15 / 9
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            const int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
            if (DIRW && DBG != 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][QA[t]], __builtin_bit_cast(bf16x8, rw[SB][j][QB[t]]), acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][QA[t]], fb[j][QB[t]], acc[i][j], 0, 0, 0);
            if (DIRW && DBG != 2 && DBG != 3) {
                // a weight plane's last product of this k-tile has issued: its registers take the plane of k-tile g + 2 (same set)
                if (m == 11) load_wq(b3_int<SB>{}, b3_int<0>{}, bp, ktb);
                if (m == 15) load_wq(b3_int<SB>{}, b3_int<2>{}, bp, ktb);
                if (m == 23) load_wq(b3_int<SB>{}, b3_int<1>{}, bp, ktb);
            }
            if (SIDE && m == 0) {
#pragma unroll
                for (int r = 0; r < RA; ++r) take_a(MM, ra[S][r], rm[S][r], v[r]);
            }
            const int op = RA == 2 ? m - 1 : ((m & 1) ? (m - 1) >> 1 : -1);
            if (SIDE && op >= 0 && op < NOPS) {
                if (op < WOPS) {
                    *reinterpret_cast<u32x4*>(wr + b_dst + op * W_PLANE) = rb[SB][op];
                } else {
                    const int u = op - WOPS;
                    const int q = u / PB, w = u % PB;
                    if (w < 2 * RA) {
                        if (DBG == 5) h[w >> 1][w & 1] = __builtin_bit_cast(unsigned, v[w >> 1][w & 1][0]) + q;
                        else h[w >> 1][w & 1] = b3_peel(v[w >> 1][w & 1]);
                    } else {
#pragma unroll
                        for (int r = 0; r < RA; ++r) *reinterpret_cast<u32x2*>(wr + a_dst + q * A_PLANE + r * (NT / 4) * B3_RS) = h[r];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DBG != 4) __syncthreads();
    };

    // start of the run: k-tile 0 of the first tile to LDS, activations of k-tiles 1 .. 3 to sets 0 .. 2, weights of k-tile 1 to set 0
    auto prime = [&](const Ctx& c, const unsigned bp, int buf) {
        short* wr = lds + buf * BUF;
        {
            f32x4 xa[RA];
            int xm[RA];
            u32x4 xb[3];
            load_a(b3_int<MASK>{}, c, 0, xa, xm);
            if (!DIRW) load_w(bp, 0, xb);
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                f32x2 v[2];
                take_a(b3_int<MASK>{}, xa[j], xm[j], v);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x2 hh;
                    hh[0] = b3_peel(v[0]);
                    hh[1] = b3_peel(v[1]);
                    *reinterpret_cast<u32x2*>(wr + a_dst + q * A_PLANE + j * (NT / 4) * B3_RS) = hh;
                }
            }
            if (!DIRW) {
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(wr + b_dst + q * W_PLANE) = xb[q];
            }
        }
        if (DIRW) {         // k-tiles 0 and 1 of the wave's fragments into sets 0 and 1
            load_wq(b3_int<0>{}, b3_int<0>{}, bp, 0); load_wq(b3_int<0>{}, b3_int<2>{}, bp, 0); load_wq(b3_int<0>{}, b3_int<1>{}, bp, 0);
            load_wq(b3_int<1>{}, b3_int<0>{}, bp, 1); load_wq(b3_int<1>{}, b3_int<2>{}, bp, 1); load_wq(b3_int<1>{}, b3_int<1>{}, bp, 1);
        } else {
            load_w(bp, 1, rb[0]);
        }
#pragma unroll
        for (int g = 1; g < 4; ++g) load_a(b3_int<MASK>{}, c, g, ra[g - 1], rm[g - 1]);
        __syncthreads();
    };

    Ctx c;
    int tile = first + l;
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    setup(m0, c);
    unsigned bp = w_tile(n0);
    unsigned bpn = bp;               // weights of the tile the activation cursor has already moved on to
    int kb = 2;
    prime(c, bp, 0);
    for (;;) {
        const int ln = l + stride;
        const bool has_next = ln < count;
        const int tile_n = first + ln;
        const int m0n = (tile_n / tiles_n) * BM, n0n = (tile_n % tiles_n) * BN;
        clear();
        // request cursors: activations four k-tiles ahead, weights two; each crosses into the next tile on its own stage (at the end of
        // the run: harmless re-loads of this tile's first k-tiles)
        int kf = 4;
        auto cross = [&]() {
            kf = 0;
            if (has_next) {
                setup(m0n, c);
                bpn = w_tile(n0n);
            }
        };
        constexpr b3_int<MASK> MM{};
        for (int kt = 0; kt < nk; kt += 4) {
            if (kf == nk) cross();
            stage(b3_int<0>{}, MM, c, kf, bp, kb, 0);
            stage(b3_int<1>{}, MM, c, kf + 1, bp, kb + 1, 1);
            kb += 2;
            if (kb == nk) {
                kb = 0;
                bp = bpn;
            }
            stage(b3_int<2>{}, MM, c, kf + 2, bp, kb, 0);
            stage(b3_int<3>{}, MM, c, kf + 3, bp, kb + 1, 1);
            kb += 2;
            if (kb == nk) {
                kb = 0;
                bp = bpn;
            }
            kf += 4;
        }
        if (!MASK || (m0 + BM <= M && n0 + BN <= p.N)) gw_epilogue<TM, TN, true>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, lane);
        else gw_epilogue<TM, TN, false>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, lane);
        if (!has_next) break;
        l = ln;
        tile = tile_n;
        m0 = m0n;
        n0 = n0n;
    }
}

// w fp32 [N][K] -> the three bf16 planes in staging order (rows past N: zeros); one thread per four k of one row
// DIRW: operand order [32-row block][k-tile][plane][64 lanes][8] instead -- lane (n % 32) + 32 * (k % 16 / 8) of the block holds k % 8 ..
template <bool DIRW>
__global__ __launch_bounds__(256) void gemm_win_b3_pack_kernel(const float* __restrict__ w, short* __restrict__ w3, int N, int K, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int kq = K / 4;
    const int n = (int)(i / kq);
    const int k = (int)(i % kq) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < N) v = *reinterpret_cast<const f32x4*>(w + (long)n * K + k);
    f32x2 v0 = {v[0], v[1]}, v1 = {v[2], v[3]};
    short* dst = DIRW ? w3 + ((long)(n / 32) * (K / B3_KB) + k / B3_KB) * (3 * 512) + ((n % 32) + 32 * ((k % B3_KB) / 8)) * 8 + k % 8
                      : w3 + ((long)(n / 128) * (K / B3_KB) + k / B3_KB) * B3_WTILE + b3_slot(n % 128) * B3_KB + k % B3_KB;     // slot g holds row b3_row(g)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x2 h;
        h[0] = b3_peel(v0);
        h[1] = b3_peel(v1);
        *reinterpret_cast<u32x2*>(dst + q * (DIRW ? 512 : 128 * B3_KB)) = h;
    }
}

// Which of the two weight layouts / kernel forms the library runs (one answer per process: the pack routine and the GEMM must agree).
// The tools build reads RST_B3_DIRW for A/B measurements.
// The shipped library runs the direct form only; the staged form (weights through LDS) is compiled into the tools build for the A/B of
// DESIGN.md 3.1b (RST_B3_DIRW=0).
static bool b3_dirw() {
    static const int v = rst_knob("RST_B3_DIRW", 1);
    return v != 0;
}
// (Round 5 also built the same stream with NO barrier per stage: the split activation planes in a ring of four LDS slots, every
// condition a barrier enforces checked against per-slot arrival counters in LDS with a full stage of slack.  Parity-green and 5 - 15 %
// SLOWER on every layer (181 vs 196 TFLOP/s, profiles/r05_b3_ring_vs_barrier.txt): waves that drift apart no longer share the weight
// fragments in the L1, and the counter traffic costs more than the barrier it replaces.  The kernel is kept as
// tools/probes/b3_ring_kernel.hip, not in the library.)

template <int TM, int TN, int WM, int WN, int KB = BK>
int launch_cfg(const GemmWinParams& p, bool vec, hipStream_t stream) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const long M = (long)p.B * p.T_out;
    const long tiles = ((M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (tiles <= 0) return RST_OK;
    if (tiles > 0x7fffffffL) {
        rst_set_error("gemm_win: too many tiles (%ld)", tiles);
        return RST_ERR_UNSUPPORTED;
    }
    const size_t lds = 2 * (BM + BN) * (KB + 4) * sizeof(float);
    auto go = [&](auto kern) {
        static RstOncePerDevice attr_once;  // > 64 KiB of dynamic LDS needs the opt-in once per kernel instantiation
        if (attr_once.first()) {
            // the kernel also has a few bytes of static LDS: stay below the 160 KiB total (the largest tile needs 83 KiB)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)(p.split_k > 1 ? p.split_k : 1)), dim3(256), lds, stream, p);
    };
    const bool elu = p.act_in == 1;
    if (vec && elu) go(gemm_win_kernel<TM, TN, WM, WN, true, true, KB>);
    else if (vec) go(gemm_win_kernel<TM, TN, WM, WN, true, false, KB>);
    else if (elu) go(gemm_win_kernel<TM, TN, WM, WN, false, true, KB>);
    else go(gemm_win_kernel<TM, TN, WM, WN, false, false, KB>);
    return rst_check_launch("gemm_win");
}

static int gw_cu_count() {
    return rst_cu_count();       // per device (rst_common.h)
}

// the 128 x 128 configuration on 16-byte-aligned operands: resident workgroups streaming through the tiles
template <int KB>
int launch_stream(const GemmWinParams& p, long tiles, hipStream_t stream) {
    if (tiles > 0x7fffffffL) {
        rst_set_error("gemm_win: too many tiles (%ld)", tiles);
        return RST_ERR_UNSUPPORTED;
    }
    const size_t lds = 2 * (128 + 128) * (KB + 4) * sizeof(float);
    static const int per_cu_env = rst_knob("RST_GEMM_STREAM_WGS", 0);      // tools build only
    const int per_cu = per_cu_env > 0 ? per_cu_env : (KB == 16 ? 3 : 2);
    const long resident = (long)per_cu * gw_cu_count();
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    auto go = [&](auto kern) {
        static RstOncePerDevice attr_once;
        if (attr_once.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p, (int)tiles);
    };
#ifdef RST_ABLATION
    // ceiling measurements: instances that skip the loads / run the MFMAs only -- WRONG RESULTS by construction, which
    // is why they are compiled into the tools build alone
    static const int dbg = rst_knob("RST_GEMM_DBG", 0);
    if (KB == 16 && dbg == 1) { go(gemm_win_stream_kernel<false, 16, 1>); return rst_check_launch("gemm_win"); }
    if (KB == 16 && dbg == 5) { go(gemm_win_stream_kernel<false, 16, 5>); return rst_check_launch("gemm_win"); }
#endif
    if (p.act_in == 1) go(gemm_win_stream_kernel<true, KB>);
    else go(gemm_win_stream_kernel<false, KB>);
    return rst_check_launch("gemm_win");
}

template <int NWN>
int launch_stream_b3_cfg(const GemmWinParams& p, hipStream_t stream) {
    constexpr int BN = 64 * NWN;
    const long M = (long)p.B * p.T_out;
    const long tiles = ((M + 127) / 128) * ((p.N + BN - 1) / BN);
    if (tiles > 0x7fffffffL) {
        rst_set_error("gemm_win: too many tiles (%ld)", tiles);
        return RST_ERR_UNSUPPORTED;
    }
    const bool dirw = b3_dirw();
    const size_t lds = 2 * 3 * (128 + (dirw ? 0 : BN)) * B3_RS * sizeof(short);       // 73 728 / 110 592 bytes; 36 864 without the weight planes
    static const int per_cu = rst_knob("RST_B3_WGS", NWN == 2 ? 2 : 1);      // tools build only
    const long resident = (long)per_cu * gw_cu_count();
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    auto go = [&](auto kern) {
        static RstOncePerDevice attr_once;
        if (attr_once.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * NWN), lds, stream, p, (int)tiles);
    };
    // rows whose window reaches into the zero padding (or past the end), ragged last tiles: the masked form for the whole launch.
    // (Measured: walking the interior tiles with the lean kernel and the others with the masked one as two launches gains 2 - 9 % on
    // the 24 kHz / 4.8 kHz layers, whose edge tiles are few, and loses 20 - 40 % on the layers below, where the two half-filled
    // launches run one after the other.)
    bool lean = p.P == 0 && M % 128 == 0 && p.N % BN == 0 && (long)(p.T_out - 1) * p.S * p.C + p.K <= (long)p.T_in * p.C;
#ifdef RST_ABLATION
    // RST_B3_FORCE_MASK=1 (tools build): the masked instance (9 - 33 spilled registers, scratch) on the edge-free launches too -- what do
    // its spills cost on a shape that never takes an edge path?  (VERDICT r5 #6d; profiles/r06_b3_masked_vs_lean.txt)
    static const int force_mask = rst_knob("RST_B3_FORCE_MASK", 0);
    if (force_mask) lean = false;
    // (the ablation instances of the SHIPPED form: weight fragments direct; RST_B3_DIRW=0 has no ablation instances)
    static const int dbg = rst_knob("RST_B3_DBG", 0);
    if (dirw && dbg == 1) { go(gemm_win_b3_stream_kernel<false, true, NWN, 1, true, true>); return rst_check_launch("gemm_win_b3"); }
    if (dirw && dbg == 2) { go(gemm_win_b3_stream_kernel<false, true, NWN, 2, true, true>); return rst_check_launch("gemm_win_b3"); }
    if (dirw && dbg == 3) { go(gemm_win_b3_stream_kernel<false, true, NWN, 3, true, true>); return rst_check_launch("gemm_win_b3"); }
    if (dirw && dbg == 4) { go(gemm_win_b3_stream_kernel<false, true, NWN, 4, true, true>); return rst_check_launch("gemm_win_b3"); }
    if (dirw && dbg == 5) { go(gemm_win_b3_stream_kernel<false, true, NWN, 5, true, true>); return rst_check_launch("gemm_win_b3"); }
    static const int bufl = rst_knob("RST_B3_BUF", 1);        // 0: plain global loads (64-bit addresses on the vector unit)
    if (!bufl && dirw && p.act_in != 1) {
        if (lean) go(gemm_win_b3_stream_kernel<false, false, NWN, 0, false, true>);
        else go(gemm_win_b3_stream_kernel<false, true, NWN, 0, false, true>);
        return rst_check_launch("gemm_win_b3");
    }
#endif
#ifdef RST_ABLATION
    if (!dirw) {
        if (p.act_in == 1) {
            if (lean) go(gemm_win_b3_stream_kernel<true, false, NWN>);
            else go(gemm_win_b3_stream_kernel<true, true, NWN>);
        } else {
            if (lean) go(gemm_win_b3_stream_kernel<false, false, NWN>);
            else go(gemm_win_b3_stream_kernel<false, true, NWN>);
        }
        return rst_check_launch("gemm_win_b3");
    }
#endif
    if (p.act_in == 1) {
        if (lean) go(gemm_win_b3_stream_kernel<true, false, NWN, 0, true, true>);
        else go(gemm_win_b3_stream_kernel<true, true, NWN, 0, true, true>);
    } else {
        if (lean) go(gemm_win_b3_stream_kernel<false, false, NWN, 0, true, true>);
        else go(gemm_win_b3_stream_kernel<false, true, NWN, 0, true, true>);
    }
    return rst_check_launch("gemm_win_b3");
}

int launch_stream_b3(const GemmWinParams& p, hipStream_t stream) {
    // 256 columns per tile when the output is at least that wide, a ragged last tile wastes at most an eighth of the columns and the
    // tiles still outnumber the CUs two to one (measured per layer: N = 640 and 8000-row launches lose with the wide tile)
    static const int wide_knob = rst_knob("RST_B3_WIDE", 1);      // tools build only
    const int waste = ((p.N + 255) / 256) * 256 - p.N;
    const long wide_tiles = (((long)p.B * p.T_out + 127) / 128) * ((p.N + 255) / 256);
    if (wide_knob && p.N >= 256 && waste * 8 <= p.N && wide_tiles >= 2L * gw_cu_count()) return launch_stream_b3_cfg<4>(p, stream);
    return launch_stream_b3_cfg<2>(p, stream);
}

}  // namespace

// Tile shape the launcher picks for (M, N): 0 = 128 x 128, 1 = 32 x 128, 2 = 128 x 64, 3 = 256 x 32.
// Medium M (one 80 ms frame of a single stream at the 24 kHz .. 1.2 kHz layers: 96 .. 1920 rows) takes the 32-row tiles for wide
// outputs: four times the workgroups of the 128-row tile, and with split-K enough of them to hide each other's load latency.
static int gw_tile_cfg(long M, int N) {
    if (N > 64) return M > 4096 ? 0 : 1;
    if (N > 32) return 2;
    return 3;
}
static void gw_tile_dims(int cfg, int& BM, int& BN) {
    BM = cfg == 0 ? 128 : cfg == 1 ? 32 : cfg == 2 ? 128 : 256;
    BN = cfg == 0 ? 128 : cfg == 1 ? 128 : cfg == 2 ? 64 : 32;
}

int rst_gemm_split_plan_impl(long M, int N, int K) {
    // Few- and medium-row calls (the streaming steps) leave most of the chip idle and walk K as a chain of exposed load latencies
    // (one k-tile per memory round trip): spread K over workgroups until ~256 of them run, keeping >= 2 k-tiles per split.
    if (M <= 0 || M > 4096 || (M <= 32 && N <= 64)) return 1;
    int BM, BN;
    gw_tile_dims(gw_tile_cfg(M, N), BM, BN);
    const long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int nk = (K + BK - 1) / BK;
    int s = 1;
    while (tiles * s * 2 <= 256 && nk / (2 * s) >= 2 && s < 64) s *= 2;
    return s;
}

// Shapes the three-plane bf16 kernel serves (pointer alignment apart): the 128 x 128 tile class (more than 4096 rows, N > 64), whole
// rotations of the four register sets (K % 64 == 0), windows that leave their utterance on a k-tile boundary (C % 16 == 0), zero padding,
// no history buffer, 16-byte-aligned rows, and operands its 32-bit buffer offsets can reach.
bool rst_gemm_win_b3_shape_ok(int B, int T_in, int T_out, int C, int K, int N, int pad_mode, long x_bstride, bool has_hist) {
    if (B <= 0 || T_in <= 0 || T_out <= 0 || C <= 0 || K <= 0 || N <= 0) return false;
    const long M = (long)B * T_out;
    if (gw_tile_cfg(M, N) != 0) return false;
    if (K % 64 != 0 || C % 16 != 0 || x_bstride % 4 != 0 || has_hist || pad_mode != 0) return false;
    const long x_bytes = ((long)(B - 1) * x_bstride + (long)T_in * C) * 4;
    const long w_bytes = rst_gemm_win_b3_weight_elems_impl(N, K) * 2;
    return x_bytes < 0xfffff000L && w_bytes < 0xfffff000L;
}

int rst_launch_gemm_win(const GemmWinParams& p, hipStream_t stream) {
    RST_REQUIRE(p.B >= 0 && p.T_in >= 0 && p.T_out >= 0 && p.C > 0 && p.K > 0 && p.N > 0 && p.S > 0 && p.P >= 0,
                "gemm_win: bad sizes B=%d T_in=%d T_out=%d C=%d K=%d N=%d S=%d P=%d", p.B, p.T_in, p.T_out, p.C,
                p.K, p.N, p.S, p.P);
    if (p.B == 0 || p.T_out == 0) return RST_OK;
    RST_REQUIRE(p.w && p.y && (p.x || p.T_in == 0), "gemm_win: null pointer");
    RST_REQUIRE((long)p.T_in * p.C < 0x7fffffffL && (long)p.T_out * p.S * p.C < 0x7fffffffL,
                "gemm_win: per-batch activation too large for 32-bit indexing");
    const bool vec = (p.C % 4 == 0) && (p.K % 4 == 0) && (p.x_bstride % 4 == 0) &&
                     ((uintptr_t)p.x % 16 == 0) && ((uintptr_t)p.w % 16 == 0) &&
                     (!p.hist || (uintptr_t)p.hist % 16 == 0);
    const long M = (long)p.B * p.T_out;
    RST_REQUIRE(p.split_k <= 1 || (p.ws && p.counters && M <= 4096), "gemm_win: split-K needs M <= 4096 and the scratch buffers");
    RST_REQUIRE(!p.w3 || gw_tile_cfg(M, p.N) == 0, "gemm_win_b3: %ld x %d is not a large-launch shape (more than 4096 rows, N > 64)", M, p.N);
    switch (gw_tile_cfg(M, p.N)) {
        case 0: {                                                            // 128 x 128
            // >= 3 tiles per CU: k-chunks of 16 (40 KB of LDS, accumulators in VGPRs) put three workgroups on a CU instead of two
            const long tiles = ((M + 127) / 128) * ((p.N + 127) / 128);
            // tools build only -- RST_GEMM_KB32=1 forces the 32-wide chunks (the calibration knob of the PMC traffic numbers, DESIGN.md
            // 3.1), RST_GEMM_STREAM=0 one workgroup per tile (the pre-streaming form, for A/B measurements)
            static const bool kb32_only = rst_knob("RST_GEMM_KB32", 0) != 0;
            static const bool stream_off = rst_knob("RST_GEMM_STREAM", 1) == 0;
            // the three-plane bf16 form when the caller passed the split weights: either it runs, or the call fails -- never a silent
            // change of instruction (callers ask rst_gemm_win_b3_supported first)
            if (p.w3) {
                RST_REQUIRE(vec && p.split_k <= 1 && (uintptr_t)p.w3 % 16 == 0 &&
                            rst_gemm_win_b3_shape_ok(p.B, p.T_in, p.T_out, p.C, p.K, p.N, p.pad_mode, p.x_bstride, p.hist != nullptr),
                            "gemm_win_b3: shape / alignment not served by the three-plane kernel (B=%d T_in=%d T_out=%d C=%d K=%d N=%d pad=%d "
                            "hist=%d); ask rst_gemm_win_b3_supported", p.B, p.T_in, p.T_out, p.C, p.K, p.N, p.pad_mode, p.hist != nullptr);
                return launch_stream_b3(p, stream);
            }
            // fewer 128 x 128 tiles than CUs (one 80 ms frame of 32 streams at the 6 kHz level: 15 360 rows x 128 columns = 120 tiles, each
            // 17 MFLOP on the f32 matrix instruction = 27 us of ONE CU while half the chip idles): the 32-row tiles of the medium-row
            // form instead, four times the workgroups (round 6; 55 -> ... us for that launch)
            static const int under_knob = rst_knob("RST_GEMM_UNDERFILL", 1);      // tools build: 0 = the 128 x 128 tile whatever the count
            if (under_knob && tiles < gw_cu_count() && p.split_k <= 1) return launch_cfg<1, 1, 1, 4>(p, vec, stream);
            if (vec && !stream_off && p.split_k <= 1) {
                if (tiles >= 768 && !kb32_only) return launch_stream<16>(p, tiles, stream);
                return launch_stream<32>(p, tiles, stream);
            }
            if (tiles >= 768 && !kb32_only) return launch_cfg<2, 2, 2, 2, 16>(p, vec, stream);
            return launch_cfg<2, 2, 2, 2>(p, vec, stream);
        }
        case 1: return launch_cfg<1, 1, 1, 4>(p, vec, stream);               // 32 x 128 (few / medium rows: streaming steps)
        case 2: return launch_cfg<1, 2, 4, 1>(p, vec, stream);               // 128 x 64
        default: return launch_cfg<2, 1, 4, 1>(p, vec, stream);              // 256 x 32
    }
}

// scratch sizes of a split launch: counters (one per tile of the shape the launcher picks)
int rst_gemm_split_tiles_impl(long M, int N) {
    int BM, BN;
    gw_tile_dims(gw_tile_cfg(M, N), BM, BN);
    return (int)(((M + BM - 1) / BM) * ((N + BN - 1) / BN));
}

// ---- split weights of the three-plane bf16 form
long rst_gemm_win_b3_weight_elems_impl(int N, int K) { return (long)((N + 255) / 256) * 256 * K * 3; }      // whole 256-row tiles (zero rows)

int rst_launch_gemm_win_b3_pack(const float* w, unsigned short* w3, int N, int K, hipStream_t stream) {
    RST_REQUIRE(w && w3 && N > 0 && K > 0 && K % 16 == 0, "gemm_win_b3_pack_weight: bad arguments (K %% 16 == 0 required, N=%d K=%d)", N, K);
    RST_REQUIRE((uintptr_t)w % 16 == 0 && (uintptr_t)w3 % 16 == 0, "gemm_win_b3_pack_weight: pointers must be 16-byte aligned");
    const long total = (long)((N + 255) / 256) * 256 * (K / 4);
#ifdef RST_ABLATION
    if (!b3_dirw()) {
        hipLaunchKernelGGL(gemm_win_b3_pack_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<short*>(w3), N, K, total);
        return rst_check_launch("gemm_win_b3_pack_weight");
    }
#endif
    hipLaunchKernelGGL(gemm_win_b3_pack_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<short*>(w3), N, K, total);
    return rst_check_launch("gemm_win_b3_pack_weight");
}
