// The temporal transformer of one batch-1 LM step -- all L layers -- as ONE persistent launch: models/model.py:364-389 (LMModel.forward_text:
// `self.transformer(input_)` at T = 1 in streaming mode), modules/transformer.py:551-592 (StreamingTransformerLayer: rms_norm_f32 -> self
// attention -> + residual, rms_norm_f32 -> ActivationGating -> + residual), :376-423 (StreamingMultiheadAttention with RoPE and the
// RingKVCache of :211-278), modules/gating.py:12-51.
//
// Why: as launches a layer is qkv GEMV | attention | out-proj GEMV | ffn-in GEMV | ffn-out GEMV = its 411 MB of bf16 weights at ~6.9 TB/s
// plus ~4 us of ramp and drain per launch (profiles/r05_lm_timeline.csv: 83.5 us per layer, 20 us of it not streaming).  Here the
// weights are ONE stream per wave that runs across op and layer boundaries: a weight wave always has TF_NBUF blocks of 16 KB in flight
// in registers (weights do not depend on activations -- the "prefetch-credit" row of MI355X_MICROARCH.md's price list), the op
// boundaries are in-launch all-to-all hand-offs through 8-byte {epoch, value} granules (persist.h, Guideline 16 form R2), and the KV
// ring rows of the attention are part of the same stream (their addresses do not depend on the new step either).
//
// Roles: workgroup = TF_WW weight waves + TF_CW hand-off ("comm") waves, one workgroup per CU, all resident.
//   * weight wave gw = wg * TF_WW + w owns rows r = gw, gw + W, ... of every op (W = TF_WW * G) and, inside the attention, a share of the
//     ring slots of (head, split) = (wg / S, wg % S).  It never polls global memory: it waits at workgroup barriers for the comm waves to
//     have staged an op's input vector in LDS, multiplies the blocks it requested long before, publishes one granule per row.  Vector
//     memory results return in issue order, so a wave with 48 KB of weights in flight must not be the one that polls.
//   * comm waves (256 threads, no weight loads in flight) sweep the granules of the next input vector, apply the RMSNorm (the summation
//     order of gemv_norm_kernel: thread t owns k = t + 256 i), rotate and append the new key, merge the attention partials.
// Arithmetic equals the launch-per-op chain's (lm_step.hip, lm_attn.hip) up to the order of additions inside the wave reductions, the
// single k chain of the plain-vector GEMVs (gemv_ksplit_kernel sums four chains) and the split / merge order of the attention.
#include <utility>

#include "lm_common.h"
#include "persist.h"

namespace {

constexpr int TF_WW = 4, TF_CW = 4;
constexpr int TF_THREADS = 64 * (TF_WW + TF_CW), TF_CT = 64 * TF_CW;
constexpr int TF_HDR = 64;         // floats in front of the LDS carve: TfShared
constexpr int TF_MAX_SPLITS = 8;
constexpr int TF_TAB_MAX = 48;     // block descriptors of one layer per weight wave kept in LDS (7B shape: 25 - 28)

struct TfShared {
    float red[4];
    float pw_m[TF_WW], pw_l[TF_WW];
    int wstate[TF_WW];     // per weight wave: the barrier count it is waiting for (monotonic) = how far its rows have got
    int dead;
    int pad[3];
};

enum { TF_OP_A = 0, TF_OP_KV = 1, TF_OP_B = 2, TF_OP_C = 3, TF_OP_D = 4 };

// One block of a weight wave's stream: 2 rows x 8 pieces of 16 bytes per lane (16 KB per wave).  All fields are wave-uniform (SGPRs:
// the struct is kept small, TF_NBUF of them are live).
struct TfBlk {
    const char* p0;       // GEMV: row 0 at its first k of the block; KV: the head's K rows
    const char* p1;       // GEMV: row 1 (or row 0 again when absent); KV: the head's V rows
    int meta;             // type (bits 0-2: TF_OP_*, 7 = past the end of the stream, 6 = pipeline not filled yet) | flags << 3 (1: first block of its
                          // rows -- clear the accumulators, 2: last -- reduce + publish, 4: row 1 present) | pieces << 6 | layer << 16
    int kb;               // GEMV: first k of the block; KV: first slot of the block
    int r0;               // GEMV: output index of row 0 (row 1: r0 + W); KV: end of the workgroup's slot range
    int need;             // workgroup barriers passed before the block may be consumed
    __device__ __forceinline__ int type() const { return meta & 7; }
    __device__ __forceinline__ int flags() const { return (meta >> 3) & 7; }
    __device__ __forceinline__ int nch() const { return (meta >> 6) & 15; }
    __device__ __forceinline__ int layer() const { return meta >> 16; }
};
constexpr int TF_END = 7;

// ---- the weight stream's registers.  The loads of a weight wave are inline asm into FIXED registers v128 .. v255 (the asm text names
// them); the compiler is kept out of them by four 32-register "pin" variables that an empty asm statement defines in exactly those
// registers ("={v[128:159]}") when a block is issued and another one uses when its last piece has been taken: to the register
// allocator v128 .. v255 hold live values in between, so nothing else -- SGPR-spill lanes included -- can be placed there.  Every
// 16-byte piece is TAKEN out of its registers by one asm statement that first waits for it (s_waitcnt vmcnt(n), n = the number of
// YOUNGER loads: vector-memory loads return in issue order, stores in between only make a wait conservative) and unpacks it into
// registers the compiler owns.  tools/check_asm_loads.py verifies on the ISA that no compiler-generated instruction touches
// v128 .. v255 in the weight waves' code.  Earlier forms that failed (tools/probes/temporal_frame_probe.py and the checker found them):
// ordinary loads -- the compiler's wait-count pass waited for the YOUNGEST load before the first multiply (paths with different numbers
// of stores merge in this loop); asm loads into ordinary variables -- the register allocator copied, and on paths that skip a piece
// reused, destination registers while the loads were in flight; unpinned fixed registers -- the compiler parked SGPR spills in them;
// accumulation registers -- once the kernel admits to using them the allocator uses them for its own values as well.
constexpr int TF_NBUF = 2;                      // blocks of 16 KB in flight per weight wave
constexpr int TF_REG0 = 128;                    // block B, piece j, row r lives in v[tf_reg(B, j, r) : tf_reg(B, j, r) + 3]
__host__ __device__ constexpr int tf_reg(int B, int j, int r) { return TF_REG0 + 64 * B + 8 * j + 4 * r; }
// younger loads behind piece j of the OLDEST block in flight: 2 (7 - j) in its own block, 16 in each of the other blocks
__host__ __device__ constexpr int tf_younger(int j) { return 2 * (7 - j) + 16 * (TF_NBUF - 1); }
typedef unsigned int tf_pin_t __attribute__((ext_vector_type(32)));

template <int REG>
__device__ __forceinline__ void tf_load(unsigned voff, unsigned long long sbase) {
    asm volatile("global_load_dwordx4 v[%c2:%c3], %0, %1 nt" ::"v"(voff), "s"(sbase), "n"(REG), "n"(REG + 3));
}
template <int REG, int IMM>
__device__ __forceinline__ void tf_load_imm(unsigned voff, unsigned long long sbase) {
    asm volatile("global_load_dwordx4 v[%c2:%c3], %0, %1 offset:%c4 nt" ::"v"(voff), "s"(sbase), "n"(REG), "n"(REG + 3), "n"(IMM));
}
template <int REG, int N>      // 8 bf16 -> fp32 (the shifts / masks the multiply needs anyway)
__device__ __forceinline__ void tf_take_bf16(float (&o)[8]) {
    asm volatile("s_waitcnt vmcnt(%c8)\n\tv_lshlrev_b32 %0, 16, v%c9\n\tv_and_b32 %1, 0xffff0000, v%c9\n\tv_lshlrev_b32 %2, 16, v%c10\n\t"
                 "v_and_b32 %3, 0xffff0000, v%c10\n\tv_lshlrev_b32 %4, 16, v%c11\n\tv_and_b32 %5, 0xffff0000, v%c11\n\t"
                 "v_lshlrev_b32 %6, 16, v%c12\n\tv_and_b32 %7, 0xffff0000, v%c12"
                 : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3]), "=v"(o[4]), "=v"(o[5]), "=v"(o[6]), "=v"(o[7])
                 : "n"(N), "n"(REG), "n"(REG + 1), "n"(REG + 2), "n"(REG + 3));
}
template <int REG, int N>      // 4 fp32
__device__ __forceinline__ void tf_take_f32(float (&o)[4]) {
    asm volatile("s_waitcnt vmcnt(%c4)\n\tv_mov_b32 %0, v%c5\n\tv_mov_b32 %1, v%c6\n\tv_mov_b32 %2, v%c7\n\tv_mov_b32 %3, v%c8"
                 : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3])
                 : "n"(N), "n"(REG), "n"(REG + 1), "n"(REG + 2), "n"(REG + 3));
}
// the pins of block B: defined (no instruction) when a weight wave starts, used (no instruction but the final wait) behind its loop
template <int B>
__device__ __forceinline__ void tf_pin_define(tf_pin_t& lo, tf_pin_t& hi) {
    if constexpr (B == 0) asm volatile("" : "={v[128:159]}"(lo), "={v[160:191]}"(hi));
    else asm volatile("" : "={v[192:223]}"(lo), "={v[224:255]}"(hi));
}
template <int B>
__device__ __forceinline__ void tf_pin_drain(const tf_pin_t& lo, const tf_pin_t& hi) {      // everything has landed
    if constexpr (B == 0) asm volatile("s_waitcnt vmcnt(0)" ::"{v[128:159]}"(lo), "{v[160:191]}"(hi) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::"{v[192:223]}"(lo), "{v[224:255]}"(hi) : "memory");
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) -- the register numbers above
// are immediates of the asm text, so the piece index must be a constant expression, not the variable of an unrolled loop
template <typename F, int... Is>
__device__ __forceinline__ void tf_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void tf_for(F&& f) { tf_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <bool KV16, int D> struct TfKv {
    static constexpr int EB = KV16 ? 2 : 4;            // bytes per element
    static constexpr int ROWB = D * EB;                // bytes per ring row
    static constexpr int LPR = ROWB / 16;              // lanes per row
    static constexpr int SPL = 64 / LPR;               // slots per load instruction
    static constexpr int EPL = 16 / EB;                // elements per lane
    static constexpr int SPB = 8 * SPL;                // slots per block
    template <int REG, int N> static __device__ __forceinline__ void take(float (&o)[EPL]) {
        if constexpr (KV16) tf_take_bf16<REG, N>(o); else tf_take_f32<REG, N>(o);
    }
};

__device__ __forceinline__ unsigned short tf_bf16_rne(float f) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ void tf_barrier() { __syncthreads(); }

// NG granules per comm thread: i = first + 256 j + tc (i < n), swept (all requested before any is looked at) until every tag == epoch
template <int NG, typename Map>
__device__ __forceinline__ int tf_gather(const u64* g, int first, int n, unsigned epoch, int tc, Map map, float (&out)[NG], TfShared& sh,
                                         unsigned* status, unsigned code) {
    u64 v[NG];
    long long t0 = 0;
    int sweeps = 0;
    while (true) {
        ++sweeps;
        bool all = true;
#pragma unroll
        // (unconditional loads from a clamped index: a load under a per-lane condition is branched around and waited for inside its
        // branch -- eight serialised round trips per sweep instead of one)
        for (int j = 0; j < NG; ++j) v[j] = __hip_atomic_load(g + map(min(first + j * TF_CT + tc, n - 1)), DF_RLX);
#pragma unroll
        for (int j = 0; j < NG; ++j) all = all && (first + j * TF_CT + tc >= n || (unsigned)(v[j] >> 32) == epoch);
        if (all) break;
        if (t0 == 0) t0 = wall_clock64();
        if (*(volatile int*)&sh.dead || wall_clock64() - t0 > DF_TIMEOUT_TICKS) {
            sh.dead = 1;
            atomicOr(status, code);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) out[j] = __uint_as_float((unsigned)v[j]);
    return sweeps;
}

// 8 weights (taken out of a 16-byte piece, in k order) against 8 activations: the k order of gemv_kernel's chain
__device__ __forceinline__ float tf_dot8(const float (&w)[8], const f32x4& x0, const f32x4& x1, float a) {
    a = fmaf(w[0], x0[0], a); a = fmaf(w[1], x0[1], a);
    a = fmaf(w[2], x0[2], a); a = fmaf(w[3], x0[3], a);
    a = fmaf(w[4], x1[0], a); a = fmaf(w[5], x1[1], a);
    a = fmaf(w[6], x1[2], a); a = fmaf(w[7], x1[3], a);
    return a;
}

// tools build only: wall-clock stamps (100 MHz) of workgroup RST_TF_STAMP_WG's first comm thread at the hand-off boundaries of every layer
// (tools/probes/temporal_frame_phases.py prints them)
#ifdef RST_ABLATION
constexpr int TF_ST_LAYER = 16;
__device__ unsigned long long tf_stamps[RST_TEMPORAL_MAX_L * TF_ST_LAYER + 2];
__device__ int tf_stamp_wg = 0;
__device__ unsigned long long tf_wstamps[4 * 64];     // weight wave 0 of that workgroup, layer 1: per block {type, entry, synced, done}
#define TF_STAMP(i) do { if (!SOLO && tc == 0 && wg == tf_stamp_wg) tf_stamps[(i)] = wall_clock64(); } while (0)
#else
#define TF_STAMP(i) do {} while (0)
#endif

template <bool KV16, int D, bool SOLO>
__global__ __launch_bounds__(TF_THREADS) void temporal_frame_kernel(const TemporalFrameParams p) {
    typedef TfKv<KV16, D> KV;
    if (SOLO && __hip_atomic_load(p.status, DF_RLX) == 0u) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    TfShared& sh = *reinterpret_cast<TfShared*>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, G = gridDim.x;
    const int E = p.E, Hd = p.Hd, H = p.H, L = p.L, cap = p.cap;
    // staged vectors are padded with zeros to whole blocks of 8 pieces (4096 k): a piece past the end of a row repeats the last valid weights
    // and multiplies zeros, so the multiply needs no per-piece condition
    const int EP = (E + 4095) & ~4095, HdP = (Hd + 4095) & ~4095;
    // LDS carve (floats): header | xA [EP] (norm1 / norm2 output) | xB [EP] (attention output) | xD [HdP] (gated activation) |
    //                     xres0 [E] (layer input) | xres1 [E] (after the attention block) | qh [3D] | pw_o [TF_WW][D] |
    //                     block descriptors [TF_WW][TF_TAB_MAX][8] (words)
    // (offsets into `lds`, not pointers: a pointer selected at run time loses its address space and turns every access into a flat one,
    // which counts against BOTH wait counters and drains the weight stream at every LDS read)
    const int oA = TF_HDR, oB = oA + EP, oD = oB + EP, oR0 = oD + HdP, oR1 = oR0 + E, oQ = oR1 + E, oPW = oQ + 3 * D, oTAB = oPW + TF_WW * D;
#define xA (lds + oA)
#define xB (lds + oB)
#define xD (lds + oD)
#define xres0 (lds + oR0)
#define xres1 (lds + oR1)
#define qh (lds + oQ)
#define pw_o (lds + oPW)
    // granules (the repair launch has its own zeroed set behind the persistent launch's)
    const long gset = 5L * E + Hd + (long)H * TF_MAX_SPLITS * (D + 2);
    u64* gX = p.gran + (SOLO ? gset : 0L);
    u64* gQKV = gX + E;
    u64* gATT = gQKV + 3L * E;
    u64* gH = gATT + E;
    u64* gPART = gH + Hd;

    // row `arr` of the device pointer table (TemporalFrameParams::tab): 0 in_proj, 1 out_proj, 2 gate_in, 3 gate_out, 4 norm1, 5 norm2, 6 K, 7 V
    // (read through the CONSTANT address space: the table is invariant, so the load is a scalar one.  As an ordinary global load it became a
    // vector-memory load in the weight waves' code, and the compiler's wait for it -- counted without the weight stream's asm loads it does
    // not know about -- could be satisfied before the pointer had arrived: a memory fault, tools/check_asm_loads.py now refuses any
    // compiler-generated vector load there)
    typedef const unsigned long long __attribute__((address_space(4))) * tf_cptr;
    const tf_cptr tabc = (tf_cptr)p.tab;
    auto tptr = [&](int arr, int l) __attribute__((always_inline)) { return reinterpret_cast<const char*>(tabc[arr * L + l]); };
    // ---- attention geometry (the same for every layer: the position is the frame's)
    const long pos = *p.pos_dev;
    const int S = SOLO ? 1 : min(TF_MAX_SPLITS, G / H);            // splits per head
    const int n_used = (int)min((long)cap, pos + 1);
    const int active = SOLO ? 1 : max(1, min(S, (n_used + TF_WW * KV::SPB - 1) / (TF_WW * KV::SPB)));
    const int slot_cur = (int)(pos % cap);
    // visibility of a ring slot to the new step (RingKVCache.complete's slot -> position map, modules/transformer.py:254-278, incl. the
    // `delta <= 0` quirk) in 32-bit arithmetic: with end_offset = pos + 1 the distance pos - pos_k is -1 - delta (delta <= 0) or
    // cap - 1 - delta; equals ring_visible_at (lm_common.h) for every (slot, pos, cap, context) -- checked exhaustively on small rings.
    const int end_index = (int)((pos + 1) % cap);
    const int pos_c = (int)min(pos, 0x7fffffffL);
    const int ctx = p.context;
    auto tf_visible = [&](int slot) __attribute__((always_inline)) {
        const int delta = slot - end_index;
        const int dl = delta <= 0 ? -1 - delta : cap - 1 - delta;
        return slot < cap && slot <= pos_c && dl <= pos_c && dl >= 0 && (ctx <= 0 || dl < ctx);
    };
    const float att_scale = 1.0f / sqrtf((float)D);
    // workgroup barriers per layer: norm1 (2) | per head of this workgroup: q / k / v staged, partials written (2) | xB | norm2 (2) | xD
    const int NH = SOLO ? H : 1;
    const int NPH = 6 + 2 * NH;

    for (int i = tid; i < EP - E; i += TF_THREADS) { xA[E + i] = 0.f; xB[E + i] = 0.f; }
    for (int i = tid; i < HdP - Hd; i += TF_THREADS) xD[Hd + i] = 0.f;
    if (tid == 0) sh.dead = 0;
    if (tid < TF_WW) sh.wstate[tid] = 0;
    tf_barrier();

    if (wave < TF_WW) {
        // =================================================================================================== weight waves
        const int gw = wg * TF_WW + wave, W = G * TF_WW;
        auto rows_of = [&](int N) __attribute__((always_inline)) { return gw < N ? (N - gw + W - 1) / W : 0; };
        const int rowsA = rows_of(3 * E), rowsB = rows_of(E), rowsC = rows_of(Hd);
        const int KBd = (Hd + 4095) >> 12;
        const int nA = (rowsA + 1) >> 1, nB = (rowsB + 1) >> 1, nC = rowsC, nD = nB * KBd;
        const int g = lane / KV::LPR, sub = lane % KV::LPR;

        // generator of the wave's block stream: per layer A | KV (per head of this workgroup) | B | C | D.  State = (layer, block within the
        // layer); everything else is decoded from it (separate op / index / head counters bumped in sibling branches were merged by the
        // compiler into one store through a selected POINTER, which kept all of them in scratch memory; an incremental form with running
        // pointers cost so many more scalar registers that the compiler spilled one of the pins).
        int gl = 0, gt = 0;
        auto kv_range = [&](int hh, int& s_lo, int& s_hi, bool& on) __attribute__((always_inline)) {
            const int h = SOLO ? hh : wg / S, split = SOLO ? 0 : wg % S;
            on = h < H && split < active;
            const int per = (n_used + active - 1) / active;
            s_lo = split * per;
            s_hi = min(n_used, s_lo + per);
        };
        int kvb;                                                  // KV blocks of this wave per head (the same for every head it serves)
        {
            int s_lo, s_hi; bool on;
            kv_range(0, s_lo, s_hi, on);
            const int span = s_hi - s_lo - wave * KV::SPB;
            kvb = on && span > 0 ? (span + TF_WW * KV::SPB - 1) / (TF_WW * KV::SPB) : 0;
        }
        const int cA = nA, cKV = cA + NH * kvb, cB = cKV + nB, cC = cB + nC, nL = cC + nD;      // prefix sums of a layer's blocks
        // block t of a layer, as everything but the layer: what `next` needs to form its descriptor
        struct TfDec { int meta, kb, r0, need_rel, arr0, arr1; long off0, off1; };
        auto decode = [&](int t, TfDec& d) __attribute__((always_inline)) {
            const int gop = t < cA ? TF_OP_A : (t < cKV ? TF_OP_KV : (t < cB ? TF_OP_B : (t < cC ? TF_OP_C : TF_OP_D)));
            int i = t - (t < cA ? 0 : (t < cKV ? cA : (t < cB ? cKV : (t < cC ? cB : cC))));
            if (gop == TF_OP_KV) {
                int gh = 0;
                if (SOLO) { gh = i / kvb; i -= gh * kvb; }
                int s_lo, s_hi; bool on;
                kv_range(gh, s_lo, s_hi, on);
                const int h = SOLO ? gh : wg / S;
                d.kb = s_lo + (i * TF_WW + wave) * KV::SPB;
                d.arr0 = 6; d.arr1 = 7;
                d.off0 = d.off1 = (long)h * cap * KV::ROWB;
                d.r0 = s_hi;
                d.meta = TF_OP_KV | (((i == 0 ? 1 : 0) | (i == kvb - 1 ? 2 : 0)) << 3);
                d.need_rel = 3 + 2 * gh;
                return;
            }
            int K = E, kb = 0, r0, flags = 3;
            bool has1 = false;
            if (gop == TF_OP_A) { d.arr0 = 0; r0 = gw + 2 * i * W; has1 = 2 * i + 1 < rowsA; d.need_rel = 2; }
            else if (gop == TF_OP_B) { d.arr0 = 1; r0 = gw + 2 * i * W; has1 = 2 * i + 1 < rowsB; d.need_rel = 2 * NH + 3; }
            else if (gop == TF_OP_C) { d.arr0 = 2; r0 = gw + i * W; d.need_rel = 2 * NH + 5; }
            else {
                d.arr0 = 3; K = Hd;
                const int ip = i / KBd, kblk = i - ip * KBd;
                r0 = gw + 2 * ip * W; has1 = 2 * ip + 1 < rowsB;
                kb = kblk << 12;
                flags = (kblk == 0 ? 1 : 0) | (kblk == KBd - 1 ? 2 : 0);
                d.need_rel = 2 * NH + 6;
            }
            d.arr1 = d.arr0;
            const int rem = min(K - kb, 4096);
            d.kb = kb;
            d.meta = gop | ((flags | (has1 ? 4 : 0)) << 3) | (((rem + 511) >> 9) << 6);
            d.off0 = ((long)r0 * K + kb) * 2;
            d.off1 = gop == TF_OP_C ? (((long)Hd + r0) * K) * 2 : (has1 ? d.off0 + (long)W * K * 2 : d.off0);
            d.r0 = r0;
        };
        // The descriptors of ONE layer of this wave in LDS (8 words per block), written once: `next` then costs two broadcast LDS reads,
        // six v_readfirstlane and two scalar loads of the layer's pointers instead of ~100 scalar instructions of decoding in front of
        // every block (a lone wave per SIMD hides none of it).  The repair launch (many more blocks per layer) decodes as it goes.
        const bool use_tab = !SOLO && nL <= TF_TAB_MAX;
        unsigned* const mytab = reinterpret_cast<unsigned*>(lds) + oTAB + wave * (TF_TAB_MAX * 8);
        if (use_tab) {
            for (int t = 0; t < nL; ++t) {
                TfDec d;
                decode(t, d);
                if (lane == 0) {
                    unsigned* e = mytab + t * 8;
                    e[0] = (unsigned)d.meta; e[1] = (unsigned)d.kb; e[2] = (unsigned)d.r0; e[3] = (unsigned)(d.need_rel | (d.arr0 << 8) | (d.arr1 << 12));
                    e[4] = (unsigned)d.off0; e[5] = (unsigned)((unsigned long long)d.off0 >> 32);
                    e[6] = (unsigned)d.off1; e[7] = (unsigned)((unsigned long long)d.off1 >> 32);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's own writes, read back by this wave only
        }
        auto next = [&](TfBlk& b) __attribute__((always_inline)) {
            if (gt >= nL) { gt = 0; gl = gl + 1; }
            if (gl >= L) {
                b.meta = TF_END; b.p0 = b.p1 = tptr(0, 0); b.kb = 0; b.r0 = 0; b.need = L * NPH;
                return;
            }
            const int t = gt;
            gt = t + 1;
            int meta, kb, r0, misc;
            long off0, off1;
            if (use_tab) {
                const u32x4 ea = *reinterpret_cast<const u32x4*>(mytab + t * 8), eb = *reinterpret_cast<const u32x4*>(mytab + t * 8 + 4);
                meta = __builtin_amdgcn_readfirstlane((int)ea[0]); kb = __builtin_amdgcn_readfirstlane((int)ea[1]);
                r0 = __builtin_amdgcn_readfirstlane((int)ea[2]); misc = __builtin_amdgcn_readfirstlane((int)ea[3]);
                off0 = (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)eb[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)eb[0]));
                off1 = (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)eb[3]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)eb[2]));
            } else {
                TfDec d;
                decode(t, d);
                meta = d.meta; kb = d.kb; r0 = d.r0; misc = d.need_rel | (d.arr0 << 8) | (d.arr1 << 12); off0 = d.off0; off1 = d.off1;
            }
            b.meta = meta | (gl << 16); b.kb = kb; b.r0 = r0; b.need = gl * NPH + (misc & 255);
            b.p0 = tptr((misc >> 8) & 15, gl) + off0;
            b.p1 = tptr((misc >> 12) & 15, gl) + off1;
        };
        // v128 .. v255 belong to the weight stream for the whole life of a weight wave: the pins are live from here to the drain behind the loop
        tf_pin_t pin00, pin01, pin10, pin11;
        tf_pin_define<0>(pin00, pin01);
        tf_pin_define<1>(pin10, pin11);
        // 16 loads per block, always (the wait counts are static): pieces past the valid range repeat the last valid one
        auto issue = [&](auto BC, const TfBlk& b) __attribute__((always_inline)) {
            constexpr int B = decltype(BC)::value;
            unsigned off0, clampv, tail, jstride;
            if (b.type() == TF_OP_KV) {
                off0 = (unsigned)(b.kb + g) * KV::ROWB; clampv = (unsigned)(cap - 1) * KV::ROWB; tail = sub * 16; jstride = KV::SPL * KV::ROWB;
            } else {
                const int rem = b.type() == TF_END ? 8 : min((b.type() == TF_OP_D ? Hd : E) - b.kb, 4096);
                off0 = lane * 16; clampv = (unsigned)(rem * 2 - 16); tail = 0; jstride = 1024;
            }
            const unsigned long long a0 = (unsigned long long)b.p0, a1 = (unsigned long long)b.p1;
            // (readfirstlane returns a signed int: without the casts a low half with its top bit set sign-extends over the high half)
            const unsigned long long s0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a0 >> 32)) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a0);
            const unsigned long long s1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a1 >> 32)) << 32) |
                                          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a1);
            if (b.type() != TF_OP_KV && b.nch() == 8) {
                // a whole 2 x 8-piece block of two weight rows: piece j sits j KB behind the lane's 16 bytes -- immediate offsets, no
                // per-piece address arithmetic
                const unsigned v0 = lane * 16, v1 = v0 + 4096;
                tf_for<8>([&](auto JC) __attribute__((always_inline)) {
                    constexpr int j = decltype(JC)::value;
                    tf_load_imm<tf_reg(B, j, 0), (j & 3) * 1024>(j < 4 ? v0 : v1, s0);
                    tf_load_imm<tf_reg(B, j, 1), (j & 3) * 1024>(j < 4 ? v0 : v1, s1);
                });
            } else {
                tf_for<8>([&](auto JC) __attribute__((always_inline)) {
                    constexpr int j = decltype(JC)::value;
                    const unsigned o = min(off0 + j * jstride, clampv) + tail;
                    tf_load<tf_reg(B, j, 0)>(o, s0);
                    tf_load<tf_reg(B, j, 1)>(o, s1);
                });
            }
        };

        int phase = 0;
#ifdef RST_ABLATION
        int wcount = 0;
#endif
        float acc0 = 0.f, acc1 = 0.f;
        float am = -INFINITY, al = 0.f, ao[KV::EPL], aq[KV::EPL];
#pragma unroll
        for (int e = 0; e < KV::EPL; ++e) { ao[e] = 0.f; aq[e] = 0.f; }

        auto consume = [&](auto BC, const TfBlk& b) __attribute__((always_inline)) {
            constexpr int B = decltype(BC)::value;
#ifdef RST_ABLATION
            const bool wst = !SOLO && wg == tf_stamp_wg && wave == 0 && lane == 0 && b.layer() == 1 && wcount < 64 && b.type() != TF_END;
            if (wst) { tf_wstamps[4 * wcount] = (unsigned long long)b.type(); tf_wstamps[4 * wcount + 1] = wall_clock64(); }
#endif
            if (phase < b.need) {
                // the rows of every earlier op of this wave are published: tell the comm waves (they start polling for the next hand-off
                // only then -- 256 workgroups sweeping 32-90 KB of granules all through a 30 us op is terabytes per second of traffic
                // that the weight stream queues behind)
                if (lane == 0) *(volatile int*)&sh.wstate[wave] = b.need;
                while (phase < b.need) { tf_barrier(); ++phase; }
            }
            const int type = b.type(), flags = b.flags(), nch = b.nch(), bl = b.layer();
#ifdef RST_ABLATION
            if (wst) tf_wstamps[4 * wcount + 2] = wall_clock64();
            struct WDone { bool on; int& c; __device__ ~WDone() { if (on) { tf_wstamps[4 * c + 3] = wall_clock64(); ++c; } } } wdone{wst, wcount};
#endif
            if (type == TF_OP_KV) {
                // ---- a share of the ring slots of this workgroup's (head, split): modules/transformer.py:376-416 at T = 1
                if (flags & 1) {
                    am = -INFINITY; al = 0.f;
#pragma unroll
                    for (int e = 0; e < KV::EPL; ++e) { ao[e] = 0.f; aq[e] = qh[sub * KV::EPL + e]; }
                }
                // the new step's own key / value (staged by the comm waves at the ring's precision) replace the stale row of its slot
                float kc_[KV::EPL], vc_[KV::EPL];
#pragma unroll
                for (int e = 0; e < KV::EPL; ++e) { kc_[e] = qh[D + sub * KV::EPL + e]; vc_[e] = qh[2 * D + sub * KV::EPL + e]; }
                float sc[8];
                float mb = -INFINITY;
                tf_for<8>([&](auto JC) __attribute__((always_inline)) {
                    constexpr int j = decltype(JC)::value;
                    const int slot = b.kb + j * KV::SPL + g;
                    float kk[KV::EPL];
                    KV::template take<tf_reg(B, j, 0), tf_younger(j)>(kk);
                    const bool cur = slot == slot_cur;
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < KV::EPL; ++e) d = fmaf(cur ? kc_[e] : kk[e], aq[e], d);
                    d = group_sum(d, KV::LPR);
                    const bool ok = slot < b.r0 && tf_visible(slot);
                    sc[j] = ok ? d * att_scale : -INFINITY;
                    mb = fmaxf(mb, sc[j]);
                });
                const float m_new = fmaxf(am, mb);
                const float m_sub = m_new == -INFINITY ? 0.f : m_new;          // nothing visible so far: every weight below is 0
                const float alpha = am == -INFINITY ? 0.f : expf(am - m_sub);
                al *= alpha;
#pragma unroll
                for (int e = 0; e < KV::EPL; ++e) ao[e] *= alpha;
                tf_for<8>([&](auto JC) __attribute__((always_inline)) {
                    constexpr int j = decltype(JC)::value;
                    const int slot = b.kb + j * KV::SPL + g;
                    const bool cur = slot == slot_cur;
                    const float pj = sc[j] == -INFINITY ? 0.f : expf(sc[j] - m_sub);
                    al += pj;
                    float vv[KV::EPL];
                    KV::template take<tf_reg(B, j, 1), tf_younger(j)>(vv);
                    // (pj == 0 rows: ring memory is zero-initialised and only ever holds finite values, 0 * v adds nothing)
#pragma unroll
                    for (int e = 0; e < KV::EPL; ++e) ao[e] = fmaf(pj, cur ? vc_[e] : vv[e], ao[e]);
                });
                am = m_new;
                if (flags & 2) {
                    float m_w = am;
#pragma unroll
                    for (int off = KV::LPR; off < 64; off <<= 1) m_w = fmaxf(m_w, __shfl_xor(m_w, off));
                    const float f = am == -INFINITY ? 0.f : expf(am - m_w);
                    float l_w = al * f;
#pragma unroll
                    for (int e = 0; e < KV::EPL; ++e) ao[e] *= f;
#pragma unroll
                    for (int off = KV::LPR; off < 64; off <<= 1) {
                        l_w += __shfl_xor(l_w, off);
#pragma unroll
                        for (int e = 0; e < KV::EPL; ++e) ao[e] += __shfl_xor(ao[e], off);
                    }
                    if (g == 0) {
#pragma unroll
                        for (int e = 0; e < KV::EPL; ++e) pw_o[wave * D + sub * KV::EPL + e] = ao[e];
                        if (sub == 0) { sh.pw_m[wave] = m_w; sh.pw_l[wave] = l_w; }
                    }
                }
                return;
            }
            // ---- two weight rows against the staged vector
            if (flags & 1) { acc0 = 0.f; acc1 = 0.f; }
            const int xo = (type == TF_OP_B ? oB : (type == TF_OP_D ? oD : oA)) + b.kb + lane * 8;
            // the staged vector one piece ahead of the weights (its LDS latency hides behind the previous piece's multiplies); one chain per
            // row in k order -- the order of gemv_kernel's chain.  (Second accumulators per row, or per-piece partial sums, would shorten the
            // dependent chain, but every form tried raised the register pressure enough for the compiler to spill one of the pins.)
            f32x4 x0[8], x1[8];
            auto stage = [&](auto JC) __attribute__((always_inline)) {
                constexpr int j = decltype(JC)::value;
                x0[j] = *reinterpret_cast<const f32x4*>(&lds[xo + j * 512]);
                x1[j] = *reinterpret_cast<const f32x4*>(&lds[xo + j * 512 + 4]);
            };
            stage(std::integral_constant<int, 0>{});
            tf_for<8>([&](auto JC) __attribute__((always_inline)) {
                constexpr int j = decltype(JC)::value;
                if constexpr (j + 1 < 8) stage(std::integral_constant<int, j + 1>{});      // one piece ahead (more of them spilled a pin)
                float w0[8], w1[8];
                tf_take_bf16<tf_reg(B, j, 0), tf_younger(j)>(w0);
                tf_take_bf16<tf_reg(B, j, 1), tf_younger(j)>(w1);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc0 = fmaf(w0[e], x0[j][e], acc0); acc1 = fmaf(w1[e], x0[j][e], acc1); }
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc0 = fmaf(w0[4 + e], x1[j][e], acc0); acc1 = fmaf(w1[4 + e], x1[j][e], acc1); }
                __builtin_amdgcn_sched_barrier(0);      // piece by piece: a scheduler that clusters the takes keeps 8 x 16 weights live
            });
            (void)nch;
            if (flags & 2) {
                const float s0 = wave_sum_fast(acc0), s1 = wave_sum_fast(acc1);
                if (lane == 0) {
                    const unsigned l1 = (unsigned)bl + 1u;
                    const int r0 = b.r0, r1 = r0 + W;
                    const bool has1 = (flags & 4) != 0;
                    if (type == TF_OP_A) {
                        df_publish(gQKV + r0, l1, s0);
                        if (has1) df_publish(gQKV + r1, l1, s1);
                    } else if (type == TF_OP_B) {
                        df_publish(gX + r0, 2u * l1 - 1u, xres0[r0] + s0);
                        if (has1) df_publish(gX + r1, 2u * l1 - 1u, xres0[r1] + s1);
                    } else if (type == TF_OP_C) {
                        df_publish(gH + r0, l1, silu(s0) * s1);
                    } else {
                        const float v0 = xres1[r0] + s0;
                        df_publish(gX + r0, 2u * l1, v0);
                        if (bl == L - 1) p.y[r0] = v0;
                        if (has1) {
                            const float v1 = xres1[r1] + s1;
                            df_publish(gX + r1, 2u * l1, v1);
                            if (bl == L - 1) p.y[r1] = v1;
                        }
                    }
                }
            }
        };

        // the pipeline: TF_NBUF blocks in flight; the first round only fills it (type -2: nothing to consume yet)
        // Two blocks in flight, alternating; no path skips a consume (a block whose pieces are never taken leaves its registers dead in
        // the compiler's eyes while the loads are in flight).  Behind the end of the stream `next` hands out dummy blocks (type TF_END:
        // 16 loads of one valid address, taken and dropped) so that the wait counts of the last real block still hold.
        TfBlk q0, q1;
        static_assert(TF_NBUF == 2, "the loop below alternates two blocks");
        typedef std::integral_constant<int, 0> B0;
        typedef std::integral_constant<int, 1> B1;
        // thin: a wave that has consumed the last block of an op waits for the NEXT op's input before it requests more (one block per
        // wave in flight across a hand-off instead of two): the comm waves' polls go through the same CU memory queue as the weight
        // loads and come back behind everything requested before them
        auto sync_to = [&](int need) __attribute__((always_inline)) {
            if (phase < need) {
                if (lane == 0) *(volatile int*)&sh.wstate[wave] = need;
                while (phase < need) { tf_barrier(); ++phase; }
            }
        };
        next(q0); issue(B0{}, q0);
        next(q1); issue(B1{}, q1);
        while (true) {
            // (the descriptor of the block that will REPLACE the one about to be consumed is formed first: its scalar loads of the layer's
            // pointers then come back behind the multiplies instead of in front of the 16 requests)
            TfBlk n0, n1;
            const bool end0 = q0.type() == TF_END;
            if (!end0) next(n0);
            consume(B0{}, q0);
            if (end0) break;
            if (p.thin) sync_to(q1.need);
            q0 = n0; issue(B0{}, q0);
            const bool end1 = q1.type() == TF_END;
            if (!end1) next(n1);
            consume(B1{}, q1);
            if (end1) break;
            if (p.thin) sync_to(q0.need);
            q1 = n1; issue(B1{}, q1);
        }
        // one block of dummy loads is still in flight (the other buffer's): its registers stay pinned until it has landed
        tf_pin_drain<0>(pin00, pin01);
        tf_pin_drain<1>(pin10, pin11);
        while (phase < L * NPH) { tf_barrier(); ++phase; }
    } else {
        // ===================================================================================================== comm waves
        const int tc = tid - 64 * TF_WW, cw = wave - TF_WW;
        auto ident = [](int i) { return i; };
        // this workgroup's weight waves have published every row in front of barrier `target` (all workgroups carry the same load, so
        // the others are about there too): only now does a sweep have a chance
        auto wait_rows = [&](int target) __attribute__((always_inline)) {
            while (true) {
                int m = *(volatile int*)&sh.wstate[0];
#pragma unroll
                for (int w = 1; w < TF_WW; ++w) m = min(m, *(volatile int*)&sh.wstate[w]);
                if (m >= target || *(volatile int*)&sh.dead) break;
                __builtin_amdgcn_s_sleep(4);
            }
        };
        float xr[16], al[16];
#ifdef RST_ABLATION
        int dbg_slot = 0;
#endif
        auto load_alpha = [&](const float* a) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 16; ++j) al[j] = a[min(tc + j * TF_CT, E - 1)];
        };
        // RMSNorm of the 16 values per thread in xr (modules/transformer.py:34-46; the summation order of gemv_norm_kernel) -> xA
        auto norm_to_xA = [&]() __attribute__((always_inline)) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) s = fmaf(xr[j], xr[j], s);
            s = wave_sum(s);
            if (lane == 0) sh.red[cw] = s;
            tf_barrier();
            const float tot = ((sh.red[0] + sh.red[1]) + sh.red[2]) + sh.red[3];
            const float r = 1.0f / sqrtf(p.eps + tot / (float)E);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = tc + j * TF_CT;
                if (k < E) xA[k] = xr[j] * (al[j] * r);
            }
            tf_barrier();
        };
        // the vector behind gX at `epoch` -> xr (registers) and dst (LDS)
        auto gather_x = [&](unsigned epoch, float* dst, unsigned code) __attribute__((always_inline)) {
            const int sweeps = tf_gather<16>(gX, 0, E, epoch, tc, ident, xr, sh, p.status, code);
            (void)sweeps;
#ifdef RST_ABLATION
            if (!SOLO && tc == 0 && wg == tf_stamp_wg) tf_stamps[dbg_slot] = (unsigned long long)sweeps;
#endif
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = tc + j * TF_CT;
                if (k < E) dst[k] = xr[j]; else xr[j] = 0.f;
            }
        };
        for (int l = 0; l < L; ++l) {
            const unsigned l1 = (unsigned)l + 1u;
#ifdef RST_ABLATION
            const int sl = l * TF_ST_LAYER;
#endif
            TF_STAMP(sl + 0);                                    // layer start
            // ---- layer input -> xres0, norm1 -> xA
            load_alpha(reinterpret_cast<const float*>(tptr(4, l)));
            if (l == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int k = tc + j * TF_CT;
                    xr[j] = k < E ? p.x[k] : 0.f;
                    if (k < E) xres0[k] = xr[j];
                }
            } else {
                wait_rows(l * NPH + 2);
                TF_STAMP(sl + 10);                               // own ffn-out rows of the previous layer published
#ifdef RST_ABLATION
                dbg_slot = sl + 11;
#endif
                gather_x(2u * l1 - 2u, xres0, 1u);
            }
            TF_STAMP(sl + 1);                                    // x gathered
            norm_to_xA();                                        // barriers 1, 2
            TF_STAMP(sl + 2);                                    // norm1 staged
            // ---- attention
            for (int hh = 0; hh < NH; ++hh) {
                const int h = SOLO ? hh : wg / S, split = SOLO ? 0 : wg % S;
                const bool att_on = h < H && split < active;
                if (tc < TF_WW) { sh.pw_m[tc] = -INFINITY; sh.pw_l[tc] = 0.f; }
                if (att_on) {
                    // q, k, v of the head (pairs (2i, 2i + 1) per thread), rotation (modules/rope.py:37-62), ring append by split 0
                    if (hh == 0) wait_rows(l * NPH + 3);
                    const int pairs = 3 * D / 2;
                    for (int pi = tc; pi < pairs; pi += TF_CT) {
                        const int part = pi / (D / 2), i2 = pi - part * (D / 2), d = 2 * i2;
                        const u64* src = gQKV + (long)part * E + h * D + d;
                        u64 v0, v1;
                        long long t0 = 0;
                        while (true) {
                            v0 = __hip_atomic_load(src, DF_RLX);
                            v1 = __hip_atomic_load(src + 1, DF_RLX);
                            if ((unsigned)(v0 >> 32) == l1 && (unsigned)(v1 >> 32) == l1) break;
                            if (t0 == 0) t0 = wall_clock64();
                            if (*(volatile int*)&sh.dead || wall_clock64() - t0 > DF_TIMEOUT_TICKS) { sh.dead = 1; atomicOr(p.status, 2u); break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        float a = __uint_as_float((unsigned)v0), c = __uint_as_float((unsigned)v1);
                        if (part < 2 && p.rope_cs) {
                            const float rc = p.rope_cs[2 * i2], rs = p.rope_cs[2 * i2 + 1];
                            const float ra = a * rc - c * rs, rb = a * rs + c * rc;
                            a = ra; c = rb;
                        }
                        if (part > 0) {
                            if (KV16) {
                                const unsigned short ha = tf_bf16_rne(a), hc = tf_bf16_rne(c);
                                a = __uint_as_float((unsigned)ha << 16); c = __uint_as_float((unsigned)hc << 16);
                                if (split == 0) {
                                    unsigned short* ring = reinterpret_cast<unsigned short*>(const_cast<char*>(tptr(part == 1 ? 6 : 7, l)));
                                    *reinterpret_cast<unsigned*>(ring + ((long)h * cap + slot_cur) * D + d) = (unsigned)ha | ((unsigned)hc << 16);
                                }
                            } else if (split == 0) {
                                float* ring = reinterpret_cast<float*>(const_cast<char*>(tptr(part == 1 ? 6 : 7, l)));
                                float* dst = ring + ((long)h * cap + slot_cur) * D + d;
                                dst[0] = a; dst[1] = c;
                            }
                        }
                        qh[part * D + d] = a;
                        qh[part * D + d + 1] = c;
                    }
                }
                TF_STAMP(sl + 3);                                // q / k / v of the head gathered and rotated
                tf_barrier();                                    // barrier 3: qh staged -> the weight waves walk their slots
                tf_barrier();                                    // barrier 4: their partials are in LDS
                TF_STAMP(sl + 4);                                // ring walked
                if (att_on && tc < D) {
                    float M = -INFINITY;
#pragma unroll
                    for (int w = 0; w < TF_WW; ++w) M = fmaxf(M, sh.pw_m[w]);
                    float Ls = 0.f, O = 0.f;
#pragma unroll
                    for (int w = 0; w < TF_WW; ++w) {
                        if (sh.pw_m[w] != -INFINITY) {
                            const float fw = expf(sh.pw_m[w] - M);
                            Ls = fmaf(sh.pw_l[w], fw, Ls);
                            O = fmaf(pw_o[w * D + tc], fw, O);
                        }
                    }
                    if (active == 1) {
                        df_publish(gATT + h * D + tc, l1, Ls > 0.f ? O / Ls : 0.f);
                    } else {
                        u64* pt = gPART + ((long)h * TF_MAX_SPLITS + split) * (D + 2);
                        df_publish(pt + 2 + tc, l1, O);
                        if (tc == 0) { df_publish(pt, l1, M); df_publish(pt + 1, l1, Ls); }
                        if (split == 0) {
                            // the head's owner: the partials of all active splits, merged in split order
                            const u64* p0 = gPART + (long)h * TF_MAX_SPLITS * (D + 2);
                            u64 gm[TF_MAX_SPLITS], gl_[TF_MAX_SPLITS], go[TF_MAX_SPLITS];
                            long long t0 = 0;
                            while (true) {
                                bool all = true;
#pragma unroll
                                for (int s = 0; s < TF_MAX_SPLITS; ++s) {
                                    const u64* ps = p0 + (long)min(s, active - 1) * (D + 2);
                                    gm[s] = __hip_atomic_load(ps, DF_RLX);
                                    gl_[s] = __hip_atomic_load(ps + 1, DF_RLX);
                                    go[s] = __hip_atomic_load(ps + 2 + tc, DF_RLX);
                                }
#pragma unroll
                                for (int s = 0; s < TF_MAX_SPLITS; ++s)
                                    all = all && (unsigned)(gm[s] >> 32) == l1 && (unsigned)(gl_[s] >> 32) == l1 && (unsigned)(go[s] >> 32) == l1;
                                if (all) break;
                                if (t0 == 0) t0 = wall_clock64();
                                if (*(volatile int*)&sh.dead || wall_clock64() - t0 > DF_TIMEOUT_TICKS) { sh.dead = 1; atomicOr(p.status, 4u); break; }
                                __builtin_amdgcn_s_sleep(1);
                            }
                            float Mg = -INFINITY;
#pragma unroll
                            for (int s = 0; s < TF_MAX_SPLITS; ++s)
                                if (s < active) Mg = fmaxf(Mg, __uint_as_float((unsigned)gm[s]));
                            float Lg = 0.f, Og = 0.f;
#pragma unroll
                            for (int s = 0; s < TF_MAX_SPLITS; ++s) {
                                const float ms = __uint_as_float((unsigned)gm[s]);
                                if (s < active && ms != -INFINITY) {
                                    const float fs = expf(ms - Mg);
                                    Lg = fmaf(__uint_as_float((unsigned)gl_[s]), fs, Lg);
                                    Og = fmaf(__uint_as_float((unsigned)go[s]), fs, Og);
                                }
                            }
                            df_publish(gATT + h * D + tc, l1, Lg > 0.f ? Og / Lg : 0.f);
                        }
                    }
                }
            }
            // ---- attention output of all heads -> xB
            {
                float t[16];
                wait_rows(l * NPH + 3);          // (the heads cannot be done before the in-projection rows are)
                __builtin_amdgcn_s_sleep(48);    // ... nor before their owners have gathered q / k / v and walked a block: ~1.3 us without a sweep
                tf_gather<16>(gATT, 0, E, l1, tc, ident, t, sh, p.status, 8u);
#pragma unroll
                for (int j = 0; j < 16; ++j) { const int k = tc + j * TF_CT; if (k < E) xB[k] = t[j]; }
            }
            load_alpha(reinterpret_cast<const float*>(tptr(5, l)));
            TF_STAMP(sl + 5);                                    // attention output of all heads gathered
            tf_barrier();                                        // barrier 5: xB staged -> out-projection rows
            // ---- x after the attention block -> xres1, norm2 -> xA
            wait_rows(l * NPH + 2 * NH + 5);
            TF_STAMP(sl + 12);                                   // own out-projection rows published
#ifdef RST_ABLATION
            dbg_slot = sl + 13;
#endif
            gather_x(2u * l1 - 1u, xres1, 16u);
            TF_STAMP(sl + 6);                                    // x after the attention block gathered
            norm_to_xA();                                        // barriers 6, 7
            TF_STAMP(sl + 7);                                    // norm2 staged
            // ---- gated activation -> xD
            wait_rows(l * NPH + 2 * NH + 6);
            TF_STAMP(sl + 14);                                   // own ffn-in rows published
            for (int first = 0; first < Hd; first += 8 * TF_CT) {       // (sweeps of 16 or of all 44 granules per thread made the compiler spill)
                float t[8];
                tf_gather<8>(gH, first, Hd, l1, tc, ident, t, sh, p.status, 32u);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int k = first + tc + j * TF_CT; if (k < Hd) xD[k] = t[j]; }
            }
            TF_STAMP(sl + 8);                                    // gated activation gathered
            tf_barrier();                                        // barrier 8: xD staged -> ffn-out rows
            TF_STAMP(sl + 9);
        }
    }
    if (SOLO) df_solo_done(p.status);
#undef xA
#undef xB
#undef xD
#undef xres0
#undef xres1
#undef qh
#undef pw_o
}

#ifdef RST_ABLATION
}  // namespace
extern "C" int rst_debug_temporal_frame_stamps(unsigned long long* out, int n, int wg) {
    if (n > RST_TEMPORAL_MAX_L * TF_ST_LAYER) n = RST_TEMPORAL_MAX_L * TF_ST_LAYER;
    const int r = hipMemcpyFromSymbol(out, HIP_SYMBOL(tf_stamps), sizeof(unsigned long long) * n) == hipSuccess ? n : -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(tf_stamp_wg), &wg, sizeof(int));
    return r;
}
extern "C" int rst_debug_temporal_frame_wstamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tf_wstamps), sizeof(unsigned long long) * 256) == hipSuccess ? 256 : -1;
}
namespace {
#endif

int tf_cu_count() { return rst_cu_count(); }

size_t tf_lds_bytes(const TemporalFrameParams& p) {
    const int EP = (p.E + 4095) & ~4095, HdP = (p.Hd + 4095) & ~4095;
    static_assert(sizeof(TfShared) <= TF_HDR * 4, "LDS header too small");
    return (size_t)(TF_HDR + 2 * EP + HdP + 2 * p.E + 3 * p.D + TF_WW * p.D + TF_WW * TF_TAB_MAX * 8) * sizeof(float);
}

}  // namespace

// gX [E] | gQKV [3E] | gATT [E] | gH [Hd] | gPART [H][8][D + 2]
long rst_temporal_frame_workspace_granules(int E, int Hd, int H, int D) { return 5L * E + Hd + (long)H * TF_MAX_SPLITS * (D + 2); }

// Workgroups of the persistent launch for a shape, 0 if it is not served.
int rst_temporal_frame_grid(const TemporalFrameParams& p) {
    if (!(p.E > 0 && p.E % 8 == 0 && p.E <= 4096 && p.Hd > 0 && p.Hd % 8 == 0 && p.H > 0 && (p.D == 64 || p.D == 128) && p.H * p.D == p.E &&
          p.L >= 1 && p.L <= RST_TEMPORAL_MAX_L && p.cap >= 1))
        return 0;
    const size_t lds = tf_lds_bytes(p);
    if (lds > 150 * 1024) return 0;
    // every weight wave owns a row of the narrowest all-to-all op (E rows); every head needs a workgroup
    int G = tf_cu_count();
    if (G > p.E / TF_WW) G = p.E / TF_WW;
    static const int cap = rst_knob("RST_TF_GRID", 0);      // tools build only
    if (cap > 0 && G > cap) G = cap;
    if (G < 1 || p.H > G) return 0;
    static signed char fits[RST_MAX_DEVICES][2];            // [device][kv_bf16]
    signed char uncached = 0;
    signed char& f = rst_device_cell(&fits[0][p.kv_bf16 ? 1 : 0], 2, uncached);
    if (f == 0) {
        const void* kern = p.kv_bf16 ? reinterpret_cast<const void*>(temporal_frame_kernel<true, 128, false>)
                                     : reinterpret_cast<const void*>(temporal_frame_kernel<false, 128, false>);
        (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        int nb = 0;
        const hipError_t e = p.kv_bf16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, temporal_frame_kernel<true, 128, false>, TF_THREADS, 150 * 1024)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, temporal_frame_kernel<false, 128, false>, TF_THREADS, 150 * 1024);
        (void)hipGetLastError();
        f = (e == hipSuccess && nb >= 1) ? 1 : -1;
    }
    return f > 0 ? G : 0;
}

int rst_launch_temporal_frame(const TemporalFrameParams& p, hipStream_t stream) {
    RST_REQUIRE(p.tab && p.x && p.y && p.pos_dev && p.gran && p.status, "temporal_frame: null buffers");
    const int G = rst_temporal_frame_grid(p);
    RST_REQUIRE(G > 0, "temporal_frame: unsupported shape (E=%d Hd=%d H=%d D=%d L=%d cap=%d) or no resident grid for it", p.E, p.Hd, p.H, p.D, p.L, p.cap);
    const size_t lds = tf_lds_bytes(p);
    TemporalFrameParams pk = p;
    static const int thin = rst_knob("RST_TF_THIN", 0);      // tools build only (A/B)
    pk.thin = thin;
    if (hipMemsetAsync(p.gran, 0, (size_t)rst_temporal_frame_workspace_granules(p.E, p.Hd, p.H, p.D) * 16, stream) != hipSuccess) {
        rst_set_error("temporal_frame: workspace memset failed");
        return RST_ERR_LAUNCH;
    }
    auto go = [&](auto kern, int grid) {
        static RstOncePerDevice attr_once;
        if (attr_once.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(TF_THREADS), lds, stream, pk);
    };
    static const int no_repair = rst_knob("RST_TF_NO_REPAIR", 0);      // tools build only
    const int key = (p.kv_bf16 ? 2 : 0) + (p.D == 128 ? 1 : 0);
    switch (key) {
        case 3: go(temporal_frame_kernel<true, 128, false>, G); if (!no_repair) go(temporal_frame_kernel<true, 128, true>, 1); break;
        case 2: go(temporal_frame_kernel<true, 64, false>, G); if (!no_repair) go(temporal_frame_kernel<true, 64, true>, 1); break;
        case 1: go(temporal_frame_kernel<false, 128, false>, G); if (!no_repair) go(temporal_frame_kernel<false, 128, true>, 1); break;
        default: go(temporal_frame_kernel<false, 64, false>, G); if (!no_repair) go(temporal_frame_kernel<false, 64, true>, 1); break;
    }
    return rst_check_launch("temporal_frame");
}
