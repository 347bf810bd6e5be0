// Residual vector quantiser: nearest-codeword search (all levels of a group fused, residual kept in LDS) and the
// decode-side gather.
//
// Arithmetic contract (bit-exact against oracle/rvq_ref.c):
//     dot(x, e)  = fmaf chain over k = 0 .. D-1 in ascending order, starting from 0
//     e2(e)      = fmaf chain  e2 = fmaf(e[k], e[k], e2), k ascending
//     score      = fmaf(-2, dot, e2)              (= |x-e|^2 - |x|^2: same argmin as cdist, core_vq.py:179-185)
//     code       = lowest index among the minimal scores;   residual -= emb[code]   (exact fp32)
// v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain, so feeding it k = 2s (lane half 0) and 2s+1 (lane half 1)
// at step s reproduces that order exactly.  The codebook is pre-packed so that the A operand of four consecutive
// MFMA steps is ONE coalesced 16-byte global load per lane (codebook levels are 2 MiB: L2 resident).
#include "rst_common.h"
#include "rst_kernels.h"
#include <math.h>

namespace {

constexpr int FR = 32;       // frames per workgroup
constexpr int NW = 8;        // waves per workgroup

__device__ __forceinline__ int pk_off(int k) { return (k >> 3) * 8 + (k & 1) * 4 + ((k >> 1) & 3); }

__global__ __launch_bounds__(256) void rvq_pack_kernel(const float* __restrict__ emb, float* __restrict__ packed,
                                                       float* __restrict__ e2, int n_codes, int D) {
    const long total = (long)n_codes * D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % D);
        const long code = idx / D;
        // [k/8][code][k&1][(k>>1)&3]
        packed[((long)(k >> 3) * n_codes + code) * 8 + (k & 1) * 4 + ((k >> 1) & 3)] = emb[idx];
    }
    for (long code = (long)blockIdx.x * 256 + threadIdx.x; code < n_codes; code += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < D; ++k) s = fmaf(emb[code * D + k], emb[code * D + k], s);
        e2[code] = s;
    }
}

// KQT: D / 8 when known at compile time (the codec's 256-d codebooks: 32) -- selects the pipelined operand loop; 0: any D
template <int KQT>
__global__ __launch_bounds__(64 * NW) void rvq_search_kernel(const RvqSearchParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = p.D, LD = D + 4;
    float* r_pk = smem;                       // [FR][LD] residual, k-permuted (pk_off)
    float* e2s = r_pk + FR * LD;              // [n_codes]
    float* red_s = e2s + p.n_codes;           // [NW][FR]
    int* red_i = reinterpret_cast<int*>(red_s + NW * FR);  // [NW][FR]
    int* win = red_i + NW * FR;               // [FR]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * FR;
    const int g = blockIdx.y;
    const int tiles = p.n_codes / 32;
    const int tpw = (tiles + NW - 1) / NW;

    for (int idx = tid; idx < FR * D; idx += 64 * NW) {
        const int f = idx / D, k = idx - f * D;
        const int m = m0 + f;
        r_pk[f * LD + pk_off(k)] = m < p.M ? p.x[(long)m * p.ldx + g * D + k] : 0.f;
    }

    for (int li = 0; li < p.group_count[g]; ++li) {
        const int lvl = p.group_begin[g] + li;
        const float* packed = p.packed + (long)lvl * p.n_codes * D;
        const float* emb = p.emb + (long)lvl * p.n_codes * D;
        for (int c = tid; c < p.n_codes; c += 64 * NW) e2s[c] = p.e2[(long)lvl * p.n_codes + c];
        __syncthreads();

        float best = INFINITY;
        int bidx = 0x7fffffff;
        const float* rrow = r_pk + j * LD + h * 4;
        // Codebook operand of a code tile: D / 8 16-byte loads per lane, 64 KB apart (k-major packing).  With ~250 workgroups of 7 levels
        // on 256 CUs a SIMD holds two waves, and a loop that requested 8 steps, waited, and ran their 32 MFMAs left the matrix pipe
        // idle through every L2 round trip (round 2: pipe busy 0.46).  KQT form: the 8 steps AFTER the ones being consumed are always
        // in flight, across tile boundaries (two register buffers, all indices compile-time).  Either way the MFMA chain consumes k
        // in ascending order: scores stay bit-identical to oracle/rvq_ref.c.
        const int ct_lo = wave * tpw, ct_hi = min(tiles, (wave + 1) * tpw);
        if (KQT > 0) {
            constexpr int HB = 8, NH = KQT > 0 ? KQT / HB : 1;          // steps per buffer, buffers per tile (even)
            f32x4 ab[2][HB];
            auto request = [&](int ct, int half, f32x4 (&dst)[HB]) {
                const float* ap = packed + (long)(ct * 32 + j) * 8 + h * 4 + (long)half * HB * p.n_codes * 8;
#pragma unroll
                for (int u = 0; u < HB; ++u) dst[u] = *reinterpret_cast<const f32x4*>(ap + (long)u * p.n_codes * 8);
            };
            if (ct_lo < ct_hi) request(ct_lo, 0, ab[0]);
            for (int ct = ct_lo; ct < ct_hi; ++ct) {
                const int c0 = ct * 32;
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int hf = 0; hf < NH; ++hf) {
                    if (hf + 1 < NH) request(ct, hf + 1, ab[(hf + 1) & 1]);
                    else if (ct + 1 < ct_hi) request(ct + 1, 0, ab[0]);          // NH even: the next tile starts in buffer 0
#pragma unroll
                    for (int u = 0; u < HB; ++u) {
                        const f32x4 a = ab[hf & 1][u];
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(rrow + (hf * HB + u) * 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bq[e], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int code = c0 + rst_mfma32_row(r, lane);
                    const float sc = fmaf(-2.0f, acc[r], e2s[code]);
                    if (sc < best) { best = sc; bidx = code; }
                }
            }
        } else {
            for (int ct = ct_lo; ct < ct_hi; ++ct) {
                const int c0 = ct * 32;
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                const float* ap = packed + (long)(c0 + j) * 8 + h * 4;
#pragma unroll 8
                for (int kq = 0; kq < D / 8; ++kq) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + (long)kq * p.n_codes * 8);
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(rrow + kq * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bq[e], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int code = c0 + rst_mfma32_row(r, lane);
                    const float sc = fmaf(-2.0f, acc[r], e2s[code]);
                    if (sc < best) { best = sc; bidx = code; }
                }
            }
        }
        {   // the two lane halves hold different codes of the same frame
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bidx, 32);
            if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (h == 0) { red_s[wave * FR + j] = best; red_i[wave * FR + j] = bidx; }
        __syncthreads();
        if (tid < FR) {
            float bs = red_s[tid];
            int bi = red_i[tid];
            for (int w = 1; w < NW; ++w) {
                const float s = red_s[w * FR + tid];
                const int i = red_i[w * FR + tid];
                if (s < bs || (s == bs && i < bi)) { bs = s; bi = i; }
            }
            if (bi < 0 || bi >= p.n_codes) bi = 0;  // only reachable with NaN inputs
            win[tid] = bi;
            const int m = m0 + tid;
            if (m < p.M) {
                const int b = m / p.F, f = m - b * p.F;
                p.codes[((long)b * p.L + lvl) * p.F + f] = bi;
                if (p.dist) p.dist[(long)lvl * p.M + m] = bs;
            }
        }
        __syncthreads();
        if (li + 1 < p.group_count[g]) {
            for (int idx = tid; idx < FR * D; idx += 64 * NW) {
                const int f = idx / D, k = idx - f * D;
                r_pk[f * LD + pk_off(k)] -= emb[(long)win[f] * D + k];
            }
        }
        __syncthreads();
    }
}

// ---- few-frame path (streaming steps: M <= a few dozen frames) ------------------------------------------------------
// The big kernel serialises 2048 codes x 7 levels inside one workgroup, which is the right shape for thousands of frames
// but ~1 ms of latency for one.  Here the CODES are spread over workgroups instead: one launch per residual level, every
// workgroup re-derives the level's residual from x and the winners of the previous levels (exact, sequential
// subtractions), scores its 128 codes with the SAME k-ordered MFMA chain, and publishes (score, index) with a 64-bit
// atomicMin on an order-preserving key -- lowest score, then lowest index: bit-identical decisions to the big kernel.
__device__ __forceinline__ unsigned long long rvq_key(float score, int code) {
    unsigned u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // monotonic float -> uint
    return ((unsigned long long)u << 32) | (unsigned)code;
}

__global__ __launch_bounds__(256) void rvq_level_kernel(const RvqSearchParams p, unsigned long long* __restrict__ keys, int step) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = p.D, LD = D + 4;
    float* r_pk = smem;                       // [FR][LD]
    int* prev = reinterpret_cast<int*>(r_pk + FR * LD);   // [FR] winner of the previous level
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int g = blockIdx.y;
    if (step >= p.group_count[g]) return;
    const int lvl = p.group_begin[g] + step;
    const int m0 = blockIdx.z * FR;
    // residual of this level: ((x - e_0[c_0]) - e_1[c_1]) - ...   in level order, exactly as the fused kernel does.  The
    // winners of ALL previous levels are fetched first (one barrier), then every element gathers its `step` codeword entries
    // with independent loads and subtracts them in order -- one memory round trip instead of one per level.
    constexpr int MAX_PREV = 16;
    // only the frames that exist are derived (one or two per stream and step in streaming use); the other rows of the MFMA
    // operand are zeros
    const int frv = min(FR, p.M - m0);
    for (int idx = tid; idx < step * frv; idx += 256) {
        const int sidx = idx / frv, f = idx - sidx * frv;
        const int lv = p.group_begin[g] + sidx;
        prev[sidx * FR + f] = (int)(keys[(long)lv * p.M + m0 + f] & 0xffffffffu);
    }
    for (int idx = tid + frv * D; idx < FR * D; idx += 256) {
        const int f = idx / D, k = idx - f * D;
        r_pk[f * LD + pk_off(k)] = 0.f;
    }
    __syncthreads();
    // (U elements per thread and pass, all their loads requested before the first subtraction: with one element per pass a
    // workgroup of 32 frames walked 32 dependent memory round trips -- 31 us per level at 32 streams, 11 us at one)
    constexpr int U = 4;
    for (int idx0 = tid; idx0 < frv * D; idx0 += 256 * U) {
        float r[U], e[U][MAX_PREV];
        int fk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = min(idx0 + u * 256, frv * D - 1);          // clamped: the surplus lanes repeat the last element
            const int f = idx / D, k = idx - f * D;
            fk[u] = f * LD + pk_off(k);
            r[u] = p.x[(long)(m0 + f) * p.ldx + g * D + k];
#pragma unroll
            for (int sidx = 0; sidx < MAX_PREV; ++sidx)
                e[u][sidx] = sidx < step ? p.emb[((long)(p.group_begin[g] + sidx) * p.n_codes + prev[sidx * FR + f]) * D + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v = r[u];
#pragma unroll
            for (int sidx = 0; sidx < MAX_PREV; ++sidx)
                if (sidx < step) v -= e[u][sidx];
            if (idx0 + u * 256 < frv * D) r_pk[fk[u]] = v;
        }
    }
    __syncthreads();
    const int c0 = (blockIdx.x * 4 + wave) * 32;
    if (c0 >= p.n_codes) return;
    const float* packed = p.packed + (long)lvl * p.n_codes * D;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* ap = packed + (long)(c0 + j) * 8 + h * 4;
    const float* rrow = r_pk + j * LD + h * 4;
    // codebook operands: 16 loads in flight per lane (the plain loop waited out one memory latency per 16 bytes); the MFMA
    // chain still consumes k in ascending order, so the scores stay bit-identical to the fused kernel / rvq_ref.c
    constexpr int UN = 16;
    for (int kq0 = 0; kq0 < D / 8; kq0 += UN) {
        f32x4 a[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            a[u] = kq0 + u < D / 8 ? *reinterpret_cast<const f32x4*>(ap + (long)(kq0 + u) * p.n_codes * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (kq0 + u < D / 8) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(rrow + (kq0 + u) * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], bq[e], acc, 0, 0, 0);
            }
        }
    }
    float best = INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int code = c0 + rst_mfma32_row(r, lane);
        const float sc = fmaf(-2.0f, acc[r], p.e2[(long)lvl * p.n_codes + code]);
        if (sc < best) { best = sc; bidx = code; }
    }
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(bidx, 32);
    if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    if (h == 0 && m0 + j < p.M) atomicMin(keys + (long)lvl * p.M + m0 + j, rvq_key(best, bidx));
}

__global__ __launch_bounds__(256) void rvq_finalize_kernel(const RvqSearchParams p, unsigned long long* __restrict__ keys) {
    const long total = (long)p.L * p.M;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int lvl = (int)(idx / p.M);
        const int m = (int)(idx - (long)lvl * p.M);
        bool used = false;
        for (int g = 0; g < p.n_groups; ++g) used = used || (lvl >= p.group_begin[g] && lvl < p.group_begin[g] + p.group_count[g]);
        const unsigned long long kv = keys[idx];
        keys[idx] = ~0ull;                                   // re-arm for the next call
        if (!used) continue;
        const int b = m / p.F, f = m - b * p.F;
        int code = (int)(kv & 0xffffffffu);
        if (code < 0 || code >= p.n_codes) code = 0;        // only reachable with NaN inputs
        p.codes[((long)b * p.L + lvl) * p.F + f] = code;
        if (p.dist) {
            unsigned u = (unsigned)(kv >> 32);
            u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            p.dist[idx] = __uint_as_float(u);
        }
    }
}

// ---- few-frame path, ONE launch for all levels (round 5) ---------------------------------------------------------------------------
// The per-level launches above cost a kernel boundary per residual level (7 dependent launches of ~13 us per 80 ms frame, ~26 us each
// at 32 streams: every launch re-derives the residual from x with `step` gathers per element).  Here the workgroups of a (group, frame
// tile) -- one per 128-code slice, all resident: at most a few dozen of them -- stay in the launch and hand the level's decision over
// in-kernel, the data being the flag (persist.h's form): every workgroup publishes its slice's best (score, index) key per frame as ONE
// 8-byte relaxed agent-scope store into its own slot, sweeps the slots of all slices until none is empty (all ones: no key is), and takes
// the minimum itself -- lowest score, then lowest index, exactly the atomicMin's decision.  The residual stays in LDS and is updated
// with ONE gather per element and level, in level order (the same sequence of exact subtractions).  Scores come from the same
// k-ordered MFMA chain: codes and winning scores are bit-identical to the other two forms and to oracle/rvq_ref.c.
// Every spin is bounded by the wall clock; a timed-out workgroup ORs a code into status[0].  The finish launch behind it (one
// workgroup) re-arms the slots and, if status[0] is set, recomputes every (group, tile) alone -- all slices in a loop, minimum in
// LDS: it waits for nobody -- before it counts the repair in status[1] (ops.persistent_poll retires the path on a device that repairs).
constexpr long long RVQ_TIMEOUT_TICKS = 10000000;       // of the 100 MHz wall clock: 0.1 s
#define RVQ_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// best key of frame j (lanes j and j + 32 of the wave agree) over the wave's 32 codes c0 .. c0 + 31 of level `lvl`
__device__ __forceinline__ unsigned long long rvq_wave_best(const RvqSearchParams& p, const float* r_pk, int LD, int lvl, int c0, int lane) {
    const int D = p.D;
    const int j = lane & 31, h = lane >> 5;
    const float* packed = p.packed + (long)lvl * p.n_codes * D;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* ap = packed + (long)(c0 + j) * 8 + h * 4;
    const float* rrow = r_pk + j * LD + h * 4;
    constexpr int UN = 16;
    for (int kq0 = 0; kq0 < D / 8; kq0 += UN) {
        f32x4 a[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            a[u] = kq0 + u < D / 8 ? *reinterpret_cast<const f32x4*>(ap + (long)(kq0 + u) * p.n_codes * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (kq0 + u < D / 8) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(rrow + (kq0 + u) * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], bq[e], acc, 0, 0, 0);
            }
        }
    }
    float best = INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int code = c0 + rst_mfma32_row(r, lane);
        const float sc = fmaf(-2.0f, acc[r], p.e2[(long)lvl * p.n_codes + code]);
        if (sc < best) { best = sc; bidx = code; }
    }
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(bidx, 32);
    if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    return rvq_key(best, bidx);
}

// All levels of group g for the frame tile at m0.  SOLO: this workgroup takes every slice itself (the repair form); otherwise slice
// `slice` of `ns`, hand-offs through `slots` [L][Mpad][ns].
template <bool SOLO>
__device__ __forceinline__ void rvq_chain_body(const RvqSearchParams& p, unsigned long long* slots, unsigned* status, float* smem,
                                               const int g, const int m0, const int slice, const int ns, const int Mpad) {
    const int D = p.D, LD = D + 4;
    float* r_pk = smem;                                                        // [FR][LD]
    unsigned long long* wbest = reinterpret_cast<unsigned long long*>(r_pk + FR * LD);     // [4 waves][FR]
    unsigned long long* fbest = wbest + 4 * FR;                                // [FR] the level's decision
    int* dead = reinterpret_cast<int*>(fbest + FR);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frv = min(FR, p.M - m0);
    if (tid == 0) *dead = 0;
    for (int idx = tid; idx < FR * D; idx += 256) {
        const int f = idx / D, k = idx - f * D;
        r_pk[f * LD + pk_off(k)] = f < frv ? p.x[(long)(m0 + f) * p.ldx + g * D + k] : 0.f;
    }
    __syncthreads();
    for (int li = 0; li < p.group_count[g]; ++li) {
        const int lvl = p.group_begin[g] + li;
        // ---- this workgroup's best key per frame
        if (SOLO) {
            if (tid < FR) fbest[tid] = ~0ull;
            __syncthreads();
            for (int sl = 0; sl < ns; ++sl) {
                const int c0 = (sl * 4 + wave) * 32;
                unsigned long long k = ~0ull;
                if (c0 < p.n_codes) k = rvq_wave_best(p, r_pk, LD, lvl, c0, lane);
                if (lane < 32) wbest[wave * FR + lane] = k;
                __syncthreads();
                if (tid < FR) {
                    unsigned long long k4 = fbest[tid];
#pragma unroll
                    for (int w = 0; w < 4; ++w) k4 = wbest[w * FR + tid] < k4 ? wbest[w * FR + tid] : k4;
                    fbest[tid] = k4;
                }
                __syncthreads();
            }
        } else {
            const int c0 = (slice * 4 + wave) * 32;
            unsigned long long k = ~0ull;
            if (c0 < p.n_codes) k = rvq_wave_best(p, r_pk, LD, lvl, c0, lane);
            if (lane < 32) wbest[wave * FR + lane] = k;
            __syncthreads();
            unsigned long long* sl_base = slots + ((long)lvl * Mpad + m0) * ns;
            if (tid < FR) {
                unsigned long long k4 = wbest[tid];
#pragma unroll
                for (int w = 1; w < 4; ++w) k4 = wbest[w * FR + tid] < k4 ? wbest[w * FR + tid] : k4;
                __hip_atomic_store(sl_base + (long)tid * ns + slice, k4, RVQ_RLX);       // (a slice beyond the codebook cannot exist: ns = ceil(n_codes / 128))
            }
            // ---- the level's decision: minimum over the slices' slots.  Thread t: frame t / 8, slices (t % 8), (t % 8) + 8, ...
            const int f = tid >> 3, s0 = tid & 7;
            unsigned long long mine = ~0ull;
            long long t0 = 0;
            for (int sl = s0; sl < ns; sl += 8) {
                unsigned long long v;
                while (true) {
                    v = __hip_atomic_load(sl_base + (long)f * ns + sl, RVQ_RLX);
                    if (v != ~0ull) break;
                    if (t0 == 0) t0 = wall_clock64();
                    if (*(volatile int*)dead || wall_clock64() - t0 > RVQ_TIMEOUT_TICKS) {
                        *dead = 1;
                        atomicOr(status, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                mine = v < mine ? v : mine;
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                const unsigned long long other = __shfl_xor(mine, o);
                mine = other < mine ? other : mine;
            }
            if (s0 == 0) fbest[f] = mine;
            __syncthreads();
        }
        // ---- codes / winning scores (one writer per frame), then the residual of the next level: one gather per element
        if ((SOLO || slice == 0) && tid < frv) {
            const unsigned long long kv = fbest[tid];
            int code = (int)(kv & 0xffffffffu);
            if (code < 0 || code >= p.n_codes) code = 0;        // only reachable with NaN inputs
            const int m = m0 + tid;
            const int b = m / p.F, fr = m - b * p.F;
            p.codes[((long)b * p.L + lvl) * p.F + fr] = code;
            if (p.dist) {
                unsigned u = (unsigned)(kv >> 32);
                u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
                p.dist[(long)lvl * p.M + m] = __uint_as_float(u);
            }
        }
        if (li + 1 < p.group_count[g]) {
            const float* emb = p.emb + (long)lvl * p.n_codes * D;
            constexpr int U = 8;
            for (int idx0 = tid; idx0 < frv * D; idx0 += 256 * U) {
                float e[U];
                int fk[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = min(idx0 + u * 256, frv * D - 1);
                    const int f = idx / D, k = idx - f * D;
                    int code = (int)(fbest[f] & 0xffffffffu);
                    if (code < 0 || code >= p.n_codes) code = 0;
                    fk[u] = f * LD + pk_off(k);
                    e[u] = emb[(long)code * D + k];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (idx0 + u * 256 < frv * D) r_pk[fk[u]] -= e[u];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void rvq_chain_kernel(const RvqSearchParams p, unsigned long long* __restrict__ slots, unsigned* __restrict__ status,
                                                        int ns, int Mpad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < ns && p.group_count[blockIdx.y] > 0)
        rvq_chain_body<false>(p, slots, status, smem, blockIdx.y, blockIdx.z * FR, blockIdx.x, ns, Mpad);
}

// one workgroup behind the chain launch: repair (if any workgroup timed out) and re-arm the slots for the next call
__global__ __launch_bounds__(256) void rvq_chain_finish_kernel(const RvqSearchParams p, unsigned long long* __restrict__ slots, unsigned* __restrict__ status,
                                                               int ns, int Mpad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned code;
    if (threadIdx.x == 0) code = __hip_atomic_load(status, RVQ_RLX);
    __syncthreads();
    if (code != 0) {
        for (int g = 0; g < p.n_groups; ++g)
            for (int m0 = 0; m0 < p.M; m0 += FR) {
                rvq_chain_body<true>(p, slots, status, smem, g, m0, 0, ns, Mpad);
                __syncthreads();
            }
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(status + 1, 1u, RVQ_RLX);
            __hip_atomic_fetch_or(status + 2, code, RVQ_RLX);
            __hip_atomic_store(status, 0u, RVQ_RLX);
        }
    }
    const long total = (long)p.L * Mpad * ns;
    for (long i = threadIdx.x; i < total; i += 256) slots[i] = ~0ull;
}

__global__ __launch_bounds__(256) void rvq_gather_kernel(const RvqGatherParams p) {
    const int ND = p.n_groups * p.D;
    const long total = (long)p.M * ND;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int col = (int)(idx % ND);
        const long m = idx / ND;
        const int g = col / p.D, k = col - g * p.D;
        const long b = m / p.F, f = m - b * p.F;
        // ResidualVectorQuantization.decode (core_vq.py:378-384): 0 + q_0 + q_1 + ... in level order
        float acc = 0.f;
        for (int li = 0; li < p.group_count[g]; ++li) {
            const int lvl = p.group_begin[g] + li;
            long c = p.codes[(b * p.L + lvl) * p.F + f];
            c = c < 0 ? 0 : (c >= p.n_codes ? p.n_codes - 1 : c);
            acc = acc + p.emb[((long)lvl * p.n_codes + c) * p.D + k];
        }
        p.out[idx] = acc;
    }
}

}  // namespace

int rst_launch_rvq_pack(const float* emb, float* packed, float* e2, int n_codes, int D, hipStream_t stream) {
    RST_REQUIRE(emb && packed && e2 && n_codes > 0 && D > 0 && D % 8 == 0, "rvq_pack: bad arguments (D must be a multiple of 8)");
    long g = ((long)n_codes * D + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(rvq_pack_kernel, dim3((unsigned)g), dim3(256), 0, stream, emb, packed, e2, n_codes, D);
    return rst_check_launch("rvq_pack");
}

int rst_launch_rvq_search(const RvqSearchParams& p, hipStream_t stream) {
    if (p.M == 0) return RST_OK;
    RST_REQUIRE(p.x && p.emb && p.packed && p.e2 && p.codes, "rvq_search: null pointer");
    RST_REQUIRE(p.M >= 0 && p.F > 0 && p.D > 0 && p.D % 8 == 0 && p.n_codes > 0 && p.n_codes % 32 == 0,
                "rvq_search: need D %% 8 == 0 and n_codes %% 32 == 0 (D=%d n_codes=%d)", p.D, p.n_codes);
    RST_REQUIRE(p.n_groups >= 1 && p.n_groups <= 2, "rvq_search: n_groups must be 1 or 2");
    RST_REQUIRE(p.M % p.F == 0, "rvq_search: M (%d) is not a multiple of F (%d)", p.M, p.F);
    if (p.M == 0) return RST_OK;
    const size_t lds = ((size_t)FR * (p.D + 4) + p.n_codes + 2 * NW * FR + FR) * sizeof(float);
    RST_REQUIRE(lds <= 160 * 1024, "rvq_search: D=%d n_codes=%d needs %zu bytes of LDS", p.D, p.n_codes, lds);
    static RstOncePerDevice attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rvq_search_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rvq_search_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const dim3 grid((p.M + FR - 1) / FR, p.n_groups);
    if (p.D == 256) hipLaunchKernelGGL(rvq_search_kernel<32>, grid, dim3(64 * NW), lds, stream, p);     // the codec's codebooks
    else hipLaunchKernelGGL(rvq_search_kernel<0>, grid, dim3(64 * NW), lds, stream, p);
    return rst_check_launch("rvq_search");
}

int rst_launch_rvq_search_small(const RvqSearchParams& p, unsigned long long* keys, hipStream_t stream) {
    if (p.M == 0) return RST_OK;
    RST_REQUIRE(p.x && p.emb && p.packed && p.e2 && p.codes && keys, "rvq_search_small: null pointer");
    RST_REQUIRE(p.F > 0 && p.D > 0 && p.D % 8 == 0 && p.n_codes > 0 && p.n_codes % 32 == 0 && p.M % p.F == 0 &&
                    p.n_groups >= 1 && p.n_groups <= 2,
                "rvq_search_small: need D %% 8 == 0 and n_codes %% 32 == 0 (D=%d n_codes=%d)", p.D, p.n_codes);
    const size_t lds = ((size_t)FR * (p.D + 4) + 16 * FR) * sizeof(float);
    int steps = 0;
    for (int g = 0; g < p.n_groups; ++g) steps = p.group_count[g] > steps ? p.group_count[g] : steps;
    const dim3 grid((p.n_codes + 127) / 128, p.n_groups, (p.M + FR - 1) / FR);
    RST_REQUIRE(steps <= 16, "rvq_search_small: at most 16 levels per group (got %d)", steps);
    for (int s = 0; s < steps; ++s) {
        hipLaunchKernelGGL(rvq_level_kernel, grid, dim3(256), lds, stream, p, keys, s);
        const int rc = rst_check_launch("rvq_level");
        if (rc) return rc;
    }
    long g = ((long)p.L * p.M + 255) / 256;
    hipLaunchKernelGGL(rvq_finalize_kernel, dim3((unsigned)(g > 64 ? 64 : g)), dim3(256), 0, stream, p, keys);
    return rst_check_launch("rvq_finalize");
}

int rst_rvq_chain_slices(int n_codes) { return (n_codes + 127) / 128; }

// shapes the one-launch chain serves (the checks of the launcher below): the library's answer, so that callers pick the per-level launches
// instead of catching an error
int rst_rvq_chain_supported_impl(int M, int n_codes, int L, int D, int n_groups) {
    if (!(M > 0 && L > 0 && D > 0 && D % 8 == 0 && n_codes > 0 && n_codes % 32 == 0 && n_groups >= 1 && n_groups <= 2)) return 0;
    const int ns = rst_rvq_chain_slices(n_codes), mt = (M + FR - 1) / FR;
    if (ns > 64 || (long)ns * n_groups * mt > rst_cu_count()) return 0;
    const size_t lds = ((size_t)FR * (D + 4)) * sizeof(float) + (5 * FR) * sizeof(unsigned long long) + 16;
    return lds <= 64 * 1024 ? 1 : 0;
}

int rst_launch_rvq_search_chain(const RvqSearchParams& p, unsigned long long* slots, unsigned* status, hipStream_t stream) {
    if (p.M == 0) return RST_OK;
    RST_REQUIRE(p.x && p.emb && p.packed && p.e2 && p.codes && slots && status, "rvq_search_chain: null pointer");
    RST_REQUIRE(p.F > 0 && p.D > 0 && p.D % 8 == 0 && p.n_codes > 0 && p.n_codes % 32 == 0 && p.M % p.F == 0 &&
                    p.n_groups >= 1 && p.n_groups <= 2,
                "rvq_search_chain: need D %% 8 == 0 and n_codes %% 32 == 0 (D=%d n_codes=%d)", p.D, p.n_codes);
    const int ns = rst_rvq_chain_slices(p.n_codes);
    const int mt = (p.M + FR - 1) / FR;
    // every workgroup waits for its peers of the same (group, tile): they must all be resident -- a few dozen on an idle device
    RST_REQUIRE(ns <= 64 && (long)ns * p.n_groups * mt <= rst_cu_count(), "rvq_search_chain: %d x %d x %d workgroups exceed the CUs", ns, p.n_groups, mt);
    const size_t lds = ((size_t)FR * (p.D + 4)) * sizeof(float) + (5 * FR) * sizeof(unsigned long long) + 16;
    RST_REQUIRE(lds <= 64 * 1024, "rvq_search_chain: D=%d needs %zu bytes of LDS", p.D, lds);
    hipLaunchKernelGGL(rvq_chain_kernel, dim3(ns, p.n_groups, mt), dim3(256), lds, stream, p, slots, status, ns, mt * FR);
    int rc = rst_check_launch("rvq_chain");
    if (rc) return rc;
    hipLaunchKernelGGL(rvq_chain_finish_kernel, dim3(1), dim3(256), lds, stream, p, slots, status, ns, mt * FR);
    return rst_check_launch("rvq_chain_finish");
}

int rst_launch_rvq_gather(const RvqGatherParams& p, hipStream_t stream) {
    if (p.M == 0) return RST_OK;
    RST_REQUIRE(p.codes && p.emb && p.out, "rvq_gather: null pointer");
    RST_REQUIRE(p.M >= 0 && p.F > 0 && p.D > 0 && p.n_codes > 0 && p.n_groups >= 1 && p.n_groups <= 2 && p.M % p.F == 0,
                "rvq_gather: bad sizes");
    const long total = (long)p.M * p.n_groups * p.D;
    if (total == 0) return RST_OK;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rvq_gather_kernel, dim3((unsigned)g), dim3(256), 0, stream, p);
    return rst_check_launch("rvq_gather");
}
