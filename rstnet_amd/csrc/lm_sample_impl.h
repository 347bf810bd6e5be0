// sample_token of MLLM_v2/utils/sampling.py for one logits row, as a device function shared by the stand-alone sampling
// kernel (lm_sample.hip) and the persistent depth-frame kernel (lm_depth.hip).
#pragma once
#include "lm_common.h"

namespace {

// One workgroup per batch row (256 threads for V <= 4096, 1024 above).  Greedy: argmax (lowest index on ties).
// Sampling (utils/sampling.py:51-105): probs = softmax(logits / temp); (p, idx) = top-k in descending order (ties: lowest
// index first); token = idx[argmax_j p_j / noise_j].  Exact top-k WITHOUT sorting, register resident: each thread keeps
// EPT order-preserving uint keys of the scaled logits; a bracketing search on the key (block-wide counts of the keys at or above a
// probe; probes alternate between interpolation on the counts and bisection, both in VALUE space between the row's minimum and
// maximum, and stop as soon as exactly k keys lie above the probe: ~8 counts for a 2048-way row where the bit-by-bit descent of
// rounds 1-3 took up to 32) finds the k-th largest key, ties at that value are resolved by a search
// on the index, the exactly-k candidates are compacted into LDS as 64-bit (key, ~index) composites and each computes its
// rank by counting the larger composites.
// LDS scratch of one sample_row call (besides the candidate list `comp`, [top_k rounded up to 8] 64-bit words)
template <int NT> struct SampleShared {
    float red_v[NT / 64];
    int red_i[NT / 64], red_j[NT / 64];
    int cnt[2 * (NT / 64)];
    int n_cand;
};

// What a SELECT call (first level of the two-level sampler for large vocabularies, lm_sample.hip) reports about its chunk of a row.
struct SampleChunk {
    unsigned max_key;       // order-preserving key of the chunk's largest (scaled) logit
    int max_idx;            // its lowest id
    float sum_exp;          // sum over the chunk of exp(logit - chunk max)
    int n_cand;             // candidates written to comp (the chunk's top-k among the ids that may be drawn)
};

// One logits row `lg[0..V)` (global or LDS), all NT threads of the workgroup: returns the token (valid in thread 0).
// SELECT: `lg[0..V)` is a CHUNK of a row whose first id is `id0`; the call stops after the exact top-k selection -- the candidates
// (composites carrying their GLOBAL ids) are left in comp[0 .. n_cand) and `*chunk` is filled (valid in every thread).
template <int NT, int EPT, bool SELECT = false>
__device__ __forceinline__ int sample_row(const float* lg, const float* noise_row, int V, int top_k, bool sampling, float temp, int limit_in,
                                          unsigned long long* comp, SampleShared<NT>& sh, int id0 = 0, SampleChunk* chunk = nullptr,
                                          unsigned long long* dbg = nullptr) {
#define SAMPLE_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[(i)] = wall_clock64(); } while (0)
    constexpr int NW = NT / 64;
    float (&red_v)[NW] = sh.red_v;
    int (&red_i)[NW] = sh.red_i;
    int (&red_j)[NW] = sh.red_j;
    int (&cnt)[2 * NW] = sh.cnt;
    int& n_cand = sh.n_cand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    auto to_key = [](float f) { unsigned u = __float_as_uint(f + 0.0f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto from_key = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    unsigned key[EPT];
    int slot = 0;                                  // block-wide counts alternate between two rows of per-wave LDS cells
    // number of elements in the block satisfying pred(j) (j = the thread's element slot): ballots + scalar popcounts per
    // wave, one LDS cell per wave, one barrier
    auto block_count = [&](auto pred) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < EPT; ++j) c += __popcll(__ballot(pred(j)));
        // two rows of cells, alternating: a row is rewritten two counts later, behind the barrier of the count in between
        if (lane == 0) cnt[(slot & 1) * NW + wave] = c;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += cnt[(slot & 1) * NW + w];
        ++slot;
        return t;
    };

    if (tid == 0) n_cand = 0;
    // keys of the (scaled) logits, element j of this thread is index j * NT + tid; key 0 (below every real key) pads the tail
    unsigned bk = 0u, mk = 0xffffffffu;            // largest key (and its lowest index) | smallest key of a real element
    int bi = 0x7fffffff;
    // (all EPT loads are unconditional -- index clamped, tail keys cleared through a mask the compiler cannot fold back into a
    // condition: a load under `i < V` is branched around and waited for on the spot, EPT exposed round trips per row)
    float fl[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) fl[j] = lg[min(j * NT + tid, V - 1)];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = j * NT + tid;
        unsigned msk = i < V ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(msk));
        float f = fl[j];
        if (sampling) f = f / temp;
        key[j] = to_key(f) & msk;
        if (key[j] > bk) { bk = key[j]; bi = i; }          // ascending i: the first maximum is kept
        mk = key[j] && key[j] < mk ? key[j] : mk;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned ok = __shfl_xor(bk, o);
        const int oi = __shfl_xor(bi, o);
        const unsigned om = __shfl_xor(mk, o);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
        mk = om < mk ? om : mk;
    }
    if (lane == 0) { red_j[wave] = (int)bk; red_i[wave] = bi; red_v[wave] = __uint_as_float(mk); }
    __syncthreads();
    bk = (unsigned)red_j[0]; bi = red_i[0]; mk = __float_as_uint(red_v[0]);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        if ((unsigned)red_j[w] > bk || ((unsigned)red_j[w] == bk && red_i[w] < bi)) { bk = (unsigned)red_j[w]; bi = red_i[w]; }
        mk = __float_as_uint(red_v[w]) < mk ? __float_as_uint(red_v[w]) : mk;
    }
    if (SELECT) { chunk->max_key = bk; chunk->max_idx = id0 + bi; chunk->sum_exp = 0.f; chunk->n_cand = 0; }
    SAMPLE_STAMP(0);
    if (!sampling) return bi;
    // softmax denominator (fp32, max-subtracted like torch.softmax)
    const float mx = from_key(bk);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) s += key[j] ? expf(from_key(key[j]) - mx) : 0.f;
    s = wave_sum(s);
    __syncthreads();                 // red_v still holds the minimum keys of the reduction above
    if (lane == 0) red_v[wave] = s;
    __syncthreads();
    float denom = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) denom += red_v[w];

    SAMPLE_STAMP(1);
    // id blanking of sample_token_audio / sample_token_audio_2048 (utils/sampling.py:107-158): the probabilities of ids >= limit
    // are overwritten after the softmax over ALL ids, so the denominator above is untouched and the ids just leave the race
    if (SELECT) chunk->sum_exp = denom;
    // (SELECT: `limit_in` is the row's limit minus id0, already clamped to [0, V] by the caller; 0 = no id of this chunk may be drawn)
    const int limit = SELECT ? limit_in : (limit_in > 0 && limit_in < V ? limit_in : V);
    if (SELECT && limit <= 0) return 0;
    if (limit < V) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) key[j] = j * NT + tid < limit ? key[j] : 0u;
    }
    // k-th largest key = the largest probe t with count(key >= t) >= k.  Bracket [lo, hi): count(>= lo) >= k > count(>= hi), from
    // [row minimum, row maximum + 1]; stop as soon as a probe isolates exactly k keys, or when the bracket is one key wide (ties)
    const int k = min(top_k > 0 ? top_k : V, limit);
    unsigned thr = 0u;
    bool exact = false;
    int clo = block_count([&](int j) { return key[j] >= mk; });
    if (clo >= k && mk <= bk && bk != 0xffffffffu) {
        unsigned lo = mk;
        unsigned long long hi = (unsigned long long)bk + 1ull;
        int chi = 0;
        float vlo = from_key(mk), vhi = from_key(bk);
        exact = clo == k;
#pragma unroll 1
        for (int round = 0; !exact && hi - lo > 1ull; ++round) {
            // value-space probes (interpolation on the counts / midpoint, alternating) for the first eight rounds only: a row with a
            // huge dynamic range (ids masked to -3.4e38, k close to the live keys) lets them creep; from then on the middle KEY,
            // which halves a bracket of at most 2^32 keys per round -- the loop is bounded by 8 + 32 block-wide counts
            float v = (round & 1) ? 0.5f * (vlo + vhi) : vlo + (vhi - vlo) * (((float)(clo - k) + 0.5f) / (float)(clo - chi));
            unsigned t = to_key(v);
            if (round >= 8 || !(t > lo && (unsigned long long)t < hi)) {   // (or no progress in value space / a non-finite end)
                t = lo + (unsigned)((hi - lo) >> 1);
                v = from_key(t);
            }
            const int n = block_count([&](int j) { return key[j] >= t; });
            if (n >= k) { lo = t; clo = n; vlo = v; } else { hi = t; chi = n; vhi = v; }
            exact = n == k;
        }
        thr = lo;
    } else {
        // fewer than k real keys (or a NaN maximum): the bit-by-bit descent of the earlier rounds, which takes what there is
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = thr | (1u << bit);
            const int n = block_count([&](int j) { return key[j] >= cand; });
            if (n >= k) thr = cand;
            if (n == k) { exact = true; break; }
        }
    }
    SAMPLE_STAMP(2);
    int idx_lim = 0x7fffffff;            // ties (key == thr) with index <= idx_lim are taken
    if (!exact) {
        const int n_gt = block_count([&](int j) { return key[j] > thr; });
        const int n_eq = block_count([&](int j) { return key[j] == thr; });
        // of the n_eq elements equal to the threshold only the need = k - n_gt with the LOWEST indices belong to the top-k
        const int need = k - n_gt;
        if (need < n_eq) {
            int lim = 0;                 // largest L with count(ties, idx < L) < need, built bit by bit
            for (int bit = 16; bit >= 0; --bit) {
                const int cand = lim | (1 << bit);
                if (block_count([&](int j) { return key[j] == thr && j * NT + tid < cand; }) < need) lim = cand;
            }
            idx_lim = lim;
        }
    }
    SAMPLE_STAMP(3);
    // compact the exactly-k candidates (any order: ranks come from comparisons)
    const int kpad = (k + 7) & ~7;
    for (int i = k + tid; i < kpad; i += NT) comp[i] = 0ull;
    int wave_total = 0;
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        wave_total += __popcll(__ballot(key[j] > thr || (key[j] == thr && j * NT + tid <= idx_lim)));
    int base = 0;
    if (lane == 0) base = atomicAdd(&n_cand, wave_total);
    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const bool take = key[j] > thr || (key[j] == thr && j * NT + tid <= idx_lim);
        const unsigned long long mk = __ballot(take);
        const int at = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (take && at < k) comp[at] = ((unsigned long long)key[j] << 32) | (unsigned)(0x7fffffff - (id0 + j * NT + tid));
        base += __popcll(mk);
    }
    __syncthreads();
    if (SELECT) { chunk->n_cand = k; return 0; }
    SAMPLE_STAMP(4);
    float win = -INFINITY;
    int win_rank = 0x7fffffff, win_tok = 0;
    for (int c = tid; c < k; c += NT) {
        const unsigned long long mine = comp[c];
        int rank = 0;
        for (int j0 = 0; j0 < kpad; j0 += 8) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = comp[j0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += v[u] > mine ? 1 : 0;
        }
        const float sc = (expf(from_key((unsigned)(mine >> 32)) - mx) / denom) / noise_row[rank];
        if (sc > win || (sc == win && rank < win_rank)) { win = sc; win_rank = rank; win_tok = 0x7fffffff - (int)(unsigned)mine; }
    }
    SAMPLE_STAMP(5);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(win, o);
        const int orank = __shfl_xor(win_rank, o);
        const int ot = __shfl_xor(win_tok, o);
        if (ov > win || (ov == win && orank < win_rank)) { win = ov; win_rank = orank; win_tok = ot; }
    }
    __syncthreads();
    if (lane == 0) { red_v[wave] = win; red_i[wave] = win_rank; red_j[wave] = win_tok; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w)
            if (red_v[w] > win || (red_v[w] == win && red_i[w] < win_rank)) { win = red_v[w]; win_rank = red_i[w]; win_tok = red_j[w]; }
    }
    SAMPLE_STAMP(6);
#undef SAMPLE_STAMP
    return win_tok;
}


}  // namespace
