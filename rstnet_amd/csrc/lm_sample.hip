// sample_token of MLLM_v2/utils/sampling.py on the device: greedy argmax or exact top-k + Exp(1) race, bit-identical to the
// oracle given the same noise (ties to the lowest index, plateaus, vocabularies up to 2^20, id blanking).
#include "lm_common.h"

namespace {

// One workgroup per batch row (256 threads for V <= 4096, 1024 above).  Greedy: argmax (lowest index on ties).
// Sampling (utils/sampling.py:51-105): probs = softmax(logits / temp); (p, idx) = top-k in descending order (ties: lowest
// index first); token = idx[argmax_j p_j / noise_j].  Exact top-k WITHOUT sorting, register resident: each thread keeps
// EPT order-preserving uint keys of the scaled logits; a bit-wise binary search on the key (block-wide counts, early exit
// as soon as exactly k keys lie above the probe) finds the k-th largest key, ties at that value are resolved by a search
// on the index, the exactly-k candidates are compacted into LDS as 64-bit (key, ~index) composites and each computes its
// rank by counting the larger composites.
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void sample_kernel(const LmSampleParams p) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned long long comp[];   // [k rounded up to 8] candidates
    __shared__ float red_v[NW];
    __shared__ int red_i[NW], red_j[NW];
    __shared__ int cnt[52 * NW];
    __shared__ int n_cand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    const float* lg = p.logits + b * p.ld;
    const bool sampling = p.use_sampling && p.temp > 0.f;
    const int V = p.V;

    auto to_key = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto from_key = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    unsigned key[EPT];
    int slot = 0;                                  // every block-wide count uses a fresh row of per-wave LDS cells
    // number of elements in the block satisfying pred(j) (j = the thread's element slot): ballots + scalar popcounts per
    // wave, one LDS cell per wave, one barrier
    auto block_count = [&](auto pred) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < EPT; ++j) c += __popcll(__ballot(pred(j)));
        if (lane == 0) cnt[slot * NW + wave] = c;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += cnt[slot * NW + w];
        ++slot;
        return t;
    };

    if (tid == 0) n_cand = 0;
    // keys of the (scaled) logits, element j of this thread is index j * NT + tid; key 0 (below every real key) pads the tail
    unsigned bk = 0u;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = j * NT + tid;
        float f = i < V ? lg[i] : 0.f;
        if (sampling) f = f / p.temp;
        key[j] = i < V ? to_key(f) : 0u;
        if (key[j] > bk) { bk = key[j]; bi = i; }          // ascending i: the first maximum is kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned ok = __shfl_xor(bk, o);
        const int oi = __shfl_xor(bi, o);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
    }
    if (lane == 0) { red_j[wave] = (int)bk; red_i[wave] = bi; }
    __syncthreads();
    bk = (unsigned)red_j[0]; bi = red_i[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if ((unsigned)red_j[w] > bk || ((unsigned)red_j[w] == bk && red_i[w] < bi)) { bk = (unsigned)red_j[w]; bi = red_i[w]; }
    if (!sampling) {
        if (tid == 0) p.tokens[b * p.tok_stride] = bi;
        return;
    }
    // softmax denominator (fp32, max-subtracted like torch.softmax)
    const float mx = from_key(bk);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) s += key[j] ? expf(from_key(key[j]) - mx) : 0.f;
    s = wave_sum(s);
    if (lane == 0) red_v[wave] = s;
    __syncthreads();
    float denom = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) denom += red_v[w];

    // id blanking of sample_token_audio / sample_token_audio_2048 (utils/sampling.py:107-158): the probabilities of ids >= limit
    // are overwritten after the softmax over ALL ids, so the denominator above is untouched and the ids just leave the race
    int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    limit = limit > 0 && limit < V ? limit : V;
    if (limit < V) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) key[j] = j * NT + tid < limit ? key[j] : 0u;
    }
    // k-th largest key: binary search from the top bit down; stop as soon as a probe isolates exactly k keys
    const int k = min(p.top_k > 0 ? p.top_k : V, limit);
    unsigned thr = 0u;
    bool exact = false;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = thr | (1u << bit);
        const int n = block_count([&](int j) { return key[j] >= cand; });
        if (n >= k) thr = cand;
        if (n == k) { exact = true; break; }
    }
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
        if (take && at < k) comp[at] = ((unsigned long long)key[j] << 32) | (unsigned)(0x7fffffff - (j * NT + tid));
        base += __popcll(mk);
    }
    __syncthreads();
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
        const float sc = (expf(from_key((unsigned)(mine >> 32)) - mx) / denom) / p.noise[b * p.noise_stride + rank];
        if (sc > win || (sc == win && rank < win_rank)) { win = sc; win_rank = rank; win_tok = 0x7fffffff - (int)(unsigned)mine; }
    }
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
        p.tokens[b * p.tok_stride] = win_tok;
    }
}

// Large vocabularies (V > 32768, e.g. the 151 936-entry Qwen head): the keys no longer fit in registers, so the row (which
// is L2-resident: 0.6 MB) is read three times -- (1) arg-max + per-thread maxima, (2) softmax denominator, (3) candidate
// collection -- and the exact top-k is taken over a short candidate list:
//   the k-th largest of the 1024 per-thread maxima is a LOWER bound of the k-th largest key (k distinct elements reach it),
//   so {key >= that bound} contains the top-k and is typically only a little larger than k.
// Every candidate then computes its exact rank (count of larger (key, ~index) composites); ranks < k are the sorted top-k.
// If the candidate list overflows (plateaus of equal logits) the predicate is replaced by the exact one, found by bit-wise
// searches with counting passes over the row (always correct, slower).
constexpr int SAMPLE_BIG_CAP = 4096;
__global__ __launch_bounds__(1024) void sample_big_kernel(const LmSampleParams p) {
    constexpr int NT = 1024, NW = 16;
    __shared__ unsigned long long comp[SAMPLE_BIG_CAP];
    __shared__ float red_v[NW];
    __shared__ int red_i[NW], red_j[NW];
    __shared__ int cnt[96 * NW];
    __shared__ int n_cand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    const float* lg = p.logits + b * p.ld;
    const bool sampling = p.use_sampling && p.temp > 0.f;
    const int V = p.V;
    auto to_key = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto from_key = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    auto key_at = [&](int i) { return to_key(sampling ? lg[i] / p.temp : lg[i]); };
    int slot = 0;
    auto block_count = [&](int c) {            // sum of a per-thread count over the block (fresh LDS row per call)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) cnt[slot * NW + wave] = c;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += cnt[slot * NW + w];
        ++slot;
        return t;
    };
    int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    limit = limit > 0 && limit < V ? limit : V;
    if (tid == 0) n_cand = 0;

    // (1) arg-max over all ids (lowest index on ties) + this thread's maximum over the ids that may be drawn
    unsigned bk = 0u, tk = 0u;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += NT) {
        const unsigned u = key_at(i);
        if (u > bk) { bk = u; bi = i; }
        if (i < limit && u > tk) tk = u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned ok = __shfl_xor(bk, o);
        const int oi = __shfl_xor(bi, o);
        if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
    }
    if (lane == 0) { red_j[wave] = (int)bk; red_i[wave] = bi; }
    __syncthreads();
    bk = (unsigned)red_j[0]; bi = red_i[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if ((unsigned)red_j[w] > bk || ((unsigned)red_j[w] == bk && red_i[w] < bi)) { bk = (unsigned)red_j[w]; bi = red_i[w]; }
    if (!sampling) {
        if (tid == 0) p.tokens[b * p.tok_stride] = bi;
        return;
    }
    // (2) softmax denominator over all ids
    const float mx = from_key(bk);
    float s = 0.f;
    for (int i = tid; i < V; i += NT) s += expf(from_key(key_at(i)) - mx);
    s = wave_sum(s);
    if (lane == 0) red_v[wave] = s;
    __syncthreads();
    float denom = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) denom += red_v[w];

    // lower bound of the k-th largest key: the k-th largest per-thread maximum
    const int k = min(min(p.top_k > 0 ? p.top_k : V, limit), NT);
    unsigned thr = 0u;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = thr | (1u << bit);
        if (block_count(tk >= cand ? 1 : 0) >= k) thr = cand;
    }
    int idx_lim = 0x7fffffff;          // ids equal to thr are taken up to this index
    bool strict_only = false;          // exact predicate: key > thr, or key == thr && i <= idx_lim
    auto take = [&](unsigned u, int i) { return i < limit && u != 0u && (u > thr || (u == thr && i <= idx_lim)); };
    // (3) collect
    auto collect = [&]() {
        for (int i = tid; i < V; i += NT) {
            const unsigned u = key_at(i);
            if (take(u, i)) {
                const int at = atomicAdd(&n_cand, 1);
                if (at < SAMPLE_BIG_CAP) comp[at] = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - i);
            }
        }
        __syncthreads();
        return n_cand;
    };
    int nc = thr != 0u ? collect() : SAMPLE_BIG_CAP + 1;
    if (nc > SAMPLE_BIG_CAP) {
        // exact k-th key by counting passes over the row, then the index bound among its ties
        __syncthreads();
        if (tid == 0) n_cand = 0;
        thr = 0u;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = thr | (1u << bit);
            int c = 0;
            for (int i = tid; i < limit; i += NT) c += key_at(i) >= cand ? 1 : 0;
            if (block_count(c) >= k) thr = cand;
        }
        int c_gt = 0;
        for (int i = tid; i < limit; i += NT) c_gt += key_at(i) > thr ? 1 : 0;
        const int need = k - block_count(c_gt);
        int lim = 0;                   // largest L with count(ties, idx < L) < need
#pragma unroll 1
        for (int bit = 20; bit >= 0; --bit) {
            const int cand = lim | (1 << bit);
            int c = 0;
            for (int i = tid; i < limit && i < cand; i += NT) c += key_at(i) == thr ? 1 : 0;
            if (block_count(c) < need) lim = cand;
        }
        idx_lim = lim;
        (void)strict_only;
        nc = collect();                // exactly k <= 1024 candidates
    }
    // exact ranks among the candidates; ranks < k are the sorted top-k
    float win = -INFINITY;
    int win_rank = 0x7fffffff, win_tok = 0;
    for (int c = tid; c < nc; c += NT) {
        const unsigned long long mine = comp[c];
        int rank = 0;
        for (int j = 0; j < nc; ++j) rank += comp[j] > mine ? 1 : 0;
        if (rank < k) {
            const float sc = (expf(from_key((unsigned)(mine >> 32)) - mx) / denom) / p.noise[b * p.noise_stride + rank];
            if (sc > win || (sc == win && rank < win_rank)) { win = sc; win_rank = rank; win_tok = 0x7fffffff - (int)(unsigned)mine; }
        }
    }
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
        p.tokens[b * p.tok_stride] = win_tok;
    }
}

}  // namespace

int rst_launch_lm_sample(const LmSampleParams& p, hipStream_t stream) {
    RST_REQUIRE(p.logits && p.tokens && p.B >= 1 && p.V > 0 && p.V <= (1 << 20), "lm_sample: bad arguments");
    RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || p.noise, "lm_sample: sampling needs the exponential noise tensor");
    const int k = p.top_k > 0 && p.top_k < p.V ? p.top_k : p.V;
    RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || k <= 8192, "lm_sample: top-k %d exceeds the 8192 candidate stage", k);
    const size_t lds = (size_t)((k + 7) & ~7) * 8;
    if (p.V <= 2048) hipLaunchKernelGGL((sample_kernel<256, 8>), dim3(p.B), dim3(256), lds, stream, p);
    else if (p.V <= 4096) hipLaunchKernelGGL((sample_kernel<256, 16>), dim3(p.B), dim3(256), lds, stream, p);
    else if (p.V <= 32768) hipLaunchKernelGGL((sample_kernel<1024, 32>), dim3(p.B), dim3(1024), lds, stream, p);
    else {
        RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || k <= 1024, "lm_sample: top-k %d > 1024 for a vocabulary of %d", k, p.V);
        hipLaunchKernelGGL(sample_big_kernel, dim3(p.B), dim3(1024), 0, stream, p);
    }
    return rst_check_launch("lm_sample");
}
