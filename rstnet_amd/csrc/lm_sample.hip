// sample_token of MLLM_v2/utils/sampling.py on the device: greedy argmax or exact top-k + Exp(1) race, bit-identical to the
// oracle given the same noise (ties to the lowest index, plateaus, vocabularies up to 2^20, id blanking).
#include "lm_common.h"
#include "lm_sample_impl.h"

namespace {

template <int NT, int EPT>
__global__ __launch_bounds__(NT) void sample_kernel(const LmSampleParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long comp[];   // [k rounded up to 8] candidates
    __shared__ SampleShared<NT> sh;
    const long b = blockIdx.x;
    const int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    const int tok = sample_row<NT, EPT>(p.logits + b * p.ld, p.noise ? p.noise + b * p.noise_stride : nullptr, p.V, p.top_k,
                                        p.use_sampling && p.temp > 0.f, p.temp, limit, comp, sh);
    if (threadIdx.x == 0) p.tokens[b * p.tok_stride] = tok;
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
    // fn(i, key) over this thread's ids i = tid, tid + NT, ... < V in ascending order, eight loads in flight at a time (index
    // clamped, the keys past the end cleared to 0 -- below every real key -- through a mask the compiler cannot turn back into a
    // condition on the load).  One load per loop iteration is one exposed L2 round trip per id: 148 of them per pass at V = 151 936.
    auto scan = [&](auto fn) {
        for (int i0 = tid; i0 < V; i0 += NT * 8) {
            float f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) f[u] = lg[min(i0 + u * NT, V - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NT;
                unsigned msk = i < V ? 0xffffffffu : 0u;
                asm volatile("" : "+v"(msk));
                fn(i, to_key(sampling ? f[u] / p.temp : f[u]) & msk);
            }
        }
    };
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
    scan([&](int i, unsigned u) {
        if (u > bk) { bk = u; bi = i; }
        if (i < limit && u > tk) tk = u;
    });
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
    scan([&](int, unsigned u) { s += u ? expf(from_key(u) - mx) : 0.f; });
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
        scan([&](int i, unsigned u) {
            if (take(u, i)) {
                const int at = atomicAdd(&n_cand, 1);
                if (at < SAMPLE_BIG_CAP) comp[at] = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - i);
            }
        });
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
