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
    auto to_key = [](float f) { unsigned u = __float_as_uint(f + 0.0f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
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


// ---- two-level form for large vocabularies (round 3) ---------------------------------------------------------------------------------
// sample_big_kernel walks the 0.6 MB row of a 151 936-entry head three times with ONE workgroup per row (119 us at batch 32: 32
// workgroups on a 256-CU chip).  Here level 1 cuts every row into chunks of <= 10 240 ids, one 256-thread workgroup each (15 x 32 =
// 480 workgroups at batch 32): the chunk's keys live in registers (sample_row<256, 40, SELECT>), it reports its maximum, its
// exp-sum relative to that maximum and its exact top-k among the ids that may be drawn; level 2 (one workgroup per row) folds the
// chunk maxima / exp-sums into the row's softmax terms and takes the exact top-k of the <= chunks x k candidates by rank counting --
// the union of the chunks' top-k contains the row's, and (key, ~id) composites order ties towards the lowest id exactly as the
// one-level kernels do.  Same tokens as sample_big_kernel: selection is exact on both paths, the race terms use the same formula (the
// exp-sum is accumulated chunk by chunk instead of thread by thread -- a common positive divisor of every race term).
constexpr int SPLIT_NT = 256, SPLIT_EPT = 40, SPLIT_CHUNK = SPLIT_NT * SPLIT_EPT, SPLIT_CAP = 4096;
struct SplitRec { unsigned max_key; int max_idx; float sum_exp; int n_cand; };

__global__ __launch_bounds__(SPLIT_NT) void sample_split_kernel(const LmSampleParams p, int chunks, int kc) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long comp[];   // [kc rounded up to 8]
    __shared__ SampleShared<SPLIT_NT> sh;
    const int c = blockIdx.x, tid = threadIdx.x;
    const long b = blockIdx.y;
    const int id0 = c * SPLIT_CHUNK, Vc = min(SPLIT_CHUNK, p.V - id0);
    int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    limit = limit > 0 && limit < p.V ? limit : p.V;
    SampleChunk ch;
    sample_row<SPLIT_NT, SPLIT_EPT, true>(p.logits + b * p.ld + id0, nullptr, Vc, kc, p.use_sampling && p.temp > 0.f, p.temp,
                                          max(0, min(limit - id0, Vc)), comp, sh, id0, &ch);
    SplitRec* rec = reinterpret_cast<SplitRec*>(p.ws) + b * chunks + c;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(reinterpret_cast<SplitRec*>(p.ws) + (long)p.B * chunks) + (b * chunks + c) * kc;
    if (tid == 0) *rec = SplitRec{ch.max_key, ch.max_idx, ch.sum_exp, ch.n_cand};
    for (int i = tid; i < kc; i += SPLIT_NT) cand[i] = i < ch.n_cand ? comp[i] : 0ull;
}

__global__ __launch_bounds__(256) void sample_merge_kernel(const LmSampleParams p, int chunks, int kc) {
    constexpr int NT = 256, NW = 4;
    __shared__ unsigned long long comp[SPLIT_CAP];
    __shared__ float red_v[NW];
    __shared__ int red_i[NW], red_j[NW];
    __shared__ float s_mx, s_denom;
    __shared__ int s_tok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    const bool sampling = p.use_sampling && p.temp > 0.f;
    auto from_key = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    const SplitRec* rec = reinterpret_cast<const SplitRec*>(p.ws) + b * chunks;
    const unsigned long long* cand = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const SplitRec*>(p.ws) + (long)p.B * chunks) + b * chunks * kc;
    const int n = chunks * kc;
    if (sampling)
        for (int i = tid; i < n; i += NT) comp[i] = cand[i];
    // the row's maximum (lowest id on ties: chunks are in id order and each reports its lowest) and softmax denominator
    if (wave == 0) {
        unsigned bk = 0u;
        int bi = 0x7fffffff;
        for (int c = lane; c < chunks; c += 64) {
            const SplitRec r = rec[c];
            if (r.max_key > bk || (r.max_key == bk && r.max_idx < bi)) { bk = r.max_key; bi = r.max_idx; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned ok = __shfl_xor(bk, o);
            const int oi = __shfl_xor(bi, o);
            if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
        }
        const float mx = from_key(bk);
        float d = 0.f;
        if (sampling)
            for (int c = lane; c < chunks; c += 64) { const SplitRec r = rec[c]; d += r.sum_exp * expf(from_key(r.max_key) - mx); }
        d = wave_sum(d);
        if (lane == 0) { s_mx = mx; s_denom = d; s_tok = bi; }
    }
    __syncthreads();
    if (!sampling) {
        if (tid == 0) p.tokens[b * p.tok_stride] = s_tok;
        return;
    }
    const float mx = s_mx, denom = s_denom;
    int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    limit = limit > 0 && limit < p.V ? limit : p.V;
    const int k = min(p.top_k > 0 ? p.top_k : p.V, limit);
    float win = -INFINITY;
    int win_rank = 0x7fffffff, win_tok = 0;
    for (int c = tid; c < n; c += NT) {
        const unsigned long long mine = comp[c];
        if ((unsigned)(mine >> 32) == 0u) continue;          // an empty slot of a chunk with fewer candidates
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += comp[j] > mine ? 1 : 0;
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
    if (lane == 0) { red_v[wave] = win; red_i[wave] = win_rank; red_j[wave] = win_tok; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w)
            if (red_v[w] > win || (red_v[w] == win && red_i[w] < win_rank)) { win = red_v[w]; win_rank = red_i[w]; win_tok = red_j[w]; }
        p.tokens[b * p.tok_stride] = win_tok;
    }
}

// ---- nucleus sampling: sample_top_p (utils/sampling.py:66-82) -----------------------------------------------------------------------
// probs = softmax(logits / temp) sorted in descending order (ties: lowest id first); entry j survives while the EXCLUSIVE prefix sum
// of the sorted probabilities is <= top_p; the survivors are renormalised and the token is idx[argmax_j q_j / noise_j] -- `noise` holds
// one Exp(1) draw per SORTED position, V per row (multinomial draws a full-width noise tensor, :44-46).  One workgroup per row: keys ->
// (key, ~id) composites in a global scratch of Vpad = 2^ceil(log2 V) words, bitonic sort there (the row stays in L2), prefix sums in
// double precision (torch's CPU cumsum accumulates float rows in double) rounded to fp32 per element, survivors race.  The
// renormalising division is by a common positive number and cannot change the winner; it is applied all the same.  Ids >= limit
// (the blanking of the audio samplers) are excluded from the nucleus -- the reference writes -inf into `probs` there, which turns
// its own cumsum into NaN; no caller combines the two.  Not a hot path: no reference caller passes top_p (a 151 936-entry row costs
// ~171 sort passes of 2 MB).
__global__ __launch_bounds__(1024) void sample_top_p_kernel(const LmSampleParams p, int vpad) {
    constexpr int NT = 1024, NW = 16;
    __shared__ float red_v[NW];
    __shared__ int red_i[NW], red_j[NW];
    __shared__ double seg[NT];
    __shared__ double s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    const float* lg = p.logits + b * p.ld;
    const int V = p.V;
    unsigned long long* a = reinterpret_cast<unsigned long long*>(p.ws) + b * vpad;
    auto to_key = [](float f) { unsigned u = __float_as_uint(f + 0.0f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto from_key = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
    int limit = p.v_limit_dev ? *p.v_limit_dev : p.v_limit;
    limit = limit > 0 && limit < V ? limit : V;
    // (1) keys, row maximum, composites (ids that may not be drawn and the padding sort last: composite 0)
    unsigned bk = 0u;
    for (int i = tid; i < vpad; i += NT) {
        unsigned long long cpos = 0ull;
        if (i < V) {
            const unsigned u = to_key(lg[i] / p.temp);
            bk = max(bk, u);
            if (i < limit) cpos = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - i);
        }
        a[i] = cpos;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bk = max(bk, (unsigned)__shfl_xor((int)bk, o));
    if (lane == 0) red_j[wave] = (int)bk;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) bk = max(bk, (unsigned)red_j[w]);
    const float mx = from_key(bk);
    // (2) softmax denominator over ALL ids
    float s = 0.f;
    for (int i = tid; i < V; i += NT) s += expf(lg[i] / p.temp - mx);
    s = wave_sum(s);
    if (lane == 0) red_v[wave] = s;
    __syncthreads();
    float denom = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) denom += red_v[w];
    // (3) bitonic sort, descending
    for (int kk = 2; kk <= vpad; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < vpad / 2; t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;      // the pair (i, i + j), bit j of i clear
                const unsigned long long x = a[i], y = a[l];
                const bool desc = (i & kk) == 0;
                if (desc ? x < y : x > y) { a[i] = y; a[l] = x; }
            }
        }
    __syncthreads();
    // (4) exclusive prefix sums of the sorted probabilities: per-thread segments, then a scan of the segment totals
    const int per = (vpad + NT - 1) / NT, i0 = tid * per, i1 = min(vpad, i0 + per);
    auto prob = [&](unsigned u) { return expf(from_key(u) - mx) / denom; };
    double local = 0.0;
    for (int i = i0; i < i1; ++i) {
        const unsigned u = (unsigned)(a[i] >> 32);
        local += u ? (double)prob(u) : 0.0;
    }
    seg[tid] = local;
    __syncthreads();
    if (tid == 0) {
        double run0 = 0.0;
        for (int t = 0; t < NT; ++t) { const double v = seg[t]; seg[t] = run0; run0 += v; }
    }
    __syncthreads();
    const double start = seg[tid];
    __syncthreads();
    // (5) the survivors' renormalising sum; mask_j = (cumsum_j - p_j > top_p) is evaluated per element, as the reference does
    double run = start, kept = 0.0;
    for (int i = i0; i < i1; ++i) {
        const unsigned u = (unsigned)(a[i] >> 32);
        if (!u) break;                                             // ids that may not be drawn / padding: the tail of the sorted row
        const float pj = prob(u);
        run += (double)pj;
        if (!((float)run - pj > p.top_p)) kept += (double)pj;
    }
    seg[tid] = kept;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NT; ++w) t += seg[w];
        s_total = t;
    }
    __syncthreads();
    const float norm = (float)s_total;
    // (6) the race over the survivors: sorted position j takes noise[j]
    float win = -INFINITY;
    int win_rank = 0x7fffffff, win_tok = 0;
    run = start;
    for (int i = i0; i < i1; ++i) {
        const unsigned long long c = a[i];
        const unsigned u = (unsigned)(c >> 32);
        if (!u) break;
        const float pj = prob(u);
        run += (double)pj;
        if ((float)run - pj > p.top_p) continue;
        const float sc = (pj / norm) / p.noise[b * p.noise_stride + i];
        if (sc > win) { win = sc; win_rank = i; win_tok = 0x7fffffff - (int)(unsigned)c; }     // ascending i: the first maximum is kept
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

static int sample_vpad(int V) {
    int v = 2;
    while (v < V) v <<= 1;
    return v;
}

// chunks of the two-level form for (V, k), 0 when the one-level kernels serve the row
static int sample_split_chunks(int V, int k) {
    if (V <= 32768) return 0;
    const int chunks = (V + SPLIT_CHUNK - 1) / SPLIT_CHUNK;
    return (long)chunks * k <= SPLIT_CAP ? chunks : 0;
}

long rst_lm_sample_workspace_bytes_impl(int B, int V, int top_k, int top_p_mode) {
    if (B < 1 || V < 1) return 0;
    if (top_p_mode) return (long)B * sample_vpad(V) * 8;
    const int k = top_k > 0 && top_k < V ? top_k : V;
    const int chunks = sample_split_chunks(V, k);
    return chunks ? (long)B * chunks * (sizeof(SplitRec) + (long)k * 8) : 0;
}

int rst_launch_lm_sample(const LmSampleParams& p, hipStream_t stream) {
    RST_REQUIRE(p.logits && p.tokens && p.B >= 1 && p.V > 0 && p.V <= (1 << 20), "lm_sample: bad arguments");
    RST_REQUIRE(!p.use_sampling || p.temp <= 0.f || p.noise, "lm_sample: sampling needs the exponential noise tensor");
    const bool sampling = p.use_sampling && p.temp > 0.f;
    if (sampling && p.top_p > 0.f) {
        // nucleus sampling takes precedence over top-k, as in sample_token (utils/sampling.py:96-99)
        const int vpad = sample_vpad(p.V);
        RST_REQUIRE(p.ws && p.ws_bytes >= (long)p.B * vpad * 8, "lm_sample: top_p needs a workspace of B * 2^ceil(log2 V) * 8 bytes");
        RST_REQUIRE(p.noise_stride >= p.V, "lm_sample: top_p needs one noise value per vocabulary entry and row (noise_stride %d < V %d)", p.noise_stride, p.V);
        hipLaunchKernelGGL(sample_top_p_kernel, dim3(p.B), dim3(1024), 0, stream, p, vpad);
        return rst_check_launch("lm_sample_top_p");
    }
    const int k = p.top_k > 0 && p.top_k < p.V ? p.top_k : p.V;
    RST_REQUIRE(!sampling || k <= 8192, "lm_sample: top-k %d exceeds the 8192 candidate stage", k);
    const size_t lds = sampling ? (size_t)((k + 7) & ~7) * 8 : 0;      // greedy keeps no candidate list
    if (p.V <= 2048) hipLaunchKernelGGL((sample_kernel<256, 8>), dim3(p.B), dim3(256), lds, stream, p);
    else if (p.V <= 4096) hipLaunchKernelGGL((sample_kernel<256, 16>), dim3(p.B), dim3(256), lds, stream, p);
    else if (p.V <= 32768) hipLaunchKernelGGL((sample_kernel<1024, 32>), dim3(p.B), dim3(1024), lds, stream, p);
    else {
        RST_REQUIRE(!sampling || k <= 1024, "lm_sample: top-k %d > 1024 for a vocabulary of %d", k, p.V);
        const int kc = sampling ? k : 1;
        const int chunks = sample_split_chunks(p.V, kc);
        if (chunks && p.ws && p.ws_bytes >= (long)p.B * chunks * ((long)sizeof(SplitRec) + (long)kc * 8)) {
            LmSampleParams q = p;
            hipLaunchKernelGGL(sample_split_kernel, dim3(chunks, p.B), dim3(SPLIT_NT), (size_t)((kc + 7) & ~7) * 8, stream, q, chunks, kc);
            hipLaunchKernelGGL(sample_merge_kernel, dim3(p.B), dim3(256), 0, stream, q, chunks, kc);
        } else {
            hipLaunchKernelGGL(sample_big_kernel, dim3(p.B), dim3(1024), 0, stream, p);
        }
    }
    return rst_check_launch("lm_sample");
}
