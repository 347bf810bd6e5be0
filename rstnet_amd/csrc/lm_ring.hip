// Delay-pattern bookkeeping of LMGen.step (MLLM_v2/models/model.py:490-562) on the device: the token ring
// cache [B][K][CT] int64, the per-codebook delays and the step counter never leave HBM, so a whole frame -- ring update,
// temporal step, text sample, 8 depth steps with sampling, ring commit + delayed gather -- is one captured graph with no host
// arithmetic in between.  B * K is a few hundred elements: one workgroup, plain loops.
#include "lm_common.h"

namespace {

// Start of a frame (model.py:506-521): user streams written at (offset + delay_k) % CT, initial tokens while
// offset <= delay_k, then the column at offset % CT is the model input of this step.
__global__ __launch_bounds__(256) void lm_ring_begin_kernel(const LmRingParams p) {
    const long off = *p.offset_dev;
    const int pos = (int)(off % p.CT);
    for (int idx = threadIdx.x; idx < p.B * p.K; idx += 256) {
        const int b = idx / p.K, k = idx - b * p.K;
        long* row = p.cache + (long)idx * p.CT;
        const int q = k - p.first_user;
        if (q >= 0 && q < p.Ki) row[(off + p.delays[k]) % p.CT] = p.user[(long)b * p.Ki + q];
        if (off <= p.delays[k]) row[pos] = p.initial[k];
        p.input_out[idx] = row[pos];
    }
}

// End of a frame (model.py:545-562): offset += 1, the generated text / audio tokens go to column offset % CT, and the
// output is the column set re-aligned by the delays: out[b][k] = cache[b][k][(offset - max_delay + delay_k) % CT].
__global__ __launch_bounds__(256) void lm_ring_commit_kernel(const LmRingParams p) {
    const long off1 = *p.offset_dev + 1;
    const int pos = (int)(off1 % p.CT);
    for (int idx = threadIdx.x; idx < p.B * p.n_out; idx += 256) {
        const int b = idx / p.n_out, k = idx - b * p.n_out;
        p.cache[((long)b * p.K + k) * p.CT + pos] = p.tokens[idx];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < p.B * p.n_out; idx += 256) {
        const int b = idx / p.n_out, k = idx - b * p.n_out;
        long g = (off1 - p.max_delay + p.delays[k]) % p.CT;
        if (g < 0) g += p.CT;
        p.out[idx] = p.cache[((long)b * p.K + k) * p.CT + g];
    }
    __syncthreads();                 // every thread has read the counter before it moves
    if (threadIdx.x == 0) *p.offset_dev = off1;
}

}  // namespace

int rst_launch_lm_ring_begin(const LmRingParams& p, hipStream_t stream) {
    RST_REQUIRE(p.cache && p.user && p.initial && p.delays && p.offset_dev && p.input_out, "lm_ring_begin: null pointer");
    RST_REQUIRE(p.B >= 1 && p.K >= 1 && p.CT >= 1 && p.Ki >= 0 && p.first_user >= 0 && p.first_user + p.Ki <= p.K,
                "lm_ring_begin: bad sizes (K=%d Ki=%d first_user=%d)", p.K, p.Ki, p.first_user);
    hipLaunchKernelGGL(lm_ring_begin_kernel, dim3(1), dim3(256), 0, stream, p);
    return rst_check_launch("lm_ring_begin");
}

int rst_launch_lm_ring_commit(const LmRingParams& p, hipStream_t stream) {
    RST_REQUIRE(p.cache && p.tokens && p.delays && p.offset_dev && p.out, "lm_ring_commit: null pointer");
    RST_REQUIRE(p.B >= 1 && p.K >= 1 && p.CT >= 1 && p.n_out >= 1 && p.n_out <= p.K && p.max_delay >= 0, "lm_ring_commit: bad sizes");
    hipLaunchKernelGGL(lm_ring_commit_kernel, dim3(1), dim3(256), 0, stream, p);
    return rst_check_launch("lm_ring_commit");
}
