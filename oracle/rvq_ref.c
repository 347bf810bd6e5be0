/* CPU oracle for the residual-VQ nearest-codeword search -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Restates EuclideanCodebook._quantize + ResidualVectorQuantization.encode of the reference
 * (MLLM_v2/tools/tokenizer/MimiCodec/model/quantization/core_vq.py:179-185 and :365-376) with a FIXED fp32
 * evaluation order, which is the arithmetic contract of rstnet_amd/csrc/rvq.hip:
 *     dot   = fmaf chain over k ascending, from 0        e2 = fmaf(e[k], e[k], e2) over k ascending
 *     score = fmaf(-2, dot, e2)                          (= |x-e|^2 - |x|^2, same argmin as cdist(x, E).argmin)
 *     code  = lowest index with the minimal score;       residual -= emb[code]
 * The reference computes the same argmin through torch.cdist (an MKL sgemm with unspecified summation order), so the
 * two agree on every decision whose top-2 gap exceeds fp32 round-off; tests/test_oracle_golden.py pins this file
 * against codes produced by the real reference.
 *
 * Build: see oracle/build.py  (gcc -O3 -mavx2 -mfma -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* x [M][D] (group latent, already projected); emb [L][n_codes][D]; codes [L][M]; dist [L][M] (may be NULL) */
int rvq_search_ref(const float* x, const float* emb, int M, int D, int n_codes, int L, int64_t* codes, float* dist) {
    float* embT = (float*)malloc((size_t)D * n_codes * sizeof(float));
    float* e2 = (float*)malloc((size_t)n_codes * sizeof(float));
    float* r = (float*)malloc((size_t)M * D * sizeof(float));
    if (!embT || !e2 || !r) return -1;
    memcpy(r, x, (size_t)M * D * sizeof(float));
    for (int l = 0; l < L; ++l) {
        const float* E = emb + (size_t)l * n_codes * D;
        for (int c = 0; c < n_codes; ++c) {
            float s = 0.0f;
            for (int k = 0; k < D; ++k) {
                s = __builtin_fmaf(E[(size_t)c * D + k], E[(size_t)c * D + k], s);
                embT[(size_t)k * n_codes + c] = E[(size_t)c * D + k];
            }
            e2[c] = s;
        }
#pragma omp parallel
        {
            float* dot = (float*)malloc((size_t)n_codes * sizeof(float));
#pragma omp for schedule(static)
            for (int m = 0; m < M; ++m) {
                float* rm = r + (size_t)m * D;
                for (int c = 0; c < n_codes; ++c) dot[c] = 0.0f;
                for (int k = 0; k < D; ++k) {
                    const float xv = rm[k];
                    const float* et = embT + (size_t)k * n_codes;
                    for (int c = 0; c < n_codes; ++c) dot[c] = __builtin_fmaf(xv, et[c], dot[c]);
                }
                float best = INFINITY;
                int bi = 0;
                for (int c = 0; c < n_codes; ++c) {
                    const float sc = __builtin_fmaf(-2.0f, dot[c], e2[c]);
                    if (sc < best) { best = sc; bi = c; }
                }
                codes[(size_t)l * M + m] = bi;
                if (dist) dist[(size_t)l * M + m] = best;
                const float* eb = E + (size_t)bi * D;
                for (int k = 0; k < D; ++k) rm[k] = rm[k] - eb[k];
            }
            free(dot);
        }
    }
    free(embT); free(e2); free(r);
    return 0;
}
