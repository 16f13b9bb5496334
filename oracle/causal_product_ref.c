/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the causal dot product that the reference's Performer
 * backbone executes on the CPU.  The algorithm lives in the third-party
 * dependency `pytorch-fast-transformers` (PyPI, unpinned by the reference
 * README.md:13-16; latest release 0.4.0), file
 * `fast_transformers/causal_product/causal_product_cpu.cpp`, which is ABSENT
 * from /root/reference and cannot be installed offline.  => "parity unpinned"
 * upstream: this file restates the published algorithm; it is anchored on the
 * reference call site stage2_accompaniment/model/fast_transformer_decoder.py:33-40
 * (att_builder.get("causal-linear")) and cross-checked in tests against two
 * independent formulations (O(T^2) masked form and the one-token recurrence).
 *
 * Semantics per (n, h):   S = 0 [E x M]
 *   forward : for t:  S += k_t (x) v_t ;  out_t = q_t^T S
 *   backward: forward sweep  S += k_t (x) v_t ;  dq_t = S . dout_t
 *             reverse sweep  R += q_t (x) dout_t ; dk_t = R . v_t ; dv_t = R^T . k_t
 * Layout: Q,K [N,H,L,E]  V,out [N,H,L,M], row-major fp32 — what
 * causal_linear_attention.py hands the native kernel after
 * `.permute(0,2,1,3).contiguous()`.
 *
 * Parallelisation follows upstream: OpenMP over the N*H independent scans,
 * strictly sequential in t.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void causal_dot_product_ref(const float *Q, const float *K, const float *V,
                            float *out, int64_t N, int64_t H, int64_t L,
                            int64_t E, int64_t M)
{
    const int64_t NH = N * H;
#pragma omp parallel for
    for (int64_t nh = 0; nh < NH; ++nh) {
        float *S = (float *)calloc((size_t)(E * M), sizeof(float));
        const float *q = Q + nh * L * E, *k = K + nh * L * E;
        const float *v = V + nh * L * M;
        float *o = out + nh * L * M;
        for (int64_t t = 0; t < L; ++t) {
            const float *kt = k + t * E, *vt = v + t * M, *qt = q + t * E;
            float *ot = o + t * M;
            for (int64_t e = 0; e < E; ++e) {
                const float ke = kt[e];
                float *Se = S + e * M;
                for (int64_t m = 0; m < M; ++m) Se[m] += ke * vt[m];
            }
            for (int64_t m = 0; m < M; ++m) ot[m] = 0.f;
            for (int64_t e = 0; e < E; ++e) {
                const float qe = qt[e];
                const float *Se = S + e * M;
                for (int64_t m = 0; m < M; ++m) ot[m] += qe * Se[m];
            }
        }
        free(S);
    }
}

void causal_dot_backward_ref(const float *Q, const float *K, const float *V,
                             const float *dO, float *dQ, float *dK, float *dV,
                             int64_t N, int64_t H, int64_t L, int64_t E,
                             int64_t M)
{
    const int64_t NH = N * H;
#pragma omp parallel for
    for (int64_t nh = 0; nh < NH; ++nh) {
        float *S = (float *)calloc((size_t)(E * M), sizeof(float));
        const float *q = Q + nh * L * E, *k = K + nh * L * E;
        const float *v = V + nh * L * M, *g = dO + nh * L * M;
        float *dq = dQ + nh * L * E, *dk = dK + nh * L * E;
        float *dv = dV + nh * L * M;
        /* forward sweep: dq_t = S_t . dout_t */
        for (int64_t t = 0; t < L; ++t) {
            const float *kt = k + t * E, *vt = v + t * M, *gt = g + t * M;
            for (int64_t e = 0; e < E; ++e) {
                const float ke = kt[e];
                float *Se = S + e * M;
                float acc = 0.f;
                for (int64_t m = 0; m < M; ++m) {
                    Se[m] += ke * vt[m];
                    acc += Se[m] * gt[m];
                }
                dq[t * E + e] = acc;
            }
        }
        /* reverse sweep: R += q_t (x) dout_t ; dk_t = R v_t ; dv_t = R^T k_t */
        memset(S, 0, (size_t)(E * M) * sizeof(float));
        for (int64_t t = L - 1; t >= 0; --t) {
            const float *qt = q + t * E, *kt = k + t * E;
            const float *vt = v + t * M, *gt = g + t * M;
            float *dvt = dv + t * M;
            for (int64_t m = 0; m < M; ++m) dvt[m] = 0.f;
            for (int64_t e = 0; e < E; ++e) {
                const float qe = qt[e], ke = kt[e];
                float *Re = S + e * M;
                float acc = 0.f;
                for (int64_t m = 0; m < M; ++m) {
                    Re[m] += qe * gt[m];
                    acc += Re[m] * vt[m];
                    dvt[m] += Re[m] * ke;
                }
                dk[t * E + e] = acc;
            }
        }
        free(S);
    }
}
