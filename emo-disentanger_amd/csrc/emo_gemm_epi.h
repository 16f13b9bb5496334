// Shared by the GEMM translation units: epilogue parameters and the row-contiguous fused epilogue
// (+bias -> aux_out -> act -> *mul(aux) -> dropout -> +residual -> store), see include/emo_hip.h (emo_epilogue_t).
#pragma once
#include "emo_common.h"

struct EpiParams {
    const float* bias;
    int act;
    void* aux_out;
    const void* mul_aux;
    int mul_mode;
    float mul_scale;
    DropCtx drop;
    const void* residual;
    int64_t ldc;
    int accumulate;
    int atomic;   // 0 = plain store; > 0 = split-K partial sums via fp32 atomics, value = XCDs per split (splitk_coords)
    // LayerNorm folded around a skinny (M <= 32) GEMM — decode path, see gemm_bf16_skinny_kernel
    const float* ln_c1;        // [N]: C = rstd[m] * (A.B^T - mean[m] * ln_c1[n]) (+ bias ...); row statistics of A computed in-kernel
    float* ln_stats_out;       // optional [M][2] (mean, rstd) of the A rows, written by block 0
    const void* rln_x;         // residual = LayerNorm(rln_x[m][n]) from rln_stats [M][2], rln_gamma / rln_beta [N]
    const float* rln_stats;
    const float* rln_gamma;
    const float* rln_beta;
    float ln_eps;
    float* a_rowsum;     // [M] += sum_k op(A)[m][k] (bias gradient riding on the wgrad GEMM), TN bf16 kernel only
    float* b_rowsum;     // [N] += sum_k op(B)[n][k] (same for the Conv1D layout, where dY is the B operand)
    int64_t ws_stride;   // > 0: split-K partials go to C + split * ws_stride with plain stores (splitk_reduce_kernel sums them)
    uint8_t* mask_out;   // M*N/8 bytes, tiled (emo_hip.h): bit mask of (value != 0) after act / dropout; mul_mode EMO_MUL_BITMASK reads one (A-stationary kernel only)
    int nt_store;        // A-stationary kernel: non-temporal output stores (set by its launcher for outputs too large to stay cached)
    int a_nt;            // streaming (non-temporal) hint on the A operand loads (A-stationary panel loads / 256 x 256 tile A tiles)
    int ablate;   // diagnostics only (EMO_GEMM_ABLATE): 1 = skip tile loads, 2 = skip MFMAs
    // LayerNorm of the A operand inside the A-stationary kernel (emo_hip.h: lna_*)
    const float* lna_gamma;
    const float* lna_beta;
    void* lna_out;
    float* lna_mean;
    float* lna_rstd;
    // per-row, per-64-column-block divisor (emo_hip.h: hdiv), A-stationary kernel only
    const float* hdiv;
    int64_t hdiv_T;
};

// ------------------------------------------------------------------------------------------------
// epilogue for 4 consecutive columns n..n+3 of row m
template <typename OutT> struct Out4;
template <> struct Out4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
};
template <> struct Out4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) { bf16x4 t = *(const bf16x4*)p; v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3]; }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) { bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; *(bf16x4*)p = t; }
};

template <typename OutT>
__device__ __forceinline__ void epi_store4(const EpiParams& ep, OutT* __restrict__ C, int64_t m, int64_t n,
                                           f32x4 acc, int64_t N) {
    const int64_t off = m * ep.ldc + n;
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if ((n + 3 < N) && ((ep.ldc & 3) == 0)) {
        // ---- fast path: 4 valid, 8/16-B aligned columns
        if (ep.bias) {
            f32x4 b = *(const f32x4*)(ep.bias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += b[i];
        }
        if (ep.aux_out) Out4<OutT>::store((OutT*)ep.aux_out + off, v);
        if (ep.act == EMO_ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        } else if (ep.act == EMO_ACT_GELU_NEW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_new_o<OutT>(v[i]);
        } else if (ep.act == EMO_ACT_GELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
        }
        if (ep.mul_mode != EMO_MUL_NONE) {
            float a[4];
            Out4<OutT>::load((const OutT*)ep.mul_aux + off, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= (ep.mul_mode == EMO_MUL_NONZERO) ? (a[i] != 0.f ? ep.mul_scale : 0.f) : (ep.mul_mode == EMO_MUL_DGELU ? dgelu_erf_f(a[i]) : dgelu_new_o<OutT>(a[i]));
        }
        if (ep.drop.thr16) {
            float dm[4];
            drop_mult4(ep.drop, (uint64_t)(m * N + n), dm);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= dm[i];
        }
        if (ep.residual) {
            float r[4];
            Out4<OutT>::load((const OutT*)ep.residual + off, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += r[i];
        }
        OutT* c = C + off;
        if constexpr (sizeof(OutT) == 4) {
            if (ep.atomic) {
#pragma unroll
                for (int i = 0; i < 4; ++i) atomicAdd((float*)c + i, v[i]);
                return;
            }
            if (ep.accumulate) {
                float o[4];
                Out4<float>::load((const float*)c, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += o[i];
            }
        }
        Out4<OutT>::store(c, v);
        return;
    }
    // ---- edge path (partial column group or unaligned ldc): scalar
    const int cnt = (int)((N - n) < 4 ? (N - n) : 4);
    for (int i = 0; i < cnt; ++i) {
        float x = v[i];
        if (ep.bias) x += ep.bias[n + i];
        if (ep.aux_out) ((OutT*)ep.aux_out)[off + i] = from_f32<OutT>(x);
        if (ep.act == EMO_ACT_RELU) x = fmaxf(x, 0.f);
        else if (ep.act == EMO_ACT_GELU_NEW) x = gelu_new_o<OutT>(x);
        else if (ep.act == EMO_ACT_GELU) x = gelu_erf_f(x);
        if (ep.mul_mode != EMO_MUL_NONE) {
            const float a = to_f32<OutT>(((const OutT*)ep.mul_aux)[off + i]);
            x *= (ep.mul_mode == EMO_MUL_NONZERO) ? (a != 0.f ? ep.mul_scale : 0.f) : (ep.mul_mode == EMO_MUL_DGELU ? dgelu_erf_f(a) : dgelu_new_o<OutT>(a));
        }
        if (ep.drop.thr16) x *= drop_mult(ep.drop, (uint64_t)(m * N + n + i));
        if (ep.residual) x += to_f32<OutT>(((const OutT*)ep.residual)[off + i]);
        if constexpr (sizeof(OutT) == 4) {
            if (ep.atomic) { atomicAdd((float*)C + off + i, x); continue; }
            if (ep.accumulate) x += ((const float*)C)[off + i];
        }
        C[off + i] = from_f32<OutT>(x);
    }
}

// ------------------------------------------------------------------------------------------------
// Tile epilogue through LDS.  Measured (EMO_GEMM_ABLATE, K=512 vs K=2048): the per-tile FIXED cost was ~7 us, most of it
// the store tail of the fragment-layout epilogue (16 x 8-B row-strided stores per lane = the store-issue-bound pattern
// of the guide's T21).  The 128x128 fp32 accumulator tile is staged in the (now idle) 64 KB of LDS with a 16-B-chunk
// XOR swizzle, then every thread owns 8 CONSECUTIVE columns of a row: bias / residual / mask loads and the output
// store are 16-B, fully coalesced (16 lanes per 256-B row), and the dropout hash count halves.
template <typename OutT>
__device__ __forceinline__ void epi_row8(const EpiParams& ep, OutT* __restrict__ C, int64_t m, int64_t n, float (&v)[8], int64_t N) {
    const int64_t off = m * ep.ldc + n;
    if ((n + 7 < N) && ((ep.ldc & 7) == 0)) {
        if (ep.bias) {
            f32x4 b0 = *(const f32x4*)(ep.bias + n), b1 = *(const f32x4*)(ep.bias + n + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] += b0[i]; v[4 + i] += b1[i]; }
        }
        if (ep.aux_out) {
            float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
            Out4<OutT>::store((OutT*)ep.aux_out + off, lo);
            Out4<OutT>::store((OutT*)ep.aux_out + off + 4, hi);
        }
        if (ep.act == EMO_ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        } else if (ep.act == EMO_ACT_GELU_NEW) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = gelu_new_o<OutT>(v[i]);
        } else if (ep.act == EMO_ACT_GELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = gelu_erf_f(v[i]);
        }
        if (ep.mul_mode != EMO_MUL_NONE) {
            float a[8];
            { float t0[4], t1[4]; Out4<OutT>::load((const OutT*)ep.mul_aux + off, t0); Out4<OutT>::load((const OutT*)ep.mul_aux + off + 4, t1);
#pragma unroll
              for (int i = 0; i < 4; ++i) { a[i] = t0[i]; a[4 + i] = t1[i]; } }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= (ep.mul_mode == EMO_MUL_NONZERO) ? (a[i] != 0.f ? ep.mul_scale : 0.f) : (ep.mul_mode == EMO_MUL_DGELU ? dgelu_erf_f(a[i]) : dgelu_new_o<OutT>(a[i]));
        }
        if (ep.drop.thr16) {
            float d0[4], d1[4];
            drop_mult4(ep.drop, (uint64_t)(m * N + n), d0);
            drop_mult4(ep.drop, (uint64_t)(m * N + n + 4), d1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] *= d0[i]; v[4 + i] *= d1[i]; }
        }
        if (ep.residual) {
            float t0[4], t1[4];
            Out4<OutT>::load((const OutT*)ep.residual + off, t0);
            Out4<OutT>::load((const OutT*)ep.residual + off + 4, t1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] += t0[i]; v[4 + i] += t1[i]; }
        }
        OutT* c = C + off;
        if constexpr (sizeof(OutT) == 4) {
            if (ep.atomic) {
#pragma unroll
                for (int i = 0; i < 8; ++i) atomicAdd((float*)c + i, v[i]);
                return;
            }
            if (ep.accumulate) {
                float t0[4], t1[4];
                Out4<float>::load((const float*)c, t0);
                Out4<float>::load((const float*)c + 4, t1);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] += t0[i]; v[4 + i] += t1[i]; }
            }
            float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
            Out4<float>::store((float*)c, lo);
            Out4<float>::store((float*)c + 4, hi);
        } else {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
            *(bf16x8*)c = o;
        }
        return;
    }
    // edge: reuse the 4-wide path (which itself falls back to scalars)
    if (n < N) epi_store4<OutT>(ep, C, m, n, (f32x4){v[0], v[1], v[2], v[3]}, N);
    if (n + 4 < N) epi_store4<OutT>(ep, C, m, n + 4, (f32x4){v[4], v[5], v[6], v[7]}, N);
}


// A-stationary K = 512 kernel (emo_gemm_astat.hip): true when the shape / epilogue is eligible and the launch was queued.
bool emo_gemm_astat_try(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int dtype_out, int64_t M, int64_t N, int64_t K,
                        const EpiParams& ep, hipStream_t st);
// 256 x 256 tile, one wave per SIMD (emo_gemm_w128.hip): long-K NT products; true when eligible and queued.
bool emo_gemm_w128_try(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int dtype_out, int64_t M, int64_t N, int64_t K,
                       const EpiParams& ep, hipStream_t st);
// persistent 256 x 256 tile walk, 32 x 32 x 16 MFMA (emo_gemm_p256.hip, r05): true when eligible and queued.
int emo_gemm_p256_try(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int dtype_out, int64_t M, int64_t N, int64_t K,
                       const EpiParams& ep, hipStream_t st);
// the same tile for the wgrad layout (A stored [K, M], B stored [K, N], fp32 out, split-K through the caller's workspace)
int64_t emo_gemm_w128_tn_splits(int64_t M, int64_t N, int64_t K);
int64_t emo_gemm_w128_tn_rs_floats(int64_t M, int64_t N, int64_t splits);
bool emo_gemm_w128_tn_try(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate,
                          float* a_rowsum, float* b_rowsum, void* ws, int64_t ws_bytes, hipStream_t st);
