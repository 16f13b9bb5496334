// HBM-bound kernels of the path: K1 embedding prologue (fwd/bwd), K6 LayerNorm (fwd/bwd),
// dropout re-application, K9 cross-entropy (fwd/bwd), K12 accuracy counts, optimizer plumbing
// (sum of squares, clip coefficient, Adam, dtype cast).  All are one-pass, 16-B-per-lane
// coalesced streams with wave64 shuffle reductions; none is reshaped into a GEMM.
#include <stdarg.h>

#include "emo_common.h"
#include "emo_nucleus.h"

// ------------------------------------------------------------------------------------------------ errors / misc
static thread_local char g_err[512] = "";
void emo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* emo_last_error(void) { return g_err; }
extern "C" int emo_version(void) { return 100; }
extern "C" int emo_build_flags(void) {
#ifdef EMO_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}
extern "C" int emo_device_cus(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) { bf16x4 t = *(const bf16x4*)p; v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3]; }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) { bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; *(bf16x4*)p = t; }
};

// ================================================================================================ K1 embedding
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ seg,
                                                        const float* __restrict__ E, const float* __restrict__ S,
                                                        const float* __restrict__ pe, T* __restrict__ out, int64_t M, int64_t T_len,
                                                        int64_t D, int64_t pos0, const int64_t* __restrict__ pos_ids, float scale, DropCtx drop) {
    const int64_t d4 = D >> 2, total = M * d4;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = it / d4, c = (it - row * d4) << 2;
        const int64_t t = row % T_len;
        const int64_t pbase = pos0 + (pos_ids ? pos_ids[row / T_len] : 0);
        float e[4], s[4] = {0, 0, 0, 0}, p[4], o[4];
        Vec4<float>::load(E + tok[row] * D + c, e);
        if (seg) Vec4<float>::load(S + seg[row] * D + c, s);
        Vec4<float>::load(pe + (pbase + t) * D + c, p);
        float dm[4];
        drop_mult4(drop, (uint64_t)(row * D + c), dm);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // reference order: emb.mul_(scale); emb += seg.mul_(scale); + pe
            float v = e[i] * scale;
            v += s[i] * scale;
            v += p[i];
            o[i] = v * dm[i];
        }
        Vec4<T>::store(out + row * D + c, o);
    }
}

extern "C" int emo_embed_fwd(const int64_t* tok, const int64_t* seg, const float* E, const float* S, const float* pe,
                             void* out, int dtype, int64_t B, int64_t T, int64_t D, int64_t V, int64_t n_seg, int64_t pos0,
                             const int64_t* pos_ids, float scale, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    (void)V; (void)n_seg;
    EMO_CHECK(tok && E && pe && out, "emo_embed_fwd: null pointer");
    EMO_CHECK((D & 3) == 0, "emo_embed_fwd: D must be a multiple of 4");
    EMO_CHECK(!(seg && !S), "emo_embed_fwd: seg ids without a segment table");
    const int64_t M = B * T, total = M * (D >> 2);
    int64_t blocks = cdiv64(total, 256);
    if (blocks > 4096) blocks = 4096;
    DropCtx drop = make_drop(p_drop, seed, offset);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EMO_F32) hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, tok, seg, E, S, pe, (float*)out, M, T, D, pos0, pos_ids, scale, drop);
    else hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, tok, seg, E, S, pe, (bf16_t*)out, M, T, D, pos0, pos_ids, scale, drop);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// backward: block = (64-column slice) x (token chunk); per-block LDS table [(V+n_seg) x 64] fp32
// accumulated with ds_add_f32, flushed once with global atomics.  The 84-KB table allows one block per CU, so the block is 8 waves and
// every wave keeps 8 token rows in flight (ids through scalar loads, all 8 gradient loads issued before the first LDS atomic): (A single-wave-per-slice variant
// with plain ds_read / add / ds_write instead of atomics was slower, 1070 us: one wave per CU cannot hide the HBM latency of 128-B rows.)
constexpr int EB_THREADS = 512, EB_U = 8;
template <typename T>
__global__ __launch_bounds__(EB_THREADS) void embed_bwd_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ seg,
                                                        const T* __restrict__ dout, float* __restrict__ dE, float* __restrict__ dS,
                                                        int64_t M, int64_t D, int64_t V, int64_t n_seg, int64_t rows_per_block,
                                                        float scale, DropCtx drop) {
    extern __shared__ float tab[];  // (V + n_seg) * 64
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = EB_THREADS / 64;
    const int64_t col = (int64_t)blockIdx.x * 64 + lane;
    const int64_t rows_tab = V + n_seg;
    for (int64_t i = threadIdx.x; i < rows_tab * 64; i += EB_THREADS) tab[i] = 0.f;
    __syncthreads();
    const int64_t mbeg = (int64_t)blockIdx.y * rows_per_block;
    int64_t mend = mbeg + rows_per_block;
    if (mend > M) mend = M;
    // LDS float atomics run at ~0.4 lane-adds per clock per CU and ARE this kernel's time (r01: 671 us with, 354 us without the segment rows,
    // independent of the id distribution and of the flush count): the <= 2 segment rows are a column sum and stay in registers
    const bool seg_regs = n_seg <= 2;
    float s0 = 0.f, s1 = 0.f;
    if (col < D) {
        for (int64_t m0 = mbeg + wave * EB_U; m0 < mend; m0 += NW * EB_U) {
            float g[EB_U];
            int64_t tk[EB_U], sg[EB_U];
#pragma unroll
            for (int u = 0; u < EB_U; ++u) {
                const int64_t m = m0 + u < mend ? m0 + u : mend - 1;
                tk[u] = tok[m];
                sg[u] = seg ? seg[m] : 0;
                g[u] = to_f32<T>(dout[m * D + col]);
            }
            // rows of this step that carry the same id (wave-uniform scalars: music tokens repeat a lot) are summed in registers first,
            // so the LDS atomic unit — the bottleneck — sees one add per distinct id
            float vv[EB_U];
            bool dead[EB_U];
#pragma unroll
            for (int u = 0; u < EB_U; ++u) {
                vv[u] = (m0 + u < mend) ? g[u] * drop_mult(drop, (uint64_t)((m0 + u) * D + col)) * scale : 0.f;
                dead[u] = !(m0 + u < mend);
            }
            float segv[EB_U];
#pragma unroll
            for (int u = 0; u < EB_U; ++u) segv[u] = vv[u];
#pragma unroll
            for (int u = EB_U - 1; u > 0; --u) {
#pragma unroll
                for (int w = u - 1; w >= 0; --w) {
                    if (!dead[u] && !dead[w] && tk[u] == tk[w]) { vv[w] += vv[u]; dead[u] = true; }
                }
            }
#pragma unroll
            for (int u = 0; u < EB_U; ++u) {
                if (m0 + u < mend) {
                    const float v = segv[u];
                    if (!dead[u]) atomicAdd(&tab[tk[u] * 64 + lane], vv[u]);
                    if (seg) {
                        if (seg_regs) { s0 += sg[u] == 0 ? v : 0.f; s1 += sg[u] == 1 ? v : 0.f; }
                        else atomicAdd(&tab[(V + sg[u]) * 64 + lane], v);
                    }
                }
            }
        }
        if (seg && seg_regs) {
            if (s0 != 0.f) atomicAdd(dS + col, s0);
            if (n_seg > 1 && s1 != 0.f) atomicAdd(dS + D + col, s1);
        }
    }
    __syncthreads();
    if (col < D) {
        for (int64_t r = wave; r < (seg_regs ? V : rows_tab); r += NW) {
            float v = tab[r * 64 + lane];
            if (v != 0.f) {
                if (r < V) atomicAdd(dE + r * D + col, v);
                else atomicAdd(dS + (r - V) * D + col, v);
            }
        }
    }
}

extern "C" int emo_embed_bwd(const int64_t* tok, const int64_t* seg, const void* dout, int dtype, float* dE, float* dS,
                             int64_t B, int64_t T, int64_t D, int64_t V, int64_t n_seg, float scale, float p_drop,
                             uint64_t seed, uint64_t offset, emo_stream_t stream) {
    EMO_CHECK(tok && dout && dE, "emo_embed_bwd: null pointer");
    if (!seg) n_seg = 0;
    EMO_CHECK(!(seg && !dS), "emo_embed_bwd: seg ids without dS");
    const size_t lds = (size_t)(V + n_seg) * 64 * sizeof(float);
    EMO_CHECK(lds <= 160 * 1024, "emo_embed_bwd: vocabulary of %lld rows does not fit the LDS table", (long long)(V + n_seg));
    const int64_t M = B * T;
    // 8 column slices x 32 token chunks = 256 blocks (one 84-KB block per CU) at the bench shape; smaller token counts keep the 32 chunks
    // (r03: with the fixed 4096-row chunk the reference's batch size 4 ran 16 blocks and took the same 335 us as batch 64)
    int64_t rpb = cdiv64(cdiv64(M, 32), 64) * 64;
    if (rpb < 256) rpb = 256;
    if (rpb > 4096) rpb = 4096;
    { const char* e = getenv("EMO_EMBED_RPB"); if (e && atoi(e) > 0) rpb = atoi(e); }
    if (rpb > M) rpb = M;
    dim3 grid((unsigned)cdiv64(D, 64), (unsigned)cdiv64(M, rpb));
    DropCtx drop = make_drop(p_drop, seed, offset);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EMO_F32) {
        static bool a = false;
        if (!a) { (void)hipFuncSetAttribute((const void*)embed_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a = true; }
        hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(EB_THREADS), lds, st, tok, seg, (const float*)dout, dE, dS, M, D, V, n_seg, rpb, scale, drop);
    } else {
        static bool a = false;
        if (!a) { (void)hipFuncSetAttribute((const void*)embed_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a = true; }
        hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, grid, dim3(EB_THREADS), lds, st, tok, seg, (const bf16_t*)dout, dE, dS, M, D, V, n_seg, rpb, scale, drop);
    }
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// ================================================================================================ K6 LayerNorm
// one wave per row; lane owns columns {lane*4 + 256*j}, j < NV (D <= 256*NV), kept in registers.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int64_t M,
                                                            int64_t D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t c = lane * 4 + 256 * j;
        if (c < D) Vec4<T>::load(x + row * D + c, v[j]);
        else v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t c = lane * 4 + 256 * j;
        if (c < D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { float d = v[j][i] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t c = lane * 4 + 256 * j;
        if (c < D) {
            float g[4], b[4], o[4];
            Vec4<float>::load(gamma + c, g);
            Vec4<float>::load(beta + c, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (v[j][i] - mu) * rs * g[i] + b[i];
            Vec4<T>::store(y + row * D + c, o);
        }
    }
}

// bf16, D = 512 (the bench width): the 1-KB row is ONE 16-B load per lane, two rows in flight per wave, gamma / beta in registers.
// r01: the generic kernel (8-B pieces, one row per wave per block) ran at 4.0 TB/s effective (67 us for 131072 rows).
__global__ __launch_bounds__(256) void layernorm_fwd_bf16_d512_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                      float* __restrict__ mean, float* __restrict__ rstd, int64_t M, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float g[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[i] = gamma[lane * 8 + i]; b[i] = beta[lane * 8 + i]; }
    for (int64_t r0 = wave * 2; r0 < M; r0 += nwaves * 2) {
        const int64_t r1 = r0 + 1 < M ? r0 + 1 : r0;
        const bf16x8 a0 = *(const bf16x8*)(x + r0 * 512 + lane * 8);
        const bf16x8 a1 = *(const bf16x8*)(x + r1 * 512 + lane * 8);
        float v0[8], v1[8], s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v0[i] = (float)a0[i]; v1[i] = (float)a1[i]; s0 += v0[i]; s1 += v1[i]; }
        const float m0 = wave_sum(s0) * (1.f / 512.f), m1 = wave_sum(s1) * (1.f / 512.f);
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d0 = v0[i] - m0, d1 = v1[i] - m1; q0 += d0 * d0; q1 += d1 * d1; }
        const float rs0 = rsqrtf(wave_sum(q0) * (1.f / 512.f) + eps), rs1 = rsqrtf(wave_sum(q1) * (1.f / 512.f) + eps);
        bf16x8 o0, o1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o0[i] = (bf16_t)((v0[i] - m0) * rs0 * g[i] + b[i]);
            o1[i] = (bf16_t)((v1[i] - m1) * rs1 * g[i] + b[i]);
        }
        *(bf16x8*)(y + r0 * 512 + lane * 8) = o0;
        if (r0 + 1 < M) *(bf16x8*)(y + r1 * 512 + lane * 8) = o1;
        if (lane == 0) {
            mean[r0] = m0; rstd[r0] = rs0;
            if (r0 + 1 < M) { mean[r1] = m1; rstd[r1] = rs1; }
        }
    }
}

extern "C" int emo_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 int dtype, int64_t M, int64_t D, float eps, emo_stream_t stream) {
    EMO_CHECK(x && gamma && beta && y && mean && rstd, "emo_layernorm_fwd: null pointer");
    EMO_CHECK((D & 3) == 0 && D <= 1024, "emo_layernorm_fwd: D must be a multiple of 4 and <= 1024 (got %lld)", (long long)D);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EMO_BF16 && D == 512 && M >= 4096 && getenv("EMO_LN_GENERIC") == nullptr && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        int64_t blocks = cdiv64(M, 32);           // 4 waves x 2 rows x 4 steps per block
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(layernorm_fwd_bf16_d512_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, eps);
        EMO_LAUNCH_CHECK();
        return EMO_OK;
    }
    dim3 grid((unsigned)cdiv64(M, 4));
#define LN_FWD(TT, NVV) hipLaunchKernelGGL((layernorm_fwd_kernel<TT, NVV>), grid, dim3(256), 0, st, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, M, D, eps)
    const int nv = (int)cdiv64(D, 256);
    if (dtype == EMO_F32) { if (nv <= 1) LN_FWD(float, 1); else if (nv == 2) LN_FWD(float, 2); else LN_FWD(float, 4); }
    else { if (nv <= 1) LN_FWD(bf16_t, 1); else if (nv == 2) LN_FWD(bf16_t, 2); else LN_FWD(bf16_t, 4); }
#undef LN_FWD
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// backward: wave per row (grid-stride over rows), per-lane dgamma/dbeta partials kept in registers
// over all rows of the block, reduced across the block's 4 waves in LDS, one atomic per column.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ dres,
                                                            T* __restrict__ dx, T* __restrict__ dx_drop, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dcol, int64_t M, int64_t D, DropCtx drop) {
    __shared__ float red[3][4][256 * NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NV][4], dg[NV][4], db[NV][4], dc[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t c = lane * 4 + 256 * j;
        if (c < D) Vec4<float>::load(gamma + c, g[j]);
        else g[j][0] = g[j][1] = g[j][2] = g[j][3] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dg[j][i] = db[j][i] = dc[j][i] = 0.f;
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float gy[NV][4], xh[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t c = lane * 4 + 256 * j;
            if (c < D) {
                float a[4], b[4];
                Vec4<T>::load(dy + row * D + c, a);
                Vec4<T>::load(x + row * D + c, b);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[j][i] = (b[i] - mu) * rs;
                    dg[j][i] += a[i] * xh[j][i];
                    db[j][i] += a[i];
                    gy[j][i] = a[i] * g[j][i];
                    s1 += gy[j][i];
                    s2 += gy[j][i] * xh[j][i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) gy[j][i] = xh[j][i] = 0.f;
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t c = lane * 4 + 256 * j;
            if (c < D) {
                float o[4], r[4] = {0, 0, 0, 0};
                if (dres) Vec4<T>::load(dres + row * D + c, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = rs * (gy[j][i] - s1 - xh[j][i] * s2) + r[i];
                Vec4<T>::store(dx + row * D + c, o);
                if (dcol && !dx_drop) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dc[j][i] += to_f32<T>(from_f32<T>(o[i]));
                }
                if (dx_drop) {
                    // the consumer re-reads dx in storage precision: mask the ROUNDED value
                    float od[4], dm[4];
                    drop_mult4(drop, (uint64_t)(row * D + c), dm);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { od[i] = to_f32<T>(from_f32<T>(o[i])) * dm[i]; dc[j][i] += to_f32<T>(from_f32<T>(od[i])); }
                    Vec4<T>::store(dx_drop + row * D + c, od);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[0][wave][j * 256 + lane * 4 + i] = dg[j][i];
            red[1][wave][j * 256 + lane * 4 + i] = db[j][i];
            red[2][wave][j * 256 + lane * 4 + i] = dc[j][i];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 256 * NV; c += 256) {
        if (c < D) {
            atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
            if (dcol) atomicAdd(dcol + c, red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
        }
    }
}

// bf16, D = 512: a lane owns 8 consecutive columns, so every access is one 16-B vector and a wave covers the row with one instruction per
// tensor (the generic kernel's 4-column ownership gives 8-B accesses: 3.4-4.1 TB/s effective at M = 131072, r01).  Same arithmetic and the
// same rounding points as the generic kernel.
__global__ __launch_bounds__(256) void layernorm_bwd_bf16_d512_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                      const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                                      bf16_t* __restrict__ dx, bf16_t* __restrict__ dx_drop, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta, float* __restrict__ dcol, int64_t M, DropCtx drop,
                                                                      float* __restrict__ part) {
    constexpr int D = 512;
    __shared__ float red[3][4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane * 8;
    float g[8], dg[8], db[8], dc[8];
    { const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { g[i] = g0[i]; g[4 + i] = g1[i]; } }
#pragma unroll
    for (int i = 0; i < 8; ++i) dg[i] = db[i] = dc[i] = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const bf16x8 a8 = *(const bf16x8*)(dy + row * D + c), b8 = *(const bf16x8*)(x + row * D + c);
        bf16x8 r8;
        if (dres) r8 = *(const bf16x8*)(dres + row * D + c);
        const float mu = mean[row], rs = rstd[row];
        float gy[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = (float)a8[i];
            xh[i] = ((float)b8[i] - mu) * rs;
            dg[i] += a * xh[i];
            db[i] += a;
            gy[i] = a * g[i];
            s1 += gy[i];
            s2 += gy[i] * xh[i];
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        bf16x8 o8;
        float of[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float o = rs * (gy[i] - s1 - xh[i] * s2) + (dres ? (float)r8[i] : 0.f);
            o8[i] = (bf16_t)o;
            of[i] = (float)o8[i];                    // the consumer re-reads dx in storage precision
        }
        *(bf16x8*)(dx + row * D + c) = o8;
        if (dx_drop) {
            float dm0[4], dm1[4];
            drop_mult4(drop, (uint64_t)(row * D + c), dm0);
            drop_mult4(drop, (uint64_t)(row * D + c + 4), dm1);
            bf16x8 d8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d8[i] = (bf16_t)(of[i] * (i < 4 ? dm0[i] : dm1[i - 4]));
                dc[i] += (float)d8[i];
            }
            *(bf16x8*)(dx_drop + row * D + c) = d8;
        } else if (dcol) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dc[i] += of[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[0][wave][c + i] = dg[i]; red[1][wave][c + i] = db[i]; red[2][wave][c + i] = dc[i]; }
    __syncthreads();
    if (part) {                                      // per-block partial column sums [gridDim][3][512]: ln_bwd_colsum_kernel adds them up in a fixed order
        float* pb = part + (int64_t)blockIdx.x * 3 * D;
        for (int cc = threadIdx.x; cc < 3 * D; cc += 256) {
            const int w = cc >> 9, c1 = cc & (D - 1);
            pb[cc] = red[w][0][c1] + red[w][1][c1] + red[w][2][c1] + red[w][3][c1];
        }
        return;
    }
    for (int cc = threadIdx.x; cc < D; cc += 256) {
        atomicAdd(dgamma + cc, red[0][0][cc] + red[0][1][cc] + red[0][2][cc] + red[0][3][cc]);
        atomicAdd(dbeta + cc, red[1][0][cc] + red[1][1][cc] + red[1][2][cc] + red[1][3][cc]);
        if (dcol) atomicAdd(dcol + cc, red[2][0][cc] + red[2][1][cc] + red[2][2][cc] + red[2][3][cc]);
    }
}

// dgamma / dbeta / dcol += the blocks' partial column sums, summed in block order (deterministic; r04: the 3 x 512 fp32 atomics per block were
// ~8 us per 256 blocks — a third of the kernel at 8192 rows — and made these three gradients depend on the arrival order).
// A block owns 32 columns of one of the three vectors: 8 column quads x 32 slices of the block list.
__global__ __launch_bounds__(256) void ln_bwd_colsum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dcol) {
    constexpr int D = 512;
    __shared__ f32x4 red[32][8];
    const int w = blockIdx.x / (D / 32), c0 = (blockIdx.x % (D / 32)) * 32 + (threadIdx.x & 7) * 4, sl = threadIdx.x >> 3;
    float* out = w == 0 ? dgamma : (w == 1 ? dbeta : dcol);
    if (!out) return;
    const float* p = part + (int64_t)w * D + c0;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a, c = a, d = a;
    int i = sl;
    for (; i + 96 < nblk; i += 128) {
        const f32x4 x0 = *(const f32x4*)(p + (int64_t)i * 3 * D), x1 = *(const f32x4*)(p + (int64_t)(i + 32) * 3 * D);
        const f32x4 x2 = *(const f32x4*)(p + (int64_t)(i + 64) * 3 * D), x3 = *(const f32x4*)(p + (int64_t)(i + 96) * 3 * D);
        a += x0; b += x1; c += x2; d += x3;
    }
    for (; i < nblk; i += 32) a += *(const f32x4*)(p + (int64_t)i * 3 * D);
    red[sl][threadIdx.x & 7] = (a + b) + (c + d);
    __syncthreads();
    if (threadIdx.x < 8) {
        f32x4 s = red[0][threadIdx.x];
        for (int j = 1; j < 32; ++j) s += red[j][threadIdx.x];
        f32x4 o = *(f32x4*)(out + c0);
        o += s;
        *(f32x4*)(out + c0) = o;
    }
}

// partial column sums of the bf16 / D = 512 kernel: [blocks][3][512] fp32 (0: the shape takes the generic kernel, which accumulates with atomics)
static int64_t ln_bwd_blocks(int64_t M) {
    int64_t b = cdiv64(M, 8);                         // two rows per wave at small M; >= 128 rows per block at the benchmark's 131072
    if (b > 1024) b = 1024;
    { const char* eb = getenv("EMO_LN_BWD_BLOCKS"); if (eb && atoi(eb) > 0) b = atoi(eb); }
    return b < 1 ? 1 : b;
}
extern "C" int64_t emo_layernorm_bwd_workspace_bytes(int dtype, int64_t M, int64_t D) {
    if (dtype != EMO_BF16 || D != 512 || getenv("EMO_LN_GENERIC") != nullptr) return 0;
    { const char* e = getenv("EMO_LN_BWD_ATOMIC"); if (e && atoi(e) != 0) return 0; }
    return ln_bwd_blocks(M) * 3 * 512 * (int64_t)sizeof(float);
}

extern "C" int emo_layernorm_bwd_ws(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                    const void* dres, void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dcol, int dtype,
                                    int64_t M, int64_t D, float p_drop, uint64_t seed, uint64_t offset, void* workspace, int64_t workspace_bytes,
                                    emo_stream_t stream) {
    EMO_CHECK(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "emo_layernorm_bwd: null pointer");
    EMO_CHECK((D & 3) == 0 && D <= 1024, "emo_layernorm_bwd: D must be a multiple of 4 and <= 1024 (got %lld)", (long long)D);
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = cdiv64(M, 4);
    if (blocks > 1024) blocks = 1024;
    dim3 grid((unsigned)blocks);
    DropCtx drop = make_drop(p_drop, seed, offset);
    if (dtype == EMO_BF16 && D == 512 && getenv("EMO_LN_GENERIC") == nullptr &&
        ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres | (uintptr_t)dx_drop | (uintptr_t)gamma) & 15) == 0)) {
        const int64_t need = emo_layernorm_bwd_workspace_bytes(dtype, M, D);
        const bool use_ws = workspace && need > 0 && workspace_bytes >= need && (((uintptr_t)workspace | (uintptr_t)dgamma | (uintptr_t)dbeta | (uintptr_t)dcol) & 15) == 0;
        int64_t b8;
        if (use_ws) b8 = ln_bwd_blocks(M);
        else {
            // every block ends with 3 x 512 fp32 atomics (dgamma, dbeta, dcol): ~25 us per 1024 blocks (r01 sweep), so a block takes >= 32 rows
            b8 = cdiv64(M, 32);
            if (b8 > 1024) b8 = 1024;
        }
        hipLaunchKernelGGL(layernorm_bwd_bf16_d512_kernel, dim3((unsigned)b8), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                           (const bf16_t*)dres, (bf16_t*)dx, (bf16_t*)dx_drop, dgamma, dbeta, dcol, M, drop, use_ws ? (float*)workspace : nullptr);
        if (use_ws)
            hipLaunchKernelGGL(ln_bwd_colsum_kernel, dim3(3 * 512 / 32), dim3(256), 0, st, (const float*)workspace, (int)b8, dgamma, dbeta, dcol);
        EMO_LAUNCH_CHECK();
        return EMO_OK;
    }
#define LN_BWD(TT, NVV) hipLaunchKernelGGL((layernorm_bwd_kernel<TT, NVV>), grid, dim3(256), 0, st, (const TT*)dy, (const TT*)x, gamma, mean, rstd, (const TT*)dres, (TT*)dx, (TT*)dx_drop, dgamma, dbeta, dcol, M, D, drop)
    const int nv = (int)cdiv64(D, 256);
    if (dtype == EMO_F32) { if (nv <= 1) LN_BWD(float, 1); else if (nv == 2) LN_BWD(float, 2); else LN_BWD(float, 4); }
    else { if (nv <= 1) LN_BWD(bf16_t, 1); else if (nv == 2) LN_BWD(bf16_t, 2); else LN_BWD(bf16_t, 4); }
#undef LN_BWD
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

extern "C" int emo_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dcol, int dtype,
                                 int64_t M, int64_t D, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    return emo_layernorm_bwd_ws(dy, x, gamma, mean, rstd, dres, dx, dx_drop, dgamma, dbeta, dcol, dtype, M, D, p_drop, seed, offset, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------ dropout re-apply
template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n, DropCtx drop) {
    const int64_t n4 = n >> 2;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n4; it += (int64_t)gridDim.x * blockDim.x) {
        float v[4], dm[4];
        Vec4<T>::load(x + it * 4, v);
        drop_mult4(drop, (uint64_t)(it * 4), dm);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= dm[i];
        Vec4<T>::store(out + it * 4, v);
    }
}
extern "C" int emo_dropout_apply(const void* x, void* out, int dtype, int64_t n, float p_drop, uint64_t seed, uint64_t offset,
                                 emo_stream_t stream) {
    EMO_CHECK(x && out && (n & 3) == 0, "emo_dropout_apply: bad args (n must be a multiple of 4)");
    int64_t blocks = cdiv64(n >> 2, 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    DropCtx drop = make_drop(p_drop, seed, offset);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EMO_F32) hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)out, n, drop);
    else hipLaunchKernelGGL(dropout_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, n, drop);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// ================================================================================================ K9 cross-entropy
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ tgt, int64_t M,
                                                       int64_t V, int64_t ignore, float* __restrict__ row_lse, float* __restrict__ acc) {
    __shared__ float part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lsum = 0.f, lcnt = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const float* l = logits + row * V;
        float mx = -INFINITY;
        for (int64_t c = lane; c < V; c += 64) mx = fmaxf(mx, l[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int64_t c = lane; c < V; c += 64) s += expf(l[c] - mx);
        s = wave_sum(s);
        const float lse = mx + logf(s);
        if (lane == 0) {
            row_lse[row] = lse;
            const int64_t t = tgt[row];
            if (t != ignore) { lsum += lse - l[t]; lcnt += 1.f; }
        }
    }
    if (lane == 0) { part[0][wave] = lsum; part[1][wave] = lcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(acc + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}
// V <= 64 * NV: the row lives in registers (one global read instead of two) and every wave keeps two rows in flight
// (r01: 126 us for 131072 x 327 fp32 logits = 1.4 TB/s with the generic kernel: one row per wave, two dependent passes).
template <int NV>
__global__ __launch_bounds__(256) void xent_fwd_regs_kernel(const float* __restrict__ logits, const int64_t* __restrict__ tgt, int64_t M,
                                                            int64_t V, int64_t ignore, float* __restrict__ row_lse, float* __restrict__ acc) {
    __shared__ float part[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lsum = 0.f, lcnt = 0.f;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 2; r0 < M; r0 += nw * 2) {
        const int64_t r1 = r0 + 1 < M ? r0 + 1 : r0;
        const float* l0 = logits + r0 * V;
        const float* l1 = logits + r1 * V;
        float a[NV], b[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t c = lane + 64 * i;
            a[i] = c < V ? l0[c] : -INFINITY;
            b[i] = c < V ? l1[c] : -INFINITY;
        }
        float ma = a[0], mb = b[0];
#pragma unroll
        for (int i = 1; i < NV; ++i) { ma = fmaxf(ma, a[i]); mb = fmaxf(mb, b[i]); }
        ma = wave_max(ma); mb = wave_max(mb);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t c = lane + 64 * i;
            if (c < V) { sa += expf(a[i] - ma); sb += expf(b[i] - mb); }
        }
        sa = wave_sum(sa); sb = wave_sum(sb);
        const float lse_a = ma + logf(sa), lse_b = mb + logf(sb);
        if (lane == 0) {
            row_lse[r0] = lse_a;
            const int64_t t0 = tgt[r0];
            if (t0 != ignore) { lsum += lse_a - l0[t0]; lcnt += 1.f; }
            if (r0 + 1 < M) {
                row_lse[r1] = lse_b;
                const int64_t t1 = tgt[r1];
                if (t1 != ignore) { lsum += lse_b - l1[t1]; lcnt += 1.f; }
            }
        }
    }
    if (lane == 0) { part[0][wave] = lsum; part[1][wave] = lcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(acc + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

extern "C" int emo_xent_fwd(const float* logits, const int64_t* tgt, int64_t M, int64_t V, int64_t ignore_index, float* row_lse,
                            float* acc, emo_stream_t stream) {
    EMO_CHECK(logits && tgt && row_lse && acc, "emo_xent_fwd: null pointer");
    int64_t blocks = cdiv64(M, 4);
    if (blocks > 2048) blocks = 2048;
    if (V <= 512 && M >= 1024) {
        if (V <= 384) hipLaunchKernelGGL(xent_fwd_regs_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, tgt, M, V, ignore_index, row_lse, acc);
        else hipLaunchKernelGGL(xent_fwd_regs_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, tgt, M, V, ignore_index, row_lse, acc);
        EMO_LAUNCH_CHECK();
        return EMO_OK;
    }
    hipLaunchKernelGGL(xent_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, tgt, M, V, ignore_index, row_lse, acc);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ tgt,
                                                       const float* __restrict__ row_lse, const float* __restrict__ gscale,
                                                       T* __restrict__ dl, int64_t ld, int64_t M, int64_t V, int64_t ignore) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float gs = gscale[0];
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const int64_t t = tgt[row];
        const float lse = row_lse[row];
        const float keep = (t != ignore) ? gs : 0.f;
        for (int64_t c = lane; c < ld; c += 64) {
            float v = 0.f;
            if (c < V) v = (expf(logits[row * V + c] - lse) - (c == t ? 1.f : 0.f)) * keep;
            dl[row * ld + c] = from_f32<T>(v);
        }
    }
}
extern "C" int emo_xent_bwd(const float* logits, const int64_t* tgt, const float* row_lse, const float* gscale, void* dlogits,
                            int64_t ld_out, int dtype_out, int64_t M, int64_t V, int64_t ignore_index, emo_stream_t stream) {
    EMO_CHECK(logits && tgt && row_lse && gscale && dlogits && ld_out >= V, "emo_xent_bwd: bad args");
    int64_t blocks = cdiv64(M, 4);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype_out == EMO_F32) hipLaunchKernelGGL(xent_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, logits, tgt, row_lse, gscale, (float*)dlogits, ld_out, M, V, ignore_index);
    else hipLaunchKernelGGL(xent_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, logits, tgt, row_lse, gscale, (bf16_t*)dlogits, ld_out, M, V, ignore_index);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// ================================================================================================ argmax / accuracy
__device__ __forceinline__ void wave_argmax(float& v, int64_t& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o, 64);
        int64_t oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}
__device__ __forceinline__ int64_t row_argmax(const float* l, int64_t V, int lane) {
    float best = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t c = lane; c < V; c += 64) {
        float x = l[c];
        if (x > best || (x != x && bi == INT64_MAX)) { best = x; bi = c; }  // first max wins inside a lane (ascending c)
    }
    wave_argmax(best, bi);
    return bi;
}
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int64_t rows, int64_t V, int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int64_t bi = row_argmax(logits + row * V, V, lane);
    if (lane == 0) out[row] = bi;
}
extern "C" int emo_argmax(const float* logits, int64_t rows, int64_t V, int64_t* out, emo_stream_t stream) {
    EMO_CHECK(logits && out && rows > 0 && V > 0, "emo_argmax: bad args");
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, rows, V, out);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

__global__ __launch_bounds__(256) void accuracy_kernel(const float* __restrict__ logits, const int64_t* __restrict__ tgt,
                                                       const int64_t* __restrict__ chord, const int64_t* __restrict__ melody, int64_t M,
                                                       int64_t V, int64_t pad, unsigned long long* __restrict__ counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long c[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        int64_t bi = row_argmax(logits + row * V, V, lane);
        if (lane == 0) {
            const int64_t t = tgt[row];
            const unsigned long long ok = (bi == t);
            if (t != pad) { c[0]++; c[1] += ok; }
            if (chord && chord[row] == 1) { c[2]++; c[3] += ok; }
            if (melody && melody[row] == 1) { c[4]++; c[5] += ok; }
        }
    }
    if (lane == 0)
        for (int i = 0; i < 6; ++i)
            if (c[i]) atomicAdd(counts + i, c[i]);
}
// same idea for the accuracy counts: two rows per wave step, the row read once into registers
template <int NV>
__global__ __launch_bounds__(256) void accuracy_regs_kernel(const float* __restrict__ logits, const int64_t* __restrict__ tgt,
                                                            const int64_t* __restrict__ chord, const int64_t* __restrict__ melody, int64_t M,
                                                            int64_t V, int64_t pad, unsigned long long* __restrict__ counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long c[6] = {0, 0, 0, 0, 0, 0};
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 2; r0 < M; r0 += nw * 2) {
        const int64_t r1 = r0 + 1 < M ? r0 + 1 : r0;
        const float* l0 = logits + r0 * V;
        const float* l1 = logits + r1 * V;
        float a[NV], b[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t cc = lane + 64 * i;
            a[i] = cc < V ? l0[cc] : -INFINITY;
            b[i] = cc < V ? l1[cc] : -INFINITY;
        }
        float ba = -INFINITY, bb = -INFINITY;
        int64_t ia = INT64_MAX, ib = INT64_MAX;
#pragma unroll
        for (int i = 0; i < NV; ++i) {          // first max wins inside a lane (ascending column), NaN as in row_argmax
            const int64_t cc = lane + 64 * i;
            if (cc < V) {
                if (a[i] > ba || (a[i] != a[i] && ia == INT64_MAX)) { ba = a[i]; ia = cc; }
                if (b[i] > bb || (b[i] != b[i] && ib == INT64_MAX)) { bb = b[i]; ib = cc; }
            }
        }
        wave_argmax(ba, ia);
        wave_argmax(bb, ib);
        if (lane == 0) {
            const int64_t t0 = tgt[r0];
            const unsigned long long ok0 = (ia == t0);
            if (t0 != pad) { c[0]++; c[1] += ok0; }
            if (chord && chord[r0] == 1) { c[2]++; c[3] += ok0; }
            if (melody && melody[r0] == 1) { c[4]++; c[5] += ok0; }
            if (r0 + 1 < M) {
                const int64_t t1 = tgt[r1];
                const unsigned long long ok1 = (ib == t1);
                if (t1 != pad) { c[0]++; c[1] += ok1; }
                if (chord && chord[r1] == 1) { c[2]++; c[3] += ok1; }
                if (melody && melody[r1] == 1) { c[4]++; c[5] += ok1; }
            }
        }
    }
    if (lane == 0)
        for (int i = 0; i < 6; ++i)
            if (c[i]) atomicAdd(counts + i, c[i]);
}
extern "C" int emo_accuracy_counts(const float* logits, const int64_t* tgt, const int64_t* chord, const int64_t* melody, int64_t M,
                                   int64_t V, int64_t pad, int64_t* counts, emo_stream_t stream) {
    EMO_CHECK(logits && tgt && counts, "emo_accuracy_counts: null pointer");
    int64_t blocks = cdiv64(M, 4);
    if (blocks > 2048) blocks = 2048;
    if (V <= 512 && M >= 1024) {
        if (V <= 384) hipLaunchKernelGGL(accuracy_regs_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, tgt, chord, melody, M, V, pad, (unsigned long long*)counts);
        else hipLaunchKernelGGL(accuracy_regs_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, tgt, chord, melody, M, V, pad, (unsigned long long*)counts);
        EMO_LAUNCH_CHECK();
        return EMO_OK;
    }
    hipLaunchKernelGGL(accuracy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, tgt, chord, melody, M, V, pad, (unsigned long long*)counts);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// ================================================================================================ K10 nucleus sampling
// One 512-thread block per stream.  probs = softmax(l/temp) (fp32, as NumPy on fp32 logits); rank sort (descending, ties by
// ascending index) of <= 1024 entries in LDS; inclusive cumsum in np.cumsum's sequential fp32 order; last_index = SECOND position
// whose cumsum exceeds top_p (reference inference.py:93-94 keeps the crossing token — SURVEY F12); where the reference would raise
// IndexError (single crossing) all sorted tokens are kept.  Draw: cdf over the renormalised (f64) candidates, searchsorted(u, right).
// Serial work is two tight prefix scans (fp32 by wave 0, f64 by wave 1, concurrently); because both prefixes are monotone the
// crossing positions are COUNTS (#{cum <= top_p}, #{run <= target}) taken by all threads.  r01: 62 us (bitonic network + three
// branchy single-thread loops) -> see profiles.
__global__ __launch_bounds__(512) void nucleus_kernel(const float* __restrict__ logits, int64_t V, float temp, float top_p,
                                                      const float* __restrict__ u, int64_t* __restrict__ out, int64_t* __restrict__ step,
                                                      int64_t* __restrict__ seq, int64_t ld_seq, int64_t col0) {
    __shared__ __attribute__((aligned(16))) char lds[EMO_NUCLEUS_LDS];
    const int64_t kstep = step ? step[blockIdx.x] : 0;            // device-side step counter of this stream (hipGraph replay)
    const int64_t tok = emo_nucleus_draw(logits + (int64_t)blockIdx.x * V, V, temp, top_p, u[kstep * gridDim.x + blockIdx.x], lds, (int)threadIdx.x,
                                         [] { __syncthreads(); });
    if (threadIdx.x == 0) {
        out[blockIdx.x] = tok;
        if (seq) seq[(int64_t)blockIdx.x * ld_seq + col0 + kstep] = tok;
        if (step) step[blockIdx.x] = kstep + 1;
    }
}
extern "C" int emo_sample_nucleus(const float* logits, int64_t rows, int64_t V, float temperature, float top_p, const float* u,
                                  int64_t* out, emo_stream_t stream) {
    EMO_CHECK(logits && u && out && rows > 0, "emo_sample_nucleus: bad args");
    EMO_CHECK(V > 0 && V <= 1024, "emo_sample_nucleus: V must be <= 1024 (got %lld)", (long long)V);
    EMO_CHECK(temperature > 0.f, "emo_sample_nucleus: temperature must be > 0");
    hipLaunchKernelGGL(nucleus_kernel, dim3((unsigned)rows), dim3(512), 0, (hipStream_t)stream, logits, V, temperature, top_p, u, out, (int64_t*)nullptr,
                       (int64_t*)nullptr, (int64_t)0, (int64_t)0);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
extern "C" int emo_sample_nucleus_step(const float* logits, int64_t rows, int64_t V, float temperature, float top_p, const float* u_steps,
                                       int64_t* step, int64_t* seq, int64_t ld_seq, int64_t col0, int64_t* out, emo_stream_t stream) {
    EMO_CHECK(logits && u_steps && out && step && rows > 0, "emo_sample_nucleus_step: bad args");
    EMO_CHECK(V > 0 && V <= 1024, "emo_sample_nucleus_step: V must be <= 1024 (got %lld)", (long long)V);
    EMO_CHECK(temperature > 0.f, "emo_sample_nucleus_step: temperature must be > 0");
    hipLaunchKernelGGL(nucleus_kernel, dim3((unsigned)rows), dim3(512), 0, (hipStream_t)stream, logits, V, temperature, top_p, u_steps, out, step, seq,
                       ld_seq, col0);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// ================================================================================================ optimizer plumbing
// Deterministic: every block stores its partial sum in acc[1 + block] and the block that draws the last ticket adds the partials in
// index order into acc[0] — the same bits on every data-parallel rank for the same gradient (an atomicAdd of the partials sums in
// arrival order, and replicas then drift apart through the clip coefficient).  acc: EMO_SUMSQ_FLOATS floats, zero-initialised ONCE by
// the caller (the ticket word acc[1025] is returned to zero by the last block).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ acc) {
    __shared__ float red[4];
    __shared__ int last;
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n4; it += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = *(const f32x4*)(x + it * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(acc + 1 + blockIdx.x, red[0] + red[1] + red[2] + red[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add((unsigned*)(acc + 1025), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (last) {
        // the last block adds the partials in a FIXED tree (thread t: partials t, t + 256, ...; xor-shuffle tree; four wave sums in index order):
        // the same bits on every rank, and 4 loads in flight per thread instead of one thread walking 1024 dependent L2 round trips (r03: the
        // serial walk was ~100 us of the kernel's 150 us)
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        float t = 0.f;
        for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __hip_atomic_load(acc + 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            acc[0] = (red[0] + red[1]) + (red[2] + red[3]);
            __hip_atomic_store((unsigned*)(acc + 1025), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
extern "C" int emo_sumsq(const float* x, int64_t n, float* acc, emo_stream_t stream) {
    EMO_CHECK(x && acc && n > 0 && ((uintptr_t)x & 15) == 0, "emo_sumsq: bad args (x must be 16-B aligned)");
    int64_t blocks = cdiv64(cdiv64(n, 4), 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, acc);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float pre, const float* denom, float* coef) {
    if (denom) pre = denom[0] > 0.f ? pre / denom[0] : 0.f;   // no non-pad target on any rank: the summed gradient is exactly zero, keep it so
    const float total = sqrtf(sumsq[0]) * pre;            // norm of the (pre-scaled) gradient
    float c = max_norm / (total + 1e-6f);                 // torch.nn.utils.clip_grad_norm_
    coef[0] = (c < 1.f ? c : 1.f) * pre;
}
extern "C" int emo_clip_coef(const float* sumsq, float max_norm, float pre, const float* denom, float* coef, emo_stream_t stream) {
    EMO_CHECK(sumsq && coef, "emo_clip_coef: null pointer");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, pre, denom, coef);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ pb, int64_t n, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt, const float* __restrict__ gscale) {
    const float gs = gscale ? gscale[0] : 1.f;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);       // torch: exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;      // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        const float pi = p[i] - step_size * (mi / denom);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (pb) pb[i] = (bf16_t)pi;
    }
}
extern "C" int emo_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                             float beta2, float eps, int64_t step, const float* gscale, emo_stream_t stream) {
    EMO_CHECK(p && g && m && v && n > 0 && step > 0, "emo_adam_step: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = cdiv64(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps,
                       (float)bc1, (float)sqrt(bc2), gscale);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

template <typename S, typename Dd>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, Dd* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = from_f32<Dd>(to_f32<S>(s[i]));
}
extern "C" int emo_cast(const void* src, int sd, void* dst, int dd, int64_t n, emo_stream_t stream) {
    EMO_CHECK(src && dst && n > 0, "emo_cast: bad args");
    int64_t blocks = cdiv64(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (sd == EMO_F32 && dd == EMO_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (sd == EMO_BF16 && dd == EMO_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (sd == EMO_F32 && dd == EMO_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// qu = x + b1, qv = x + b2 (per-column fp32 biases, the sum rounded once to the storage type): the two biased query copies the key-tile
// and distance-window passes of the relative-position attention backward read (emo_relpos_attn_bwd_kv / _r), in ONE launch instead of a
// float copy, two adds and two casts (ATen, 5 launches per layer in the stage-1 step).  x: [M, D] view with row pitch ld; outputs contiguous.
template <typename T>
__global__ __launch_bounds__(256) void add_bias2_kernel(const T* __restrict__ x, int64_t ld, const float* __restrict__ b1, const float* __restrict__ b2,
                                                        T* __restrict__ o1, T* __restrict__ o2, int64_t M, int64_t D) {
    const int64_t n4 = M * (D >> 2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / (D >> 2), c = (i % (D >> 2)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = to_f32<T>(x[m * ld + c + e]);
            o1[m * D + c + e] = from_f32<T>(v + b1[c + e]);
            o2[m * D + c + e] = from_f32<T>(v + b2[c + e]);
        }
    }
}
extern "C" int emo_add_bias2(const void* x, int64_t ld, const float* b1, const float* b2, void* o1, void* o2, int dtype, int64_t M, int64_t D,
                             emo_stream_t stream) {
    EMO_CHECK(x && b1 && b2 && o1 && o2 && M > 0 && D > 0 && (D & 3) == 0 && ld >= D, "emo_add_bias2: bad args (D %% 4 == 0, ld >= D)");
    int64_t blocks = cdiv64(M * (D >> 2), 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EMO_F32) hipLaunchKernelGGL(add_bias2_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, ld, b1, b2, (float*)o1, (float*)o2, M, D);
    else hipLaunchKernelGGL(add_bias2_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, ld, b1, b2, (bf16_t*)o1, (bf16_t*)o2, M, D);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// ------------------------------------------------------------------------------------------------
// Batched bf16 transposes (the transposed weight mirrors of engine.ParamStore.wT: 4 per Performer layer, refreshed after every optimizer
// step): ONE launch for all matrices instead of one ATen copy kernel each (48 x 7.6 us per training step).  desc: n records of six int64
// {src, dst, rows, cols, first tile, tiles per row}; a block transposes one 64 x 64 tile through LDS (16-B global accesses on both sides).
__global__ __launch_bounds__(256) void transpose_batch_kernel(const int64_t* __restrict__ desc, int n) {
    __shared__ bf16_t tile[64][64 + 2];
    const int64_t b = blockIdx.x;
    int d = 0;
    while (d + 1 < n && desc[(d + 1) * 6 + 4] <= b) ++d;
    const bf16_t* src = (const bf16_t*)desc[d * 6];
    bf16_t* dst = (bf16_t*)desc[d * 6 + 1];
    const int64_t rows = desc[d * 6 + 2], cols = desc[d * 6 + 3], t = b - desc[d * 6 + 4], tpr = desc[d * 6 + 5];
    const int64_t r0 = (t / tpr) * 64, c0 = (t % tpr) * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {                       // 64 rows x 8 chunks of 8 columns
        const int idx = tid + 256 * it, r = idx >> 3, c = (idx & 7) * 8;
        if (r0 + r < rows && c0 + c + 7 < cols && (cols & 7) == 0) {
            const bf16x8 v = *(const bf16x8*)(src + (r0 + r) * cols + c0 + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[r][c + e] = v[e];
        } else {
            for (int e = 0; e < 8; ++e) tile[r][c + e] = (r0 + r < rows && c0 + c + e < cols) ? src[(r0 + r) * cols + c0 + c + e] : (bf16_t)0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {                       // output row = source column
        const int idx = tid + 256 * it, c = idx >> 3, r = (idx & 7) * 8;
        if (c0 + c >= cols) continue;
        if (r0 + r + 7 < rows && (rows & 7) == 0) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[r + e][c];
            *(bf16x8*)(dst + (c0 + c) * rows + r0 + r) = v;
        } else {
            for (int e = 0; e < 8; ++e)
                if (r0 + r + e < rows) dst[(c0 + c) * rows + r0 + r + e] = tile[r + e][c];
        }
    }
}

// `waiter` continues only after everything queued on `signaler` so far: one event record + one stream wait.  Events come from a small per-thread
// ring per device (a wait holds the record it was issued behind, so re-recording an event 32 forks later cannot disturb it).
extern "C" int emo_stream_wait(emo_stream_t waiter, emo_stream_t signaler) {
    constexpr int RING = 32, MAXDEV = 16;
    // (one ring per device: an event belongs to the device that was current when it was created, and recording it on another device's stream fails —
    // a thread that switches devices gets that device's own ring)
    static thread_local hipEvent_t ev[MAXDEV][RING] = {};
    static thread_local int next[MAXDEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
        emo_set_error("emo_stream_wait: device index %d outside [0, %d)", dev, MAXDEV);
        return EMO_ERR_INVALID;
    }
    hipEvent_t& e = ev[dev][next[dev]];
    next[dev] = (next[dev] + 1) % RING;
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        emo_set_error("emo_stream_wait: hipEventCreate failed");
        return EMO_ERR_LAUNCH;
    }
    if (hipEventRecord(e, (hipStream_t)signaler) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiter, e, 0) != hipSuccess) {
        emo_set_error("emo_stream_wait: %s", hipGetErrorString(hipGetLastError()));
        return EMO_ERR_LAUNCH;
    }
    return EMO_OK;
}

extern "C" int emo_transpose_batch(const int64_t* desc, int n, int64_t total_tiles, emo_stream_t stream) {
    EMO_CHECK(desc && n > 0 && total_tiles > 0, "emo_transpose_batch: bad args");
    hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, desc, n);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
