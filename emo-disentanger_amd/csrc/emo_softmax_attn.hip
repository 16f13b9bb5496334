// K5 — causal softmax attention of the GPT-2 backbone (HF GPT2Attention._attn): flash-style
// forward / backward that never materialises the T x T score matrix, plus a single-query decode
// kernel over a KV cache.  64-query x 64-key tiles, 4 waves; each wave owns 16 query rows (forward,
// dQ pass) or 16 key rows (dK/dV pass) so that all softmax statistics are lane-local + two
// cross-lane-group shuffles.  K/V/Q/dO tiles are staged in padded LDS images and every contraction
// is an NT product on MFMA (emo_lds_mma.h); masked key tiles above the diagonal are skipped.
// Attention-prob dropout is regenerated from (seed, offset, ((b*H+h)*T+i)*T+j).
#include "emo_lds_mma.h"

template <typename CT, int DH> struct SaDims {
    static constexpr int DHP = CMax<DH, Img<CT>::KMIN>::v;
    static constexpr int LDX = DHP + Img<CT>::PAD;
    static constexpr int LDC = 64 + Img<CT>::PAD;
};

// D[t] = sum_d dO[t][d] * O[t][d]   (64 rows, 4 threads per row)
template <typename CT, int DH>
__device__ __forceinline__ void rows_dot(float* Dv, const CT* __restrict__ a, const CT* __restrict__ b, int64_t ld, int valid, int tid) {
    const int r = tid >> 2, part = tid & 3;
    float s = 0.f;
    if (r < valid)
        for (int d = part * (DH / 4); d < (part + 1) * (DH / 4); ++d) s += to_f32<CT>(a[(int64_t)r * ld + d]) * to_f32<CT>(b[(int64_t)r * ld + d]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0) Dv[r] = s;
}

// =============================================================================================== forward
template <typename CT, int DH>
__global__ __launch_bounds__(256) void sattn_fwd_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                        CT* __restrict__ out, int64_t ld_out, float* __restrict__ lse_g, int64_t T, int64_t H,
                                                        DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Qi = (CT*)smem;          // [64][LDX]
    CT* Ki = Qi + 64 * LDX;      // [64][LDX]
    CT* VT = Ki + 64 * LDX;      // [DH][LDC]
    CT* Pi = VT + DH * LDC;      // [64][LDC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qt = (int64_t)gridDim.x - 1 - blockIdx.x;   // longest tiles first
    const int64_t bh = blockIdx.y, b = bh / H, h = bh % H;
    const int64_t q0 = qt * 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const int qvalid = (int)((T - q0) < 64 ? (T - q0) : 64);
    load_rows<CT, DH, DHP>(Qi, LDX, qb + q0 * ld, ld, 64, qvalid, tid);
    const float sqrt_dh = sqrtf((float)DH);
    const int tl = wave * 16 + (lane & 15);        // local query row
    const int64_t tg = q0 + tl;                    // global query index
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 oacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int64_t kt = 0; kt <= qt; ++kt) {
        const int64_t k0 = kt * 64;
        const int kvalid = (int)((T - k0) < 64 ? (T - k0) : 64);
        __syncthreads();
        load_rows<CT, DH, DHP>(Ki, LDX, kb + k0 * ld, ld, 64, kvalid, tid);
        load_rows_T<CT, DH>(VT, LDC, vb + k0 * ld, ld, 64, kvalid, tid);
        __syncthreads();
        float s[4][4];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mm16<CT>(acc, Ki, LDX, jt * 16, Qi, LDX, wave * 16, DHP, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t jg = k0 + jt * 16 + (lane >> 4) * 4 + r;
                float val = acc[r] / sqrt_dh;
                if (jg > tg || jg >= T) val = -INFINITY;
                s[jt][r] = val;
                mx = fmaxf(mx, val);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float m_new = fmaxf(m_run, mx);
        if (m_new == -INFINITY) m_new = 0.f;
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = expf(s[jt][r] - m_new);
                psum += p[r];
                if (drop.thr16) {
                    const int64_t jg = k0 + jt * 16 + (lane >> 4) * 4 + r;
                    p[r] *= drop_mult(drop, (uint64_t)((bh * T + tg) * T + jg));
                }
            }
            Img<CT>::store4(Pi + tl * LDC + jt * 16 + (lane >> 4) * 4, p[0], p[1], p[2], p[3]);
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < ND; ++i) oacc[i] *= alpha;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ND; ++i) mm16<CT>(oacc[i], VT, LDC, i * 16, Pi, LDC, wave * 16, 64, lane);
    }
    if (tg < T) {
        const float inv = 1.f / l_run;
        CT* ob = out + (b * T + tg) * ld_out + h * DH;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = i * 16 + (lane >> 4) * 4;
            Img<CT>::store4(ob + d0, oacc[i][0] * inv, oacc[i][1] * inv, oacc[i][2] * inv, oacc[i][3] * inv);
        }
        if ((lane >> 4) == 0) lse_g[bh * T + tg] = m_run + logf(l_run);
    }
}

// =============================================================================================== backward: dQ (per query tile)
template <typename CT, int DH>
__global__ __launch_bounds__(256) void sattn_bwd_dq_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                           const CT* __restrict__ out, const CT* __restrict__ dout, int64_t ld_out,
                                                           const float* __restrict__ lse_g, CT* __restrict__ dq, int64_t ld_d, int64_t T, int64_t H,
                                                           DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Qi = (CT*)smem;           // [64][LDX]
    CT* dOi = Qi + 64 * LDX;      // [64][LDX]
    CT* Ki = dOi + 64 * LDX;      // [64][LDX]
    CT* Vi = Ki + 64 * LDX;       // [64][LDX]
    CT* KT = Vi + 64 * LDX;       // [DH][LDC]
    CT* dSi = KT + DH * LDC;      // [64][LDC]
    float* Dv = (float*)(dSi + 64 * LDC);   // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qt = (int64_t)gridDim.x - 1 - blockIdx.x;
    const int64_t bh = blockIdx.y, b = bh / H, h = bh % H;
    const int64_t q0 = qt * 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* ob = out + (b * T) * ld_out + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    const int qvalid = (int)((T - q0) < 64 ? (T - q0) : 64);
    load_rows<CT, DH, DHP>(Qi, LDX, qb + q0 * ld, ld, 64, qvalid, tid);
    load_rows<CT, DH, DHP>(dOi, LDX, gb + q0 * ld_out, ld_out, 64, qvalid, tid);
    rows_dot<CT, DH>(Dv, gb + q0 * ld_out, ob + q0 * ld_out, ld_out, qvalid, tid);
    __syncthreads();
    const float sqrt_dh = sqrtf((float)DH);
    const int tl = wave * 16 + (lane & 15);
    const int64_t tg = q0 + tl;
    const float lse = tg < T ? lse_g[bh * T + tg] : 0.f;
    const float Dt = Dv[tl];
    f32x4 dqacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) dqacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int64_t kt = 0; kt <= qt; ++kt) {
        const int64_t k0 = kt * 64;
        const int kvalid = (int)((T - k0) < 64 ? (T - k0) : 64);
        __syncthreads();
        load_rows<CT, DH, DHP>(Ki, LDX, kb + k0 * ld, ld, 64, kvalid, tid);
        load_rows<CT, DH, DHP>(Vi, LDX, vb + k0 * ld, ld, 64, kvalid, tid);
        load_rows_T<CT, DH>(KT, LDC, kb + k0 * ld, ld, 64, kvalid, tid);
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            mm16<CT>(sa, Ki, LDX, jt * 16, Qi, LDX, wave * 16, DHP, lane);
            mm16<CT>(dp, Vi, LDX, jt * 16, dOi, LDX, wave * 16, DHP, lane);
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t jg = k0 + jt * 16 + (lane >> 4) * 4 + r;
                float p = 0.f, dpe = dp[r];
                if (jg <= tg && jg < T && tg < T) p = expf(sa[r] / sqrt_dh - lse);
                if (drop.thr16) dpe *= drop_mult(drop, (uint64_t)((bh * T + tg) * T + jg));
                ds[r] = p * (dpe - Dt);
            }
            Img<CT>::store4(dSi + tl * LDC + jt * 16 + (lane >> 4) * 4, ds[0], ds[1], ds[2], ds[3]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ND; ++i) mm16<CT>(dqacc[i], KT, LDC, i * 16, dSi, LDC, wave * 16, 64, lane);
    }
    if (tg < T) {
        CT* db = dq + (b * T + tg) * ld_d + h * DH;
        const float inv = 1.f / sqrt_dh;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = i * 16 + (lane >> 4) * 4;
            Img<CT>::store4(db + d0, dqacc[i][0] * inv, dqacc[i][1] * inv, dqacc[i][2] * inv, dqacc[i][3] * inv);
        }
    }
}

// =============================================================================================== backward: dK, dV (per key tile)
template <typename CT, int DH>
__global__ __launch_bounds__(256) void sattn_bwd_dkv_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                            const CT* __restrict__ out, const CT* __restrict__ dout, int64_t ld_out,
                                                            const float* __restrict__ lse_g, CT* __restrict__ dk, CT* __restrict__ dv, int64_t ld_d,
                                                            int64_t T, int64_t H, DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Ki = (CT*)smem;           // [64][LDX]
    CT* Vi = Ki + 64 * LDX;       // [64][LDX]
    CT* Qi = Vi + 64 * LDX;       // [64][LDX]
    CT* dOi = Qi + 64 * LDX;      // [64][LDX]
    CT* QT = dOi + 64 * LDX;      // [DH][LDC]
    CT* dOT = QT + DH * LDC;      // [DH][LDC]
    CT* PdT = dOT + DH * LDC;     // [64][LDC]
    CT* dST = PdT + 64 * LDC;     // [64][LDC]
    float* Dv = (float*)(dST + 64 * LDC);   // [64]
    float* Lv = Dv + 64;                    // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t kt = blockIdx.x;
    const int64_t bh = blockIdx.y, b = bh / H, h = bh % H;
    const int64_t k0 = kt * 64;
    const int64_t nqt = (T + 63) / 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* ob = out + (b * T) * ld_out + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    const int kvalid = (int)((T - k0) < 64 ? (T - k0) : 64);
    load_rows<CT, DH, DHP>(Ki, LDX, kb + k0 * ld, ld, 64, kvalid, tid);
    load_rows<CT, DH, DHP>(Vi, LDX, vb + k0 * ld, ld, 64, kvalid, tid);
    const float sqrt_dh = sqrtf((float)DH);
    const int jl = wave * 16 + (lane & 15);
    const int64_t jg = k0 + jl;
    f32x4 dkacc[ND], dvacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { dkacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int64_t qt = kt; qt < nqt; ++qt) {
        const int64_t q0 = qt * 64;
        const int qvalid = (int)((T - q0) < 64 ? (T - q0) : 64);
        __syncthreads();
        load_rows<CT, DH, DHP>(Qi, LDX, qb + q0 * ld, ld, 64, qvalid, tid);
        load_rows<CT, DH, DHP>(dOi, LDX, gb + q0 * ld_out, ld_out, 64, qvalid, tid);
        load_rows_T<CT, DH>(QT, LDC, qb + q0 * ld, ld, 64, qvalid, tid);
        load_rows_T<CT, DH>(dOT, LDC, gb + q0 * ld_out, ld_out, 64, qvalid, tid);
        rows_dot<CT, DH>(Dv, gb + q0 * ld_out, ob + q0 * ld_out, ld_out, qvalid, tid);
        if (tid < 64) Lv[tid] = (q0 + tid) < T ? lse_g[bh * T + q0 + tid] : 0.f;
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            mm16<CT>(sa, Qi, LDX, tt * 16, Ki, LDX, wave * 16, DHP, lane);
            mm16<CT>(dp, dOi, LDX, tt * 16, Vi, LDX, wave * 16, DHP, lane);
            float pd[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tl = tt * 16 + (lane >> 4) * 4 + r;
                const int64_t tg = q0 + tl;
                float p = 0.f, mult = 1.f;
                if (jg <= tg && tg < T && jg < T) p = expf(sa[r] / sqrt_dh - Lv[tl]);
                if (drop.thr16) mult = drop_mult(drop, (uint64_t)((bh * T + tg) * T + jg));
                pd[r] = p * mult;
                ds[r] = p * (dp[r] * mult - Dv[tl]);
            }
            Img<CT>::store4(PdT + jl * LDC + tt * 16 + (lane >> 4) * 4, pd[0], pd[1], pd[2], pd[3]);
            Img<CT>::store4(dST + jl * LDC + tt * 16 + (lane >> 4) * 4, ds[0], ds[1], ds[2], ds[3]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            mm16<CT>(dvacc[i], dOT, LDC, i * 16, PdT, LDC, wave * 16, 64, lane);
            mm16<CT>(dkacc[i], QT, LDC, i * 16, dST, LDC, wave * 16, 64, lane);
        }
    }
    if (jg < T) {
        CT* dkb = dk + (b * T + jg) * ld_d + h * DH;
        CT* dvb = dv + (b * T + jg) * ld_d + h * DH;
        const float inv = 1.f / sqrt_dh;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = i * 16 + (lane >> 4) * 4;
            Img<CT>::store4(dkb + d0, dkacc[i][0] * inv, dkacc[i][1] * inv, dkacc[i][2] * inv, dkacc[i][3] * inv);
            Img<CT>::store4(dvb + d0, dvacc[i][0], dvacc[i][1], dvacc[i][2], dvacc[i][3]);
        }
    }
}

// =============================================================================================== decode (one query per stream)
template <typename CT>
__global__ __launch_bounds__(256) void sattn_decode_kernel(const CT* __restrict__ q, int64_t ld_q, const CT* __restrict__ kc, const CT* __restrict__ vc,
                                                           int64_t T_max, const int64_t* __restrict__ lens, CT* __restrict__ out, int64_t ld_out,
                                                           int64_t H, int dh) {
    extern __shared__ float sc[];            // [T_max] scores, then [256] partials
    __shared__ float qs[128], red[4], part[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sh = blockIdx.x, s = sh / H, h = sh % H;
    const int64_t len = lens[s];
    const int64_t HD = H * dh;
    if (tid < dh) qs[tid] = to_f32<CT>(q[s * ld_q + h * dh + tid]);
    __syncthreads();
    const float sqrt_dh = sqrtf((float)dh);
    float mx = -INFINITY;
    for (int64_t j = tid; j < len; j += 256) {
        const CT* kr = kc + (s * T_max + j) * HD + h * dh;
        float a = 0.f;
        for (int d = 0; d < dh; ++d) a += qs[d] * to_f32<CT>(kr[d]);
        a = a / sqrt_dh;
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int64_t j = tid; j < len; j += 256) { float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int d = tid % dh, pt = tid / dh, npt = 256 / dh;
    float acc = 0.f;
    for (int64_t j = pt; j < len; j += npt) acc += sc[j] * to_f32<CT>(vc[(s * T_max + j) * HD + h * dh + d]);
    part[tid] = acc;
    __syncthreads();
    if (tid < dh) {
        float a = 0.f;
        for (int p = 0; p < npt; ++p) a += part[p * dh + tid];
        out[s * ld_out + h * dh + tid] = from_f32<CT>(a / tot);
    }
}

// =============================================================================================== host
template <typename CT, int DH> static size_t sa_fwd_lds() { typedef SaDims<CT, DH> D; return sizeof(CT) * (size_t)(2 * 64 * D::LDX + DH * D::LDC + 64 * D::LDC); }
template <typename CT, int DH> static size_t sa_dq_lds() { typedef SaDims<CT, DH> D; return sizeof(CT) * (size_t)(4 * 64 * D::LDX + DH * D::LDC + 64 * D::LDC) + 64 * sizeof(float); }
template <typename CT, int DH> static size_t sa_dkv_lds() { typedef SaDims<CT, DH> D; return sizeof(CT) * (size_t)(4 * 64 * D::LDX + 2 * DH * D::LDC + 2 * 64 * D::LDC) + 128 * sizeof(float); }

template <typename CT, int DH>
static int run_sattn(int which, const void* q, const void* k, const void* v, int64_t ld, const void* out, const void* dout, int64_t ld_out, float* lse,
                     void* dq, void* dk, void* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H, DropCtx drop, hipStream_t st) {
    dim3 grid((unsigned)((T + 63) / 64), (unsigned)(B * H));
    static bool attr = false;
    const size_t lfwd = sa_fwd_lds<CT, DH>(), ldq = sa_dq_lds<CT, DH>(), ldkv = sa_dkv_lds<CT, DH>();
    auto kfwd = sattn_fwd_kernel<CT, DH>;
    auto kdq = sattn_bwd_dq_kernel<CT, DH>;
    auto kdkv = sattn_bwd_dkv_kernel<CT, DH>;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kfwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lfwd);
        (void)hipFuncSetAttribute((const void*)kdq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldq);
        (void)hipFuncSetAttribute((const void*)kdkv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldkv);
        attr = true;
    }
    if (which == 0) {
        hipLaunchKernelGGL(kfwd, grid, dim3(256), lfwd, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, (CT*)out, ld_out,
                           lse, T, H, drop);
    } else {
        hipLaunchKernelGGL(kdq, grid, dim3(256), ldq, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, (const CT*)out,
                           (const CT*)dout, ld_out, lse, (CT*)dq, ld_d, T, H, drop);
        hipLaunchKernelGGL(kdkv, grid, dim3(256), ldkv, st, (const CT*)q, (const CT*)k, (const CT*)v, ld,
                           (const CT*)out, (const CT*)dout, ld_out, lse, (CT*)dk, (CT*)dv, ld_d, T, H, drop);
    }
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

static int dispatch_sattn(int which, int dtype, int64_t dh, const void* q, const void* k, const void* v, int64_t ld, const void* out, const void* dout,
                          int64_t ld_out, float* lse, void* dq, void* dk, void* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H, DropCtx drop,
                          hipStream_t st) {
#define SA_CASE(DHv)                                                                                                                      \
    if (dh == DHv) {                                                                                                                      \
        if (dtype == EMO_BF16) return run_sattn<bf16_t, DHv>(which, q, k, v, ld, out, dout, ld_out, lse, dq, dk, dv, ld_d, B, T, H, drop, st); \
        return run_sattn<float, DHv>(which, q, k, v, ld, out, dout, ld_out, lse, dq, dk, dv, ld_d, B, T, H, drop, st);                    \
    }
    SA_CASE(64)
    SA_CASE(32)
    SA_CASE(16)
#undef SA_CASE
    emo_set_error("softmax attention: unsupported d_head=%lld (built: 16, 32, 64)", (long long)dh);
    return EMO_ERR_UNSUPPORTED;
}

static int sattn_check(const void* q, const void* k, const void* v, int64_t ld, int64_t ld_out, int dtype, int64_t dh) {
    EMO_CHECK(q && k && v, "softmax attention: null pointer");
    EMO_CHECK(dtype == EMO_F32 || dtype == EMO_BF16, "softmax attention: bad dtype");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK(ld % ve == 0 && ld_out % ve == 0 && dh % ve == 0, "softmax attention: ld/dh must keep rows 16-B aligned");
    EMO_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "softmax attention: q/k/v must be 16-B aligned");
    return EMO_OK;
}

extern "C" int emo_softmax_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, void* out, int64_t ld_out, float* lse, int dtype, int64_t B,
                                    int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    int rc = sattn_check(q, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(out && lse && ((uintptr_t)out & 15) == 0, "emo_softmax_attn_fwd: bad out/lse");
    return dispatch_sattn(0, dtype, dh, q, k, v, ld, out, nullptr, ld_out, lse, nullptr, nullptr, nullptr, 0, B, T, H, make_drop(p_drop, seed, offset),
                          (hipStream_t)stream);
}

extern "C" int emo_softmax_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* out, const void* dout, int64_t ld_out,
                                    const float* lse, void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H, int64_t dh,
                                    float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    int rc = sattn_check(q, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(out && dout && lse && dq && dk && dv, "emo_softmax_attn_bwd: null pointer");
    EMO_CHECK(ld_d % 4 == 0 && (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)out | (uintptr_t)dout) & 15) == 0,
              "emo_softmax_attn_bwd: gradients must be 16-B aligned with ld_d %% 4 == 0");
    return dispatch_sattn(1, dtype, dh, q, k, v, ld, out, dout, ld_out, (float*)lse, dq, dk, dv, ld_d, B, T, H, make_drop(p_drop, seed, offset),
                          (hipStream_t)stream);
}

extern "C" int emo_softmax_attn_decode(const void* q, int64_t ld_q, const void* kcache, const void* vcache, int64_t T_max, const int64_t* lens, void* out,
                                       int64_t ld_out, int dtype, int64_t n_streams, int64_t H, int64_t dh, emo_stream_t stream) {
    EMO_CHECK(q && kcache && vcache && lens && out, "emo_softmax_attn_decode: null pointer");
    EMO_CHECK(dh <= 128 && 256 % dh == 0, "emo_softmax_attn_decode: d_head must divide 256 and be <= 128");
    EMO_CHECK(T_max * 4 <= 128 * 1024, "emo_softmax_attn_decode: T_max too large for the LDS score buffer");
    dim3 grid((unsigned)(n_streams * H));
    const size_t lds = (size_t)T_max * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EMO_F32) {
        static bool a = false;
        if (!a) { (void)hipFuncSetAttribute((const void*)sattn_decode_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); a = true; }
        hipLaunchKernelGGL(sattn_decode_kernel<float>, grid, dim3(256), lds, st, (const float*)q, ld_q, (const float*)kcache, (const float*)vcache, T_max, lens,
                           (float*)out, ld_out, H, (int)dh);
    } else {
        static bool a = false;
        if (!a) { (void)hipFuncSetAttribute((const void*)sattn_decode_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); a = true; }
        hipLaunchKernelGGL(sattn_decode_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)q, ld_q, (const bf16_t*)kcache, (const bf16_t*)vcache, T_max,
                           lens, (bf16_t*)out, ld_out, H, (int)dh);
    }
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
