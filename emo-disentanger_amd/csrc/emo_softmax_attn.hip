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
// bf16 speed mode works in the base-2 domain: scores are scaled by log2(e)/sqrt(dh) once, so every probability is ONE v_exp_f32 after one
// subtract / fma (the kernels are VALU-bound: ~20 VALU ops per score element against 0.25 clk of MFMA).  Parity mode (fp32) keeps expf and the
// reference's division by sqrt(dh).
#define EMO_LOG2E 1.4426950408889634f
#define EMO_LN2 0.6931471805599453f
template <typename CT> struct SaK { static constexpr int NS64 = 64 / Img<CT>::KSTEP; };   // MFMA steps over a 64-wide k range

// Occupancy: the kernels are latency-bound per wave (r02: halving the LDS traffic per flop with 32 query rows per wave made the forward 22 %
// SLOWER at 2 waves/SIMD, dropping one of the two barriers per key tile changed nothing, while capping the registers at 128 for a fourth
// wave per SIMD gave +13 %), so the bf16 forward and dQ kernels are bounded to 4 workgroups per CU.
template <typename CT, int DH>
__global__ __launch_bounds__(256, (sizeof(CT) == 2 ? 4 : 1)) void sattn_fwd_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                        CT* __restrict__ out, int64_t ld_out, float* __restrict__ lse_g, int64_t T, int64_t H,
                                                        DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16, NQ = DHP / Img<CT>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Qi = (CT*)smem;          // [64][LDX]
    CT* Ki = Qi + 64 * LDX;      // [64][LDX]
    CT* VT = Ki + 64 * LDX;      // fp32: V^T [DH][LDC]; bf16: V row-major [64][LDX] read through ds_read_b64_tr_b16
    constexpr bool TR = sizeof(CT) == 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qt = (int64_t)gridDim.y - 1 - blockIdx.y;   // longest tiles first
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;      // (b, h) fastest in dispatch order: the longest sweeps of the whole grid go first
    const int64_t q0 = qt * 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const int qvalid = (int)((T - q0) < 64 ? (T - q0) : 64);
    RowPrefetch<CT, DH, DHP, 64, 256> pk;
    RowPrefetch<CT, DH, DHP, 64, 256, !TR> pv;               // fp32: only stored transposed (row-fast mapping)
    {
        const int kv0 = (int)(T < 64 ? T : 64);
        pk.load(kb, ld, kv0, tid);
        pv.load(vb, ld, kv0, tid);
    }
    load_rows<CT, DH, DHP>(Qi, LDX, qb + q0 * ld, ld, 64, qvalid, tid);
    __syncthreads();
    typename Img<CT>::V qf[NQ];                              // this wave's 16 query rows stay in registers for the whole sweep
#pragma unroll
    for (int kk = 0; kk < NQ; ++kk) qf[kk] = Img<CT>::load(Qi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
    const float sqrt_dh = sqrtf((float)DH), rsqrt_dh = 1.f / sqrt_dh, c2 = rsqrt_dh * EMO_LOG2E;
    const int tl = wave * 16 + (lane & 15);        // local query row
    const int64_t tg = q0 + tl;                    // global query index
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 oacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int64_t kt = 0; kt <= qt; ++kt) {
        const int64_t k0 = kt * 64;
        __syncthreads();                                     // every wave is done with the previous K / V images
        pk.store_rows(Ki, LDX, tid);
        if constexpr (TR) pv.store_rows(VT, LDX, tid); else pv.store_T(VT, LDC, tid);
        if (kt < qt) {                                       // next key tile stays in flight during this tile's math
            const int64_t kn = k0 + 64;
            const int nv = (int)((T - kn) < 64 ? (T - kn) : 64);
            pk.load(kb + kn * ld, ld, nv, tid);
            pv.load(vb + kn * ld, ld, nv, tid);
        }
        __syncthreads();
        float s[4][4];
        float mx = -INFINITY;
        const bool diag = kt == qt;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (!(diag && jt > wave)) {                      // key sub-tile entirely above this wave's rows: masked anyway
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) acc = Img<CT>::mma(Img<CT>::load(Ki, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), qf[kk], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float val = sizeof(CT) == 2 ? acc[r] * c2 : acc[r] / sqrt_dh;           // bf16: base-2 domain; parity mode keeps the reference's division
                if (diag) {                              // only the diagonal tile touches the causal boundary or the end of the sequence
                    const int jl = jt * 16 + (lane >> 4) * 4 + r;
                    if (jl > tl || k0 + jl >= T) val = -INFINITY;
                }
                s[jt][r] = val;
                mx = fmaxf(mx, val);
            }
        }
        mx = rows4_max(mx);
        float m_new = fmaxf(m_run, mx);
        if (m_new == -INFINITY) m_new = 0.f;
        const float alpha = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(m_run - m_new) : __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16) drop_mult4(drop, (uint64_t)((bh * T + tg) * T + k0 + jt * 16 + (lane >> 4) * 4), dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(s[jt][r] - m_new) : Img<CT>::ex(s[jt][r] - m_new);
                psum += p;
                s[jt][r] = p * dm[r];
            }
        }
        l_run = l_run * alpha + psum;              // per-lane partial row sum: the four row groups are added once, after the sweep
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < ND; ++i) oacc[i] *= alpha;
        // O^T[d][t] += sum_j V^T[d][j] P[t][j] : P from registers, V^T read in the same permuted k order
#pragma unroll
        for (int st = 0; st < SaK<CT>::NS64; ++st) {
            const typename Img<CT>::V pf = reg_perm<CT>(s, st);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if constexpr (TR) oacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)VT, LDX, i * 16, st, lane), pf, oacc[i]);
                else oacc[i] = Img<CT>::mma(load_perm<CT>(VT, LDC, i * 16, st, lane), pf, oacc[i]);
            }
        }
    }
    l_run = rows4_sum(l_run);
    if (tg < T) {
        const float inv = 1.f / l_run;
        CT* ob = out + (b * T + tg) * ld_out + h * DH;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = i * 16 + (lane >> 4) * 4;
            Img<CT>::store4(ob + d0, oacc[i][0] * inv, oacc[i][1] * inv, oacc[i][2] * inv, oacc[i][3] * inv);
        }
        if ((lane >> 4) == 0) lse_g[bh * T + tg] = (sizeof(CT) == 2 ? m_run * EMO_LN2 : m_run) + logf(l_run);
    }
}

// =============================================================================================== backward: dQ (per query tile)
// Also exports delta[t] = dO[t].O[t] (one value per query row) for the dK/dV pass, which used to recompute it for every (key tile, query tile) pair.
template <typename CT, int DH>
__global__ __launch_bounds__(256, (sizeof(CT) == 2 ? 4 : 2)) void sattn_bwd_dq_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                           const CT* __restrict__ out, const CT* __restrict__ dout, int64_t ld_out,
                                                           const float* __restrict__ lse_g, float* __restrict__ delta_g, CT* __restrict__ dq, int64_t ld_d,
                                                           int64_t T, int64_t H, DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16, NQ = DHP / Img<CT>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Qi = (CT*)smem;           // [64][LDX]  (only to build the register fragments)
    CT* dOi = Qi + 64 * LDX;      // [64][LDX]
    CT* Ki = dOi + 64 * LDX;      // [64][LDX]
    CT* Vi = Ki + 64 * LDX;       // [64][LDX]
    CT* KT = Vi + 64 * LDX;       // [DH][LDC]
    float* Dv = (float*)(KT + (sizeof(CT) == 2 ? 0 : DH * LDC));   // [64]  (bf16: no K^T image)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qt = (int64_t)gridDim.y - 1 - blockIdx.y;
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;      // (b, h) fastest in dispatch order: the longest sweeps of the whole grid go first
    const int64_t q0 = qt * 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* ob = out + (b * T) * ld_out + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    const int qvalid = (int)((T - q0) < 64 ? (T - q0) : 64);
    constexpr bool TR = sizeof(CT) == 2;                     // bf16: K^T fragments come from Ki through ds_read_b64_tr_b16
    RowPrefetch<CT, DH, DHP, 64, 256> pk, pv;
    RowPrefetch<CT, DH, DHP, 64, 256, true> pkT;
    {
        const int kv0 = (int)(T < 64 ? T : 64);
        pk.load(kb, ld, kv0, tid); pv.load(vb, ld, kv0, tid);
        if constexpr (!TR) pkT.load(kb, ld, kv0, tid);
    }
    load_rows<CT, DH, DHP>(Qi, LDX, qb + q0 * ld, ld, 64, qvalid, tid);
    load_rows<CT, DH, DHP>(dOi, LDX, gb + q0 * ld_out, ld_out, 64, qvalid, tid);
    rows_dot<CT, DH>(Dv, gb + q0 * ld_out, ob + q0 * ld_out, ld_out, qvalid, tid);
    __syncthreads();
    typename Img<CT>::V qf[NQ], gf[NQ];
#pragma unroll
    for (int kk = 0; kk < NQ; ++kk) {
        qf[kk] = Img<CT>::load(Qi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
        gf[kk] = Img<CT>::load(dOi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
    }
    const float sqrt_dh = sqrtf((float)DH), rsqrt_dh = 1.f / sqrt_dh;
    const int tl = wave * 16 + (lane & 15);
    const int64_t tg = q0 + tl;
    const float lse = tg < T ? lse_g[bh * T + tg] : INFINITY;      // rows past the end: p = exp(-inf) = 0
    const float c2 = rsqrt_dh * EMO_LOG2E, lse2 = lse * EMO_LOG2E;
    const float Dt = Dv[tl];
    if (delta_g && tid < qvalid) delta_g[bh * T + q0 + tid] = Dv[tid];
    f32x4 dqacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) dqacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int64_t kt = 0; kt <= qt; ++kt) {
        const int64_t k0 = kt * 64;
        __syncthreads();
        pk.store_rows(Ki, LDX, tid);
        pv.store_rows(Vi, LDX, tid);
        if constexpr (!TR) pkT.store_T(KT, LDC, tid);
        if (kt < qt) {
            const int64_t kn = k0 + 64;
            const int nv = (int)((T - kn) < 64 ? (T - kn) : 64);
            pk.load(kb + kn * ld, ld, nv, tid); pv.load(vb + kn * ld, ld, nv, tid);
            if constexpr (!TR) pkT.load(kb + kn * ld, ld, nv, tid);
        }
        __syncthreads();
        const bool diag = kt == qt;
        float ds[4][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            if (!(diag && jt > wave)) {
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) {
                    sa = Img<CT>::mma(Img<CT>::load(Ki, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), qf[kk], sa);
                    dp = Img<CT>::mma(Img<CT>::load(Vi, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), gf[kk], dp);
                }
            }
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16) drop_mult4(drop, (uint64_t)((bh * T + tg) * T + k0 + jt * 16 + (lane >> 4) * 4), dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(sa[r] * c2 - lse2) : Img<CT>::ex(sa[r] / sqrt_dh - lse);
                if (diag) {                              // causal boundary / end of the sequence only on the diagonal tile
                    const int jl = jt * 16 + (lane >> 4) * 4 + r;
                    if (jl > tl || k0 + jl >= T) p = 0.f;
                }
                ds[jt][r] = p * (dp[r] * dm[r] - Dt);
            }
        }
        // dQ^T[d][t] += sum_j K^T[d][j] dS[t][j] : dS from registers
#pragma unroll
        for (int st = 0; st < SaK<CT>::NS64; ++st) {
            const typename Img<CT>::V df = reg_perm<CT>(ds, st);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if constexpr (TR) dqacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)Ki, LDX, i * 16, st, lane), df, dqacc[i]);
                else dqacc[i] = Img<CT>::mma(load_perm<CT>(KT, LDC, i * 16, st, lane), df, dqacc[i]);
            }
        }
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
__global__ __launch_bounds__(256, (sizeof(CT) == 2 ? 2 : 1)) void sattn_bwd_dkv_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                            const CT* __restrict__ dout, int64_t ld_out, const float* __restrict__ lse_g,
                                                            const float* __restrict__ delta_g, CT* __restrict__ dk, CT* __restrict__ dv, int64_t ld_d,
                                                            int64_t T, int64_t H, DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16, NQ = DHP / Img<CT>::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Ki = (CT*)smem;           // [64][LDX]  (Ki / Vi only to build the register fragments)
    CT* Vi = Ki + 64 * LDX;       // [64][LDX]
    CT* Qi = Vi + 64 * LDX;       // [64][LDX]
    CT* dOi = Qi + 64 * LDX;      // [64][LDX]
    CT* QT = dOi + 64 * LDX;      // [DH][LDC]
    CT* dOT = QT + DH * LDC;      // [DH][LDC]
    float* Dv = (float*)(QT + (sizeof(CT) == 2 ? 0 : 2 * DH * LDC));   // [64]  (bf16: no Q^T / dO^T images)
    float* Lv = Dv + 64;                    // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t kt = blockIdx.y;
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;      // (b, h) fastest in dispatch order: the longest sweeps of the whole grid go first
    const int64_t k0 = kt * 64;
    const int64_t nqt = (T + 63) / 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    const int kvalid = (int)((T - k0) < 64 ? (T - k0) : 64);
    constexpr bool TR = sizeof(CT) == 2;                     // bf16: Q^T / dO^T fragments come from Qi / dOi through ds_read_b64_tr_b16
    RowPrefetch<CT, DH, DHP, 64, 256> pq, pg;
    RowPrefetch<CT, DH, DHP, 64, 256, true> pqT, pgT;
    float pl = 0.f, pd_ = 0.f;                                // lse / delta of row q0 + tid (tid < 64)
    auto fetch = [&](int64_t q0n) {
        const int nv = (int)((T - q0n) < 64 ? (T - q0n) : 64);
        pq.load(qb + q0n * ld, ld, nv, tid);
        pg.load(gb + q0n * ld_out, ld_out, nv, tid);
        if constexpr (!TR) { pqT.load(qb + q0n * ld, ld, nv, tid); pgT.load(gb + q0n * ld_out, ld_out, nv, tid); }
        if (tid < 64) {
            const bool ok = q0n + tid < T;
            pl = ok ? lse_g[bh * T + q0n + tid] * (sizeof(CT) == 2 ? EMO_LOG2E : 1.f) : INFINITY;      // rows past the end: p = exp(-inf) = 0
            pd_ = ok ? delta_g[bh * T + q0n + tid] : 0.f;
        }
    };
    fetch(k0);                                                // first query tile = the diagonal one
    load_rows<CT, DH, DHP>(Ki, LDX, kb + k0 * ld, ld, 64, kvalid, tid);
    load_rows<CT, DH, DHP>(Vi, LDX, vb + k0 * ld, ld, 64, kvalid, tid);
    __syncthreads();
    typename Img<CT>::V kf[NQ], vf[NQ];
#pragma unroll
    for (int kk = 0; kk < NQ; ++kk) {
        kf[kk] = Img<CT>::load(Ki, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
        vf[kk] = Img<CT>::load(Vi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
    }
    const float sqrt_dh = sqrtf((float)DH), rsqrt_dh = 1.f / sqrt_dh, c2 = rsqrt_dh * EMO_LOG2E;
    const int jl = wave * 16 + (lane & 15);
    const int64_t jg = k0 + jl;
    f32x4 dkacc[ND], dvacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { dkacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int64_t qt = kt; qt < nqt; ++qt) {
        const int64_t q0 = qt * 64;
        __syncthreads();
        pq.store_rows(Qi, LDX, tid);
        pg.store_rows(dOi, LDX, tid);
        if constexpr (!TR) { pqT.store_T(QT, LDC, tid); pgT.store_T(dOT, LDC, tid); }
        if (tid < 64) { Lv[tid] = pl; Dv[tid] = pd_; }
        if (qt + 1 < nqt) fetch(q0 + 64);
        __syncthreads();
        const bool diag = qt == kt;
        float pd[4][4], ds[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            if (!(diag && tt < wave)) {                       // query sub-tile entirely before this wave's keys: masked anyway
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) {
                    sa = Img<CT>::mma(Img<CT>::load(Qi, LDX, tt * 16, kk * Img<CT>::KSTEP, lane), kf[kk], sa);
                    dp = Img<CT>::mma(Img<CT>::load(dOi, LDX, tt * 16, kk * Img<CT>::KSTEP, lane), vf[kk], dp);
                }
            }
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16 && (T & 3) == 0)                  // one hash per lane for the 4 rows (shared inside the key quad)
                drop_mult_col4(drop, (uint64_t)((bh * T + q0 + tt * 16 + (lane >> 4) * 4 + (lane & 3)) * T + (jg & ~(int64_t)3)), lane, dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tl = tt * 16 + (lane >> 4) * 4 + r;
                const int64_t tg = q0 + tl;
                float p = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(sa[r] * c2 - Lv[tl]) : Img<CT>::ex(sa[r] / sqrt_dh - Lv[tl]), mult = dm[r];
                if (diag && jl > tl) p = 0.f;            // causal boundary only on the diagonal tile (rows past the end carry lse = +inf)
                if (drop.thr16 && (T & 3) != 0) mult = drop_mult(drop, (uint64_t)((bh * T + tg) * T + jg));
                pd[tt][r] = p * mult;
                ds[tt][r] = p * (dp[r] * mult - Dv[tl]);
            }
        }
        // dV^T[d][j] += sum_t dO^T[d][t] Pd[t][j] ; dK^T[d][j] += sum_t Q^T[d][t] dS[t][j] : Pd / dS from registers
#pragma unroll
        for (int st = 0; st < SaK<CT>::NS64; ++st) {
            const typename Img<CT>::V pf = reg_perm<CT>(pd, st), df = reg_perm<CT>(ds, st);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if constexpr (TR) {
                    dvacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)dOi, LDX, i * 16, st, lane), pf, dvacc[i]);
                    dkacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)Qi, LDX, i * 16, st, lane), df, dkacc[i]);
                } else {
                    dvacc[i] = Img<CT>::mma(load_perm<CT>(dOT, LDC, i * 16, st, lane), pf, dvacc[i]);
                    dkacc[i] = Img<CT>::mma(load_perm<CT>(QT, LDC, i * 16, st, lane), df, dkacc[i]);
                }
            }
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
// One workgroup per (stream, head).  len = lens[s] + lens_off keys are valid INCLUDING the new token, whose k / v rows (k_new / v_new, may be
// NULL) the kernel appends to the cache itself at position len - 1 (saves two index_copy launches per layer).  Score pass: dh/VE lanes share
// a key row (16-B loads, a wave reads 64/(dh/VE) whole rows per instruction), shuffle-reduced; value pass: 16-B loads, 256/(dh/VE) keys in flight.
template <typename CT, int NT>
__global__ __launch_bounds__(NT) void sattn_decode_kernel(const CT* __restrict__ q, int64_t ld_q, CT* __restrict__ kc, CT* __restrict__ vc,
                                                           int64_t T_max, const int64_t* __restrict__ lens, int64_t lens_off,
                                                           const CT* __restrict__ k_new, const CT* __restrict__ v_new, int64_t ld_new,
                                                           CT* __restrict__ out, int64_t ld_out, int64_t H, int dh, int head_major) {
    constexpr int VE = 16 / sizeof(CT);
    extern __shared__ float sc[];            // [T_max] scores
    __shared__ float qs[128], red[NT / 64];
    __shared__ float part[NT * VE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sh = blockIdx.x, s = sh / H, h = sh % H;
    const int64_t len = lens[s] + lens_off;
    const int64_t HD = H * dh;
    // cache layout: [n, T_max, H * dh] (row stride H * dh: the layout of the projections) or, head_major, [n, H, T_max, dh] — a (stream, head)'s
    // keys are then ONE contiguous run of len * dh elements instead of dh-element pieces H * dh apart
    const int64_t rstride = head_major ? dh : HD;
    kc += head_major ? (s * H + h) * T_max * dh : s * T_max * HD + h * dh;
    vc += head_major ? (s * H + h) * T_max * dh : s * T_max * HD + h * dh;
    if (tid < dh) {
        qs[tid] = to_f32<CT>(q[s * ld_q + h * dh + tid]);
        if (k_new) {
            kc[(len - 1) * rstride + tid] = k_new[s * ld_new + h * dh + tid];
            vc[(len - 1) * rstride + tid] = v_new[s * ld_new + h * dh + tid];
        }
    }
    __syncthreads();
    const float sqrt_dh = sqrtf((float)dh);
    const int LPR = dh / VE;                 // lanes per key row (power of two: dh in {16,32,64,128})
    const int rl = tid / LPR, cl = (tid % LPR) * VE, RPB = NT / LPR;
    float qv[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) qv[e] = qs[cl + e];
    // Both sweeps keep SU key / value rows per thread in flight (r05: with one 16-B load per thread and iteration a CU had 4 KB outstanding
    // against the ~40 KB that cover the HBM round trip at its share of the bandwidth — rocprofv3 of the GPT-2 generation: 18 us per layer at
    // ~300 keys, 40 % of the token step).  Rows past `len` are clamped to the last valid row and discarded.
    constexpr int SU = NT >= 1024 ? 4 : 8;
    typedef typename std::conditional<sizeof(CT) == 2, bf16x8, f32x4>::type RowV;
    float mx = -INFINITY;
    for (int64_t j0 = 0; j0 < len; j0 += (int64_t)RPB * SU) {
        RowV kv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int64_t j = j0 + (int64_t)u * RPB + rl;
            kv[u] = *(const RowV*)(kc + (j < len ? j : len - 1) * rstride + cl);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int64_t j = j0 + (int64_t)u * RPB + rl;
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < VE; ++e) a += qv[e] * (float)kv[u][e];
            for (int o = LPR >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            a = a / sqrt_dh;
            if (j < len) {
                if ((tid % LPR) == 0) sc[j] = a;
                mx = fmaxf(mx, a);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) mx = fmaxf(mx, red[i]);
    __syncthreads();
    float sum = 0.f;
    for (int64_t j = tid; j < len; j += NT) { float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) tot += red[i];
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    for (int64_t j0 = 0; j0 < len; j0 += (int64_t)RPB * SU) {
        RowV vv[SU];
        float pw[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int64_t j = j0 + (int64_t)u * RPB + rl;
            vv[u] = *(const RowV*)(vc + (j < len ? j : len - 1) * rstride + cl);
            pw[u] = j < len ? sc[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u)
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] += pw[u] * (float)vv[u][e];
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) part[rl * dh + cl + e] = acc[e];
    __syncthreads();
    if (tid < dh) {
        float a = 0.f;
        for (int p = 0; p < RPB; ++p) a += part[p * dh + tid];
        out[s * ld_out + h * dh + tid] = from_f32<CT>(a / tot);
    }
}

// =============================================================================================== host
template <typename CT, int DH> static size_t sa_fwd_lds() { typedef SaDims<CT, DH> D; return sizeof(CT) * (size_t)(2 * 64 * D::LDX + CMax<DH * D::LDC, 64 * D::LDX>::v); }
template <typename CT, int DH> static size_t sa_dq_lds() { typedef SaDims<CT, DH> D; return sizeof(CT) * (size_t)(4 * 64 * D::LDX + (sizeof(CT) == 2 ? 0 : DH * D::LDC)) + 64 * sizeof(float); }
template <typename CT, int DH> static size_t sa_dkv_lds() { typedef SaDims<CT, DH> D; return sizeof(CT) * (size_t)(4 * 64 * D::LDX + (sizeof(CT) == 2 ? 0 : 2 * DH * D::LDC)) + 128 * sizeof(float); }

// emo_softmax_attn32.hip: bf16 / d_head 64 / T % 128 == 0 kernels on 32 x 32 x 16 tiles (false: not covered -> the kernels of this file)
bool emo_sattn32_dkv_try(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dout, int64_t ld_out, const float* lse, const float* delta,
                         bf16_t* dk, bf16_t* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H, DropCtx drop, const uint32_t* keep, hipStream_t st);
bool emo_sattn32_try(int which, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ld_out, float* lse, int64_t B, int64_t T,
                     int64_t H, DropCtx drop, uint32_t* keep, hipStream_t st);
int64_t emo_sattn32_keep_bytes(int64_t B, int64_t T, int64_t H, int64_t dh, float p_drop);
bool emo_sattn32_dq_try(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* out, const bf16_t* dout, int64_t ld_out, const float* lse,
                        float* delta, bf16_t* dq, int64_t ld_d, int64_t B, int64_t T, int64_t H, DropCtx drop, const uint32_t* keep, hipStream_t st);

template <typename CT, int DH>
static int run_sattn(int which, const void* q, const void* k, const void* v, int64_t ld, const void* out, const void* dout, int64_t ld_out, float* lse,
                     float* delta, void* dq, void* dk, void* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H, DropCtx drop, uint32_t* keep, hipStream_t st) {
    if constexpr (sizeof(CT) == 2 && DH == 64) {
        if (which == 0 && emo_sattn32_try(0, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, (bf16_t*)out, ld_out, lse, B, T, H, drop, keep, st)) {
            EMO_LAUNCH_CHECK();
            return EMO_OK;
        }
    }
    // keep words exist only in the 32 x 32 kernels' layout: a forward that cannot write them must not pretend to, a backward must not half-use them
    EMO_CHECK(!(which == 0 && keep), "softmax attention: keep words requested but the call is not served by the 32 x 32 kernels (bf16, d_head 64, T %% 128 == 0, 16-B aligned views)");
    dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64));
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
        bool dq32 = false;
        if constexpr (sizeof(CT) == 2 && DH == 64)
            dq32 = emo_sattn32_dq_try((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, (const bf16_t*)out, (const bf16_t*)dout, ld_out, lse, delta, (bf16_t*)dq,
                                      ld_d, B, T, H, drop, keep, st);
        if (!dq32)
            hipLaunchKernelGGL(kdq, grid, dim3(256), ldq, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, (const CT*)out,
                               (const CT*)dout, ld_out, lse, delta, (CT*)dq, ld_d, T, H, drop);
        bool dkv32 = false;
        if constexpr (sizeof(CT) == 2 && DH == 64)
            dkv32 = emo_sattn32_dkv_try((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, (const bf16_t*)dout, ld_out, lse, delta, (bf16_t*)dk, (bf16_t*)dv, ld_d,
                                        B, T, H, drop, keep, st);
        EMO_CHECK(dkv32 || !keep, "softmax attention backward: keep words given but the 32 x 32 dK/dV kernel does not serve the call");
        if (!dkv32)
            hipLaunchKernelGGL(kdkv, grid, dim3(256), ldkv, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, (const CT*)dout, ld_out, lse, delta, (CT*)dk,
                               (CT*)dv, ld_d, T, H, drop);
    }
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

static int dispatch_sattn(int which, int dtype, int64_t dh, const void* q, const void* k, const void* v, int64_t ld, const void* out, const void* dout,
                          int64_t ld_out, float* lse, float* delta, void* dq, void* dk, void* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H,
                          DropCtx drop, uint32_t* keep, hipStream_t st) {
#define SA_CASE(DHv)                                                                                                                      \
    if (dh == DHv) {                                                                                                                      \
        if (dtype == EMO_BF16) return run_sattn<bf16_t, DHv>(which, q, k, v, ld, out, dout, ld_out, lse, delta, dq, dk, dv, ld_d, B, T, H, drop, keep, st); \
        return run_sattn<float, DHv>(which, q, k, v, ld, out, dout, ld_out, lse, delta, dq, dk, dv, ld_d, B, T, H, drop, keep, st);             \
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

extern "C" int64_t emo_softmax_attn_keep_bytes(int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, float p_drop) {
    return dtype == EMO_BF16 ? emo_sattn32_keep_bytes(B, T, H, dh, p_drop) : 0;
}

extern "C" int emo_softmax_attn_fwd_keep(const void* q, const void* k, const void* v, int64_t ld, void* out, int64_t ld_out, float* lse, int dtype, int64_t B,
                                         int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint64_t offset, void* keep, int64_t keep_bytes,
                                         emo_stream_t stream) {
    int rc = sattn_check(q, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(out && lse && ((uintptr_t)out & 15) == 0, "emo_softmax_attn_fwd: bad out/lse");
    if (keep) {
        const int64_t need = emo_softmax_attn_keep_bytes(dtype, B, T, H, dh, p_drop);
        EMO_CHECK(need > 0 && keep_bytes >= need && ((uintptr_t)keep & 15) == 0, "emo_softmax_attn_fwd_keep: keep buffer of %lld bytes, need %lld (emo_softmax_attn_keep_bytes; 0 = not available for this call)",
                  (long long)keep_bytes, (long long)need);
    }
    return dispatch_sattn(0, dtype, dh, q, k, v, ld, out, nullptr, ld_out, lse, nullptr, nullptr, nullptr, nullptr, 0, B, T, H, make_drop(p_drop, seed, offset),
                          (uint32_t*)keep, (hipStream_t)stream);
}
extern "C" int emo_softmax_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, void* out, int64_t ld_out, float* lse, int dtype, int64_t B,
                                    int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    return emo_softmax_attn_fwd_keep(q, k, v, ld, out, ld_out, lse, dtype, B, T, H, dh, p_drop, seed, offset, nullptr, 0, stream);
}

extern "C" int emo_softmax_attn_bwd_keep(const void* q, const void* k, const void* v, int64_t ld, const void* out, const void* dout, int64_t ld_out,
                                         const float* lse, float* delta_ws, void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H,
                                         int64_t dh, float p_drop, uint64_t seed, uint64_t offset, const void* keep, int64_t keep_bytes, emo_stream_t stream) {
    int rc = sattn_check(q, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(out && dout && lse && delta_ws && dq && dk && dv, "emo_softmax_attn_bwd: null pointer");
    EMO_CHECK(ld_d % 4 == 0 && (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)out | (uintptr_t)dout) & 15) == 0,
              "emo_softmax_attn_bwd: gradients must be 16-B aligned with ld_d %% 4 == 0");
    if (keep) {
        const int64_t need = emo_softmax_attn_keep_bytes(dtype, B, T, H, dh, p_drop);
        EMO_CHECK(need > 0 && keep_bytes >= need && ((uintptr_t)keep & 15) == 0, "emo_softmax_attn_bwd_keep: keep buffer of %lld bytes, need %lld", (long long)keep_bytes, (long long)need);
    }
    return dispatch_sattn(1, dtype, dh, q, k, v, ld, out, dout, ld_out, (float*)lse, delta_ws, dq, dk, dv, ld_d, B, T, H, make_drop(p_drop, seed, offset),
                          (uint32_t*)keep, (hipStream_t)stream);
}
extern "C" int emo_softmax_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* out, const void* dout, int64_t ld_out,
                                    const float* lse, float* delta_ws, void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H,
                                    int64_t dh, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    return emo_softmax_attn_bwd_keep(q, k, v, ld, out, dout, ld_out, lse, delta_ws, dq, dk, dv, ld_d, dtype, B, T, H, dh, p_drop, seed, offset, nullptr, 0, stream);
}

extern "C" int emo_softmax_attn_decode_layout(const void* q, int64_t ld_q, void* kcache, void* vcache, int64_t T_max, const int64_t* lens, int64_t lens_off,
                                              const void* k_new, const void* v_new, int64_t ld_new, void* out, int64_t ld_out, int dtype, int64_t n_streams,
                                              int64_t H, int64_t dh, int head_major, emo_stream_t stream);
extern "C" int emo_softmax_attn_decode(const void* q, int64_t ld_q, void* kcache, void* vcache, int64_t T_max, const int64_t* lens, int64_t lens_off,
                                       const void* k_new, const void* v_new, int64_t ld_new, void* out, int64_t ld_out, int dtype, int64_t n_streams,
                                       int64_t H, int64_t dh, emo_stream_t stream) {
    return emo_softmax_attn_decode_layout(q, ld_q, kcache, vcache, T_max, lens, lens_off, k_new, v_new, ld_new, out, ld_out, dtype, n_streams, H, dh, 0, stream);
}
extern "C" int emo_softmax_attn_decode_layout(const void* q, int64_t ld_q, void* kcache, void* vcache, int64_t T_max, const int64_t* lens, int64_t lens_off,
                                              const void* k_new, const void* v_new, int64_t ld_new, void* out, int64_t ld_out, int dtype, int64_t n_streams,
                                              int64_t H, int64_t dh, int head_major, emo_stream_t stream) {
    EMO_CHECK(q && kcache && vcache && lens && out, "emo_softmax_attn_decode: null pointer");
    EMO_CHECK(dh == 16 || dh == 32 || dh == 64 || dh == 128, "emo_softmax_attn_decode: d_head must be 16, 32, 64 or 128");
    EMO_CHECK(!k_new == !v_new, "emo_softmax_attn_decode: k_new and v_new go together");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK((((uintptr_t)kcache | (uintptr_t)vcache) & 15) == 0 && (H * dh) % ve == 0, "emo_softmax_attn_decode: caches must be 16-B aligned");
    EMO_CHECK(T_max * 4 <= 128 * 1024, "emo_softmax_attn_decode: T_max too large for the LDS score buffer");
    dim3 grid((unsigned)(n_streams * H));
    const size_t lds = (size_t)T_max * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    // one workgroup per (stream, head) = at most one per CU at 32 streams x 8 heads: 16 waves instead of 4 give the CU's memory pipe four
    // times the requests in flight (r05; EMO_SATTN_DECODE_NT=256 keeps the 4-wave instance).  GPT-2 generation, 32 streams x 2048: 25 us per
    // layer with 8 rows per thread in flight, 20 us with 16 waves (the CU's own ~10 B/clk HBM rate gives 12 us for its 270 KB at the mean
    // context); a single-sweep online-softmax variant (K and V rows in flight together, no score buffer) measured the same 20 us and was dropped.
    // (r06: a single-sweep form — lens, q, the new rows and the first 1024 key AND value rows requested speculatively in the first round trip,
    // thread-local online softmax, one merge at the end — measured 0.675 against 0.558 ms per token step of the 32 x 2048 generation: it moves
    // 1.5 x the bytes (every step fetches 1024 rows per (stream, head) whatever its context) and the kernel is bound by the CU's fetch rate, not
    // by its chain of round trips; removed, tools/ab_gen_gpt2.sh kept.  Requesting the next iteration's rows before the current ones are used
    // (software-pipelined sweeps) measured 0.585 against 0.566: more requests in flight per CU do not help either)
    int nt = 1024;
    { const char* e = getenv("EMO_SATTN_DECODE_NT"); if (e && atoi(e) == 256) nt = 256; }
    if (T_max * 4 > 96 * 1024) nt = 256;                             // (two-pass kernel: score buffer + the 1024-thread instance's partial sums must fit the LDS)
#define SD_LAUNCH(CTv, NTv)                                                                                                                \
    do {                                                                                                                                   \
        static bool a = false;                                                                                                             \
        if (!a) { (void)hipFuncSetAttribute((const void*)sattn_decode_kernel<CTv, NTv>, hipFuncAttributeMaxDynamicSharedMemorySize, (NTv == 1024 ? 96 : 128) * 1024); a = true; } \
        hipLaunchKernelGGL((sattn_decode_kernel<CTv, NTv>), grid, dim3(NTv), lds, st, (const CTv*)q, ld_q, (CTv*)kcache, (CTv*)vcache, T_max, lens, lens_off, \
                           (const CTv*)k_new, (const CTv*)v_new, ld_new, (CTv*)out, ld_out, H, (int)dh, head_major);                      \
    } while (0)
    if (dtype == EMO_F32) { if (nt == 1024) SD_LAUNCH(float, 1024); else SD_LAUNCH(float, 256); }
    else { if (nt == 1024) SD_LAUNCH(bf16_t, 1024); else SD_LAUNCH(bf16_t, 256); }
#undef SD_LAUNCH
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// =============================================================================================== relative-position attention (stage-1 Transformer-XL)
// RelPartialLearnableMultiHeadAttn (stage1_compose/model/optimus_txl_decoder.py:301-391), forward + one-query decode.
//   score[i][j] = ((q_i + u).k_j + (q_i + v).R[dist])/sqrt(dh),  dist = i - j >= 0   (u = r_w_bias, v = r_r_bias, R = r_net(pos_emb))
// The reference materialises BD = (q+v).R^T for all positions and re-labels it with the pad-and-view trick `_rel_shift` (:280-293); R only
// depends on the distance, so the caller passes r_dist [n_dist, H*dh] indexed BY DISTANCE and the kernel gathers.  Tile kernel = the flash
// forward above plus, per key tile, one more product P2[c][t] = Rwin[c].(q_t + v) over the 80 distances c a wave's 16 rows can see; the
// skew c = t - j + 63 moves P2 between lanes through a small wave-private LDS buffer.  Probabilities: softmax -> dropout -> p / (sum p + 1e-8)
// (:361-363); the kernel tracks l = sum e and E = sum drop(e) online, out = sum drop(e) v / (E + 1e-8 l); zden = E / l + 1e-8 saved with lse.
template <typename CT, int DH>
__global__ __launch_bounds__(256) void relattn_fwd_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                          const CT* __restrict__ rd, int64_t ld_r, int64_t n_dist, const float* __restrict__ ub,
                                                          const float* __restrict__ vb_, CT* __restrict__ out, int64_t ld_out, float* __restrict__ lse_g,
                                                          float* __restrict__ zden_g, int64_t T, int64_t H, DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16, NQ = DHP / Img<CT>::KSTEP, SKW = 84;
    constexpr bool TR = sizeof(CT) == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Qi = (CT*)smem;                       // [64][LDX]  (q + u, then q + v: only to build the register fragments)
    CT* Ki = Qi + 64 * LDX;                   // [64][LDX]
    CT* VT = Ki + 64 * LDX;                   // fp32: V^T [DH][LDC]; bf16: V row-major [64][LDX]
    CT* Rw = VT + CMax<DH * LDC, 64 * LDX>::v;   // [128][LDX]  R rows of the distance window d0 .. d0+127
    float* sk = (float*)(Rw + 128 * LDX);     // [4 waves][16 t][SKW]  skew buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qt = (int64_t)gridDim.y - 1 - blockIdx.y;
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;      // (b, h) fastest in dispatch order: the longest sweeps of the whole grid go first
    const int64_t q0 = qt * 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* rb = rd + h * DH;
    const int qvalid = (int)((T - q0) < 64 ? (T - q0) : 64);
    RowPrefetch<CT, DH, DHP, 64, 256> pk;
    RowPrefetch<CT, DH, DHP, 64, 256, !TR> pv;
    constexpr int VE = 16 / sizeof(CT), CH = DH / VE, RNI = (128 * CH + 255) / 256;
    CT rr[RNI][VE];                           // prefetched R window rows
    auto fetch_r = [&](int64_t d0) {
#pragma unroll
        for (int i = 0; i < RNI; ++i) {
            const int it = tid + 256 * i;
            const int row = it / CH, c = (it % CH) * VE;
            const int64_t dist = d0 + row;
#pragma unroll
            for (int e = 0; e < VE; ++e) rr[i][e] = from_f32<CT>(0.f);
            if (it < 128 * CH && dist >= 0 && dist < n_dist) {
                if constexpr (sizeof(CT) == 2) *(bf16x8*)rr[i] = *(const bf16x8*)(rb + dist * ld_r + c);
                else *(f32x4*)rr[i] = *(const f32x4*)(rb + dist * ld_r + c);
            }
        }
    };
    auto store_r = [&]() {
#pragma unroll
        for (int i = 0; i < RNI; ++i) {
            const int it = tid + 256 * i;
            if (it < 128 * CH) {
                const int row = it / CH, c = (it % CH) * VE;
                if constexpr (sizeof(CT) == 2) *(bf16x8*)(Rw + row * LDX + c) = *(const bf16x8*)rr[i];
                else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) Rw[row * LDX + c + e] = rr[i][e];
                }
            }
        }
        if constexpr (DHP > DH) {
            for (int it = tid; it < 128 * (DHP - DH); it += 256) Rw[(it / (DHP - DH)) * LDX + DH + it % (DHP - DH)] = from_f32<CT>(0.f);
        }
    };
    {
        const int kv0 = (int)(T < 64 ? T : 64);
        pk.load(kb, ld, kv0, tid);
        pv.load(vb, ld, kv0, tid);
        fetch_r(q0 - 63);
    }
    // fragments of (q + u) and (q + v) for this wave's 16 query rows
    typename Img<CT>::V quf[NQ], qvf[NQ];
    for (int pass = 0; pass < 2; ++pass) {
        const float* bias = (pass == 0 ? ub : vb_) + h * DH;
        __syncthreads();
        for (int it = tid; it < 64 * DHP; it += 256) {
            const int r = it / DHP, d = it % DHP;
            float x = 0.f;
            if (r < qvalid && d < DH) x = to_f32<CT>(qb[(q0 + r) * ld + d]) + bias[d];
            Qi[r * LDX + d] = from_f32<CT>(x);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) {
            if (pass == 0) quf[kk] = Img<CT>::load(Qi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
            else qvf[kk] = Img<CT>::load(Qi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
        }
    }
    const float sqrt_dh = sqrtf((float)DH), rsqrt_dh = 1.f / sqrt_dh, c2 = rsqrt_dh * EMO_LOG2E;
    const int tl = wave * 16 + (lane & 15);
    const int64_t tg = q0 + tl;
    float m_run = -INFINITY, l_run = 0.f, e_run = 0.f;
    f32x4 oacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* skw = sk + (wave * 16 + (lane & 15)) * SKW;       // this lane's query row of the wave-private skew buffer

    for (int64_t kt = 0; kt <= qt; ++kt) {
        const int64_t k0 = kt * 64;
        __syncthreads();
        pk.store_rows(Ki, LDX, tid);
        if constexpr (TR) pv.store_rows(VT, LDX, tid); else pv.store_T(VT, LDC, tid);
        store_r();
        if (kt < qt) {
            const int64_t kn = k0 + 64;
            const int nv = (int)((T - kn) < 64 ? (T - kn) : 64);
            pk.load(kb + kn * ld, ld, nv, tid);
            pv.load(vb + kn * ld, ld, nv, tid);
            fetch_r(q0 - kn - 63);
        }
        __syncthreads();
        // P2[c][t] = Rwin[c].(q_t + v) for the window rows this wave can reach: c = t_l - j_l + 63 in [16w, 16w + 78] -> tiles w .. w+4
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) a2 = Img<CT>::mma(Img<CT>::load(Rw, LDX, (wave + ci) * 16, kk * Img<CT>::KSTEP, lane), qvf[kk], a2);
            *(f32x4*)(skw + ci * 16 + (lane >> 4) * 4) = a2;                 // sk[t][c - 16w]
        }
        __builtin_amdgcn_wave_barrier();
        float s[4][4];
        float mx = -INFINITY;
        const bool diag = kt == qt;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (!(diag && jt > wave)) {
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) acc = Img<CT>::mma(Img<CT>::load(Ki, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), quf[kk], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = jt * 16 + (lane >> 4) * 4 + r;
                const float bd = skw[(lane & 15) + 63 - jl];               // c - 16w = (16w + (l&15)) - jl + 63 - 16w
                float val = sizeof(CT) == 2 ? (acc[r] + bd) * c2 : (acc[r] + bd) / sqrt_dh;
                if (diag && (jl > tl || k0 + jl >= T)) val = -INFINITY;
                s[jt][r] = val;
                mx = fmaxf(mx, val);
            }
        }
        __builtin_amdgcn_wave_barrier();
        mx = rows4_max(mx);
        float m_new = fmaxf(m_run, mx);
        if (m_new == -INFINITY) m_new = 0.f;
        const float alpha = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(m_run - m_new) : __expf(m_run - m_new);
        float psum = 0.f, esum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16) drop_mult4(drop, (uint64_t)((bh * T + tg) * T + k0 + jt * 16 + (lane >> 4) * 4), dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(s[jt][r] - m_new) : Img<CT>::ex(s[jt][r] - m_new);
                psum += p;
                s[jt][r] = p * dm[r];
                esum += s[jt][r];
            }
        }
        l_run = l_run * alpha + psum;              // per-lane partial sums: the four row groups are added once, after the sweep
        e_run = e_run * alpha + esum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < ND; ++i) oacc[i] *= alpha;
#pragma unroll
        for (int st = 0; st < SaK<CT>::NS64; ++st) {
            const typename Img<CT>::V pf = reg_perm<CT>(s, st);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if constexpr (TR) oacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)VT, LDX, i * 16, st, lane), pf, oacc[i]);
                else oacc[i] = Img<CT>::mma(load_perm<CT>(VT, LDC, i * 16, st, lane), pf, oacc[i]);
            }
        }
    }
    l_run = rows4_sum(l_run);
    e_run = rows4_sum(e_run);
    if (tg < T) {
        const float inv = 1.f / (e_run + 1e-8f * l_run);
        CT* ob = out + (b * T + tg) * ld_out + h * DH;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = i * 16 + (lane >> 4) * 4;
            Img<CT>::store4(ob + d0, oacc[i][0] * inv, oacc[i][1] * inv, oacc[i][2] * inv, oacc[i][3] * inv);
        }
        if ((lane >> 4) == 0) {
            lse_g[bh * T + tg] = (sizeof(CT) == 2 ? m_run * EMO_LN2 : m_run) + logf(l_run);
            if (zden_g) zden_g[bh * T + tg] = e_run / l_run + 1e-8f;
        }
    }
}

// Backward, query-tile organised (first version of the stage-1 training path).  With a_ij = m_ij p_ij / Z_i the final attention weight
// (m = dropout multiplier, Z_i = sum_j m_ij p_ij + 1e-8 saved by the forward), dA_ij = dO_i.v_j and r1_i = dO_i.O_i:
//     ds_ij = p_ij ( m_ij (dA_ij - r1_i) / Z_i - 1e-8 r1_i / Z_i )
// The kernel recomputes p (content + relative term, as the forward), forms ds in registers and accumulates BOTH parts of dq:
//   dq_content[t] = sum_j ds[t][j] k_j / sqrt(dh)                       (ds stays in registers as the MFMA operand)
//   dq_rel[t]     = sum_j ds[t][j] R[t - j] / sqrt(dh) = sum_c S2[t][c] Rwin[c]   with the skew column c = t_l - j_l + 63:
// each wave scatters its 16 x 64 ds values into a [16 t][c' = c - 16w] image (it re-uses the wave's P2 skew buffer, which is dead by then)
// and multiplies it with the TRANSPOSED window rows (ds_read_b64_tr_b16 in bf16 mode).  dq = dq_content + dq_rel; dq_rel also goes out on its
// own because d r_r_bias = colsum(dq_rel) and d r_w_bias = colsum(dq) - colsum(dq_rel).  dK / dV: relattn_bwd_dkv_kernel; dR: relattn_bwd_dr_kernel.
template <typename CT, int DH>
__global__ __launch_bounds__(256, (sizeof(CT) == 2 ? 2 : 1)) void relattn_bwd_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                             const CT* __restrict__ rd, int64_t ld_r, int64_t n_dist, const float* __restrict__ ub,
                                                             const float* __restrict__ vb_, const CT* __restrict__ out, const CT* __restrict__ dout,
                                                             int64_t ld_out, const float* __restrict__ lse_g, const float* __restrict__ zden_g,
                                                             CT* __restrict__ dq, int64_t ld_d, CT* __restrict__ dq_rel, int64_t ld_rel,
                                                             float* __restrict__ delta_g, int64_t T, int64_t H, DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16, NQ = DHP / Img<CT>::KSTEP, SKW = 84;
    constexpr bool TR = sizeof(CT) == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Qi = (CT*)smem;                       // [64][LDX]  scratch image for the register fragments (q + u, q + v, dO)
    CT* Ki = Qi + 64 * LDX;                   // [64][LDX]
    CT* Vi = Ki + 64 * LDX;                   // [64][LDX]
    CT* KT = Vi + 64 * LDX;                   // fp32 only: K^T [DH][LDC]
    CT* Rw = KT + (TR ? 0 : DH * LDC);        // [128][LDX]
    float* sk = (float*)(Rw + 128 * LDX);     // [4][16][SKW]
    float* Dv = sk + 4 * 16 * SKW;            // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qt = (int64_t)gridDim.y - 1 - blockIdx.y;
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;      // (b, h) fastest in dispatch order: the longest sweeps of the whole grid go first
    const int64_t q0 = qt * 64;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* ob = out + (b * T) * ld_out + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    const CT* rb = rd + h * DH;
    const int qvalid = (int)((T - q0) < 64 ? (T - q0) : 64);
    RowPrefetch<CT, DH, DHP, 64, 256> pk, pv;
    RowPrefetch<CT, DH, DHP, 64, 256, true> pkT;
    constexpr int VE = 16 / sizeof(CT), CH = DH / VE, RNI = (128 * CH + 255) / 256;
    CT rr[RNI][VE];
    auto fetch_r = [&](int64_t d0) {
#pragma unroll
        for (int i = 0; i < RNI; ++i) {
            const int it = tid + 256 * i;
            const int row = it / CH, c = (it % CH) * VE;
            const int64_t dist = d0 + row;
#pragma unroll
            for (int e = 0; e < VE; ++e) rr[i][e] = from_f32<CT>(0.f);
            if (it < 128 * CH && dist >= 0 && dist < n_dist) {
                if constexpr (sizeof(CT) == 2) *(bf16x8*)rr[i] = *(const bf16x8*)(rb + dist * ld_r + c);
                else *(f32x4*)rr[i] = *(const f32x4*)(rb + dist * ld_r + c);
            }
        }
    };
    auto store_r = [&]() {
#pragma unroll
        for (int i = 0; i < RNI; ++i) {
            const int it = tid + 256 * i;
            if (it < 128 * CH) {
                const int row = it / CH, c = (it % CH) * VE;
                if constexpr (sizeof(CT) == 2) *(bf16x8*)(Rw + row * LDX + c) = *(const bf16x8*)rr[i];
                else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) Rw[row * LDX + c + e] = rr[i][e];
                }
            }
        }
        if constexpr (DHP > DH) {
            for (int it = tid; it < 128 * (DHP - DH); it += 256) Rw[(it / (DHP - DH)) * LDX + DH + it % (DHP - DH)] = from_f32<CT>(0.f);
        }
    };
    {
        const int kv0 = (int)(T < 64 ? T : 64);
        pk.load(kb, ld, kv0, tid); pv.load(vb, ld, kv0, tid);
        if constexpr (!TR) pkT.load(kb, ld, kv0, tid);
        fetch_r(q0 - 63);
    }
    rows_dot<CT, DH>(Dv, gb + q0 * ld_out, ob + q0 * ld_out, ld_out, qvalid, tid);
    typename Img<CT>::V quf[NQ], qvf[NQ], gf[NQ];
    for (int pass = 0; pass < 3; ++pass) {
        const float* bias = (pass == 0 ? ub : vb_) + h * DH;
        __syncthreads();
        for (int it = tid; it < 64 * DHP; it += 256) {
            const int r = it / DHP, d = it % DHP;
            float x = 0.f;
            if (r < qvalid && d < DH) x = pass < 2 ? to_f32<CT>(qb[(q0 + r) * ld + d]) + bias[d] : to_f32<CT>(gb[(q0 + r) * ld_out + d]);
            Qi[r * LDX + d] = from_f32<CT>(x);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) {
            const typename Img<CT>::V f = Img<CT>::load(Qi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
            if (pass == 0) quf[kk] = f; else if (pass == 1) qvf[kk] = f; else gf[kk] = f;
        }
    }
    const float sqrt_dh = sqrtf((float)DH), rsqrt_dh = 1.f / sqrt_dh, c2 = rsqrt_dh * EMO_LOG2E;
    const int tl = wave * 16 + (lane & 15);
    const int64_t tg = q0 + tl;
    const bool row_ok = tg < T;
    const float lse = row_ok ? lse_g[bh * T + tg] : INFINITY;
    const float lse2 = lse * EMO_LOG2E;
    const float zinv = row_ok ? 1.f / zden_g[bh * T + tg] : 0.f;
    const float r1 = Dv[tl], r2 = 1e-8f * r1 * zinv;
    if (delta_g && tid < qvalid) delta_g[bh * T + q0 + tid] = Dv[tid];
    f32x4 dqacc[ND], dracc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) dqacc[i] = dracc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* skw = sk + (wave * 16 + (lane & 15)) * SKW;
    // skew image of ds: [16 t][LD2] over the wave's P2 buffer (16 * SKW floats).  bf16: k padded to 96 (3 MFMA steps), f32: k = 80.
    constexpr int LD2 = TR ? 104 : 84;
    static_assert(16 * LD2 * sizeof(CT) <= 16 * SKW * sizeof(float), "ds skew image must fit the P2 skew buffer");
    CT* s2 = (CT*)(sk + wave * 16 * SKW);
    for (int64_t kt = 0; kt <= qt; ++kt) {
        const int64_t k0 = kt * 64;
        __syncthreads();
        pk.store_rows(Ki, LDX, tid);
        pv.store_rows(Vi, LDX, tid);
        if constexpr (!TR) pkT.store_T(KT, LDC, tid);
        store_r();
        if (kt < qt) {
            const int64_t kn = k0 + 64;
            const int nv = (int)((T - kn) < 64 ? (T - kn) : 64);
            pk.load(kb + kn * ld, ld, nv, tid); pv.load(vb + kn * ld, ld, nv, tid);
            if constexpr (!TR) pkT.load(kb + kn * ld, ld, nv, tid);
            fetch_r(q0 - kn - 63);
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) a2 = Img<CT>::mma(Img<CT>::load(Rw, LDX, (wave + ci) * 16, kk * Img<CT>::KSTEP, lane), qvf[kk], a2);
            *(f32x4*)(skw + ci * 16 + (lane >> 4) * 4) = a2;
        }
        __builtin_amdgcn_wave_barrier();
        const bool diag = kt == qt;
        float ds[4][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const bool live = !(diag && jt > wave);
            if (live) {
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) {
                    sa = Img<CT>::mma(Img<CT>::load(Ki, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), quf[kk], sa);
                    dp = Img<CT>::mma(Img<CT>::load(Vi, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), gf[kk], dp);
                }
            }
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16) drop_mult4(drop, (uint64_t)((bh * T + tg) * T + k0 + jt * 16 + (lane >> 4) * 4), dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = jt * 16 + (lane >> 4) * 4 + r;
                const float sc = sa[r] + skw[(lane & 15) + 63 - jl];
                float p = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(sc * c2 - lse2) : Img<CT>::ex(sc / sqrt_dh - lse);
                if (diag && (jl > tl || k0 + jl >= T)) p = 0.f;
                ds[jt][r] = row_ok ? p * (dm[r] * (dp[r] - r1) * zinv - r2) : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ds -> skew image (the P2 values are dead): zero, then one element per (t, j) at column c' = t_l - j_l + 63 (in [0, 78])
        {
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            for (int i = lane; i < (int)(16 * LD2 * sizeof(CT) / 16); i += 64) ((u32x4*)s2)[i] = z4;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s2[(lane & 15) * LD2 + (lane & 15) + 63 - (jt * 16 + (lane >> 4) * 4 + r)] = from_f32<CT>(ds[jt][r]);
        __builtin_amdgcn_wave_barrier();
        // dQ_rel^T[d][t] += sum_c' Rwin^T[d][16w + c'] S2[t][c']
        if constexpr (TR) {
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const bf16x8 sf = load_perm<CT>(s2, LD2, 0, st, lane);
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    bf16x8 rf;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {            // load_perm_tr with the window row clamped (rows past 127 only meet zero columns)
                        int row = wave * 16 + (2 * st + hh) * 16 + (lane >> 4) * 4 + ((lane & 15) >> 2);
                        row = row > 127 ? 127 : row;
                        const bf16_t* pr = (const bf16_t*)Rw + row * LDX + i * 16 + (lane & 3) * 4;
                        const short4v t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)pr);
                        const bf16x4 tb = __builtin_bit_cast(bf16x4, t4);
                        rf[hh * 4 + 0] = tb[0]; rf[hh * 4 + 1] = tb[1]; rf[hh * 4 + 2] = tb[2]; rf[hh * 4 + 3] = tb[3];
                    }
                    dracc[i] = Img<CT>::mma(rf, sf, dracc[i]);
                }
            }
        } else {
#pragma unroll 4
            for (int st = 0; st < 20; ++st) {
                const float sf = s2[(lane & 15) * LD2 + 4 * st + (lane >> 4)];
#pragma unroll
                for (int i = 0; i < ND; ++i)
                    dracc[i] = Img<CT>::mma(Rw[(wave * 16 + 4 * st + (lane >> 4)) * LDX + i * 16 + (lane & 15)], sf, dracc[i]);
            }
        }
        // dQ_content^T[d][t] += sum_j K^T[d][j] dS[t][j]
#pragma unroll
        for (int st = 0; st < SaK<CT>::NS64; ++st) {
            const typename Img<CT>::V df = reg_perm<CT>(ds, st);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if constexpr (TR) dqacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)Ki, LDX, i * 16, st, lane), df, dqacc[i]);
                else dqacc[i] = Img<CT>::mma(load_perm<CT>(KT, LDC, i * 16, st, lane), df, dqacc[i]);
            }
        }
    }
    if (row_ok) {
        CT* db = dq + (b * T + tg) * ld_d + h * DH;
        CT* dr = dq_rel + (b * T + tg) * ld_rel + h * DH;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = i * 16 + (lane >> 4) * 4;
            Img<CT>::store4(dr + d0, dracc[i][0] * rsqrt_dh, dracc[i][1] * rsqrt_dh, dracc[i][2] * rsqrt_dh, dracc[i][3] * rsqrt_dh);
            Img<CT>::store4(db + d0, (dqacc[i][0] + dracc[i][0]) * rsqrt_dh, (dqacc[i][1] + dracc[i][1]) * rsqrt_dh,
                            (dqacc[i][2] + dracc[i][2]) * rsqrt_dh, (dqacc[i][3] + dracc[i][3]) * rsqrt_dh);
        }
    }
}

// Backward, key-tile pass: dK and dV (replaces the dense a / ds matrices + per-(b, h) GEMMs of the first version).  qu = q + r_w_bias and
// qv = q + r_r_bias arrive materialised ([B*T, H*dh], pitch ld_q).  A wave owns 16 key rows; for a 16-row query sub-tile tt the relative term
// only touches two 16-row tiles of the distance window (c = t - j + 63 in [16(tt-w)+48, 16(tt-w)+78]), skewed through a [16][32] buffer.
template <typename CT, int DH>
__global__ __launch_bounds__(256, 1) void relattn_bwd_dkv_kernel(
    const CT* __restrict__ qu, const CT* __restrict__ qv, int64_t ld_q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
    const CT* __restrict__ rd, int64_t ld_r, int64_t n_dist, const CT* __restrict__ dout, int64_t ld_out, const float* __restrict__ lse_g,
    const float* __restrict__ zden_g, const float* __restrict__ delta_g, CT* __restrict__ dk, CT* __restrict__ dv, int64_t ld_d, int64_t T, int64_t H,
    DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, LDC = D::LDC, DHP = D::DHP, ND = DH / 16, NQ = DHP / Img<CT>::KSTEP, SKW = 36;
    constexpr bool TR = sizeof(CT) == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Ki = (CT*)smem;           // [64][LDX]  (Ki / Vi only to build the register fragments)
    CT* Vi = Ki + 64 * LDX;       // [64][LDX]
    CT* Qu = Vi + 64 * LDX;       // [64][LDX]
    CT* Qv = Qu + 64 * LDX;       // [64][LDX]
    CT* dOi = Qv + 64 * LDX;      // [64][LDX]
    CT* Rw = dOi + 64 * LDX;      // [128][LDX]
    CT* QT = Rw + 128 * LDX;      // fp32 only: (q+u)^T, dO^T [DH][LDC] each
    CT* dOT = QT + DH * LDC;
    float* sk = (float*)(QT + (TR ? 0 : 2 * DH * LDC));   // [4][16][SKW]
    float* Dv = sk + 4 * 16 * SKW;  // [64] r1
    float* Lv = Dv + 64;            // [64] lse (base-2 in bf16 mode)
    float* Zv = Lv + 64;            // [64] 1 / zden
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t kt = blockIdx.y;
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;      // (b, h) fastest in dispatch order: the longest sweeps of the whole grid go first
    const int64_t k0 = kt * 64;
    const int64_t nqt = (T + 63) / 64;
    const CT* qub = qu + (b * T) * ld_q + h * DH;
    const CT* qvb = qv + (b * T) * ld_q + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    const CT* rb = rd + h * DH;
    const int kvalid = (int)((T - k0) < 64 ? (T - k0) : 64);
    RowPrefetch<CT, DH, DHP, 64, 256> pq, pq2, pg;
    RowPrefetch<CT, DH, DHP, 64, 256, true> pqT, pgT;
    constexpr int VE = 16 / sizeof(CT), CH = DH / VE, RNI = (128 * CH + 255) / 256;
    CT rr[RNI][VE];
    float pl = 0.f, pz = 0.f, pd_ = 0.f;
    auto fetch = [&](int64_t q0n) {
        const int nv = (int)((T - q0n) < 64 ? (T - q0n) : 64);
        pq.load(qub + q0n * ld_q, ld_q, nv, tid);
        pq2.load(qvb + q0n * ld_q, ld_q, nv, tid);
        pg.load(gb + q0n * ld_out, ld_out, nv, tid);
        if constexpr (!TR) { pqT.load(qub + q0n * ld_q, ld_q, nv, tid); pgT.load(gb + q0n * ld_out, ld_out, nv, tid); }
        const int64_t d0 = q0n - k0 - 63;
#pragma unroll
        for (int i = 0; i < RNI; ++i) {
            const int it = tid + 256 * i;
            const int row = it / CH, c = (it % CH) * VE;
            const int64_t dist = d0 + row;
#pragma unroll
            for (int e = 0; e < VE; ++e) rr[i][e] = from_f32<CT>(0.f);
            if (it < 128 * CH && dist >= 0 && dist < n_dist) {
                if constexpr (sizeof(CT) == 2) *(bf16x8*)rr[i] = *(const bf16x8*)(rb + dist * ld_r + c);
                else *(f32x4*)rr[i] = *(const f32x4*)(rb + dist * ld_r + c);
            }
        }
        if (tid < 64) {
            const bool ok = q0n + tid < T;
            pl = ok ? lse_g[bh * T + q0n + tid] * (sizeof(CT) == 2 ? EMO_LOG2E : 1.f) : INFINITY;
            pz = ok ? 1.f / zden_g[bh * T + q0n + tid] : 0.f;
            pd_ = ok ? delta_g[bh * T + q0n + tid] : 0.f;
        }
    };
    fetch(k0);
    load_rows<CT, DH, DHP>(Ki, LDX, kb + k0 * ld, ld, 64, kvalid, tid);
    load_rows<CT, DH, DHP>(Vi, LDX, vb + k0 * ld, ld, 64, kvalid, tid);
    __syncthreads();
    typename Img<CT>::V kf[NQ], vf[NQ];
#pragma unroll
    for (int kk = 0; kk < NQ; ++kk) {
        kf[kk] = Img<CT>::load(Ki, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
        vf[kk] = Img<CT>::load(Vi, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
    }
    const float sqrt_dh = sqrtf((float)DH), rsqrt_dh = 1.f / sqrt_dh, c2 = rsqrt_dh * EMO_LOG2E;
    const int jl = wave * 16 + (lane & 15);
    const int64_t jg = k0 + jl;
    float* skw = sk + wave * 16 * SKW;
    f32x4 dkacc[ND], dvacc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) { dkacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int64_t qt = kt; qt < nqt; ++qt) {
        const int64_t q0 = qt * 64;
        __syncthreads();
        pq.store_rows(Qu, LDX, tid);
        pq2.store_rows(Qv, LDX, tid);
        pg.store_rows(dOi, LDX, tid);
        if constexpr (!TR) { pqT.store_T(QT, LDC, tid); pgT.store_T(dOT, LDC, tid); }
#pragma unroll
        for (int i = 0; i < RNI; ++i) {
            const int it = tid + 256 * i;
            if (it < 128 * CH) {
                const int row = it / CH, c = (it % CH) * VE;
                if constexpr (sizeof(CT) == 2) *(bf16x8*)(Rw + row * LDX + c) = *(const bf16x8*)rr[i];
                else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) Rw[row * LDX + c + e] = rr[i][e];
                }
            }
        }
        if constexpr (DHP > DH) {
            for (int it = tid; it < 128 * (DHP - DH); it += 256) Rw[(it / (DHP - DH)) * LDX + DH + it % (DHP - DH)] = from_f32<CT>(0.f);
        }
        if (tid < 64) { Lv[tid] = pl; Zv[tid] = pz; Dv[tid] = pd_; }
        if (qt + 1 < nqt) fetch(q0 + 64);
        __syncthreads();
        const bool diag = qt == kt;
        float pd[4][4], ds[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const bool live = !(diag && tt < wave);
            if (live) {
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) {
                    sa = Img<CT>::mma(Img<CT>::load(Qu, LDX, tt * 16, kk * Img<CT>::KSTEP, lane), kf[kk], sa);
                    dp = Img<CT>::mma(Img<CT>::load(dOi, LDX, tt * 16, kk * Img<CT>::KSTEP, lane), vf[kk], dp);
                }
                const int ct0 = tt - wave + 3;                // window tiles ct0, ct0 + 1 hold c = t - j + 63 for this (tt, wave)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < NQ; ++kk)
                        a2 = Img<CT>::mma(Img<CT>::load(Rw, LDX, (ct0 + ci) * 16, kk * Img<CT>::KSTEP, lane), Img<CT>::load(Qv, LDX, tt * 16, kk * Img<CT>::KSTEP, lane), a2);
                    *(f32x4*)(skw + (lane & 15) * SKW + ci * 16 + (lane >> 4) * 4) = a2;        // buf[t_loc][c - 16 ct0]
                }
            }
            __builtin_amdgcn_wave_barrier();
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16 && (T & 3) == 0)
                drop_mult_col4(drop, (uint64_t)((bh * T + q0 + tt * 16 + (lane >> 4) * 4 + (lane & 3)) * T + (jg & ~(int64_t)3)), lane, dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tq = (lane >> 4) * 4 + r;           // query row inside the sub-tile
                const int tl = tt * 16 + tq;
                const int64_t tg = q0 + tl;
                const float bd = live ? skw[tq * SKW + tq - (lane & 15) + 15] : 0.f;
                const float sc = sa[r] + bd;
                float p = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(sc * c2 - Lv[tl]) : Img<CT>::ex(sc / sqrt_dh - Lv[tl]), mult = dm[r];
                if (diag && jl > tl) p = 0.f;
                if (drop.thr16 && (T & 3) != 0) mult = drop_mult(drop, (uint64_t)((bh * T + tg) * T + jg));
                const float zi = Zv[tl], r1 = Dv[tl];
                pd[tt][r] = p * mult * zi;
                ds[tt][r] = p * (mult * (dp[r] - r1) * zi - 1e-8f * r1 * zi);
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int st = 0; st < SaK<CT>::NS64; ++st) {
            const typename Img<CT>::V pf = reg_perm<CT>(pd, st), df = reg_perm<CT>(ds, st);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if constexpr (TR) {
                    dvacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)dOi, LDX, i * 16, st, lane), pf, dvacc[i]);
                    dkacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)Qu, LDX, i * 16, st, lane), df, dkacc[i]);
                } else {
                    dvacc[i] = Img<CT>::mma(load_perm<CT>(dOT, LDC, i * 16, st, lane), pf, dvacc[i]);
                    dkacc[i] = Img<CT>::mma(load_perm<CT>(QT, LDC, i * 16, st, lane), df, dkacc[i]);
                }
            }
        }
    }
    if (jg < T) {
        CT* dkb = dk + (b * T + jg) * ld_d + h * DH;
        CT* dvb = dv + (b * T + jg) * ld_d + h * DH;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d0 = i * 16 + (lane >> 4) * 4;
            Img<CT>::store4(dkb + d0, dkacc[i][0] * rsqrt_dh, dkacc[i][1] * rsqrt_dh, dkacc[i][2] * rsqrt_dh, dkacc[i][3] * rsqrt_dh);
            Img<CT>::store4(dvb + d0, dvacc[i][0], dvacc[i][1], dvacc[i][2], dvacc[i][3]);
        }
    }
}

// one query per stream against a KV cache, keys j in [j0, len): score_j = ((q+u).k_j + (q+v).R[len-1-j]) / sqrt(dh)
// Backward, distance-window pass: dR[dist] = sum_{b, i} ds[b, i, i - dist] (q_i + v) / sqrt(dh), the gradient of the by-distance rows r_dist.
// For a pair (query tile qt, key tile kt) the 64 x 64 ds values land on the 127 distances 64 (qt - kt) - 63 .. + 63, i.e. on a window that only
// depends on the diagonal delta = qt - kt.  One workgroup therefore owns ONE diagonal of one (b, h): it walks the pairs (kt + delta, kt),
// recomputes ds like the query-tile pass (the R window stays in LDS for the whole walk), scatters it into a block-wide skew image
// S2[64 t][128 c] (c = t_l - j_l + 63) and accumulates  dRwin[c][d] += sum_t S2^T[c][t] qv[t][d]  in MFMA accumulators over the whole
// diagonal (both operands through transposed LDS reads) — the partial windows [B*H, n_tiles, 128, dh] go to a workspace and
// relattn_dr_reduce_kernel adds the (at most two) windows covering each distance over the batch, in a fixed order (deterministic).
// This replaces the dense [H, B*T, T] skewed ds matrix + one GEMM per head of the first version.
template <typename CT, int DH>
__global__ __launch_bounds__(256) void relattn_bwd_dr_kernel(const CT* __restrict__ qu, const CT* __restrict__ qv, int64_t ld_q, const CT* __restrict__ k,
                                                             const CT* __restrict__ v, int64_t ld, const CT* __restrict__ rd, int64_t ld_r, int64_t n_dist,
                                                             const CT* __restrict__ dout, int64_t ld_out, const float* __restrict__ lse_g,
                                                             const float* __restrict__ zden_g, const float* __restrict__ delta_g, float* __restrict__ part,
                                                             int64_t T, int64_t H, DropCtx drop) {
    typedef SaDims<CT, DH> D;
    constexpr int LDX = D::LDX, DHP = D::DHP, ND = DH / 16, NQ = DHP / Img<CT>::KSTEP, SKW = 84;
    constexpr bool TR = sizeof(CT) == 2;
    constexpr int LDS2 = TR ? 136 : 130;                      // S2 row pitch (128 columns + pad)
    constexpr int S2SZ = CMax<2 * 64 * LDX, 64 * LDS2>::v;    // the skew image re-uses the (q + u) and dO images
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* Qu = (CT*)smem;                       // [64][LDX]  q + u   } only feed the register fragments; then S2 [64][LDS2] lives here
    CT* Go = Qu + 64 * LDX;                   // [64][LDX]  dO      }
    CT* S2 = Qu;
    CT* Qv = Qu + S2SZ;                       // [64][LDX]  q + v (fragments AND the operand of the dR product)
    CT* Ki = Qv + 64 * LDX;                   // [64][LDX]
    CT* Vi = Ki + 64 * LDX;                   // [64][LDX]
    CT* Rw = Vi + 64 * LDX;                   // [128][LDX] window rows: distance 64 delta - 63 + c
    float* sk = (float*)(Rw + 128 * LDX);     // [4][16][SKW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nt = (T + 63) / 64;
    const int64_t dl = blockIdx.y;            // diagonal: qt = kt + dl (dl = 0 is the longest walk and starts first)
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;      // (b, h) fastest in dispatch order: the longest sweeps of the whole grid go first
    const CT* qub = qu + (b * T) * ld_q + h * DH;
    const CT* qvb = qv + (b * T) * ld_q + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    const CT* rb = rd + h * DH;
    {   // the window rows, once
        constexpr int VE = 16 / sizeof(CT), CH = DH / VE;
        for (int it = tid; it < 128 * CH; it += 256) {
            const int row = it / CH, c = (it % CH) * VE;
            const int64_t dist = 64 * dl - 63 + row;
            CT tmp[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) tmp[e] = from_f32<CT>(0.f);
            if (dist >= 0 && dist < n_dist) {
                if constexpr (sizeof(CT) == 2) *(bf16x8*)tmp = *(const bf16x8*)(rb + dist * ld_r + c);
                else *(f32x4*)tmp = *(const f32x4*)(rb + dist * ld_r + c);
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) Rw[row * LDX + c + e] = tmp[e];
        }
        if constexpr (DHP > DH) {
            for (int it = tid; it < 128 * (DHP - DH); it += 256) Rw[(it / (DHP - DH)) * LDX + DH + it % (DHP - DH)] = from_f32<CT>(0.f);
        }
    }
    RowPrefetch<CT, DH, DHP, 64, 256> pqu, pqv, pg, pk, pv;
    float pl = INFINITY, pz = 0.f, pdl = 0.f;
    auto fetch = [&](int64_t kt) {
        const int64_t q0 = (kt + dl) * 64, k0 = kt * 64;
        const int nq = (int)((T - q0) < 64 ? (T - q0) : 64), nk = (int)((T - k0) < 64 ? (T - k0) : 64);
        pqu.load(qub + q0 * ld_q, ld_q, nq, tid);
        pqv.load(qvb + q0 * ld_q, ld_q, nq, tid);
        pg.load(gb + q0 * ld_out, ld_out, nq, tid);
        pk.load(kb + k0 * ld, ld, nk, tid);
        pv.load(vb + k0 * ld, ld, nk, tid);
        const int64_t tg = q0 + wave * 16 + (lane & 15);
        const bool ok = tg < T;
        pl = ok ? lse_g[bh * T + tg] : INFINITY;
        pz = ok ? 1.f / zden_g[bh * T + tg] : 0.f;
        pdl = ok ? delta_g[bh * T + tg] : 0.f;
    };
    fetch(0);
    const float sqrt_dh = sqrtf((float)DH), rsqrt_dh = 1.f / sqrt_dh, c2 = rsqrt_dh * EMO_LOG2E;
    const int tl = wave * 16 + (lane & 15);
    f32x4 racc[2][ND];                        // window rows 32 w .. 32 w + 31 x all d
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < ND; ++i) racc[a][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* skw = sk + (wave * 16 + (lane & 15)) * SKW;
    const bool diag = dl == 0;
    for (int64_t kt = 0; kt + dl < nt; ++kt) {
        const int64_t q0 = (kt + dl) * 64, k0 = kt * 64;
        const int64_t tg = q0 + tl;
        __syncthreads();                      // the previous pair's dR product has read S2 / Qv
        pqu.store_rows(Qu, LDX, tid);
        pqv.store_rows(Qv, LDX, tid);
        pg.store_rows(Go, LDX, tid);
        pk.store_rows(Ki, LDX, tid);
        pv.store_rows(Vi, LDX, tid);
        const float lse = pl, lse2 = pl * EMO_LOG2E, zinv = pz, r1 = pdl, r2 = 1e-8f * pdl * pz;
        const bool row_ok = tg < T;
        if (kt + 1 + dl < nt) fetch(kt + 1);
        __syncthreads();
        typename Img<CT>::V quf[NQ], qvf[NQ], gf[NQ];
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) {
            quf[kk] = Img<CT>::load(Qu, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
            qvf[kk] = Img<CT>::load(Qv, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
            gf[kk] = Img<CT>::load(Go, LDX, wave * 16, kk * Img<CT>::KSTEP, lane);
        }
        __syncthreads();                      // Qu / Go are dead: S2 may overwrite them
        {
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            u32x4* zr = (u32x4*)(S2 + wave * 16 * LDS2);
            for (int i = lane; i < (int)(16 * LDS2 * sizeof(CT) / 16); i += 64) zr[i] = z4;
        }
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NQ; ++kk) a2 = Img<CT>::mma(Img<CT>::load(Rw, LDX, (wave + ci) * 16, kk * Img<CT>::KSTEP, lane), qvf[kk], a2);
            *(f32x4*)(skw + ci * 16 + (lane >> 4) * 4) = a2;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            if (!(diag && jt > wave)) {
#pragma unroll
                for (int kk = 0; kk < NQ; ++kk) {
                    sa = Img<CT>::mma(Img<CT>::load(Ki, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), quf[kk], sa);
                    dp = Img<CT>::mma(Img<CT>::load(Vi, LDX, jt * 16, kk * Img<CT>::KSTEP, lane), gf[kk], dp);
                }
            }
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16) drop_mult4(drop, (uint64_t)((bh * T + tg) * T + k0 + jt * 16 + (lane >> 4) * 4), dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = jt * 16 + (lane >> 4) * 4 + r;
                const float sc = sa[r] + skw[(lane & 15) + 63 - jl];
                float p = sizeof(CT) == 2 ? __builtin_amdgcn_exp2f(sc * c2 - lse2) : Img<CT>::ex(sc / sqrt_dh - lse);
                if ((diag && jl > tl) || k0 + jl >= T) p = 0.f;
                const float dsv = row_ok ? p * (dm[r] * (dp[r] - r1) * zinv - r2) * rsqrt_dh : 0.f;
                S2[tl * LDS2 + tl - jl + 63] = from_f32<CT>(dsv);
            }
        }
        __syncthreads();                      // S2 complete
        // dRwin[c][d] += sum_t S2[t][c] qv[t][d]
        if constexpr (TR) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 qf[ND];
#pragma unroll
                for (int i = 0; i < ND; ++i) qf[i] = load_perm_tr((const bf16_t*)Qv, LDX, i * 16, st, lane);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const bf16x8 sf = load_perm_tr((const bf16_t*)S2, LDS2, (2 * wave + a) * 16, st, lane);
#pragma unroll
                    for (int i = 0; i < ND; ++i) racc[a][i] = Img<CT>::mma(sf, qf[i], racc[a][i]);
                }
            }
        } else {
#pragma unroll 4
            for (int st = 0; st < 16; ++st) {
                const int t = 4 * st + (lane >> 4);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const float sf = S2[t * LDS2 + (2 * wave + a) * 16 + (lane & 15)];
#pragma unroll
                    for (int i = 0; i < ND; ++i) racc[a][i] = Img<CT>::mma(sf, Qv[t * LDX + i * 16 + (lane & 15)], racc[a][i]);
                }
            }
        }
    }
    float* po = part + ((bh * nt + dl) * 128) * DH;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) po[((2 * wave + a) * 16 + (lane >> 4) * 4 + r) * DH + i * 16 + (lane & 15)] = racc[a][i][r];
}

// dR[dist][h * dh + d] = sum_b (window delta_hi row c + window delta_hi - 1 row c + 64),  delta_hi = (dist + 63) / 64,  c = dist + 63 - 64 delta_hi
__global__ __launch_bounds__(256) void relattn_dr_reduce_kernel(const float* __restrict__ part, float* __restrict__ dR, int64_t ld_dr, int64_t B, int64_t T,
                                                                int64_t H, int dh) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t HD = H * dh;
    if (idx >= T * HD) return;
    const int64_t dist = idx / HD, h = (idx % HD) / dh;
    const int d = (int)(idx % dh);
    const int64_t nt = (T + 63) / 64, dhi = (dist + 63) / 64;
    const int c = (int)(dist + 63 - 64 * dhi);
    float s = 0.f;
    for (int64_t b = 0; b < B; ++b) {
        const float* pb = part + ((b * H + h) * nt) * 128 * dh;
        if (dhi < nt) s += pb[(dhi * 128 + c) * dh + d];
        if (dhi >= 1) s += pb[((dhi - 1) * 128 + c + 64) * dh + d];
    }
    dR[dist * ld_dr + h * dh + d] = s;
}

template <typename CT>
__global__ __launch_bounds__(256) void relattn_decode_kernel(const CT* __restrict__ q, int64_t ld_q, CT* __restrict__ kc, CT* __restrict__ vc, int64_t T_max,
                                                             const int64_t* __restrict__ lens, int64_t lens_off, int64_t mem_len,
                                                             const CT* __restrict__ k_new, const CT* __restrict__ v_new, int64_t ld_new,
                                                             const CT* __restrict__ rd, int64_t ld_r, const float* __restrict__ ub,
                                                             const float* __restrict__ vb_, CT* __restrict__ out, int64_t ld_out, int64_t H, int dh) {
    constexpr int VE = 16 / sizeof(CT);
    extern __shared__ float sc[];            // [T_max] scores
    __shared__ float qu[128], qv[128], red[4];
    __shared__ float part[256 * VE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sh = blockIdx.x, s = sh / H, h = sh % H;
    const int64_t len = lens[s] + lens_off;
    const int64_t j0 = (mem_len > 0 && len - 1 - mem_len > 0) ? len - 1 - mem_len : 0;      // the memory keeps the last mem_len positions
    const int64_t HD = H * dh;
    if (tid < dh) {
        const float x = to_f32<CT>(q[s * ld_q + h * dh + tid]);
        qu[tid] = x + ub[h * dh + tid];
        qv[tid] = x + vb_[h * dh + tid];
        if (k_new) {
            kc[(s * T_max + len - 1) * HD + h * dh + tid] = k_new[s * ld_new + h * dh + tid];
            vc[(s * T_max + len - 1) * HD + h * dh + tid] = v_new[s * ld_new + h * dh + tid];
        }
    }
    __syncthreads();
    const float sqrt_dh = sqrtf((float)dh);
    const int LPR = dh / VE;
    const int rl = tid / LPR, cl = (tid % LPR) * VE, RPB = 256 / LPR;
    float mx = -INFINITY;
    for (int64_t jb = j0; jb < len; jb += RPB) {
        const int64_t j = jb + rl;
        float a = 0.f;
        if (j < len) {
            const CT* kr = kc + (s * T_max + j) * HD + h * dh + cl;
            const CT* rr = rd + (len - 1 - j) * ld_r + h * dh + cl;
            if constexpr (sizeof(CT) == 2) { const bf16x8 kv = *(const bf16x8*)kr, rv = *(const bf16x8*)rr;
#pragma unroll
                for (int e = 0; e < 8; ++e) a += qu[cl + e] * (float)kv[e] + qv[cl + e] * (float)rv[e]; }
            else { const f32x4 kv = *(const f32x4*)kr, rv = *(const f32x4*)rr;
#pragma unroll
                for (int e = 0; e < 4; ++e) a += qu[cl + e] * kv[e] + qv[cl + e] * rv[e]; }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        a = a / sqrt_dh;
        if (j < len) {
            if ((tid % LPR) == 0) sc[j - j0] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int64_t j = tid; j < len - j0; j += 256) { float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    for (int64_t j = j0 + rl; j < len; j += RPB) {
        const CT* vr = vc + (s * T_max + j) * HD + h * dh + cl;
        const float p = sc[j - j0];
        if constexpr (sizeof(CT) == 2) { const bf16x8 vv = *(const bf16x8*)vr;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += p * (float)vv[e]; }
        else { const f32x4 vv = *(const f32x4*)vr;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += p * vv[e]; }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) part[rl * dh + cl + e] = acc[e];
    __syncthreads();
    if (tid < dh) {
        float a = 0.f;
        for (int p = 0; p < RPB; ++p) a += part[p * dh + tid];
        out[s * ld_out + h * dh + tid] = from_f32<CT>(a / tot / (1.f + 1e-8f));     // eval: p / (sum p + 1e-8) with sum p = 1
    }
}

template <typename CT, int DH> static size_t ra_fwd_lds() {
    typedef SaDims<CT, DH> D;
    return sizeof(CT) * (size_t)(2 * 64 * D::LDX + CMax<DH * D::LDC, 64 * D::LDX>::v + 128 * D::LDX) + sizeof(float) * 4 * 16 * 84;
}
template <typename CT, int DH>
static int run_relattn(const void* q, const void* k, const void* v, int64_t ld, const void* rd, int64_t ld_r, int64_t n_dist, const float* ub, const float* vb,
                       void* out, int64_t ld_out, float* lse, float* zden, int64_t B, int64_t T, int64_t H, DropCtx drop, hipStream_t st) {
    dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64));
    const size_t lds = ra_fwd_lds<CT, DH>();
    auto kf = relattn_fwd_kernel<CT, DH>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(kf, grid, dim3(256), lds, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, (const CT*)rd, ld_r, n_dist, ub, vb, (CT*)out, ld_out, lse,
                       zden, T, H, drop);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

extern "C" int emo_relpos_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, const void* r_dist, int64_t ld_r, int64_t n_dist,
                                   const float* r_w_bias, const float* r_r_bias, void* out, int64_t ld_out, float* lse, float* zden, int dtype, int64_t B,
                                   int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    int rc = sattn_check(q, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(r_dist && r_w_bias && r_r_bias && out && lse, "emo_relpos_attn_fwd: null pointer");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK(ld_r % ve == 0 && ((uintptr_t)r_dist & 15) == 0 && ((uintptr_t)out & 15) == 0, "emo_relpos_attn_fwd: r_dist / out must keep rows 16-B aligned");
    EMO_CHECK(n_dist >= T, "emo_relpos_attn_fwd: r_dist needs a row for every distance 0 .. T-1");
    const DropCtx drop = make_drop(p_drop, seed, offset);
    hipStream_t st = (hipStream_t)stream;
#define RA_CASE(DHv)                                                                                                                                   \
    if (dh == DHv) {                                                                                                                                   \
        if (dtype == EMO_BF16) return run_relattn<bf16_t, DHv>(q, k, v, ld, r_dist, ld_r, n_dist, r_w_bias, r_r_bias, out, ld_out, lse, zden, B, T, H, drop, st); \
        return run_relattn<float, DHv>(q, k, v, ld, r_dist, ld_r, n_dist, r_w_bias, r_r_bias, out, ld_out, lse, zden, B, T, H, drop, st);               \
    }
    RA_CASE(64)
    RA_CASE(32)
    RA_CASE(16)
#undef RA_CASE
    emo_set_error("emo_relpos_attn_fwd: unsupported d_head=%lld (built: 16, 32, 64)", (long long)dh);
    return EMO_ERR_UNSUPPORTED;
}

template <typename CT, int DH> static size_t ra_bwd_lds() {
    typedef SaDims<CT, DH> D;
    return sizeof(CT) * (size_t)(3 * 64 * D::LDX + (sizeof(CT) == 2 ? 0 : DH * D::LDC) + 128 * D::LDX) + sizeof(float) * (4 * 16 * 84 + 64);
}
template <typename CT, int DH>
static int run_relattn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* rd, int64_t ld_r, int64_t n_dist, const float* ub,
                           const float* vb, const void* out, const void* dout, int64_t ld_out, const float* lse, const float* zden, void* dq, int64_t ld_d,
                           void* dq_rel, int64_t ld_rel, float* delta, int64_t B, int64_t T, int64_t H, DropCtx drop, hipStream_t st) {
    dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64));
    const size_t lds = ra_bwd_lds<CT, DH>();
    auto kf = relattn_bwd_kernel<CT, DH>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(kf, grid, dim3(256), lds, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, (const CT*)rd, ld_r, n_dist, ub, vb, (const CT*)out,
                       (const CT*)dout, ld_out, lse, zden, (CT*)dq, ld_d, (CT*)dq_rel, ld_rel, delta, T, H, drop);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

extern "C" int emo_relpos_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const void* r_dist, int64_t ld_r, int64_t n_dist,
                                   const float* r_w_bias, const float* r_r_bias, const void* out, const void* dout, int64_t ld_out, const float* lse,
                                   const float* zden, void* dq, int64_t ld_d, void* dq_rel, int64_t ld_rel, float* delta, int dtype,
                                   int64_t B, int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    int rc = sattn_check(q, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(r_dist && r_w_bias && r_r_bias && out && dout && lse && zden && dq && dq_rel, "emo_relpos_attn_bwd: null pointer");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK(ld_r % ve == 0 && ld_d % 4 == 0 && ld_rel % 4 == 0 && (((uintptr_t)r_dist | (uintptr_t)dq | (uintptr_t)dq_rel | (uintptr_t)out | (uintptr_t)dout) & 15) == 0,
              "emo_relpos_attn_bwd: pointers must be 16-B aligned");
    EMO_CHECK(n_dist >= T, "emo_relpos_attn_bwd: r_dist needs a row for every distance 0 .. T-1");
    const DropCtx drop = make_drop(p_drop, seed, offset);
    hipStream_t st = (hipStream_t)stream;
#define RAB_CASE(DHv)                                                                                                                                  \
    if (dh == DHv) {                                                                                                                                   \
        if (dtype == EMO_BF16)                                                                                                                         \
            return run_relattn_bwd<bf16_t, DHv>(q, k, v, ld, r_dist, ld_r, n_dist, r_w_bias, r_r_bias, out, dout, ld_out, lse, zden, dq, ld_d, dq_rel, ld_rel,  \
                                                delta, B, T, H, drop, st);                                                          \
        return run_relattn_bwd<float, DHv>(q, k, v, ld, r_dist, ld_r, n_dist, r_w_bias, r_r_bias, out, dout, ld_out, lse, zden, dq, ld_d, dq_rel, ld_rel,       \
                                           delta, B, T, H, drop, st);                                                               \
    }
    RAB_CASE(64)
    RAB_CASE(32)
    RAB_CASE(16)
#undef RAB_CASE
    emo_set_error("emo_relpos_attn_bwd: unsupported d_head=%lld (built: 16, 32, 64)", (long long)dh);
    return EMO_ERR_UNSUPPORTED;
}

template <typename CT, int DH> static size_t ra_dr_lds() {
    typedef SaDims<CT, DH> D;
    return sizeof(CT) * (size_t)(CMax<2 * 64 * D::LDX, 64 * (sizeof(CT) == 2 ? 136 : 130)>::v + 3 * 64 * D::LDX + 128 * D::LDX) + sizeof(float) * (4 * 16 * 84);
}
template <typename CT, int DH>
static int run_relattn_dr(const void* qu, const void* qv, int64_t ld_q, const void* k, const void* v, int64_t ld, const void* rd, int64_t ld_r, int64_t n_dist,
                          const void* dout, int64_t ld_out, const float* lse, const float* zden, const float* delta, float* dR, int64_t ld_dr, float* part,
                          int64_t B, int64_t T, int64_t H, DropCtx drop, hipStream_t st) {
    dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64));
    const size_t lds = ra_dr_lds<CT, DH>();
    auto kf = relattn_bwd_dr_kernel<CT, DH>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(kf, grid, dim3(256), lds, st, (const CT*)qu, (const CT*)qv, ld_q, (const CT*)k, (const CT*)v, ld, (const CT*)rd, ld_r, n_dist,
                       (const CT*)dout, ld_out, lse, zden, delta, part, T, H, drop);
    EMO_LAUNCH_CHECK();
    const int64_t n = T * H * DH;
    hipLaunchKernelGGL(relattn_dr_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)part, dR, ld_dr, B, T, H, DH);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

extern "C" int64_t emo_relpos_attn_bwd_r_workspace_bytes(int64_t B, int64_t T, int64_t H, int64_t dh) {
    return B * H * ((T + 63) / 64) * 128 * dh * (int64_t)sizeof(float);
}

extern "C" int emo_relpos_attn_bwd_r(const void* qu, const void* qv, int64_t ld_q, const void* k, const void* v, int64_t ld, const void* r_dist, int64_t ld_r,
                                     int64_t n_dist, const void* dout, int64_t ld_out, const float* lse, const float* zden, const float* delta, float* dR,
                                     int64_t ld_dr, void* workspace, int64_t workspace_bytes, int dtype, int64_t B, int64_t T, int64_t H, int64_t dh,
                                     float p_drop, uint64_t seed, uint64_t offset, emo_stream_t stream) {
    int rc = sattn_check(qu, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(qv && r_dist && dout && lse && zden && delta && dR && workspace, "emo_relpos_attn_bwd_r: null pointer");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK(ld_r % ve == 0 && ld_q % ve == 0 && (((uintptr_t)r_dist | (uintptr_t)qu | (uintptr_t)qv | (uintptr_t)dout | (uintptr_t)workspace) & 15) == 0,
              "emo_relpos_attn_bwd_r: pointers must be 16-B aligned");
    EMO_CHECK(n_dist >= T && ld_dr >= H * dh, "emo_relpos_attn_bwd_r: r_dist needs a row for every distance 0 .. T-1, dR a column for every (h, d)");
    EMO_CHECK(workspace_bytes >= emo_relpos_attn_bwd_r_workspace_bytes(B, T, H, dh), "emo_relpos_attn_bwd_r: workspace too small (emo_relpos_attn_bwd_r_workspace_bytes)");
    const DropCtx drop = make_drop(p_drop, seed, offset);
    hipStream_t st = (hipStream_t)stream;
#define RAR_CASE(DHv)                                                                                                                                  \
    if (dh == DHv) {                                                                                                                                   \
        if (dtype == EMO_BF16)                                                                                                                         \
            return run_relattn_dr<bf16_t, DHv>(qu, qv, ld_q, k, v, ld, r_dist, ld_r, n_dist, dout, ld_out, lse, zden, delta, dR, ld_dr, (float*)workspace, B, T, \
                                               H, drop, st);                                                                                           \
        return run_relattn_dr<float, DHv>(qu, qv, ld_q, k, v, ld, r_dist, ld_r, n_dist, dout, ld_out, lse, zden, delta, dR, ld_dr, (float*)workspace, B, T, H,  \
                                          drop, st);                                                                                                   \
    }
    RAR_CASE(64)
    RAR_CASE(32)
    RAR_CASE(16)
#undef RAR_CASE
    emo_set_error("emo_relpos_attn_bwd_r: unsupported d_head=%lld (built: 16, 32, 64)", (long long)dh);
    return EMO_ERR_UNSUPPORTED;
}

template <typename CT, int DH> static size_t ra_dkv_lds() {
    typedef SaDims<CT, DH> D;
    return sizeof(CT) * (size_t)(5 * 64 * D::LDX + 128 * D::LDX + (sizeof(CT) == 2 ? 0 : 2 * DH * D::LDC)) + sizeof(float) * (4 * 16 * 36 + 3 * 64);
}
template <typename CT, int DH>
static int run_relattn_dkv(const void* qu, const void* qv, int64_t ld_q, const void* k, const void* v, int64_t ld, const void* rd, int64_t ld_r, int64_t n_dist,
                           const void* dout, int64_t ld_out, const float* lse, const float* zden, const float* delta, void* dk, void* dv, int64_t ld_d,
                           int64_t B, int64_t T, int64_t H, DropCtx drop, hipStream_t st) {
    dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64));
    const size_t lds = ra_dkv_lds<CT, DH>();
    auto kf = relattn_bwd_dkv_kernel<CT, DH>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(kf, grid, dim3(256), lds, st, (const CT*)qu, (const CT*)qv, ld_q, (const CT*)k, (const CT*)v, ld, (const CT*)rd, ld_r, n_dist,
                       (const CT*)dout, ld_out, lse, zden, delta, (CT*)dk, (CT*)dv, ld_d, T, H, drop);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

extern "C" int emo_relpos_attn_bwd_kv(const void* qu, const void* qv, int64_t ld_q, const void* k, const void* v, int64_t ld, const void* r_dist, int64_t ld_r,
                                      int64_t n_dist, const void* dout, int64_t ld_out, const float* lse, const float* zden, const float* delta, void* dk,
                                      void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, float p_drop, uint64_t seed,
                                      uint64_t offset, emo_stream_t stream) {
    int rc = sattn_check(qu, k, v, ld, ld_out, dtype, dh);
    if (rc) return rc;
    EMO_CHECK(qv && r_dist && dout && lse && zden && delta && dk && dv, "emo_relpos_attn_bwd_kv: null pointer");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK(ld_r % ve == 0 && ld_q % ve == 0 && ld_d % 4 == 0 &&
              (((uintptr_t)r_dist | (uintptr_t)qu | (uintptr_t)qv | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)dout) & 15) == 0,
              "emo_relpos_attn_bwd_kv: pointers must be 16-B aligned");
    EMO_CHECK(n_dist >= T, "emo_relpos_attn_bwd_kv: r_dist needs a row for every distance 0 .. T-1");
    const DropCtx drop = make_drop(p_drop, seed, offset);
    hipStream_t st = (hipStream_t)stream;
#define RAK_CASE(DHv)                                                                                                                                  \
    if (dh == DHv) {                                                                                                                                   \
        if (dtype == EMO_BF16)                                                                                                                         \
            return run_relattn_dkv<bf16_t, DHv>(qu, qv, ld_q, k, v, ld, r_dist, ld_r, n_dist, dout, ld_out, lse, zden, delta, dk, dv, ld_d, B, T, H, drop, st); \
        return run_relattn_dkv<float, DHv>(qu, qv, ld_q, k, v, ld, r_dist, ld_r, n_dist, dout, ld_out, lse, zden, delta, dk, dv, ld_d, B, T, H, drop, st);      \
    }
    RAK_CASE(64)
    RAK_CASE(32)
    RAK_CASE(16)
#undef RAK_CASE
    emo_set_error("emo_relpos_attn_bwd_kv: unsupported d_head=%lld (built: 16, 32, 64)", (long long)dh);
    return EMO_ERR_UNSUPPORTED;
}

extern "C" int emo_relpos_attn_decode(const void* q, int64_t ld_q, void* kcache, void* vcache, int64_t T_max, const int64_t* lens, int64_t lens_off,
                                      int64_t mem_len, const void* k_new, const void* v_new, int64_t ld_new, const void* r_dist, int64_t ld_r, int64_t n_dist,
                                      const float* r_w_bias, const float* r_r_bias, void* out, int64_t ld_out, int dtype, int64_t n_streams, int64_t H,
                                      int64_t dh, emo_stream_t stream) {
    EMO_CHECK(q && kcache && vcache && lens && out && r_dist && r_w_bias && r_r_bias, "emo_relpos_attn_decode: null pointer");
    EMO_CHECK(dh == 16 || dh == 32 || dh == 64 || dh == 128, "emo_relpos_attn_decode: d_head must be 16, 32, 64 or 128");
    EMO_CHECK(T_max * 4 <= 128 * 1024, "emo_relpos_attn_decode: T_max too large for the LDS score buffer");
    EMO_CHECK(!k_new == !v_new, "emo_relpos_attn_decode: k_new and v_new go together");
    EMO_CHECK(n_dist >= T_max, "emo_relpos_attn_decode: r_dist needs a row for every distance 0 .. T_max-1");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK((((uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)r_dist) & 15) == 0 && (H * dh) % ve == 0 && ld_r % ve == 0,
              "emo_relpos_attn_decode: caches / r_dist must be 16-B aligned");
    EMO_CHECK(T_max * 4 <= 128 * 1024, "emo_softmax_attn_decode: T_max too large for the LDS score buffer");
    dim3 grid((unsigned)(n_streams * H));
    const size_t lds = (size_t)T_max * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EMO_F32) {
        static bool a = false;
        if (!a) { (void)hipFuncSetAttribute((const void*)relattn_decode_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); a = true; }
        hipLaunchKernelGGL(relattn_decode_kernel<float>, grid, dim3(256), lds, st, (const float*)q, ld_q, (float*)kcache, (float*)vcache, T_max, lens, lens_off,
                           mem_len, (const float*)k_new, (const float*)v_new, ld_new, (const float*)r_dist, ld_r, r_w_bias, r_r_bias, (float*)out, ld_out, H,
                           (int)dh);
    } else {
        static bool a = false;
        if (!a) { (void)hipFuncSetAttribute((const void*)relattn_decode_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); a = true; }
        hipLaunchKernelGGL(relattn_decode_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)q, ld_q, (bf16_t*)kcache, (bf16_t*)vcache, T_max, lens,
                           lens_off, mem_len, (const bf16_t*)k_new, (const bf16_t*)v_new, ld_new, (const bf16_t*)r_dist, ld_r, r_w_bias, r_r_bias,
                           (bf16_t*)out, ld_out, H, (int)dh);
    }
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
