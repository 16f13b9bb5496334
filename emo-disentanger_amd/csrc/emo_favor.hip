// K3+K4 — FAVOR+ feature map fused with causal linear attention (Performer), forward + backward.
//
// Math per (batch b, head h), F = n_feat features, m = F/2, c = dh^-1/4:
//   phi(x)  = [exp(u - off), exp(-u - off)],  u = (c x) W  (W = omega [dh,m]),  off = |c x|^2/2 + ln(F)/2
//   S_t = sum_{j<=t} phi(k_j) (x) v_j   z_t = sum_{j<=t} phi(k_j)
//   out_t = phi(q_t)^T S_t / den_t,   den_t = phi(q_t).z_t + eps
// Chunked prefix-sum evaluation (chunk C tokens): the intra-chunk part is a masked C x C product,
// the inter-chunk part goes through the running state, which lives in MFMA accumulators (fp32
// registers) for the whole scan and is mirrored to LDS once per chunk as an MFMA operand.
// One workgroup (8 waves) per (b,h); every contraction is an "NT" product of two LDS images
// whose reduction index is contiguous, so fragments are single ds_read_b128 (bf16) / ds_read_b32
// (exact-f32 mode on v_mfma_f32_16x16x4_f32).  Image row strides are padded (+16 B) so that the
// 16 rows of a fragment read hit distinct bank groups.
// Backward = two independent scans: a forward sweep producing dq and a reverse sweep producing
// dk, dv (state R_t = sum_{s>t} phi(q_s) (x) [dN_s, dD_s]); the feature-map Jacobian is fused.
//   dN_t = dout_t/den_t,  dD_t = -(dout_t.out_t)/den_t
#include "emo_common.h"

#include "emo_lds_mma.h"

// 8 waves per workgroup, one workgroup per (b, h, time segment).  Backward kernels: 120-155 KB of LDS => one workgroup per CU (two waves
// per SIMD); bf16 forward: 81 KB => two workgroups per CU (see favor_fwd_kernel).
constexpr int FT = 512, FW = FT / 64;
// bf16 chunk sizes (tokens per scan step) of the forward / dq / dk,dv kernels at d_head 64, 128 features
#ifndef FAVOR_CFB
#define FAVOR_CFB 64
#endif
#ifndef FAVOR_CQB
#define FAVOR_CQB 64
#endif
#ifndef FAVOR_CKB
#define FAVOR_CKB 64
#endif
#ifndef FAVOR_PAD16
#define FAVOR_PAD16 0
#endif

// per-row feature offset: off[t] = 0.5*c^2*|x_t|^2 + 0.5*ln(F); TPR threads per row, shuffle reduce
template <typename CT, int DH, int C>
__device__ __forceinline__ void row_offsets(const CT* X, int ld, float* off, float c2, float half_ln_f, int tid) {
    constexpr int TPR = FT / C;           // threads per row (C in {16,32,64} -> 16,8,4)
    constexpr int EPT = CMax<DH / TPR, 1>::v;
    const int r = tid / TPR, part = tid % TPR;
    float s = 0.f;
    if constexpr (sizeof(CT) == 2 && EPT == 8 && DH % 8 == 0) {       // one 16-B read instead of 8 scalar ones
        const bf16x8 v = *(const bf16x8*)(X + r * ld + part * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = (float)v[e]; s += x * x; }
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int d = part * EPT + e;
            if (d < DH) { float x = to_f32<CT>(X[r * ld + d]); s += x * x; }
        }
    }
#pragma unroll
    for (int o = TPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (part == 0) off[r] = 0.5f * c2 * s + half_ln_f;
}

// feature images.  WT image [MF][LDX] holds omega^T (k = d).  X image [C][LDX].
// row-major:  Ff[t][f]  (rows<->m via R=WT, col<->t via C=X)
template <typename CT, int DHP, int MF, int C>
__device__ __forceinline__ void features_rowmajor(CT* Ff, int ldf, const CT* WT, const CT* X, int ldx, const float* off, float cs,
                                                  int valid_rows, int wave, int lane) {
    constexpr int NT = (MF / 16) * (C / 16);
    static_assert(FW % (C / 16) == 0, "a wave's tiles share the token block");
    const int ct = wave % (C / 16);
    Frags<CT, DHP> xf;
    xf.load(X, ldx, ct * 16, lane);
    _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
        const int rt = tile / (C / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mm16_c<CT, DHP>(acc, WT, ldx, rt * 16, xf, lane);
        const int t = ct * 16 + (lane & 15), m0 = rt * 16 + (lane >> 4) * 4;
        const float o = off[t];
        const float ok = t < valid_rows ? 1.f : 0.f;
        float p[4], n[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { float u = cs * acc[r]; p[r] = Img<CT>::ex(u - o) * ok; n[r] = Img<CT>::ex(-u - o) * ok; }
        Img<CT>::store4(Ff + t * ldf + m0, p[0], p[1], p[2], p[3]);
        Img<CT>::store4(Ff + t * ldf + MF + m0, n[0], n[1], n[2], n[3]);
    }
}
// both layouts of the same features from ONE product + ONE set of exps: Ff[t][f] (8-B stores) and FTt[f][t] (2-B stores, 16 lanes = 32
// contiguous bytes).  r01 ablation: the feature phase (MFMA -> 8 quarter-rate v_exp per lane -> store, all dependent) was 35 % of the
// forward kernel; the second orientation cost a full second pass.
template <typename CT, int DHP, int MF, int C>
__device__ __forceinline__ void features_both(CT* Ff, int ldf, CT* FTt, int ldt, const CT* WT, const CT* X, int ldx, const float* off, float cs,
                                              int valid_rows, int wave, int lane) {
    constexpr int NT = (MF / 16) * (C / 16);
    static_assert(FW % (C / 16) == 0, "a wave's tiles share the token block");
    const int ct = wave % (C / 16);
    Frags<CT, DHP> xf;
    xf.load(X, ldx, ct * 16, lane);
#pragma unroll
    for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
        const int tile = wave + FW * tile_i;
        if (NT % FW != 0 && tile >= NT) break;
        const int rt = tile / (C / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mm16_c<CT, DHP>(acc, WT, ldx, rt * 16, xf, lane);
        const int t = ct * 16 + (lane & 15), m0 = rt * 16 + (lane >> 4) * 4;
        const float o = off[t];
        const float ok = t < valid_rows ? 1.f : 0.f;
        float p[4], n[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { float u = cs * acc[r]; p[r] = Img<CT>::ex(u - o) * ok; n[r] = Img<CT>::ex(-u - o) * ok; }
        Img<CT>::store4(Ff + t * ldf + m0, p[0], p[1], p[2], p[3]);
        Img<CT>::store4(Ff + t * ldf + MF + m0, n[0], n[1], n[2], n[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            FTt[(m0 + r) * ldt + t] = from_f32<CT>(p[r]);
            FTt[(MF + m0 + r) * ldt + t] = from_f32<CT>(n[r]);
        }
    }
}
// transposed: FT[f][t]  (rows<->t via R=X, col<->m via C=WT)
template <typename CT, int DHP, int MF, int C>
__device__ __forceinline__ void features_transposed(CT* FT, int ldt, const CT* WT, const CT* X, int ldx, const float* off, float cs,
                                                    int valid_rows, int wave, int lane) {
    constexpr int NT = (MF / 16) * (C / 16);
    static_assert(FW % (MF / 16) == 0, "a wave's tiles share the feature block");
    const int ct = wave % (MF / 16);
    Frags<CT, DHP> wf;
    wf.load(WT, ldx, ct * 16, lane);
    _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
        const int rt = tile / (MF / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mm16_c<CT, DHP>(acc, X, ldx, rt * 16, wf, lane);
        const int t0 = rt * 16 + (lane >> 4) * 4, m = ct * 16 + (lane & 15);
        float p[4], n[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float o = off[t0 + r];
            const float ok = (t0 + r) < valid_rows ? 1.f : 0.f;
            float u = cs * acc[r];
            p[r] = Img<CT>::ex(u - o) * ok;
            n[r] = Img<CT>::ex(-u - o) * ok;
        }
        Img<CT>::store4(FT + m * ldt + t0, p[0], p[1], p[2], p[3]);
        Img<CT>::store4(FT + (MF + m) * ldt + t0, n[0], n[1], n[2], n[3]);
    }
}

// KIND: 0 = forward, 1 = dq, 2 = dk/dv kernel.  bf16 row pads: ds_read_b128 operand reads (lane -> row l&15, 16-B chunk l>>4) are
// bank-conflict-free only for row strides = 8 (mod 16) dwords, i.e. a pad of 16 bf16 on the 64- / 128-wide rows (PMC r01: with the 8-element
// pad 41-45 % of all LDS cycles were bank conflicts and the LDS array was busy 64-86 % of the kernel time).  The dk/dv kernel's images do
// not fit 160 KB with that pad and keep 8.
template <typename CT, int DH, int MF, int C, int KIND> struct FavorDims {
    static constexpr int F = 2 * MF;
    static constexpr int KMIN = Img<CT>::KMIN, PAD = (sizeof(CT) == 2 && KIND != 2 && FAVOR_PAD16) ? 16 : Img<CT>::PAD;
    static constexpr int DHP = CMax<DH, KMIN>::v;
    static constexpr int MFP = CMax<MF, KMIN>::v;
    static constexpr int CP = CMax<C, KMIN>::v;
    static constexpr int LDX = DHP + PAD;  // k = d
    static constexpr int LDM = MFP + PAD;  // k = m
    static constexpr int LDC = CP + PAD;   // k = j / t (chunk index)
    static constexpr int LDF = F + PAD;    // k = f
};

template <typename CT>
__device__ __forceinline__ void zero_img(CT* img, int n, int tid) {
    for (int i = tid; i < n; i += FT) img[i] = from_f32<CT>(0.f);
}

// =============================================================================================== forward
// sum over segments [p_lo, p_hi) of one element of the per-segment state increments (workspace [.., P, stride])
__device__ __forceinline__ float seg_sum(const float* __restrict__ ws, int64_t stride, int p_lo, int p_hi, int64_t idx) {
    float s = 0.f;
    for (int p = p_lo; p < p_hi; ++p) s += ws[p * stride + idx];
    return s;
}

// Segment-parallel scan (P > 1): the T axis is cut into P segments of Ts tokens (Ts a multiple of the chunk size).  A first
// "state only" launch (SO = true) computes every segment's state increment into S_ws [B*H, P, F, DH] / z_ws [B*H, P, F]; the main
// launch then starts each segment from the sum of the increments before it.  P = 1 is the plain single-pass scan.
// SO = true: "state only" pass of the segment-parallel scan (writes the segment's state increment to S_ws / z_ws).
// bf16 (TRV): TWO workgroups per CU.  A barrier-delimited phase costs about the same latency whatever its size (chunk 32 on two workgroups =
// chunk 64 on one, 16-wave workgroups are slower), so the way to more throughput is a second, independent scan on the same CU — which needs
// the images under 80 KB: no transposed copies (K features and V are kept row-major only; the products that contract over the token index
// read them with ds_read_b64_tr_b16 in the permuted k order, the other operand with two 8-B reads), the A matrix re-uses the q rows and the
// V rows re-use the k rows (both dead after the feature phase; V goes to LDS one phase later).  79.9 KB + 1.3 KB of vectors.
template <typename CT, int DH, int MF, int C, bool SO>
__global__ __launch_bounds__(FT, (sizeof(CT) == 2 && C == 64) ? 2 : 1) void favor_fwd_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                        const float* __restrict__ omega, CT* __restrict__ out, int64_t ld_out,
                                                        float* __restrict__ den_g, float* __restrict__ state_S, float* __restrict__ state_z,
                                                        int64_t T, int64_t H, float eps, float* __restrict__ S_ws, float* __restrict__ z_ws,
                                                        int P, int64_t Ts) {
    typedef FavorDims<CT, DH, MF, C, 0> D;
    constexpr int F = D::F, LDX = D::LDX, LDC = D::LDC, LDF = D::LDF, DHP = D::DHP, CP = D::CP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool TRV = sizeof(CT) == 2 && C == 64;
    CT* WT = (CT*)smem;                 // [MF][LDX]
    CT* Xq = WT + MF * LDX;             // [C][LDX]   TRV: re-used by Am [C][LDC] after the feature phase
    CT* Xk = Xq + C * CMax<LDX, LDC>::v;   // [C][LDX]   TRV: re-used by the V rows Vr [C][LDX] after the feature phase
    CT* VT = Xk + C * LDX;              // [DH][LDC]  (not TRV)
    CT* Vr = Xk;
    CT* Qf = TRV ? VT : VT + DH * LDC;  // [C][LDF]
    CT* Kf = Qf + C * LDF;              // [C][LDF]
    CT* KfT = Kf + C * LDF;             // [F][LDC]   (not TRV)
    CT* Am = TRV ? Xq : KfT + F * LDC;  // [C][LDC]
    CT* ST = TRV ? KfT : Am + C * LDC;  // [DH][LDF]   S^T mirror (k = f)
    float* offq = (float*)(ST + DH * LDF);
    float* offk = offq + C;
    float* dens = offk + C;
    float* zz = dens + C;               // [F]

    const int abl = P >> 8;   // diagnostics (EMO_FAVOR_ABLATE): skip phases, results are garbage
    P &= 0xFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // segment-parallel scan: block = (b, h, segment p); the segment covers tokens [tbeg, tend)
    const int64_t bh = blockIdx.x / P, b = bh / H, h = bh % H;
    const int p_seg = (int)(blockIdx.x % P);
    const int64_t tbeg = (int64_t)p_seg * Ts;
    const int64_t tend = (tbeg + Ts < T) ? tbeg + Ts : T;
    if (SO && p_seg == P - 1) return;               // nobody consumes the last segment's increment
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    CT* ob = out + (b * T) * ld_out + h * DH;
    float* dg = den_g + bh * T;
    const float cs = rsqrtf(sqrtf((float)DH));   // dh^-1/4
    const float half_ln_f = 0.5f * logf((float)F);

    // omega^T image, zero state
    zero_img(WT, MF * LDX, tid);
    zero_img(ST, DH * LDF, tid);
    if constexpr (!TRV) {
        zero_img(VT, DH * LDC, tid);
        zero_img(KfT, F * LDC, tid);
        zero_img(Am, C * LDC, tid);
    }
    for (int i = tid; i < F; i += FT) zz[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < DH * MF; i += FT) { const int d = i / MF, m = i % MF; WT[m * LDX + d] = from_f32<CT>(omega[i]); }

    constexpr int NTS = (F / 16) * (DH / 16);          // state tiles: rows<->f (R=KfT), col<->d (C=VT)
    constexpr int NTS_W = (NTS + FW - 1) / FW;
    f32x4 sacc[NTS_W];
#pragma unroll
    for (int i = 0; i < NTS_W; ++i) sacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (!SO && p_seg > 0) {                         // state carried in = sum of the increments of segments 0..p-1
        const float* Si = S_ws + bh * P * (int64_t)F * DH;
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int ft = tile / (DH / 16), dt = tile % (DH / 16);
                const int d = dt * 16 + (lane & 15), f0 = ft * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[i][r] = seg_sum(Si, (int64_t)F * DH, 0, p_seg, (f0 + r) * DH + d);
                Img<CT>::store4(ST + d * LDF + f0, sacc[i][0], sacc[i][1], sacc[i][2], sacc[i][3]);
            }
        }
        for (int f = tid; f < F; f += FT) zz[f] = seg_sum(z_ws + bh * P * (int64_t)F, F, 0, p_seg, f);
    }
    RowPrefetch<CT, DH, DHP, C, FT> pq, pk;
    RowPrefetch<CT, DH, DHP, C, FT, !TRV> pv;   // not TRV: only ever stored transposed
    if (tbeg < tend) {
        const int v0 = (int)((tend - tbeg) < C ? (tend - tbeg) : C);
        if constexpr (!SO) pq.load(qb + tbeg * ld, ld, v0, tid);
        pk.load(kb + tbeg * ld, ld, v0, tid);
        pv.load(vb + tbeg * ld, ld, v0, tid);
    }
    for (int64_t t0 = tbeg; t0 < tend; t0 += C) {
        const int valid = (int)((tend - t0) < C ? (tend - t0) : C);
        __syncthreads();
        if (!(abl & 1)) {
        if constexpr (!SO) pq.store_rows_off(Xq, LDX, tid, offq, cs * cs, half_ln_f);
        pk.store_rows_off(Xk, LDX, tid, offk, cs * cs, half_ln_f);
        }
        if constexpr (!TRV) { if (!(abl & 2)) pv.store_T(VT, LDC, tid); }
        const bool more = t0 + C < tend;
        const int vn = more ? (int)((tend - t0 - C) < C ? (tend - t0 - C) : C) : 0;
        if (more) {                                // next chunk's q/k/v stay in flight during this chunk's compute
            if constexpr (!SO) pq.load(qb + (t0 + C) * ld, ld, vn, tid);
            pk.load(kb + (t0 + C) * ld, ld, vn, tid);
            if constexpr (!TRV) pv.load(vb + (t0 + C) * ld, ld, vn, tid);
        }
        __syncthreads();
        if (!(abl & 8)) {
        if constexpr (TRV) {
            if constexpr (!SO) features_rowmajor<CT, DHP, MF, C>(Qf, LDF, WT, Xq, LDX, offq, cs, C, wave, lane);
            features_rowmajor<CT, DHP, MF, C>(Kf, LDF, WT, Xk, LDX, offk, cs, valid, wave, lane);
        } else if constexpr (!SO) {
            features_rowmajor<CT, DHP, MF, C>(Qf, LDF, WT, Xq, LDX, offq, cs, C, wave, lane);
            features_both<CT, DHP, MF, C>(Kf, LDF, KfT, LDC, WT, Xk, LDX, offk, cs, valid, wave, lane);
        } else {
            features_transposed<CT, DHP, MF, C>(KfT, LDC, WT, Xk, LDX, offk, cs, valid, wave, lane);
        }
        }
        __syncthreads();
        if constexpr (TRV) {                       // the k rows are dead: the V rows take their place (read from the out phase on)
            pv.store_rows(Vr, LDX, tid);
            if (more) pv.load(vb + (t0 + C) * ld, ld, vn, tid);
            if constexpr (SO) __syncthreads();
        }
        if constexpr (!SO) {
        // A[t][j] = Qf[t].Kf[j] masked j<=t : rows<->j (R=Kf), col<->t (C=Qf)
        if (!(abl & 16)) {
            constexpr int NT = (C / 16) * (C / 16);
            const int tt = wave % (C / 16);
            Frags<CT, F> qff;
            qff.load(Qf, LDF, tt * 16, lane);
            _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
                const int jt = tile / (C / 16);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (jt <= tt) mm16_c<CT, F>(acc, Kf, LDF, jt * 16, qff, lane);
                const int t = tt * 16 + (lane & 15), j0 = jt * 16 + (lane >> 4) * 4;
                Img<CT>::store4(Am + t * LDC + j0, j0 <= t ? acc[0] : 0.f, j0 + 1 <= t ? acc[1] : 0.f, j0 + 2 <= t ? acc[2] : 0.f,
                                j0 + 3 <= t ? acc[3] : 0.f);
            }
        }
        __syncthreads();
        // den[t] = rowsum(A[t]) + Qf[t].z_prev + eps
        if (!(abl & 32)) {
            constexpr int TPR = FT / C;
            constexpr int VE = 16 / sizeof(CT);
            const int r = tid / TPR, part = tid % TPR;
            float s = 0.f;
            if constexpr ((C / TPR) % VE == 0 && (F / TPR) % VE == 0 && sizeof(CT) == 2) {
                s = sum_contig<CT, C / TPR>(Am + r * LDC + part * (C / TPR)) +
                    dot_contig<CT, F / TPR>(Qf + r * LDF + part * (F / TPR), zz + part * (F / TPR));
            } else {
                for (int j = part; j < C; j += TPR) s += to_f32<CT>(Am[r * LDC + j]);
                for (int f = part; f < F; f += TPR) s += to_f32<CT>(Qf[r * LDF + f]) * zz[f];
            }
#pragma unroll
            for (int o = TPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (part == 0) {
                dens[r] = s + eps;
                if (r < valid) dg[t0 + r] = s + eps;
            }
        }
        __syncthreads();
        // out^T: rows<->d, col<->t : VT.Am^T (K=C) + ST.Qf^T (K=F)
        if (!(abl & 64)) {
            constexpr int NT = (DH / 16) * (C / 16);
            const int tt = wave % (C / 16);
            Frags<CT, CP> amf;
            Frags<CT, F> qff;
            typename Img<CT>::V amp[2];
            if constexpr (TRV) { amp[0] = load_perm<CT>(Am, LDC, tt * 16, 0, lane); amp[1] = load_perm<CT>(Am, LDC, tt * 16, 1, lane); }
            else amf.load(Am, LDC, tt * 16, lane);
            qff.load(Qf, LDF, tt * 16, lane);
            _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
                const int dt = tile / (C / 16);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if constexpr (TRV) {
                    acc = Img<CT>::mma(load_perm_tr((const bf16_t*)Vr, LDX, dt * 16, 0, lane), amp[0], acc);
                    acc = Img<CT>::mma(load_perm_tr((const bf16_t*)Vr, LDX, dt * 16, 1, lane), amp[1], acc);
                } else {
                    mm16_c<CT, CP>(acc, VT, LDC, dt * 16, amf, lane);
                }
                mm16_c<CT, F>(acc, ST, LDF, dt * 16, qff, lane);
                const int t = tt * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
                if (t < valid) {
                    const float inv = 1.f / dens[t];
                    Img<CT>::store4(ob + (t0 + t) * ld_out + d0, acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
                }
            }
        }
        __syncthreads();
        }   // !SO
        // state: S[f][d] += sum_j KfT[f][j] VT[d][j]; mirror ST[d][f]; z += colsum
        if (!(abl & 128)) {
        static_assert(FW % (DH / 16) == 0, "a wave's state tiles share the d block");
        Frags<CT, CP> vtf;
        typename Img<CT>::V vtr[2];
        if constexpr (TRV) {
            vtr[0] = load_perm_tr((const bf16_t*)Vr, LDX, (wave % (DH / 16)) * 16, 0, lane);
            vtr[1] = load_perm_tr((const bf16_t*)Vr, LDX, (wave % (DH / 16)) * 16, 1, lane);
        } else {
            vtf.load(VT, LDC, (wave % (DH / 16)) * 16, lane);
        }
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int ft = tile / (DH / 16), dt = tile % (DH / 16);
                if constexpr (TRV) {
                    sacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)Kf, LDF, ft * 16, 0, lane), vtr[0], sacc[i]);
                    sacc[i] = Img<CT>::mma(load_perm_tr((const bf16_t*)Kf, LDF, ft * 16, 1, lane), vtr[1], sacc[i]);
                } else {
                    mm16_c<CT, CP>(sacc[i], KfT, LDC, ft * 16, vtf, lane);
                }
                if constexpr (!SO) {
                    const int d = dt * 16 + (lane & 15), f0 = ft * 16 + (lane >> 4) * 4;
                    Img<CT>::store4(ST + d * LDF + f0, sacc[i][0], sacc[i][1], sacc[i][2], sacc[i][3]);
                }
            }
        }
        }
        if constexpr (TRV) {                       // z[f] += column sums of the row-major K features: 4 lanes per f, 16 rows each
            const int f = tid >> 2, part = tid & 3;
            if (f < F) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) s += (float)Kf[(part * 16 + j) * LDF + f];
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (part == 0) zz[f] += s;
            }
        } else if constexpr (sizeof(CT) == 2 && C == 64 && 4 * F <= FT) {
            const int f = tid >> 2, part = tid & 3;
            if (f < F) {
                float s = sum_contig<CT, 16>(KfT + f * LDC + part * 16);
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (part == 0) zz[f] += s;
            }
        } else {
            for (int f = tid; f < F; f += FT) {
                float s = 0.f;
                for (int j = 0; j < C; ++j) s += to_f32<CT>(KfT[f * LDC + j]);
                zz[f] += s;
            }
        }
    }
    __syncthreads();
    float* So = SO ? S_ws + (bh * P + p_seg) * (int64_t)F * DH : ((state_S && p_seg == P - 1) ? state_S + bh * (int64_t)F * DH : nullptr);
    float* zo = SO ? z_ws + (bh * P + p_seg) * (int64_t)F : ((state_S && p_seg == P - 1) ? state_z + bh * (int64_t)F : nullptr);
    if (So) {
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int ft = tile / (DH / 16), dt = tile % (DH / 16);
                const int d = dt * 16 + (lane & 15), f0 = ft * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) So[(f0 + r) * DH + d] = sacc[i][r];
            }
        }
        for (int f = tid; f < F; f += FT) zo[f] = zz[f];
    }
}

// =============================================================================================== backward, shared pieces
// load dout/out rows -> G image (dN = dout/den) [C][LDX] (+ optional transposed GT [DH][LDC]) and dD[t]
template <typename CT, int DH, int DHP, int C>
__device__ __forceinline__ void load_grads(CT* G, int ldg, CT* GT, int ldgt, float* dD, const CT* __restrict__ dout, const CT* __restrict__ outp,
                                           int64_t ld_out, const float* __restrict__ den, int valid, int tid) {
    constexpr int TPR = FT / C;
    constexpr int EPT = CMax<DHP / TPR, 1>::v;
    const int r = tid / TPR, part = tid % TPR;
    const float inv = r < valid ? 1.f / den[r] : 0.f;
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int d = part * EPT + e;
        if (d < DHP) {
            float g = 0.f, o = 0.f;
            if (r < valid && d < DH) { g = to_f32<CT>(dout[(int64_t)r * ld_out + d]); o = to_f32<CT>(outp[(int64_t)r * ld_out + d]); }
            dot += g * o;
            const CT gn = from_f32<CT>(g * inv);
            G[r * ldg + d] = gn;
            if (GT && d < DH) GT[d * ldgt + r] = gn;
        }
    }
#pragma unroll
    for (int o = TPR >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (part == 0) dD[r] = -dot * inv;
}

// prefetched variant of load_grads: dout / out rows arrive through RowPrefetch registers, den through `dn`
template <typename CT, int DH, int DHP, int C, int NI_>
__device__ __forceinline__ void store_grads(const RowPrefetch<CT, DH, DHP, C, FT>& pg, const RowPrefetch<CT, DH, DHP, C, FT>& po,
                                            const float (&dn)[NI_], int valid, CT* G, int ldg, CT* GT, int ldgt, float* dD, int tid) {
    typedef RowPrefetch<CT, DH, DHP, C, FT> P;
#pragma unroll
    for (int i = 0; i < P::NI; ++i) {
        const int it = tid + FT * i;
        const bool act = it < C * P::CH;
        const int row = act ? it / P::CH : 0, c = (it % P::CH) * P::VE;
        const float inv = (act && row < valid) ? 1.f / dn[i] : 0.f;
        float dot = 0.f, g[P::VE];
#pragma unroll
        for (int e = 0; e < P::VE; ++e) { g[e] = to_f32<CT>(pg.r[i][e]); dot += g[e] * to_f32<CT>(po.r[i][e]); }
#pragma unroll
        for (int o = P::CH >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        if (act) {
            if (c == 0) dD[row] = -dot * inv;
#pragma unroll
            for (int e = 0; e < P::VE; ++e) {
                const CT gn = from_f32<CT>(g[e] * inv);
                G[row * ldg + c + e] = gn;
                if (GT) GT[(c + e) * ldgt + row] = gn;
            }
        }
    }
    if constexpr (DHP > DH) {
        for (int it = tid; it < C * (DHP - DH); it += FT) G[(it / (DHP - DH)) * ldg + DH + it % (DHP - DH)] = from_f32<CT>(0.f);
    }
}
template <typename CT, int DH, int DHP, int C, int NI_>
__device__ __forceinline__ void load_den(float (&dn)[NI_], const float* __restrict__ den, int valid, int tid) {
    typedef RowPrefetch<CT, DH, DHP, C, FT> P;
#pragma unroll
    for (int i = 0; i < P::NI; ++i) {
        const int it = tid + FT * i;
        const int row = it / P::CH;
        dn[i] = (it < C * P::CH && row < valid) ? den[row] : 1.f;
    }
}

// a = dPhi * Phi ; Adiff[t][m] = a+ - a- ; sumA[t] += a+ + a-   (lane owns f0..f0+3 (plus) and MF+f0.. (minus) of column t)
template <typename CT, int MF>
__device__ __forceinline__ void jac_epilogue(const f32x4& accP, const f32x4& accM, const CT* Ff, int ldf, CT* Adiff, int lda, float* sumA,
                                             const float* extra_vec /* z or r, [F] */, float extra_scale, int t, int m0, int lane) {
    float ad[4], s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float ap = (accP[r] + extra_vec[m0 + r] * extra_scale) * to_f32<CT>(Ff[t * ldf + m0 + r]);
        const float am = (accM[r] + extra_vec[MF + m0 + r] * extra_scale) * to_f32<CT>(Ff[t * ldf + MF + m0 + r]);
        ad[r] = ap - am;
        s += ap + am;
    }
    Img<CT>::store4(Adiff + t * lda + m0, ad[0], ad[1], ad[2], ad[3]);
    s = rows4_sum(s);
    if ((lane >> 4) == 0) atomicAdd(&sumA[t], s);
}

// pad columns m in [MF, MFP) of the (aliased) Adiff image must be finite zeros
template <typename CT, int MF, int MFP, int C>
__device__ __forceinline__ void zero_adiff_pad(CT* Adiff, int lda, int tid) {
    if constexpr (MFP > MF) {
        for (int i = tid; i < C * (MFP - MF); i += FT) Adiff[(i / (MFP - MF)) * lda + MF + i % (MFP - MF)] = from_f32<CT>(0.f);
    }
}

// dx[t][d] = c * ( sum_m W[d][m] Adiff[t][m] - (c x[t][d]) sumA[t] ) : rows<->d (R=W), col<->t (C=Adiff)
template <typename CT, int DH, int MFP, int C>
__device__ __forceinline__ void dx_from_adiff(const CT* W, int ldw, const CT* Adiff, int lda, const float* sumA, const CT* __restrict__ xg,
                                              int64_t ld, CT* __restrict__ dxg, int64_t ld_d, float cs, int valid, int wave, int lane) {
    constexpr int NT = (DH / 16) * (C / 16);
    const int tt = wave % (C / 16);
    Frags<CT, MFP> adf;
    adf.load(Adiff, lda, tt * 16, lane);
    _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
        const int dt = tile / (C / 16);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mm16_c<CT, MFP>(acc, W, ldw, dt * 16, adf, lane);
        const int t = tt * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
        if (t < valid) {
            const float sa = sumA[t];
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = cs * (acc[r] - cs * to_f32<CT>(xg[(int64_t)t * ld + d0 + r]) * sa);
            Img<CT>::store4(dxg + (int64_t)t * ld_d + d0, o[0], o[1], o[2], o[3]);
        }
    }
}

// =============================================================================================== backward: dq (forward sweep)
template <typename CT, int DH, int MF, int C>
__global__ __launch_bounds__(FT) void favor_bwd_dq_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                           const float* __restrict__ omega, const CT* __restrict__ out, const CT* __restrict__ dout,
                                                           int64_t ld_out, const float* __restrict__ den_g, CT* __restrict__ dq, int64_t ld_d,
                                                           int64_t T, int64_t H, const float* __restrict__ S_ws, const float* __restrict__ z_ws,
                                                           int P, int64_t Ts) {
    typedef FavorDims<CT, DH, MF, C, 1> D;
    constexpr int F = D::F, LDX = D::LDX, LDC = D::LDC, LDF = D::LDF, LDM = D::LDM, DHP = D::DHP, CP = D::CP, MFP = D::MFP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* WT = (CT*)smem;            // [MF][LDX]  omega^T (k=d)
    CT* W = WT + MF * LDX;         // [DH][LDM]  omega   (k=m)
    CT* Xq = W + DH * LDM;         // [C][LDX]
    CT* Xk = Xq + C * LDX;         // [C][LDX]   (aliased by Adiff [C][LDM] after the feature step)
    constexpr int XKSZ = C * CMax<LDX, LDM>::v;
    CT* Adiff = Xk;
    CT* Vr = Xk + XKSZ;            // [C][LDX]   v rows (k=d)
    CT* VT = Vr + C * LDX;         // [DH][LDC]
    CT* G = VT + DH * LDC;         // [C][LDX]   dN rows (k=d)
    CT* Qf = G + C * LDX;          // [C][LDF]
    CT* KfT = Qf + C * LDF;        // [F][LDC]
    CT* Pm = KfT + F * LDC;        // [C][LDC]
    CT* SF = Pm + C * LDC;         // [F][LDX]   S mirror (k=d)
    float* offq = (float*)(SF + F * LDX);
    float* offk = offq + C;
    float* dD = offk + C;
    float* sumA = dD + C;
    float* zz = sumA + C;          // [F]

    const int abl = P >> 8;   // diagnostics (EMO_FAVOR_ABLATE_DQ)
    P &= 0xFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bh = blockIdx.x / P, b = bh / H, h = bh % H;
    const int p_seg = (int)(blockIdx.x % P);
    const int64_t tbeg = (int64_t)p_seg * Ts;
    const int64_t tend = (tbeg + Ts < T) ? tbeg + Ts : T;
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* ob = out + (b * T) * ld_out + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    CT* dqb = dq + (b * T) * ld_d + h * DH;
    const float* dg = den_g + bh * T;
    const float cs = rsqrtf(sqrtf((float)DH));
    const float half_ln_f = 0.5f * logf((float)F);

    zero_img(WT, MF * LDX, tid);
    zero_img(W, DH * LDM, tid);
    zero_img(SF, F * LDX, tid);
    zero_img(VT, DH * LDC, tid);
    zero_img(KfT, F * LDC, tid);
    zero_img(Pm, C * LDC, tid);
    for (int i = tid; i < F; i += FT) zz[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < DH * MF; i += FT) {
        const int d = i / MF, m = i % MF;
        const CT w = from_f32<CT>(omega[i]);
        WT[m * LDX + d] = w;
        W[d * LDM + m] = w;
    }
    constexpr int NTS = (DH / 16) * (F / 16);   // state tiles: rows<->d (R=VT), col<->f (C=KfT) -> SF[f][d0..]
    constexpr int NTS_W = (NTS + FW - 1) / FW;
    f32x4 sacc[NTS_W];
#pragma unroll
    for (int i = 0; i < NTS_W; ++i) sacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (p_seg > 0) {                                // K-state carried in from segments 0..p-1
        const float* Si = S_ws + bh * P * (int64_t)F * DH;
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int dt = tile / (F / 16), ft = tile % (F / 16);
                const int f = ft * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[i][r] = seg_sum(Si, (int64_t)F * DH, 0, p_seg, f * DH + d0 + r);
                Img<CT>::store4(SF + f * LDX + d0, sacc[i][0], sacc[i][1], sacc[i][2], sacc[i][3]);
            }
        }
        for (int f = tid; f < F; f += FT) zz[f] = seg_sum(z_ws + bh * P * (int64_t)F, F, 0, p_seg, f);
    }
    typedef RowPrefetch<CT, DH, DHP, C, FT> PF;
    PF pq, pk, pv, pg, po;
    float dn[PF::NI];
    if (tbeg < tend) {
        const int v0 = (int)((tend - tbeg) < C ? (tend - tbeg) : C);
        pq.load(qb + tbeg * ld, ld, v0, tid); pk.load(kb + tbeg * ld, ld, v0, tid); pv.load(vb + tbeg * ld, ld, v0, tid);
        pg.load(gb + tbeg * ld_out, ld_out, v0, tid); po.load(ob + tbeg * ld_out, ld_out, v0, tid);
        load_den<CT, DH, DHP, C>(dn, dg + tbeg, v0, tid);
    }
    for (int64_t t0 = tbeg; t0 < tend; t0 += C) {
        const int valid = (int)((tend - t0) < C ? (tend - t0) : C);
        __syncthreads();
        if (!(abl & 1)) {
        pq.store_rows_off(Xq, LDX, tid, offq, cs * cs, half_ln_f);
        pk.store_rows_off(Xk, LDX, tid, offk, cs * cs, half_ln_f);
        pv.store_rows(Vr, LDX, tid);
        }
        if (!(abl & 2)) pv.store_T(VT, LDC, tid);
        if (!(abl & 4)) store_grads<CT, DH, DHP, C>(pg, po, dn, valid, G, LDX, (CT*)nullptr, 0, dD, tid);
        for (int i = tid; i < C; i += FT) sumA[i] = 0.f;
        if (t0 + C < tend) {
            const int64_t tn_ = t0 + C;
            const int vn = (int)((tend - tn_) < C ? (tend - tn_) : C);
            pq.load(qb + tn_ * ld, ld, vn, tid); pk.load(kb + tn_ * ld, ld, vn, tid); pv.load(vb + tn_ * ld, ld, vn, tid);
            pg.load(gb + tn_ * ld_out, ld_out, vn, tid); po.load(ob + tn_ * ld_out, ld_out, vn, tid);
            load_den<CT, DH, DHP, C>(dn, dg + tn_, vn, tid);
        }
        __syncthreads();
        if (!(abl & 8)) {
        features_rowmajor<CT, DHP, MF, C>(Qf, LDF, WT, Xq, LDX, offq, cs, C, wave, lane);
        features_transposed<CT, DHP, MF, C>(KfT, LDC, WT, Xk, LDX, offk, cs, valid, wave, lane);
        }
        // P[t][j] = dN_t.v_j + dD_t, masked j<=t : rows<->j (R=Vr), col<->t (C=G)
        if (!(abl & 16)) {
            constexpr int NT = (C / 16) * (C / 16);
            const int tt = wave % (C / 16);
            Frags<CT, DHP> gf;
            gf.load(G, LDX, tt * 16, lane);
            _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
                const int jt = tile / (C / 16);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (jt <= tt) mm16_c<CT, DHP>(acc, Vr, LDX, jt * 16, gf, lane);
                const int t = tt * 16 + (lane & 15), j0 = jt * 16 + (lane >> 4) * 4;
                const float dd = dD[t];
                Img<CT>::store4(Pm + t * LDC + j0, j0 <= t ? acc[0] + dd : 0.f, j0 + 1 <= t ? acc[1] + dd : 0.f,
                                j0 + 2 <= t ? acc[2] + dd : 0.f, j0 + 3 <= t ? acc[3] + dd : 0.f);
            }
        }
        __syncthreads();
        // dPhi_q^T: rows<->f, col<->t : KfT.Pm^T (K=C) + SF.G^T (K=DHP) + z[f] dD[t]; fused Jacobian -> Adiff, sumA
        zero_adiff_pad<CT, MF, MFP, C>(Adiff, LDM, tid);
        if (!(abl & 32)) {
            constexpr int NP = (MF / 16) * (C / 16);
            const int tt = wave % (C / 16);
            Frags<CT, CP> pmf;
            Frags<CT, DHP> gf;
            pmf.load(Pm, LDC, tt * 16, lane);
            gf.load(G, LDX, tt * 16, lane);
            _Pragma("unroll") for (int pr_i = 0; pr_i < (NP + FW - 1) / FW; ++pr_i) {
            const int pr = wave + FW * pr_i;
            if (NP % FW != 0 && pr >= NP) break;
                const int ft = pr / (C / 16);
                f32x4 aP = {0.f, 0.f, 0.f, 0.f}, aM = {0.f, 0.f, 0.f, 0.f};
                mm16_c<CT, CP>(aP, KfT, LDC, ft * 16, pmf, lane);
                mm16_c<CT, DHP>(aP, SF, LDX, ft * 16, gf, lane);
                mm16_c<CT, CP>(aM, KfT, LDC, MF + ft * 16, pmf, lane);
                mm16_c<CT, DHP>(aM, SF, LDX, MF + ft * 16, gf, lane);
                const int t = tt * 16 + (lane & 15), m0 = ft * 16 + (lane >> 4) * 4;
                jac_epilogue<CT, MF>(aP, aM, Qf, LDF, Adiff, LDM, sumA, zz, dD[t], t, m0, lane);
            }
        }
        __syncthreads();
        if (!(abl & 64)) dx_from_adiff<CT, DH, MFP, C>(W, LDM, Adiff, LDM, sumA, qb + t0 * ld, ld, dqb + t0 * ld_d, ld_d, cs, valid, wave, lane);
        // state S[f][d] (+)= ; mirror SF[f][d0..] ; z += colsum(KfT)
        if (!(abl & 128)) {
        static_assert(FW % (F / 16) == 0 || (F / 16) % FW == 0, "state tiles of a wave share the feature block");
        constexpr bool SH = FW % (F / 16) == 0;
        Frags<CT, CP> kff;
        if constexpr (SH) kff.load(KfT, LDC, (wave % (F / 16)) * 16, lane);
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int dt = tile / (F / 16), ft = tile % (F / 16);
                if constexpr (SH) mm16_c<CT, CP>(sacc[i], VT, LDC, dt * 16, kff, lane);
                else mm16<CT, CP>(sacc[i], VT, LDC, dt * 16, KfT, LDC, ft * 16, lane);
            }
        }
        }
        __syncthreads();   // all reads of SF / zz for this chunk are done
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int dt = tile / (F / 16), ft = tile % (F / 16);
                const int f = ft * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
                Img<CT>::store4(SF + f * LDX + d0, sacc[i][0], sacc[i][1], sacc[i][2], sacc[i][3]);
            }
        }
        if constexpr (sizeof(CT) == 2 && C == 64 && 4 * F <= FT) {
            const int f = tid >> 2, part = tid & 3;
            if (f < F) {
                float s = sum_contig<CT, 16>(KfT + f * LDC + part * 16);
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (part == 0) zz[f] += s;
            }
        } else {
            for (int f = tid; f < F; f += FT) {
                float s = 0.f;
                for (int j = 0; j < C; ++j) s += to_f32<CT>(KfT[f * LDC + j]);
                zz[f] += s;
            }
        }
    }
}

// =============================================================================================== backward: dk, dv (reverse sweep)
// SO = true: "state only" pass of the reverse segment scan: the segment's increments of R [F, DH] and r [F] go to R_ws / r_ws.
template <typename CT, int DH, int MF, int C, bool SO>
__global__ __launch_bounds__(FT) void favor_bwd_dkv_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                            const float* __restrict__ omega, const CT* __restrict__ out, const CT* __restrict__ dout,
                                                            int64_t ld_out, const float* __restrict__ den_g, CT* __restrict__ dk, CT* __restrict__ dv,
                                                            int64_t ld_d, int64_t T, int64_t H, float* __restrict__ R_ws, float* __restrict__ r_ws, int P,
                                                            int64_t Ts) {
    typedef FavorDims<CT, DH, MF, C, 2> D;
    constexpr int F = D::F, LDX = D::LDX, LDC = D::LDC, LDF = D::LDF, LDM = D::LDM, DHP = D::DHP, CP = D::CP, MFP = D::MFP;
    constexpr int LXC = CMax<LDX, LDC>::v, LXM = CMax<LDX, LDM>::v;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT* WT = (CT*)smem;             // [MF][LDX]
    CT* W = WT + MF * LDX;          // [DH][LDM]
    CT* Xq = W + DH * LDM;          // [C][LDX] -> aliased by PmT [C][LDC] after the feature step
    CT* PmT = Xq;
    CT* Xk = Xq + C * LXC;          // [C][LDX] -> aliased by AmT [C][LDC]
    CT* AmT = Xk;
    CT* Vr = Xk + C * LXC;          // [C][LDX]
    CT* G = Vr + C * LDX;           // [C][LDX] -> aliased by Adiff [C][LDM] after P is formed
    CT* Adiff = G;
    CT* GT = G + C * LXM;           // [DH][LDC]
    CT* Qf = GT + DH * LDC;         // [C][LDF]
    CT* Kf = Qf + C * LDF;          // [C][LDF]
    CT* QfT = Kf + C * LDF;         // [F][LDC]
    CT* RF = QfT + F * LDC;         // [F][LDX]  R mirror (k=d)
    CT* RT = RF + F * LDX;          // [DH][LDF] R^T mirror (k=f)
    float* offq = (float*)(RT + DH * LDF);
    float* offk = offq + C;
    float* dD = offk + C;
    float* sumA = dD + C;
    float* rr = sumA + C;           // [F]  r[f] = sum_{t>chunk} Qf_t[f] dD_t

    const int abl = P >> 8;   // diagnostics (EMO_FAVOR_ABLATE_DKV)
    P &= 0xFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bh = blockIdx.x / P, b = bh / H, h = bh % H;
    const int p_seg = (int)(blockIdx.x % P);
    const int64_t tbeg = (int64_t)p_seg * Ts;
    const int64_t tend = (tbeg + Ts < T) ? tbeg + Ts : T;
    if (SO && p_seg == 0) return;                   // nobody consumes the first segment's increment
    const CT* qb = q + (b * T) * ld + h * DH;
    const CT* kb = k + (b * T) * ld + h * DH;
    const CT* vb = v + (b * T) * ld + h * DH;
    const CT* ob = out + (b * T) * ld_out + h * DH;
    const CT* gb = dout + (b * T) * ld_out + h * DH;
    CT* dkb = dk + (b * T) * ld_d + h * DH;
    CT* dvb = dv + (b * T) * ld_d + h * DH;
    const float* dg = den_g + bh * T;
    const float cs = rsqrtf(sqrtf((float)DH));
    const float half_ln_f = 0.5f * logf((float)F);

    zero_img(WT, MF * LDX, tid);
    zero_img(W, DH * LDM, tid);
    zero_img(RF, F * LDX, tid);
    zero_img(RT, DH * LDF, tid);
    zero_img(GT, DH * LDC, tid);
    zero_img(QfT, F * LDC, tid);
    for (int i = tid; i < F; i += FT) rr[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < DH * MF; i += FT) {
        const int d = i / MF, m = i % MF;
        const CT w = from_f32<CT>(omega[i]);
        WT[m * LDX + d] = w;
        W[d * LDM + m] = w;
    }
    constexpr int NTS = (DH / 16) * (F / 16);   // rows<->d (R=GT), col<->f (C=QfT)
    constexpr int NTS_W = (NTS + FW - 1) / FW;
    f32x4 racc[NTS_W];
#pragma unroll
    for (int i = 0; i < NTS_W; ++i) racc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (!SO && p_seg < P - 1) {                     // R-state carried in from segments p+1..P-1
        const float* Ri = R_ws + bh * P * (int64_t)F * DH;
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int dt = tile / (F / 16), ft = tile % (F / 16);
                const int f = ft * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) racc[i][r] = seg_sum(Ri, (int64_t)F * DH, p_seg + 1, P, f * DH + d0 + r);
                Img<CT>::store4(RF + f * LDX + d0, racc[i][0], racc[i][1], racc[i][2], racc[i][3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) RT[(d0 + r) * LDF + f] = from_f32<CT>(racc[i][r]);
            }
        }
        for (int f = tid; f < F; f += FT) rr[f] = seg_sum(r_ws + bh * P * (int64_t)F, F, p_seg + 1, P, f);
    }
    const int64_t nchunks = (tend > tbeg) ? (tend - tbeg + C - 1) / C : 0;
    typedef RowPrefetch<CT, DH, DHP, C, FT> PF;
    PF pq, pk, pv, pg, po;
    float dn[PF::NI];
    if (nchunks > 0) {
        const int64_t tl_ = tbeg + (nchunks - 1) * C;
        const int v0 = (int)(tend - tl_);
        pq.load(qb + tl_ * ld, ld, v0, tid);
        if constexpr (!SO) { pk.load(kb + tl_ * ld, ld, v0, tid); pv.load(vb + tl_ * ld, ld, v0, tid); }
        pg.load(gb + tl_ * ld_out, ld_out, v0, tid); po.load(ob + tl_ * ld_out, ld_out, v0, tid);
        load_den<CT, DH, DHP, C>(dn, dg + tl_, v0, tid);
    }
    for (int64_t ci = nchunks - 1; ci >= 0; --ci) {
        const int64_t t0 = tbeg + ci * C;
        const int valid = (int)((tend - t0) < C ? (tend - t0) : C);
        __syncthreads();
        if (!(abl & 1)) {
        pq.store_rows_off(Xq, LDX, tid, offq, cs * cs, half_ln_f);
        if constexpr (!SO) {
            pk.store_rows_off(Xk, LDX, tid, offk, cs * cs, half_ln_f);
            pv.store_rows(Vr, LDX, tid);
        }
        }
        if (!(abl & 4)) store_grads<CT, DH, DHP, C>(pg, po, dn, valid, G, LDX, GT, LDC, dD, tid);
        for (int i = tid; i < C; i += FT) sumA[i] = 0.f;
        if (ci > 0) {                                  // previous (earlier) chunk: full, stays in flight during this chunk's compute
            const int64_t tn_ = t0 - C;
            pq.load(qb + tn_ * ld, ld, C, tid);
            if constexpr (!SO) { pk.load(kb + tn_ * ld, ld, C, tid); pv.load(vb + tn_ * ld, ld, C, tid); }
            pg.load(gb + tn_ * ld_out, ld_out, C, tid); po.load(ob + tn_ * ld_out, ld_out, C, tid);
            load_den<CT, DH, DHP, C>(dn, dg + tn_, C, tid);
        }
        __syncthreads();
        if (!(abl & 8)) {
        if constexpr (!SO) {
            features_both<CT, DHP, MF, C>(Qf, LDF, QfT, LDC, WT, Xq, LDX, offq, cs, valid, wave, lane);
            features_rowmajor<CT, DHP, MF, C>(Kf, LDF, WT, Xk, LDX, offk, cs, valid, wave, lane);
        } else {
            features_transposed<CT, DHP, MF, C>(QfT, LDC, WT, Xq, LDX, offq, cs, valid, wave, lane);
        }
        }
        __syncthreads();   // Xq / Xk dead from here: PmT / AmT may overwrite them
        if constexpr (!SO) {
        // PmT[j][t] = dN_t.v_j + dD_t (t>=j) : rows<->t (R=G), col<->j (C=Vr)
        // AmT[j][t] = Qf_t.Kf_j       (t>=j) : rows<->t (R=Qf), col<->j (C=Kf)
        if (!(abl & 16)) {
            constexpr int NT = (C / 16) * (C / 16);
            const int jt = wave % (C / 16);
            Frags<CT, DHP> vrf;
            Frags<CT, F> kff;
            vrf.load(Vr, LDX, jt * 16, lane);
            kff.load(Kf, LDF, jt * 16, lane);
            _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
                const int tt = tile / (C / 16);
                f32x4 aP = {0.f, 0.f, 0.f, 0.f}, aA = {0.f, 0.f, 0.f, 0.f};
                if (tt >= jt) {
                    mm16_c<CT, DHP>(aP, G, LDX, tt * 16, vrf, lane);
                    mm16_c<CT, F>(aA, Qf, LDF, tt * 16, kff, lane);
                }
                const int j = jt * 16 + (lane & 15), tb = tt * 16 + (lane >> 4) * 4;
                float p[4], a[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool keep = (tb + r) >= j && (tb + r) < valid;
                    p[r] = keep ? aP[r] + dD[tb + r] : 0.f;
                    a[r] = keep ? aA[r] : 0.f;
                }
                // stores go to the aliased Xq/Xk storage: wait until every wave finished reading G/Vr/Qf/Kf? (G,Vr,Qf,Kf are
                // distinct buffers; Xq/Xk are dead) -> safe without an extra barrier
                Img<CT>::store4(PmT + j * LDC + tb, p[0], p[1], p[2], p[3]);
                Img<CT>::store4(AmT + j * LDC + tb, a[0], a[1], a[2], a[3]);
            }
        }
        __syncthreads();   // G dead from here: Adiff may overwrite it
        // dPhi_k^T: rows<->f, col<->j : QfT.PmT^T (K=C) + RF.Vr^T (K=DHP) + r[f]; fused Jacobian (uses Kf)
        zero_adiff_pad<CT, MF, MFP, C>(Adiff, LDM, tid);
        if (!(abl & 32)) {
            constexpr int NP = (MF / 16) * (C / 16);
            const int jt = wave % (C / 16);
            Frags<CT, CP> pmf;
            Frags<CT, DHP> vrf;
            pmf.load(PmT, LDC, jt * 16, lane);
            vrf.load(Vr, LDX, jt * 16, lane);
            _Pragma("unroll") for (int pr_i = 0; pr_i < (NP + FW - 1) / FW; ++pr_i) {
            const int pr = wave + FW * pr_i;
            if (NP % FW != 0 && pr >= NP) break;
                const int ft = pr / (C / 16);
                f32x4 aP = {0.f, 0.f, 0.f, 0.f}, aM = {0.f, 0.f, 0.f, 0.f};
                mm16_c<CT, CP>(aP, QfT, LDC, ft * 16, pmf, lane);
                mm16_c<CT, DHP>(aP, RF, LDX, ft * 16, vrf, lane);
                mm16_c<CT, CP>(aM, QfT, LDC, MF + ft * 16, pmf, lane);
                mm16_c<CT, DHP>(aM, RF, LDX, MF + ft * 16, vrf, lane);
                const int j = jt * 16 + (lane & 15), m0 = ft * 16 + (lane >> 4) * 4;
                jac_epilogue<CT, MF>(aP, aM, Kf, LDF, Adiff, LDM, sumA, rr, 1.f, j, m0, lane);
            }
        }
        // dV^T: rows<->d, col<->j : GT.AmT^T (K=C) + RT.Kf^T (K=F)
        if (!(abl & 2)) {
            constexpr int NT = (DH / 16) * (C / 16);
            const int jt = wave % (C / 16);
            Frags<CT, CP> amf;
            Frags<CT, F> kff;
            amf.load(AmT, LDC, jt * 16, lane);
            kff.load(Kf, LDF, jt * 16, lane);
            _Pragma("unroll") for (int tile_i = 0; tile_i < (NT + FW - 1) / FW; ++tile_i) {
            const int tile = wave + FW * tile_i;
            if (NT % FW != 0 && tile >= NT) break;
                const int dt = tile / (C / 16);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                mm16_c<CT, CP>(acc, GT, LDC, dt * 16, amf, lane);
                mm16_c<CT, F>(acc, RT, LDF, dt * 16, kff, lane);
                const int j = jt * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
                if (j < valid) Img<CT>::store4(dvb + (t0 + j) * ld_d + d0, acc[0], acc[1], acc[2], acc[3]);
            }
        }
        __syncthreads();
        if (!(abl & 64)) dx_from_adiff<CT, DH, MFP, C>(W, LDM, Adiff, LDM, sumA, kb + t0 * ld, ld, dkb + t0 * ld_d, ld_d, cs, valid, wave, lane);
        }   // !SO
        // state R[f][d] += sum_t QfT[f][t] GT[d][t]
        if (!(abl & 128)) {
        constexpr bool SH = FW % (F / 16) == 0;
        Frags<CT, CP> qtf;
        if constexpr (SH) qtf.load(QfT, LDC, (wave % (F / 16)) * 16, lane);
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int dt = tile / (F / 16), ft = tile % (F / 16);
                if constexpr (SH) mm16_c<CT, CP>(racc[i], GT, LDC, dt * 16, qtf, lane);
                else mm16<CT, CP>(racc[i], GT, LDC, dt * 16, QfT, LDC, ft * 16, lane);
            }
        }
        }
        if constexpr (!SO) {
        __syncthreads();   // all reads of RF / RT / rr for this chunk are done
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int dt = tile / (F / 16), ft = tile % (F / 16);
                const int f = ft * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
                Img<CT>::store4(RF + f * LDX + d0, racc[i][0], racc[i][1], racc[i][2], racc[i][3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) RT[(d0 + r) * LDF + f] = from_f32<CT>(racc[i][r]);
            }
        }
        }
        if constexpr (sizeof(CT) == 2 && C == 64 && 4 * F <= FT) {
            const int f = tid >> 2, part = tid & 3;
            if (f < F) {
                float s = dot_contig<CT, 16>(QfT + f * LDC + part * 16, dD + part * 16);
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (part == 0) rr[f] += s;
            }
        } else {
            for (int f = tid; f < F; f += FT) {
                float s = 0.f;
                for (int t = 0; t < C; ++t) s += to_f32<CT>(QfT[f * LDC + t]) * dD[t];
                rr[f] += s;
            }
        }
    }
    if constexpr (SO) {
        __syncthreads();
        float* Ro = R_ws + (bh * P + p_seg) * (int64_t)F * DH;
#pragma unroll
        for (int i = 0; i < NTS_W; ++i) {
            const int tile = wave + FW * i;
            if (tile < NTS) {
                const int dt = tile / (F / 16), ft = tile % (F / 16);
                const int f = ft * 16 + (lane & 15), d0 = dt * 16 + (lane >> 4) * 4;
                *(f32x4*)(Ro + f * DH + d0) = racc[i];
            }
        }
        for (int f = tid; f < F; f += FT) r_ws[(bh * P + p_seg) * (int64_t)F + f] = rr[f];
    }
}

// =============================================================================================== decode step (recurrent form)
// One workgroup (256 threads) per (stream, head).  Phase 1: phi(q), phi(k) (thread f < F).  Phase 2: ONE coalesced pass
// over the fp32 state S [F x dh]: thread (d = tid % dh, g = tid / dh) walks rows f = g, g+G, ... : S[f][d] += phi_k[f] v[d],
// num[d] += phi_q[f] S[f][d] — the state is read once and written once per token (6.4 MB per stream-step over 12 layers).
template <typename CT>
__global__ __launch_bounds__(256) void favor_decode_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                           const float* __restrict__ omega, float* __restrict__ state_S, float* __restrict__ state_z,
                                                           CT* __restrict__ out, int64_t ld_out, int64_t H, int dh, int mf, float eps) {
    __shared__ float xq[64], xk[64], xv[64], fq[128], fk[128], num[16][64], dpart[4];
    const int tid = threadIdx.x, F = 2 * mf;
    const int64_t sh = blockIdx.x, s = sh / H, h = sh % H;
    if (tid < dh) {
        xq[tid] = to_f32<CT>(q[s * ld + h * dh + tid]);
        xk[tid] = to_f32<CT>(k[s * ld + h * dh + tid]);
        xv[tid] = to_f32<CT>(v[s * ld + h * dh + tid]);
    }
    __syncthreads();
    const float cs = rsqrtf(sqrtf((float)dh)), half_ln_f = 0.5f * logf((float)F);
    float dn = 0.f;
    if (tid < F) {
        const int m = tid % mf;
        const float sgn = tid < mf ? 1.f : -1.f;
        float uq = 0.f, uk = 0.f, nq = 0.f, nk = 0.f;
        for (int d = 0; d < dh; ++d) {
            const float w = omega[d * mf + m];
            uq += xq[d] * w; uk += xk[d] * w;
            nq += xq[d] * xq[d]; nk += xk[d] * xk[d];
        }
        const float pq = __expf(sgn * cs * uq - (0.5f * cs * cs * nq + half_ln_f));
        const float pk = __expf(sgn * cs * uk - (0.5f * cs * cs * nk + half_ln_f));
        fq[tid] = pq;
        fk[tid] = pk;
        const float z = state_z[sh * F + tid] + pk;
        state_z[sh * F + tid] = z;
        dn = pq * z;
    }
    dn = wave_sum(dn);
    if ((tid & 63) == 0) dpart[tid >> 6] = dn;
    __syncthreads();
    const int G = 256 / dh, d = tid % dh, g = tid / dh;
    float acc = 0.f;
    float* Sb = state_S + sh * F * dh;
    const float vd = xv[d];
    for (int f = g; f < F; f += G) {
        const float sv = Sb[f * dh + d] + fk[f] * vd;
        Sb[f * dh + d] = sv;
        acc += fq[f] * sv;
    }
    num[g][d] = acc;
    __syncthreads();
    if (tid < dh) {
        float a = 0.f;
        for (int p = 0; p < G; ++p) a += num[p][tid];
        out[s * ld_out + h * dh + tid] = from_f32<CT>(a / (dpart[0] + dpart[1] + dpart[2] + dpart[3] + eps));
    }
}

// Compile-time (dh, mf) variant of the decode step: the whole state slice of the block (F x dh fp32 = 32 KB at 128 x 64) is requested
// with 16-B loads BEFORE the feature phase, so the HBM round trip overlaps the phi(q), phi(k) computation; everything is unrolled
// (r01: the generic kernel took 18 us per layer for 16.8 MB of state traffic = 0.9 TB/s, latency-bound on 32 dependent 4-B accesses).
template <typename CT, int DH, int MF>
__global__ __launch_bounds__(256) void favor_decode_fast_kernel(const CT* __restrict__ q, const CT* __restrict__ k, const CT* __restrict__ v, int64_t ld,
                                                                const float* __restrict__ omega, float* __restrict__ state_S,
                                                                float* __restrict__ state_z, CT* __restrict__ out, int64_t ld_out, int64_t H, float eps) {
    constexpr int F = 2 * MF, TPRW = DH / 4, R = 256 / TPRW, NP = (F + R - 1) / R;   // threads per state row, rows per pass, passes
    __shared__ float xq[DH], xk[DH], xv[DH], fq[F], fk[F], dpart[4];
    __shared__ f32x4 num[R][TPRW];
    const int tid = threadIdx.x;
    const int64_t sh = blockIdx.x, s = sh / H, h = sh % H;
    const int d4 = (tid % TPRW) * 4, g = tid / TPRW;
    float* Sb = state_S + sh * F * DH;
    f32x4 st[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int f = g + R * i;
        st[i] = f < F ? *(const f32x4*)(Sb + f * DH + d4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the omega column and z of this thread's feature do not depend on q / k either: requested before the barrier, next to the state
    float wcol[DH], zold = 0.f;
    if (tid < F) {
#pragma unroll
        for (int d = 0; d < DH; ++d) wcol[d] = omega[d * MF + (tid % MF)];
        zold = state_z[sh * F + tid];
    }
    if (tid < DH) {
        xq[tid] = to_f32<CT>(q[s * ld + h * DH + tid]);
        xk[tid] = to_f32<CT>(k[s * ld + h * DH + tid]);
        xv[tid] = to_f32<CT>(v[s * ld + h * DH + tid]);
    }
    __syncthreads();
    const float cs = rsqrtf(sqrtf((float)DH)), half_ln_f = 0.5f * logf((float)F);
    float dn = 0.f;
    if (tid < F) {
        const float sgn = tid < MF ? 1.f : -1.f;
        float uq = 0.f, uk = 0.f, nq = 0.f, nk = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            const float w = wcol[d];
            uq += xq[d] * w; uk += xk[d] * w;
            nq += xq[d] * xq[d]; nk += xk[d] * xk[d];
        }
        const float pq = __expf(sgn * cs * uq - (0.5f * cs * cs * nq + half_ln_f));
        const float pk = __expf(sgn * cs * uk - (0.5f * cs * cs * nk + half_ln_f));
        fq[tid] = pq;
        fk[tid] = pk;
        const float z = zold + pk;
        state_z[sh * F + tid] = z;
        dn = pq * z;
    }
    dn = wave_sum(dn);
    if ((tid & 63) == 0) dpart[tid >> 6] = dn;
    __syncthreads();
    const f32x4 vd = {xv[d4], xv[d4 + 1], xv[d4 + 2], xv[d4 + 3]};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int f = g + R * i;
        if (f < F) {
            const f32x4 sv = st[i] + fk[f] * vd;
            *(f32x4*)(Sb + f * DH + d4) = sv;
            acc += fq[f] * sv;
        }
    }
    num[g][tid % TPRW] = acc;
    __syncthreads();
    if (tid < DH) {
        float a = 0.f;
#pragma unroll
        for (int p = 0; p < R; ++p) a += num[p][tid >> 2][tid & 3];
        out[s * ld_out + h * DH + tid] = from_f32<CT>(a / (dpart[0] + dpart[1] + dpart[2] + dpart[3] + eps));
    }
}

// =============================================================================================== omega draw
// FAVOR+ orthogonal random features (fast-transformers orthogonal_random_matrix_): per block of dh columns,
// G ~ N(0,1)^{dh x dh} (supplied by the caller's RNG), Q = orthonormal basis of G's columns, column j scaled by
// the norm of ROW j of G.  Gram-Schmidt with re-orthogonalisation (CGS2) in LDS replaces the Householder QR of
// torch.qr: the two differ only by column signs, to which the feature set {exp(+u), exp(-u)} is invariant.
// grid = (n_layers * n_blocks); one wave per block; lane = row index.
__global__ __launch_bounds__(64) void favor_omega_kernel(const float* __restrict__ gauss, float* __restrict__ omega, int dh, int cols, int nblocks) {
    // Q is kept twice, row-major (Qr[row][col]: the update x -= Q c reads a lane's row) and column-major (Qc[col][row]: the projections c = Q^T x
    // read a lane's column), rows padded to 68 floats, so that both loops run on 16-B LDS reads: 32 + j / 2 LDS instructions per pass instead
    // of 128 + 2 j one-float reads (r04: 177 -> ~65 us for the 12 x (64 x 64) blocks of the benchmark model; the kernel is a chain of 64
    // dependent column steps on one wave per block — latency, not throughput).  Unused rows / columns (dh < 64) are zero.
    constexpr int LD = 68;
    __shared__ __attribute__((aligned(16))) float Qr[64 * LD];
    __shared__ __attribute__((aligned(16))) float Qc[64 * LD];
    __shared__ __attribute__((aligned(16))) float v[64], cf[64];
    const int lane = threadIdx.x;
    const int layer = blockIdx.x / nblocks, blk = blockIdx.x % nblocks;
    const float* G = gauss + (int64_t)blockIdx.x * dh * dh;
    for (int e = lane; e < 64 * LD; e += 64) { Qr[e] = 0.f; Qc[e] = 0.f; }
    float rn = 0.f;
    if (lane < dh)
        for (int b = 0; b < dh; ++b) { const float g = G[lane * dh + b]; rn += g * g; }
    rn = sqrtf(rn);                                   // norm of row `lane` of G
    const int start = blk * dh;
    const int ncol = (cols - start) < dh ? (cols - start) : dh;
    __syncthreads();
    for (int j = 0; j < ncol; ++j) {
        float x = lane < dh ? G[lane * dh + j] : 0.f; // column j, element `lane`
        const int j4 = (j + 3) >> 2;                  // (columns >= j of Q are still zero: whole quads can be summed)
        for (int pass = 0; pass < 2; ++pass) {
            v[lane] = x;
            __syncthreads();
            f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int a = 0; a < 16; ++a) c4 += *(const f32x4*)(Qc + lane * LD + 4 * a) * *(const f32x4*)(v + 4 * a);
            cf[lane] = (c4[0] + c4[1]) + (c4[2] + c4[3]);       // (lane >= j: column lane of Q is zero, so is the projection)
            __syncthreads();
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < j4; ++i) s4 += *(const f32x4*)(cf + 4 * i) * *(const f32x4*)(Qr + lane * LD + 4 * i);
            x -= (s4[0] + s4[1]) + (s4[2] + s4[3]);
            __syncthreads();
        }
        const float nrm = sqrtf(wave_sum(lane < dh ? x * x : 0.f));
        const float qv = x / nrm;
        if (lane < dh) { Qr[lane * LD + j] = qv; Qc[j * LD + lane] = qv; }
        const float scale = __shfl(rn, j, 64);
        if (lane < dh) omega[((int64_t)layer * dh + lane) * cols + start + j] = qv * scale;
        __syncthreads();
    }
}

extern "C" int emo_favor_draw_omega(const float* gauss, float* omega, int64_t n_layers, int64_t dh, int64_t n_feat, emo_stream_t stream) {
    EMO_CHECK(gauss && omega && dh > 0 && dh <= 64 && n_feat % 2 == 0, "emo_favor_draw_omega: bad args (d_head <= 64)");
    const int cols = (int)(n_feat / 2);
    const int nblocks = (cols + (int)dh - 1) / (int)dh;
    hipLaunchKernelGGL(favor_omega_kernel, dim3((unsigned)(n_layers * nblocks)), dim3(64), 0, (hipStream_t)stream, gauss, omega, (int)dh, cols, nblocks);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// =============================================================================================== host
template <typename CT, int DH, int MF, int C> static size_t fwd_lds() {
    typedef FavorDims<CT, DH, MF, C, 0> D;
    if (sizeof(CT) == 2 && C == 64)   // TRV layout (two workgroups per CU)
        return sizeof(CT) * (size_t)(MF * D::LDX + C * CMax<D::LDX, D::LDC>::v + C * D::LDX + 2 * C * D::LDF + DH * D::LDF) + sizeof(float) * (size_t)(3 * C + D::F);
    return sizeof(CT) * (size_t)(MF * D::LDX + C * CMax<D::LDX, D::LDC>::v + C * D::LDX + DH * D::LDC + 2 * C * D::LDF + D::F * D::LDC + C * D::LDC + DH * D::LDF) +
           sizeof(float) * (size_t)(3 * C + D::F);
}
template <typename CT, int DH, int MF, int C> static size_t dq_lds() {
    typedef FavorDims<CT, DH, MF, C, 1> D;
    return sizeof(CT) * (size_t)(MF * D::LDX + DH * D::LDM + C * D::LDX + C * CMax<D::LDX, D::LDM>::v + C * D::LDX + DH * D::LDC + C * D::LDX +
                                 C * D::LDF + D::F * D::LDC + C * D::LDC + D::F * D::LDX) +
           sizeof(float) * (size_t)(4 * C + D::F);
}
template <typename CT, int DH, int MF, int C> static size_t dkv_lds() {
    typedef FavorDims<CT, DH, MF, C, 2> D;
    return sizeof(CT) * (size_t)(MF * D::LDX + DH * D::LDM + 2 * C * CMax<D::LDX, D::LDC>::v + C * D::LDX + C * CMax<D::LDX, D::LDM>::v + DH * D::LDC +
                                 2 * C * D::LDF + D::F * D::LDC + D::F * D::LDX + DH * D::LDF) +
           sizeof(float) * (size_t)(4 * C + D::F);
}

#define EMO_MAX_LDS (160 * 1024)

// emo_favor_fs.hip: the bf16 / d_head 64 / 128-feature "slice" kernels (false: shape or mode not covered -> the generic kernels below)
int emo_favor_fs_try(int which, int stage, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const float* omega, bf16_t* out, int64_t ld_out,
                     float* den, float* sS, float* sz, const bf16_t* dout, bf16_t* dq, bf16_t* dk, bf16_t* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H,
                     float eps, const float* ws_S, const float* ws_z, int P, int64_t Ts, hipStream_t st);

// ---- segment-parallel scan geometry (see favor_fwd_kernel).  One workgroup per (b, h, segment); segments only when B*H alone cannot
// fill the 256 CUs.  Ts is a multiple of 64 (every chunk size divides it) and at least 2 chunks long.
#define EMO_FAVOR_SEG_ALIGN 64
#define EMO_FAVOR_MAX_SEGMENTS 16
static void favor_segments(int64_t B, int64_t T, int64_t H, int* P_out, int64_t* Ts_out) {
    int64_t want = 1;
    const char* e = getenv("EMO_FAVOR_SEGMENTS");
    if (e && atoi(e) > 0) want = atoi(e);
    else if (B * H < 256) want = (256 + B * H - 1) / (B * H);
    if (want > EMO_FAVOR_MAX_SEGMENTS) want = EMO_FAVOR_MAX_SEGMENTS;
    const int64_t max_p = T / (2 * EMO_FAVOR_SEG_ALIGN);
    if (want > max_p) want = max_p;
    if (want < 1) want = 1;
    int64_t Ts = (T + want - 1) / want;
    Ts = (Ts + EMO_FAVOR_SEG_ALIGN - 1) / EMO_FAVOR_SEG_ALIGN * EMO_FAVOR_SEG_ALIGN;
    if (Ts < EMO_FAVOR_SEG_ALIGN) Ts = EMO_FAVOR_SEG_ALIGN;
    const int64_t P = T > 0 ? (T + Ts - 1) / Ts : 1;
    *P_out = (int)(P < 1 ? 1 : P);
    *Ts_out = Ts;
}

extern "C" int64_t emo_favor_attn_workspace_bytes(int64_t B, int64_t T, int64_t H, int64_t dh, int64_t n_feat) {
    int P; int64_t Ts;
    favor_segments(B, T, H, &P, &Ts);
    if (P <= 1) return 0;
    return B * H * P * (n_feat * dh + n_feat) * (int64_t)sizeof(float);
}

template <typename CT, int DH, int MF, int CF, int CQ, int CK>
static int run_favor(int which, const void* q, const void* k, const void* v, int64_t ld, const float* omega, void* out, int64_t ld_out, float* den,
                     float* sS, float* sz, const void* dout, void* dq, void* dk, void* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H, float eps,
                     void* workspace, int64_t workspace_bytes, hipStream_t st, bool kstate_valid = false) {
    constexpr int F = 2 * MF;
    int P = 1; int64_t Ts = T > 0 ? T : 1;
    if (workspace) {
        favor_segments(B, T, H, &P, &Ts);
        const int64_t need = P > 1 ? B * H * P * (int64_t)(F * DH + F) * (int64_t)sizeof(float) : 0;
        EMO_CHECK(workspace_bytes >= need, "favor attention: workspace %lld B < %lld B (emo_favor_attn_workspace_bytes)", (long long)workspace_bytes,
                  (long long)need);
        EMO_CHECK(((uintptr_t)workspace & 15) == 0, "favor attention: workspace must be 16-B aligned");
    }
    if (P <= 1) { P = 1; Ts = T > 0 ? T : 1; }
    float* wsS = (float*)workspace;
    float* wsz = wsS ? wsS + B * H * P * (int64_t)F * DH : nullptr;
    dim3 grid((unsigned)(B * H * P));
    // bf16 / d_head 64 / 128 features: the main passes run on the slice kernels (emo_favor_fs.hip); with P > 1 they start every segment from the
    // increments that the generic state-only passes below leave in the workspace
    constexpr bool FS = sizeof(CT) == 2 && DH == 64 && MF == 64;
    auto fs_try = [&](int stage) -> bool {
        if constexpr (FS) {
            if (emo_favor_fs_try(which, stage, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, omega, (bf16_t*)out, ld_out, den, sS, sz, (const bf16_t*)dout,
                                 (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, ld_d, B, T, H, eps, wsS, wsz, P, Ts, st))
                return true;
        }
        const char* e = getenv("EMO_FAVOR_FS");
        if (e && atoi(e) == 2) emo_set_error("favor attention: EMO_FAVOR_FS=2 but the slice kernels do not cover this call");
        return false;
    };
    auto fs_required_failed = [&]() { const char* e = getenv("EMO_FAVOR_FS"); return e && atoi(e) == 2; };
    if (which == 0) {
        const size_t lds = fwd_lds<CT, DH, MF, CF>();
        EMO_CHECK(lds <= EMO_MAX_LDS, "favor fwd: LDS %zu too large", lds);
        auto kf = favor_fwd_kernel<CT, DH, MF, CF, false>;
        auto ks = favor_fwd_kernel<CT, DH, MF, CF, true>;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        if (P > 1)
            hipLaunchKernelGGL(ks, grid, dim3(FT), lds, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, omega, (CT*)out, ld_out, den, sS, sz, T, H, eps, wsS,
                               wsz, P, Ts);
        const char* ab = getenv("EMO_FAVOR_ABLATE");   // diagnostics only
        if (!fs_try(0)) {
            if (fs_required_failed()) return EMO_ERR_INVALID;
            hipLaunchKernelGGL(kf, grid, dim3(FT), lds, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, omega, (CT*)out, ld_out, den, sS, sz, T, H, eps, wsS, wsz,
                               P | (ab ? atoi(ab) << 8 : 0), Ts);
        }
    } else {
        const size_t l0 = fwd_lds<CT, DH, MF, CF>(), l1 = dq_lds<CT, DH, MF, CQ>(), l2 = dkv_lds<CT, DH, MF, CK>();
        EMO_CHECK(l0 <= EMO_MAX_LDS && l1 <= EMO_MAX_LDS && l2 <= EMO_MAX_LDS, "favor bwd: LDS %zu / %zu / %zu too large", l0, l1, l2);
        auto k0 = favor_fwd_kernel<CT, DH, MF, CF, true>;
        auto k1 = favor_bwd_dq_kernel<CT, DH, MF, CQ>;
        auto k2 = favor_bwd_dkv_kernel<CT, DH, MF, CK, false>;
        auto k2s = favor_bwd_dkv_kernel<CT, DH, MF, CK, true>;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0);
            (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
            (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
            (void)hipFuncSetAttribute((const void*)k2s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
            attr = true;
        }
        if (den == nullptr) {
            // emo_favor_attn_bwd_dn: `dout` is dN = dout / den already — only the slice kernels' single-segment instances take that form
            EMO_CHECK(FS && P == 1, "emo_favor_attn_bwd_dn: needs bf16, d_head 64, 128 features and a single-segment scan (B * H >= 256): emo_favor_attn_bwd_dn_supported()");
            if (!fs_try(1) || !fs_try(2)) {
                emo_set_error("emo_favor_attn_bwd_dn: the slice kernels refused this call (T %% 32, 16-B alignment, EMO_FAVOR_FS / EMO_FAVOR_FS_BWD = 0)");
                return EMO_ERR_UNSUPPORTED;
            }
            EMO_LAUNCH_CHECK();
            return EMO_OK;
        }
        // P > 1: the K-state increments are recomputed (state-only forward pass), then the workspace is reused for the R-state increments
        // (kstate_valid: the caller kept the forward's workspace — the same increments — for this call: emo_favor_attn_bwd_kstate)
        if (P > 1 && !kstate_valid)
            hipLaunchKernelGGL(k0, grid, dim3(FT), l0, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, omega, (CT*)nullptr, ld_out, (float*)nullptr,
                               (float*)nullptr, (float*)nullptr, T, H, eps, wsS, wsz, P, Ts);
        if (!fs_try(1)) {
            if (fs_required_failed()) return EMO_ERR_INVALID;
            hipLaunchKernelGGL(k1, grid, dim3(FT), l1, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, omega, (const CT*)out, (const CT*)dout, ld_out, den,
                               (CT*)dq, ld_d, T, H, (const float*)wsS, (const float*)wsz, P | (getenv("EMO_FAVOR_ABLATE_DQ") ? atoi(getenv("EMO_FAVOR_ABLATE_DQ")) << 8 : 0), Ts);
        }
        if (P > 1)
            hipLaunchKernelGGL(k2s, grid, dim3(FT), l2, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, omega, (const CT*)out, (const CT*)dout, ld_out, den,
                               (CT*)dk, (CT*)dv, ld_d, T, H, wsS, wsz, P, Ts);
        if (!fs_try(2)) {
            if (fs_required_failed()) return EMO_ERR_INVALID;
            hipLaunchKernelGGL(k2, grid, dim3(FT), l2, st, (const CT*)q, (const CT*)k, (const CT*)v, ld, omega, (const CT*)out, (const CT*)dout, ld_out, den,
                               (CT*)dk, (CT*)dv, ld_d, T, H, wsS, wsz, P | (getenv("EMO_FAVOR_ABLATE_DKV") ? atoi(getenv("EMO_FAVOR_ABLATE_DKV")) << 8 : 0), Ts);
        }
    }
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

static int dispatch_favor(int which, int dtype, int64_t dh, int64_t mf, const void* q, const void* k, const void* v, int64_t ld, const float* omega,
                          void* out, int64_t ld_out, float* den, float* sS, float* sz, const void* dout, void* dq, void* dk, void* dv, int64_t ld_d,
                          int64_t B, int64_t T, int64_t H, float eps, void* ws, int64_t ws_bytes, hipStream_t st, bool kstate_valid = false) {
#define FAVOR_CASE(DHv, MFv, CFb, CQb, CKb, CFf, CQf, CKf)                                                                                   \
    if (dh == DHv && mf == MFv) {                                                                                                            \
        if (dtype == EMO_BF16)                                                                                                               \
            return run_favor<bf16_t, DHv, MFv, CFb, CQb, CKb>(which, q, k, v, ld, omega, out, ld_out, den, sS, sz, dout, dq, dk, dv, ld_d, B, T, H, eps, ws, \
                                                              ws_bytes, st, kstate_valid);                                                   \
        return run_favor<float, DHv, MFv, CFf, CQf, CKf>(which, q, k, v, ld, omega, out, ld_out, den, sS, sz, dout, dq, dk, dv, ld_d, B, T, H, eps, ws,      \
                                                         ws_bytes, st, kstate_valid);                                                        \
    }
    FAVOR_CASE(64, 64, FAVOR_CFB, FAVOR_CQB, FAVOR_CKB, 32, 32, 16)
    FAVOR_CASE(32, 64, 64, 64, 64, 32, 32, 32)
    FAVOR_CASE(32, 32, 64, 64, 64, 32, 32, 32)
    FAVOR_CASE(16, 16, 64, 64, 64, 32, 32, 32)
    FAVOR_CASE(16, 32, 64, 64, 64, 32, 32, 32)
#undef FAVOR_CASE
    emo_set_error("favor attention: unsupported (d_head=%lld, n_feat=%lld); built: (64,128) (32,128) (32,64) (16,32) (16,64)", (long long)dh,
                  (long long)(2 * mf));
    return EMO_ERR_UNSUPPORTED;
}

static int favor_check(const void* q, const void* k, const void* v, int64_t ld, int64_t ld_out, int dtype, int64_t dh, int64_t n_feat) {
    EMO_CHECK(q && k && v, "favor attention: null pointer");
    EMO_CHECK(dtype == EMO_F32 || dtype == EMO_BF16, "favor attention: bad dtype");
    const int64_t ve = dtype == EMO_BF16 ? 8 : 4;
    EMO_CHECK(ld % ve == 0 && ld_out % 4 == 0 && dh % ve == 0, "favor attention: ld/dh must keep rows 16-B aligned");
    EMO_CHECK(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0, "favor attention: q/k/v must be 16-B aligned");
    EMO_CHECK(n_feat % 2 == 0, "favor attention: n_feat must be even");
    return EMO_OK;
}

extern "C" int emo_favor_attn_fwd(const void* q, const void* k, const void* v, int64_t ld, const float* omega, void* out, int64_t ld_out, float* den,
                                  float* state_S, float* state_z, int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, int64_t n_feat, float eps,
                                  void* workspace, int64_t workspace_bytes, emo_stream_t stream) {
    int rc = favor_check(q, k, v, ld, ld_out, dtype, dh, n_feat);
    if (rc) return rc;
    EMO_CHECK(omega && out && den, "emo_favor_attn_fwd: null pointer");
    EMO_CHECK(((uintptr_t)out & 15) == 0, "emo_favor_attn_fwd: out must be 16-B aligned");
    EMO_CHECK(!(state_S && !state_z), "emo_favor_attn_fwd: state_S without state_z");
    return dispatch_favor(0, dtype, dh, n_feat / 2, q, k, v, ld, omega, out, ld_out, den, state_S, state_z, nullptr, nullptr, nullptr, nullptr, 0, B, T, H,
                          eps, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int emo_favor_attn_bwd_kstate(const void* q, const void* k, const void* v, int64_t ld, const float* omega, const void* out, const void* dout,
                                         int64_t ld_out, const float* den, void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H,
                                         int64_t dh, int64_t n_feat, float eps, void* workspace, int64_t workspace_bytes, int kstate_valid, emo_stream_t stream) {
    int rc = favor_check(q, k, v, ld, ld_out, dtype, dh, n_feat);
    if (rc) return rc;
    EMO_CHECK(omega && out && dout && den && dq && dk && dv, "emo_favor_attn_bwd: null pointer");
    EMO_CHECK(ld_d % 4 == 0, "emo_favor_attn_bwd: ld_d must be a multiple of 4");
    EMO_CHECK((((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)out | (uintptr_t)dout) & 15) == 0, "emo_favor_attn_bwd: pointers must be 16-B aligned");
    return dispatch_favor(1, dtype, dh, n_feat / 2, q, k, v, ld, omega, (void*)out, ld_out, (float*)den, nullptr, nullptr, dout, dq, dk, dv, ld_d, B, T, H,
                          eps, workspace, workspace_bytes, (hipStream_t)stream, kstate_valid != 0 && workspace != nullptr);
}
// 1 when emo_favor_attn_bwd_dn serves this problem (the single-segment slice kernels), else 0
extern "C" int emo_favor_attn_bwd_dn_supported(int dtype, int64_t B, int64_t T, int64_t H, int64_t dh, int64_t n_feat) {
    if (dtype != EMO_BF16 || dh != 64 || n_feat != 128 || T < 32 || (T % 32) != 0 || B * H <= 0) return 0;
    const char* e = getenv("EMO_FAVOR_FS");
    const char* e2 = getenv("EMO_FAVOR_FS_BWD");
    if ((e && atoi(e) == 0) || (e2 && atoi(e2) == 0)) return 0;
    int P; int64_t Ts;
    favor_segments(B, T, H, &P, &Ts);
    return P <= 1 ? 1 : 0;
}
extern "C" int emo_favor_attn_bwd_dn(const void* q, const void* k, const void* v, int64_t ld, const float* omega, const void* out, const void* dn,
                                     int64_t ld_out, void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H,
                                     int64_t dh, int64_t n_feat, float eps, emo_stream_t stream) {
    int rc = favor_check(q, k, v, ld, ld_out, dtype, dh, n_feat);
    if (rc) return rc;
    EMO_CHECK(omega && out && dn && dq && dk && dv, "emo_favor_attn_bwd_dn: null pointer");
    EMO_CHECK(emo_favor_attn_bwd_dn_supported(dtype, B, T, H, dh, n_feat), "emo_favor_attn_bwd_dn: problem not in the supported class (emo_favor_attn_bwd_dn_supported)");
    EMO_CHECK(ld_d % 4 == 0, "emo_favor_attn_bwd_dn: ld_d must be a multiple of 4");
    EMO_CHECK((((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)out | (uintptr_t)dn) & 15) == 0, "emo_favor_attn_bwd_dn: pointers must be 16-B aligned");
    return dispatch_favor(1, dtype, dh, n_feat / 2, q, k, v, ld, omega, (void*)out, ld_out, (float*)nullptr, nullptr, nullptr, dn, dq, dk, dv, ld_d, B, T, H,
                          eps, nullptr, 0, (hipStream_t)stream, false);
}
extern "C" int emo_favor_attn_bwd(const void* q, const void* k, const void* v, int64_t ld, const float* omega, const void* out, const void* dout,
                                  int64_t ld_out, const float* den, void* dq, void* dk, void* dv, int64_t ld_d, int dtype, int64_t B, int64_t T, int64_t H,
                                  int64_t dh, int64_t n_feat, float eps, void* workspace, int64_t workspace_bytes, emo_stream_t stream) {
    return emo_favor_attn_bwd_kstate(q, k, v, ld, omega, out, dout, ld_out, den, dq, dk, dv, ld_d, dtype, B, T, H, dh, n_feat, eps, workspace, workspace_bytes, 0, stream);
}

extern "C" int emo_favor_decode_step(const void* q, const void* k, const void* v, int64_t ld, const float* omega, float* state_S, float* state_z, void* out,
                                     int64_t ld_out, int dtype, int64_t n_streams, int64_t H, int64_t dh, int64_t n_feat, float eps, emo_stream_t stream) {
    EMO_CHECK(q && k && v && omega && state_S && state_z && out, "emo_favor_decode_step: null pointer");
    EMO_CHECK(dh <= 64 && dh >= 16 && 256 % dh == 0 && n_feat <= 128 && n_feat % 2 == 0, "emo_favor_decode_step: needs 16<=d_head<=64 dividing 256, n_feat<=128");
    dim3 grid((unsigned)(n_streams * H));
    hipStream_t st = (hipStream_t)stream;
#define DECODE_LAUNCH(CTv, DHv, MFv)                                                                                                            \
    hipLaunchKernelGGL((favor_decode_fast_kernel<CTv, DHv, MFv>), grid, dim3(256), 0, st, (const CTv*)q, (const CTv*)k, (const CTv*)v, ld, omega,             \
                       state_S, state_z, (CTv*)out, ld_out, H, eps)
#define DECODE_CASE(DHv, MFv)                                                                                                                   \
    if (dh == DHv && n_feat == 2 * MFv) {                                                                                                       \
        if (dtype == EMO_F32) DECODE_LAUNCH(float, DHv, MFv); else DECODE_LAUNCH(bf16_t, DHv, MFv);                                             \
        EMO_LAUNCH_CHECK();                                                                                                                     \
        return EMO_OK;                                                                                                                          \
    }
    if (getenv("EMO_FAVOR_DECODE_GENERIC") == nullptr && (((uintptr_t)state_S) & 15) == 0) {
        DECODE_CASE(64, 64)
        DECODE_CASE(32, 64)
        DECODE_CASE(32, 32)
        DECODE_CASE(16, 16)
        DECODE_CASE(16, 32)
    }
#undef DECODE_CASE
#undef DECODE_LAUNCH
    if (dtype == EMO_F32)
        hipLaunchKernelGGL(favor_decode_kernel<float>, grid, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, ld, omega, state_S, state_z,
                           (float*)out, ld_out, H, (int)dh, (int)(n_feat / 2), eps);
    else
        hipLaunchKernelGGL(favor_decode_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, omega, state_S,
                           state_z, (bf16_t*)out, ld_out, H, (int)dh, (int)(n_feat / 2), eps);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
