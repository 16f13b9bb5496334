// K2/K7/K8 — dense GEMM on the gfx950 matrix cores with fused epilogues.
//
//   bf16 path : 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 bf16
//               tiles, fp32 accumulate.  Operands are staged global -> VGPR -> LDS (double
//               buffered, one barrier per K step).  An operand whose reduction index is
//               contiguous in memory ("K-contig": activations [M,K], nn.Linear weights [N,K])
//               lands in a 16-B-chunk XOR-swizzled [row][k] image and is read with ds_read_b128;
//               an operand whose OUTPUT index is contiguous ("MN-contig": Conv1D weights [K,N],
//               both operands of every wgrad) keeps its [k][row] image (32-B granule swizzle)
//               and is read with the gfx950 transpose read ds_read_b64_tr_b16.
//               MFMA is issued with swapped operands so that each lane owns 4 CONSECUTIVE
//               output columns -> 8/16-byte epilogue accesses.
//   f32 path  : parity mode.  64x64x16 tile on v_mfma_f32_16x16x4_f32 — bit-for-bit an fmaf
//               chain in k order (exact fp32), arbitrary strides.
//   epilogue  : +bias -> aux_out -> act -> *mul(aux) -> dropout -> +residual ; fp32 atomics for
//               split-K (wgrad: the reduction runs over B*T tokens).
//   grid      : XCD-aware tile order — the n-tiles of one A row panel run back-to-back on the
//               same XCD so the panel is fetched from HBM once and re-read from that XCD's L2.
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
    int atomic;
};

// ------------------------------------------------------------------------------------------------
// epilogue for 4 consecutive columns n..n+3 of row m
template <typename OutT>
__device__ __forceinline__ void epi_store4(const EpiParams& ep, OutT* __restrict__ C, int64_t m, int64_t n,
                                           f32x4 acc, int64_t N) {
    const int64_t off = m * ep.ldc + n;
    const bool vec = (n + 3 < N) && ((ep.ldc & 3) == 0);
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    const int cnt = vec ? 4 : (int)((N - n) < 4 ? (N - n) : 4);
    if (ep.bias) {
        if (vec) {
            f32x4 b = *(const f32x4*)(ep.bias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += b[i];
        } else {
            for (int i = 0; i < cnt; ++i) v[i] += ep.bias[n + i];
        }
    }
    if (ep.aux_out) {
        OutT* ao = (OutT*)ep.aux_out + off;
        for (int i = 0; i < cnt; ++i) ao[i] = from_f32<OutT>(v[i]);
    }
    if (ep.act == EMO_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (ep.act == EMO_ACT_GELU_NEW) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_new_f(v[i]);
    }
    if (ep.mul_mode != EMO_MUL_NONE) {
        const OutT* ma = (const OutT*)ep.mul_aux + off;
        for (int i = 0; i < cnt; ++i) {
            float a = to_f32<OutT>(ma[i]);
            v[i] *= (ep.mul_mode == EMO_MUL_NONZERO) ? (a != 0.f ? ep.mul_scale : 0.f) : dgelu_new_f(a);
        }
    }
    if (ep.drop.thr16) {
        for (int i = 0; i < cnt; ++i) v[i] *= drop_mult(ep.drop, (uint64_t)(m * N + n + i));
    }
    if (ep.residual) {
        const OutT* r = (const OutT*)ep.residual + off;
        if (vec && sizeof(OutT) == 2) {
            bf16x4 rv = *(const bf16x4*)r;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
        } else {
            for (int i = 0; i < cnt; ++i) v[i] += to_f32<OutT>(r[i]);
        }
    }
    OutT* c = C + off;
    if constexpr (sizeof(OutT) == 4) {
        if (ep.atomic) {
            for (int i = 0; i < cnt; ++i) atomicAdd((float*)c + i, v[i]);
        } else if (ep.accumulate) {
            for (int i = 0; i < cnt; ++i) ((float*)c)[i] += v[i];
        } else if (vec) {
            *(f32x4*)c = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
            for (int i = 0; i < cnt; ++i) ((float*)c)[i] = v[i];
        }
    } else {
        if (vec) {
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *(bf16x4*)c = o;
        } else {
            for (int i = 0; i < cnt; ++i) c[i] = from_f32<OutT>(v[i]);
        }
    }
}

// XCD-aware (m_tile, n_tile) from the linear block id (dispatcher places block b on XCD b%8).
__device__ __forceinline__ void tile_coords(int64_t tiles_m, int64_t tiles_n, int64_t& tm, int64_t& tn) {
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, local = bid >> 3;
    tn = local % tiles_n;
    tm = (local / tiles_n) * 8 + xcd;
    (void)tiles_m;
}

// ================================================================================================
// f32 parity kernel: exact fp32 (v_mfma_f32_16x16x4_f32), arbitrary strides.
// A(m,k) = A[m*sam + k*sak] ; B(k,n) = B[n*sbn + k*sbk]
template <typename OutT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbn, int64_t sbk,
                                                       OutT* __restrict__ C, int64_t M, int64_t N, int64_t K,
                                                       int64_t k_per_split, EpiParams ep) {
    constexpr int BM = 64, BN = 64, BK = 16, LD = BK + 1;
    __shared__ float As[BM * LD];
    __shared__ float Bs[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    int64_t tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * BM, n0 = tn * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, 32x32 each
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // loader mapping: thread -> (row r = tid/4 .. , 4 k's) ; choose the mapping that walks the
    // contiguous dimension with consecutive threads
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + i * 256;  // 0..1023
            int r, kk;
            if (sak == 1) { r = e >> 4; kk = e & 15; } else { r = e & 63; kk = e >> 6; }
            int64_t gm = m0 + r, gk = k0 + kk;
            As[r * LD + kk] = (gm < M && gk < kend) ? A[gm * sam + gk * sak] : 0.f;
            if (sbk == 1) { r = e >> 4; kk = e & 15; } else { r = e & 63; kk = e >> 6; }
            int64_t gn = n0 + r;
            gk = k0 + kk;
            Bs[r * LD + kk] = (gn < N && gk < kend) ? B[gn * sbn + gk * sbk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + (lane & 15)) * LD + ks + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(wn * 32 + j * 16 + (lane & 15)) * LD + ks + (lane >> 4)];
            // swapped operands: D'[n][m] so that a lane owns 4 consecutive n
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int64_t m = m0 + wm * 32 + i * 16 + (lane & 15);
            int64_t n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
            if (m < M && n < N) epi_store4<OutT>(ep, C, m, n, acc[i][j], N);
        }
}

// ================================================================================================
// bf16 MFMA kernel
constexpr int GB_M = 128, GB_N = 128, GB_K = 64;

__device__ __forceinline__ int swz_k(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ---- global -> registers: 4 x 16 B per thread per operand
// K-contig operand: tile [128 rows][64 k]; chunk c = 8 k's.  idx = tid + 256*i : row = idx>>3, chunk = idx&7
template <bool KC>
__device__ __forceinline__ void gload_tile(const bf16_t* __restrict__ P, int64_t ld, int64_t row0, int64_t nrows,
                                           int64_t k0, int64_t kend, int tid, bf16x8 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (KC) {
            const int row = idx >> 3, ch = idx & 7;
            const int64_t gr = row0 + row, gk = k0 + ch * 8;
            if (gr < nrows && gk < kend) {
                v = *(const bf16x8*)(P + gr * ld + gk);
                if (gk + 8 > kend) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (gk + e >= kend) v[e] = (bf16_t)0.f;
                }
            }
        } else {
            // MN-contig: tile [64 k][128 rows]; idx -> k = idx>>4, rchunk = idx&15 (8 rows each)
            const int k = idx >> 4, rc = idx & 15;
            const int64_t gk = k0 + k, gr = row0 + rc * 8;
            if (gk < kend && gr < nrows) v = *(const bf16x8*)(P + gk * ld + gr);
        }
        r[i] = v;
    }
}

template <bool KC>
__device__ __forceinline__ void lstore_tile(char* lds, int tid, const bf16x8 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        if constexpr (KC) {
            const int row = idx >> 3, ch = idx & 7;
            *(bf16x8*)(lds + row * 128 + ((ch ^ (row & 7)) << 4)) = r[i];
        } else {
            const int k = idx >> 4, rc = idx & 15;  // granule g = rc>>1 (16 rows), half = rc&1
            *(bf16x8*)(lds + k * 256 + (((rc >> 1) ^ swz_k(k)) << 5) + ((rc & 1) << 4)) = r[i];
        }
    }
}

// fragment (8 bf16 along k for this lane's row) for rows rbase..rbase+15, k-step ks (32 k's)
template <bool KC, bool SAFE>
__device__ __forceinline__ bf16x8 lfrag(const char* lds, int rbase, int ks, int lane) {
    if constexpr (KC) {
        const int row = rbase + (lane & 15), ch = ks * 4 + (lane >> 4);
        return *(const bf16x8*)(lds + row * 128 + ((ch ^ (row & 7)) << 4));
    } else if constexpr (SAFE) {
        bf16x8 v;
        const int r = rbase + (lane & 15);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 32 + (lane >> 4) * 8 + e;
            v[e] = *(const bf16_t*)(lds + k * 256 + (((r >> 4) ^ swz_k(k)) << 5) + ((r & 15) << 1));
        }
        return v;
    } else {
        // ds_read_b64_tr_b16: per 16-lane group a [4 k][16 rows] block; lane i supplies the address of
        // (k0 + i/4, rows (i%4)*4..+3) and receives column i (4 k's of row rbase+i).
        const int i = lane & 15, g = rbase >> 4;
        bf16x8 v;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = ks * 32 + (lane >> 4) * 8 + h * 4 + (i >> 2);
            const char* p = lds + k * 256 + ((g ^ swz_k(k)) << 5) + ((i & 3) << 3);
            short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
            bf16x4 tb = __builtin_bit_cast(bf16x4, t);
            v[h * 4 + 0] = tb[0]; v[h * 4 + 1] = tb[1]; v[h * 4 + 2] = tb[2]; v[h * 4 + 3] = tb[3];
        }
        return v;
    }
}

template <bool A_KC, bool B_KC, bool SAFE, typename OutT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                        const bf16_t* __restrict__ B, int64_t ldb,
                                                        OutT* __restrict__ C, int64_t M, int64_t N, int64_t K,
                                                        int64_t k_per_split, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (16 KB A + 16 KB B)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tiles_n = (N + GB_N - 1) / GB_N, tiles_m = (M + GB_M - 1) / GB_M;
    int64_t tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * GB_M, n0 = tn * GB_N;
    const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 ra[4], rb[4];
    gload_tile<A_KC>(A, lda, m0, M, kbeg, kend, tid, ra);
    gload_tile<B_KC>(B, ldb, n0, N, kbeg, kend, tid, rb);
    lstore_tile<A_KC>(smem, tid, ra);
    lstore_tile<B_KC>(smem + 16384, tid, rb);
    __syncthreads();
    int cur = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += GB_K) {
        const bool more = (k0 + GB_K) < kend;
        if (more) {
            gload_tile<A_KC>(A, lda, m0, M, k0 + GB_K, kend, tid, ra);
            gload_tile<B_KC>(B, ldb, n0, N, k0 + GB_K, kend, tid, rb);
        }
        const char* la = smem + cur * 32768;
        const char* lb = la + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = lfrag<A_KC, SAFE>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = lfrag<B_KC, SAFE>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* na = smem + (cur ^ 1) * 32768;
            lstore_tile<A_KC>(na, tid, ra);
            lstore_tile<B_KC>(na + 16384, tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
            int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (m < M && n < N) epi_store4<OutT>(ep, C, m, n, acc[i][j], N);
        }
}

// ================================================================================================
static bool g_safe_tr = false;
static bool g_safe_tr_init = false;
static bool use_safe_tr() {
    if (!g_safe_tr_init) {
        const char* e = getenv("EMO_GEMM_SAFE_TR");
        g_safe_tr = e && e[0] == '1';
        g_safe_tr_init = true;
    }
    return g_safe_tr;
}

template <bool A_KC, bool B_KC, bool SAFE, typename OutT>
static void launch_bf16(dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C,
                        int64_t M, int64_t N, int64_t K, int64_t kps, const EpiParams& ep) {
    auto kfn = gemm_bf16_kernel<A_KC, B_KC, SAFE, OutT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), 65536, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
}

template <bool SAFE, typename OutT>
static void dispatch_bf16(bool akc, bool bkc, dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B,
                          int64_t ldb, void* C, int64_t M, int64_t N, int64_t K, int64_t kps, const EpiParams& ep) {
    if (akc && bkc) launch_bf16<true, true, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (akc && !bkc) launch_bf16<true, false, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (!akc && bkc) launch_bf16<false, true, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else launch_bf16<false, false, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
}

extern "C" int emo_gemm(const void* A, int a_trans, int64_t lda, const void* B, int b_trans, int64_t ldb, void* C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, int dtype_in, int dtype_out, int accumulate,
                        const emo_epilogue_t* e, emo_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    EMO_CHECK(A && B && C, "emo_gemm: null pointer");
    EMO_CHECK(M > 0 && N > 0 && K > 0, "emo_gemm: bad dims M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    EMO_CHECK(dtype_in == EMO_F32 || dtype_in == EMO_BF16, "emo_gemm: bad dtype_in");
    EMO_CHECK(dtype_out == EMO_F32 || dtype_out == EMO_BF16, "emo_gemm: bad dtype_out");
    EMO_CHECK(!(accumulate && dtype_out != EMO_F32), "emo_gemm: accumulate needs fp32 output");
    EMO_CHECK(!(dtype_in == EMO_F32 && dtype_out != EMO_F32), "emo_gemm: fp32 inputs need fp32 output");
    EpiParams ep;
    memset(&ep, 0, sizeof(ep));
    ep.ldc = ldc;
    ep.accumulate = accumulate;
    ep.drop = make_drop(0.f, 0, 0);
    bool has_epi = false;
    if (e) {
        ep.bias = e->bias; ep.act = e->act; ep.aux_out = e->aux_out; ep.mul_aux = e->mul_aux;
        ep.mul_mode = e->mul_aux ? e->mul_mode : EMO_MUL_NONE; ep.mul_scale = e->mul_scale;
        ep.drop = make_drop(e->p_drop, e->seed, e->offset); ep.residual = e->residual;
        has_epi = e->bias || e->act || e->aux_out || ep.mul_mode || ep.drop.thr16 || e->residual;
    }
    const bool big = dtype_in == EMO_BF16;
    const int64_t BMt = big ? GB_M : 64, BNt = big ? GB_N : 64, BKt = big ? GB_K : 16;
    const int64_t tiles_m = cdiv64(M, BMt), tiles_n = cdiv64(N, BNt);
    const int64_t tiles_m8 = cdiv64(tiles_m, 8) * 8;
    // split-K: only for plain fp32 outputs (wgrad) when the tile grid cannot fill the chip
    int64_t splits = 1;
    if (dtype_out == EMO_F32 && !has_epi && tiles_m * tiles_n < 256 && K >= 8 * BKt) {
        splits = cdiv64(512, tiles_m * tiles_n);
        const int64_t max_splits = K / (4 * BKt);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    int64_t kps = cdiv64(cdiv64(K, splits), BKt) * BKt;
    splits = cdiv64(K, kps);
    if (splits > 1) {
        EMO_CHECK(ldc == N, "emo_gemm: split-K needs contiguous C");
        if (!accumulate) {
            hipError_t me = hipMemsetAsync(C, 0, (size_t)(M * N) * sizeof(float), st);
            EMO_CHECK(me == hipSuccess, "emo_gemm: memset failed");
        }
        ep.atomic = 1;
    }
    dim3 grid((unsigned)(tiles_m8 * tiles_n), 1, (unsigned)splits);
    if (dtype_in == EMO_F32) {
        const float* a = (const float*)A;
        const float* b = (const float*)B;
        const int64_t sam = a_trans ? 1 : lda, sak = a_trans ? lda : 1;
        const int64_t sbn = b_trans ? 1 : ldb, sbk = b_trans ? ldb : 1;
        hipLaunchKernelGGL(gemm_f32_kernel<float>, grid, dim3(256), 0, st, a, sam, sak, b, sbn, sbk, (float*)C, M, N, K,
                           kps, ep);
    } else {
        EMO_CHECK((lda & 7) == 0 && (ldb & 7) == 0, "emo_gemm(bf16): lda/ldb must be multiples of 8 (16-B rows)");
        EMO_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "emo_gemm(bf16): A/B must be 16-B aligned");
        const bf16_t* a = (const bf16_t*)A;
        const bf16_t* b = (const bf16_t*)B;
        const bool akc = !a_trans, bkc = !b_trans;
        const bool safe = use_safe_tr();
        if (dtype_out == EMO_F32) {
            if (safe) dispatch_bf16<true, float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_bf16<false, float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else {
            if (safe) dispatch_bf16<true, bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_bf16<false, bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        }
    }
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): grid (col blocks of 256 cols? no: 64 cols) x row splits, atomics.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t M, int64_t N, int64_t ld,
                                                     float* __restrict__ out, int64_t rows_per_block) {
    // block: 64 columns x 4 row-lanes ; thread (c = tid&63, r = tid>>6)
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    const int64_t mbeg = (int64_t)blockIdx.y * rows_per_block;
    int64_t mend = mbeg + rows_per_block;
    if (mend > M) mend = M;
    float s = 0.f;
    if (n < N)
        for (int64_t m = mbeg + r; m < mend; m += 4) s += to_f32<T>(X[m * ld + n]);
    part[r][c] = s;
    __syncthreads();
    if (r == 0 && n < N) atomicAdd(out + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

extern "C" int emo_colsum(const void* X, int dtype, int64_t M, int64_t N, int64_t ld, float* out, int accumulate,
                          emo_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    EMO_CHECK(X && out && M > 0 && N > 0, "emo_colsum: bad args");
    if (!accumulate) {
        hipError_t me = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), st);
        EMO_CHECK(me == hipSuccess, "emo_colsum: memset failed");
    }
    const int64_t cb = cdiv64(N, 64);
    int64_t rsplit = cdiv64(2048, cb);
    if (rsplit > cdiv64(M, 64)) rsplit = cdiv64(M, 64);
    if (rsplit < 1) rsplit = 1;
    const int64_t rpb = cdiv64(M, rsplit);
    dim3 grid((unsigned)cb, (unsigned)cdiv64(M, rpb));
    if (dtype == EMO_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)X, M, N, ld, out, rpb);
    else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)X, M, N, ld, out, rpb);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
