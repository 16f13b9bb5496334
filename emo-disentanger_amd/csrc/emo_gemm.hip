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
#include "emo_gemm_epi.h"

// acc[i][j]: lane owns row (wm*64 + i*16 + (lane&15)), columns (wn*64 + j*16 + (lane>>4)*4 .. +3) of the 128x128 tile
template <typename OutT, int PASSES = 1>
__device__ __forceinline__ void epilogue_tile128(const EpiParams& ep, OutT* __restrict__ C, int64_t m0, int64_t n0, int64_t M, int64_t N,
                                                 const f32x4 (&acc)[4][4], char* lds, int tid, int wm, int wn, int lane) {
    if (ep.atomic) {
        // split-K partial sums: fp32 atomics straight from the fragment layout (row-contiguous atomics measured 20-40 % slower:
        // 16 lanes of one instruction then hit the same 128-B lines)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
                const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                if (m < M && n < N) epi_store4<OutT>(ep, C, m, n, acc[i][j], N);
            }
        return;
    }
    // PASSES = 2: the tile goes through LDS in two 64-row halves (32 KB) so that kernels with a 48 KB operand ring
    // (3 blocks per CU) can use the same epilogue.
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        constexpr int ROWS = 128 / PASSES;
        __syncthreads();                  // every wave is done reading operand tiles (or the previous half) from this LDS
        if (PASSES == 1 || wm == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (PASSES == 1 ? wm * 64 : 0) + i * 16 + (lane & 15);
                    const int chunk = (wn * 64 + j * 16 + (lane >> 4) * 4) >> 2;          // 16-B chunk index 0..31
                    *(f32x4*)(lds + row * 512 + ((chunk ^ (row & 7)) << 4)) = acc[i][j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8 / PASSES; ++it) {
            const int idx = tid + 256 * it;   // row = idx >> 4, 8-column group = idx & 15
            const int row = idx >> 4, grp = idx & 15;
            const int64_t m = m0 + pass * ROWS + row, n = n0 + grp * 8;
            if (m < M && n < N) {
                const f32x4 lo = *(const f32x4*)(lds + row * 512 + (((2 * grp) ^ (row & 7)) << 4));
                const f32x4 hi = *(const f32x4*)(lds + row * 512 + (((2 * grp + 1) ^ (row & 7)) << 4));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                epi_row8<OutT>(ep, C, m, n, v, N);
            }
        }
    }
}

// out-of-line copy for kernels with many accumulator tiles (keeps the unrolled epilogue small enough that the
// accumulator array stays in registers instead of scratch)
template <typename OutT>
__device__ __noinline__ void epi_store4_call(const EpiParams& ep, OutT* __restrict__ C, int64_t m, int64_t n, f32x4 acc, int64_t N) {
    epi_store4<OutT>(ep, C, m, n, acc, N);
}

// XCD-aware (m_tile, n_tile) from the linear block id (dispatcher places block b on XCD b%8).
__device__ __forceinline__ void tile_coords(int64_t tiles_m, int64_t tiles_n, int64_t& tm, int64_t& tn) {
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, local = bid >> 3;
    tn = local % tiles_n;
    tm = (local / tiles_n) * 8 + xcd;
    (void)tiles_m;
}
// Split-K (wgrad) placement: ALL output tiles of one K-split read the same token range of dY / X, so they are put on
// ONE XCD (split = xcd + 8*round) and run back-to-back there: the range is fetched from HBM once into that XCD's L2
// instead of once per XCD (measured: wgrad of the 512x512 projection was HBM-bound at 253 TFLOP/s with x-fastest order).
// grid.x = tiles_m*tiles_n * roundup(splits, 8), grid.z = 1.
__device__ __forceinline__ void splitk_coords(int64_t tiles_m, int64_t tiles_n, int g, int64_t& tm, int64_t& tn, int64_t& split) {
    const int64_t bid = blockIdx.x, ntile = tiles_m * tiles_n;
    const int64_t xcd = bid & 7, q = bid >> 3;
    int64_t t;
    if (g <= 1) {                     // >= 8 splits: one split per XCD per round
        split = xcd + 8 * (q / ntile);
        t = q % ntile;
    } else {                          // 8/g splits (g = 2, 4): a split owns g XCDs, its tiles alternate between them
        split = xcd / g;
        t = q * g + (xcd % g);
        if (t >= ntile) { tm = tiles_m; tn = 0; return; }   // caller returns on tm >= tiles_m
    }
    tm = t / tiles_n;
    tn = t % tiles_n;
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
    int64_t tm, tn, split = 0;
    if (ep.atomic) {
        splitk_coords(tiles_m, tiles_n, ep.atomic, tm, tn, split);
        if (ep.ws_stride) { C += split * ep.ws_stride; ep.atomic = 0; }
    }
    else tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * BM, n0 = tn * BN;
    const int64_t kbeg = split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;                 // surplus split of the rounded-up grid: no K range, and no workspace slice either
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
// chunk swizzle of the 64-B-row (BK=32) K-contiguous image: f(row>>2) = {0,3,2,1}.  ds_read_b128 is serviced in the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; with this map the 16 lanes of every group hit 16 distinct 16-B slots
// of the 256-B bank row (the plain (row>>2)&3 map measured 2-way: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).
__device__ __forceinline__ int swz32(int row) { return (0 - (row >> 2)) & 3; }

// ---- global -> registers: 4 x 16 B per thread per operand
// K-contig operand: tile [128 rows][64 k]; chunk c = 8 k's.  idx = tid + 256*i : row = idx>>3, chunk = idx&7
template <bool KC, int BKT = GB_K>
__device__ __forceinline__ void gload_tile(const bf16_t* __restrict__ P, int64_t ld, int64_t row0, int64_t nrows,
                                           int64_t k0, int64_t kend, int tid, bf16x8 (&r)[BKT / 16]) {
    static_assert(BKT == 64 || !KC, "the 32-deep register-staged tile is only built for MN-contiguous operands");
#pragma unroll
    for (int i = 0; i < BKT / 16; ++i) {
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

template <bool KC, int BKT = GB_K>
__device__ __forceinline__ void lstore_tile(char* lds, int tid, const bf16x8 (&r)[BKT / 16]) {
#pragma unroll
    for (int i = 0; i < BKT / 16; ++i) {
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

// RS: 0 = plain, 1 = also a_rowsum, 2 = also b_rowsum (separate instances so that the plain wgrad carries neither the extra accumulators nor
// the branches; the register cap keeps two blocks per CU)
template <bool A_KC, bool B_KC, bool SAFE, typename OutT, int BKT = GB_K, int RS = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                        const bf16_t* __restrict__ B, int64_t ldb,
                                                        OutT* __restrict__ C, int64_t M, int64_t N, int64_t K,
                                                        int64_t k_per_split, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A tile + B tile), 128 x BKT bf16 each
    constexpr int OPB = 128 * BKT * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tiles_n = (N + GB_N - 1) / GB_N, tiles_m = (M + GB_M - 1) / GB_M;
    int64_t tm, tn, split = 0;
    if (ep.atomic) {
        splitk_coords(tiles_m, tiles_n, ep.atomic, tm, tn, split);
        if (ep.ws_stride) { C += split * ep.ws_stride; ep.atomic = 0; }
    }
    else tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * GB_M, n0 = tn * GB_N;
    const int64_t kbeg = split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias gradient for free: rowsum_k A[m][k] = A . 1 is one more MFMA per A fragment with an all-ones operand (the A fragments are in
    // registers anyway; r01: the separate column-sum launches were 2 ms/step).  All tiles_n blocks of a row of tiles see the same A tile, so
    // block tn takes the K steps with step % tiles_n == tn and each of its two waves along n takes half of the 4 fragments: +2 MFMAs on
    // 1/tiles_n of the steps, evenly spread (doing it all in the tn == 0 blocks made those blocks 18 % slower and the kernel with them).
    constexpr bool do_rs = RS == 1, do_bs = RS == 2;
    f32x4 rsacc[2];
    rsacc[0] = rsacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t one_b = (bf16_t)1.f;
    const bf16x8 ones = {one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b};

    bf16x8 ra[BKT / 16], rb[BKT / 16];
    gload_tile<A_KC, BKT>(A, lda, m0, M, kbeg, kend, tid, ra);
    gload_tile<B_KC, BKT>(B, ldb, n0, N, kbeg, kend, tid, rb);
    lstore_tile<A_KC, BKT>(smem, tid, ra);
    lstore_tile<B_KC, BKT>(smem + OPB, tid, rb);
    __syncthreads();
    int cur = 0;
    int rs_n = (int)((kbeg / BKT) % tiles_n), rs_m = (int)((kbeg / BKT) % tiles_m);   // which block of the tile row / column owns this K step's rowsum
    for (int64_t k0 = kbeg; k0 < kend; k0 += BKT) {
        const bool more = (k0 + BKT) < kend;
        if (more) {
            gload_tile<A_KC, BKT>(A, lda, m0, M, k0 + BKT, kend, tid, ra);
            gload_tile<B_KC, BKT>(B, ldb, n0, N, k0 + BKT, kend, tid, rb);
        }
        const char* la = smem + cur * 2 * OPB;
        const char* lb = la + OPB;
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
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
            if (do_rs && rs_n == (int)tn) {
                if (wn == 0) {                         // wave-uniform branches: constant register indices, no selects
                    rsacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[0], rsacc[0], 0, 0, 0);
                    rsacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[1], rsacc[1], 0, 0, 0);
                } else {
                    rsacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[2], rsacc[0], 0, 0, 0);
                    rsacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[3], rsacc[1], 0, 0, 0);
                }
            } else if (do_bs && rs_m == (int)tm) {
                if (wm == 0) {
                    rsacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], ones, rsacc[0], 0, 0, 0);
                    rsacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], ones, rsacc[1], 0, 0, 0);
                } else {
                    rsacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], ones, rsacc[0], 0, 0, 0);
                    rsacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[3], ones, rsacc[1], 0, 0, 0);
                }
            }
        }
        if (more) {
            char* na = smem + (cur ^ 1) * 2 * OPB;
            lstore_tile<A_KC, BKT>(na, tid, ra);
            lstore_tile<B_KC, BKT>(na + OPB, tid, rb);
        }
        __syncthreads();
        cur ^= 1;
        if (++rs_n == (int)tiles_n) rs_n = 0;
        if (++rs_m == (int)tiles_m) rs_m = 0;
    }
    if (do_rs && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t m = m0 + wm * 64 + (2 * wn + i) * 16 + (lane & 15);
            if (m < M) atomicAdd(ep.a_rowsum + m, rsacc[i][0]);
        }
    } else if (do_bs && !do_rs && (lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = n0 + wn * 64 + (2 * wm + j) * 16 + (lane >> 4) * 4 + r;
                if (n < N) atomicAdd(ep.b_rowsum + n, rsacc[j][r]);
            }
    }
    epilogue_tile128<OutT, (BKT == 64 ? 1 : 2)>(ep, C, m0, n0, M, N, acc, smem, tid, wm, wn, lane);
}

// ================================================================================================
// bf16 MFMA kernel, v2: LDS-DMA (global_load_lds, 16 B/lane) into a 4-stage ring of 128x128x32 tiles.
// The register-staged kernel above keeps only ONE K-tile in flight, so with K=512 every K-step pays
// a full HBM/L2 round trip (measured ~1.2 us per step vs 0.2 us of MFMA work).  Here three K-tiles
// are in flight behind counted `s_waitcnt vmcnt(N)` + raw s_barrier (never vmcnt(0) in the loop, never
// __syncthreads), the staging costs no VGPRs, and LDS stays at 64 KB/block (2 blocks per CU).
// LDS images (per stage, per operand 8 KB): K-contig [128 rows][4 x 16-B chunks], chunk ^= (row>>2)&3;
// MN-contig [32 k][128 rows] with the 32-B granule swizzle of the v1 kernel.  The DMA writes LDS
// lane-linearly, so the swizzle is applied to the per-lane SOURCE address and again on the read.
// Requires K (per split) % 32 == 0; edge rows are clamped to valid addresses (their outputs are dropped).
constexpr int G2_BK = 32;   // K granularity the host guarantees for this kernel family

// per-lane BYTE offsets of the 16-B pieces this lane fetches for one operand tile (constant over the K loop);
// the K position is carried by a wave-uniform base pointer so that a K step costs no per-lane address arithmetic.
template <bool KC, int BK>
__device__ __forceinline__ void glds_offsets(int64_t ld, int64_t row0, int64_t nrows, int wave, int lane, uint32_t (&off)[BK / 16]) {
    constexpr int NI = BK / 16;               // wave-instructions per wave per operand tile (1 KiB each)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = wave * NI + i;
        if constexpr (KC) {
            int row, c;
            if constexpr (BK == 32) { row = j * 16 + (lane >> 2); c = (lane & 3) ^ swz32(row); }
            else { row = j * 8 + (lane >> 3); c = (lane & 7) ^ (row & 7); }
            int64_t gr = row0 + row;
            if (gr > nrows - 1) gr = nrows - 1;
            off[i] = (uint32_t)((gr * ld + c * 8) * 2);
        } else {
            const int k = j * 4 + (lane >> 4), p16 = lane & 15;
            const int g = (p16 >> 1) ^ swz_k(k);
            int64_t gr = row0 + (g * 2 + (p16 & 1)) * 8;
            if (gr > nrows - 1) gr = ((nrows - 1) >> 3) << 3;
            off[i] = (uint32_t)(((int64_t)k * ld + gr) * 2);
        }
    }
}
template <int NI>
__device__ __forceinline__ void glds_issue2(const char* __restrict__ base, const uint32_t (&off)[NI], char* stage_op, int wave) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[i]),
                                         (__attribute__((address_space(3))) void*)(stage_op + (wave * NI + i) * 1024), 16, 0, 0);
}

template <bool KC, int BK>
__device__ __forceinline__ bf16x8 lfrag2(const char* lds, int rbase, int ks, int lane) {
    if constexpr (KC) {
        const int row = rbase + (lane & 15);
        if constexpr (BK == 32) {
            const int pc = (lane >> 4) ^ swz32(row);
            return *(const bf16x8*)(lds + row * 64 + (pc << 4));
        } else {
            const int ch = ks * 4 + (lane >> 4);
            return *(const bf16x8*)(lds + row * 128 + ((ch ^ (row & 7)) << 4));
        }
    } else {
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

template <int N> __device__ __forceinline__ void wait_vmcnt();
template <> __device__ __forceinline__ void wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }

// BK = 32, ST = 4 : 64 KB ring, three 32-deep tiles in flight;  BK = 64, ST = 2 : 64 KB double buffer of full 128-B lines.
template <bool A_KC, bool B_KC, typename OutT, int BK, int ST>
__global__ __launch_bounds__(256, (ST == 2 && BK == 32) ? 4 : 2) void gemm_bf16_glds_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                             OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int64_t k_per_split,
                                                             EpiParams ep) {
    constexpr int OPB = 128 * BK * 2, STAGE = 2 * OPB, NI = BK / 16, LPT = 2 * NI;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ST x (A tile + B tile)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tiles_n = (N + GB_N - 1) / GB_N, tiles_m = (M + GB_M - 1) / GB_M;
    int64_t tm, tn, split = 0;
    if (ep.atomic) {
        splitk_coords(tiles_m, tiles_n, ep.atomic, tm, tn, split);
        if (ep.ws_stride) { C += split * ep.ws_stride; ep.atomic = 0; }
    }
    else tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * GB_M, n0 = tn * GB_N;
    const int64_t kbeg = split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg) / BK);
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint32_t offA[NI], offB[NI];
    glds_offsets<A_KC, BK>(lda, m0, M, wave, lane, offA);
    glds_offsets<B_KC, BK>(ldb, n0, N, wave, lane, offB);
    // wave-uniform running base pointers (bytes)
    const char* gA = (const char*)A + (A_KC ? kbeg : kbeg * lda) * 2;
    const char* gB = (const char*)B + (B_KC ? kbeg : kbeg * ldb) * 2;
    const int64_t stepA = (A_KC ? (int64_t)BK : (int64_t)BK * lda) * 2;
    const int64_t stepB = (B_KC ? (int64_t)BK : (int64_t)BK * ldb) * 2;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) {
        if (s < nk) {
            glds_issue2<NI>(gA, offA, smem + s * STAGE, wave);
            glds_issue2<NI>(gB, offB, smem + s * STAGE + OPB, wave);
            gA += stepA;
            gB += stepB;
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int newer = (nk - 1 - kt) < (ST - 2) ? (nk - 1 - kt) : (ST - 2);
        if (newer >= 2) wait_vmcnt<2 * LPT>();
        else if (newer == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int nxt = kt + ST - 1;
        if (nxt < nk && ep.ablate != 1 && ep.ablate != 3) {
            char* st = smem + (nxt % ST) * STAGE;
            glds_issue2<NI>(gA, offA, st, wave);
            glds_issue2<NI>(gB, offB, st + OPB, wave);
            gA += stepA;
            gB += stepB;
        }
        const char* la = smem + (kt % ST) * STAGE;
        const char* lb = la + OPB;
        if (ep.ablate == 2) continue;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 fa[4], fb[4];
            if (ep.ablate == 3) {          // diagnostics: no LDS fragment reads either (MFMA + barrier + epilogue only)
#pragma unroll
                for (int i = 0; i < 4; ++i) { fa[i] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1}; fb[i] = (bf16x8){1, 1, 1, 1, 1, 1, 1, 1}; }
            } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = lfrag2<A_KC, BK>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = lfrag2<B_KC, BK>(lb, wn * 64 + j * 16, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if constexpr (ST == 2) {   // double buffer: the stage just computed is overwritten by the NEXT iteration's issue
            asm volatile("" ::: "memory");
        }
    }
    epilogue_tile128<OutT, (ST * STAGE >= 65536 ? 1 : 2)>(ep, C, m0, n0, M, N, acc, smem, tid, wm, wn, lane);
}

// ================================================================================================
// bf16 MFMA kernel, v3: 256x256x64 block tile, 8 waves (2x4, wave tile 128x64 = 8x4 MFMA tiles), LDS-DMA
// double buffer (2 x 64 KB).  PMC on v1/v2 showed the 128^2 tile is bound by the CU's vector-memory front
// end (TA busy 61 % at only 12 B/clk/CU: half-line 64-B row pieces, 1 B of tile traffic per 64 FLOP); the
// 256^2 tile halves the tile bytes per FLOP and fetches full 128-B lines (BK=64), and one K step carries
// 64 MFMAs per wave (1024 MFMA cycles), enough to cover the L2 round trip of the next stage with a plain
// double buffer: wait vmcnt(0) -> barrier -> issue stage k+1 -> compute stage k.
constexpr int G3_M = 256, G3_N = 256, G3_K = 64, G3_OP = 32768;   // bytes per operand per stage

template <bool KC>
__device__ __forceinline__ void g3_offsets(int64_t ld, int64_t row0, int64_t nrows, int wave, int lane, uint32_t (&off)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wave * 4 + i;   // wave-instruction 0..31 (1 KiB each)
        if constexpr (KC) {
            const int row = j * 8 + (lane >> 3), pc = lane & 7;
            const int c = pc ^ (row & 7);
            int64_t gr = row0 + row;
            if (gr > nrows - 1) gr = nrows - 1;
            off[i] = (uint32_t)((gr * ld + c * 8) * 2);
        } else {
            const int k = j * 2 + (lane >> 5), p16 = lane & 31;
            const int g = (p16 >> 1) ^ swz_k(k);
            int64_t gr = row0 + (g * 2 + (p16 & 1)) * 8;
            if (gr > nrows - 1) gr = ((nrows - 1) >> 3) << 3;
            off[i] = (uint32_t)(((int64_t)k * ld + gr) * 2);
        }
    }
}
__device__ __forceinline__ void g3_issue(const char* __restrict__ base, const uint32_t (&off)[4], char* stage_op, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[i]),
                                         (__attribute__((address_space(3))) void*)(stage_op + (wave * 4 + i) * 1024), 16, 0, 0);
}
template <bool KC>
__device__ __forceinline__ bf16x8 g3_frag(const char* lds, int rbase, int ks, int lane) {
    if constexpr (KC) {
        const int row = rbase + (lane & 15), ch = ks * 4 + (lane >> 4);
        return *(const bf16x8*)(lds + row * 128 + ((ch ^ (row & 7)) << 4));
    } else {
        const int i = lane & 15, g = rbase >> 4;
        bf16x8 v;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = ks * 32 + (lane >> 4) * 8 + h * 4 + (i >> 2);
            const char* p = lds + k * 512 + ((g ^ swz_k(k)) << 5) + ((i & 3) << 3);
            short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
            bf16x4 tb = __builtin_bit_cast(bf16x4, t);
            v[h * 4 + 0] = tb[0]; v[h * 4 + 1] = tb[1]; v[h * 4 + 2] = tb[2]; v[h * 4 + 3] = tb[3];
        }
        return v;
    }
}

template <bool A_KC, bool B_KC, typename OutT>
__global__ __launch_bounds__(512) void gemm_bf16_g3_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                           OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int64_t k_per_split, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (32 KB A + 32 KB B)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tiles_n = (N + G3_N - 1) / G3_N, tiles_m = (M + G3_M - 1) / G3_M;
    int64_t tm, tn, split = 0;
    if (ep.atomic) {
        splitk_coords(tiles_m, tiles_n, ep.atomic, tm, tn, split);
        if (ep.ws_stride) { C += split * ep.ws_stride; ep.atomic = 0; }
    }
    else tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * G3_M, n0 = tn * G3_N;
    const int64_t kbeg = split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg) / G3_K);
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t offA[4], offB[4];
    g3_offsets<A_KC>(lda, m0, M, wave, lane, offA);
    g3_offsets<B_KC>(ldb, n0, N, wave, lane, offB);
    const char* gA = (const char*)A + (A_KC ? kbeg : kbeg * lda) * 2;
    const char* gB = (const char*)B + (B_KC ? kbeg : kbeg * ldb) * 2;
    const int64_t stepA = (A_KC ? (int64_t)G3_K : (int64_t)G3_K * lda) * 2;
    const int64_t stepB = (B_KC ? (int64_t)G3_K : (int64_t)G3_K * ldb) * 2;
    g3_issue(gA, offA, smem, wave);
    g3_issue(gB, offB, smem + G3_OP, wave);
    gA += stepA;
    gB += stepB;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) {
            char* st = smem + ((kt + 1) & 1) * 2 * G3_OP;
            g3_issue(gA, offA, st, wave);
            g3_issue(gB, offB, st + G3_OP, wave);
            gA += stepA;
            gB += stepB;
        }
        const char* la = smem + (kt & 1) * 2 * G3_OP;
        const char* lb = la + G3_OP;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[8], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = g3_frag<B_KC>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = g3_frag<A_KC>(la, wm * 128 + i * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < 8; ++i)
#pragma clang loop unroll(full)
        for (int j = 0; j < 4; ++j) {
            int64_t m = m0 + wm * 128 + i * 16 + (lane & 15);
            int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (m < M && n < N) epi_store4_call<OutT>(ep, C, m, n, acc[i][j], N);
        }
}

template <bool A_KC, bool B_KC, typename OutT>
static void launch_g3(dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t M, int64_t N,
                      int64_t K, int64_t kps, const EpiParams& ep) {
    auto kfn = gemm_bf16_g3_kernel<A_KC, B_KC, OutT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G3_OP);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(512), 4 * G3_OP, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
}
template <typename OutT>
static void dispatch_g3(bool akc, bool bkc, dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C,
                        int64_t M, int64_t N, int64_t K, int64_t kps, const EpiParams& ep) {
    if (akc && bkc) launch_g3<true, true, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (akc && !bkc) launch_g3<true, false, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (!akc && bkc) launch_g3<false, true, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else launch_g3<false, false, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
}

// ================================================================================================
// bf16 MFMA kernel, v6: 256x256x64 block tile, 8 waves, FOUR PHASES PER K-TILE with two wave groups running one barrier apart.
// The 128^2 kernels above top out at ~840 TFLOP/s even at K=2048 (one barrier-synchronised {wait, read fragments, MFMA} step per
// K-tile: every wave of the block reads LDS at the same time and then every wave issues MFMAs at the same time, so the matrix
// pipe idles during the read sections), and the plain 256^2 double buffer (v3) is no better.  Here
//   * the K-tile is staged as four 16-KB half-tiles (A rows 0-127 / 128-255, B cols 0-127 / 128-255) in a 2 x 64 KB ring;
//     a wave owns 64 rows of EACH A half and 32 columns of EACH B half (wave tile 128 x 64);
//   * one phase = one k-half (32) of one A half against all four B fragments = 16 MFMAs on 16 distinct accumulators:
//     P1 reads B.k0 + A0.k0 (8 fragments), P2 B.k1 + A0.k1 (8), P3 A1.k0 (4), P4 A1.k1 (4).  (The first version split by
//     quadrant, 12 / 4 / 8 / 0 fragments: the 12-fragment phase was 24 transposing reads on the wgrad layout — more than the 15
//     outstanding LDS ops a wave can have — and took 1050 cycles against 470 for the others.)
//   * a half-tile slot is re-staged (LDS-DMA, 2 instructions per wave) two phases after its last fragment read, one half-tile per
//     phase: P1: B0 of t+1, P2: B1 of t+1, P3: A1 of t+1, P4: A0 of t+2 — loads live for 2-4 phases and are retired by two COUNTED
//     waits per K-tile (P2: vmcnt(6) -> A1 of t, read in P3;  P4: vmcnt(4) -> A0, B0, B1 of t+1, read in the next P1); the queue
//     never drains inside the loop;
//   * waves 4-7 (the second wave of every SIMD: HW_ID.SIMD_ID of waves w and w+4 is equal) run one s_barrier behind waves 0-3:
//     while one wave of a SIMD is in its MFMA section the other is in its read / stage section (s_setprio favours the MFMA wave).
// Hazards (E_n = n-th barrier; group 0: R1 E1 M1 E2 R2 E3 M2 E4 R3 E5 M3 E6 R4 E7 M4 E8, group 1 the same shifted by one E):
//   RAW: a wave's wait sits before the first barrier of its phase p (E_{2p-1} for group 0, E_{2p} for group 1); the data is first read
//        in R_{p+1}, which group 0 starts after E_{2p}.  WAR: a slot read in R_p has its reads retired at the latest after E_{2p}
//        (group 1's lgkmcnt before its M_p) and is re-staged in R_{p+2} or later, which group 0 starts after E_{2p+2}.
constexpr int G6_HT = 16384, G6_STAGE = 65536;
#ifndef G6_PROFILE
#define G6_PROFILE 0      // 1: per-section cycle stamps (EMO_GEMM_ABLATE & 8, tools/g6_phases.py); costs registers, diagnostics builds only
#endif

template <bool KC>
__device__ __forceinline__ void g6_offsets(int64_t ld, int64_t row0, int64_t nrows, int wave, int lane, uint32_t (&off)[2][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave * 2 + i;       // wave-instruction 0..15 of the half-tile (1 KiB each)
            if constexpr (KC) {               // [128 rows][64 k]: 8 rows x 128 B per instruction, chunk ^= row & 7
                const int row = j * 8 + (lane >> 3), c = (lane & 7) ^ (row & 7);
                int64_t gr = row0 + h * 128 + row;
                if (gr > nrows - 1) gr = nrows - 1;
                off[h][i] = (uint32_t)((gr * ld + c * 8) * 2);
            } else {                          // [64 k][128 rows]: 4 k x 256 B per instruction, 32-B granule swizzle
                const int k = j * 4 + (lane >> 4), p16 = lane & 15;
                const int g = (p16 >> 1) ^ swz_k(k);
                int64_t gr = row0 + h * 128 + (g * 2 + (p16 & 1)) * 8;
                if (gr > nrows - 1) gr = ((nrows - 1) >> 3) << 3;
                off[h][i] = (uint32_t)(((int64_t)k * ld + gr) * 2);
            }
        }
}
__device__ __forceinline__ void g6_issue(const char* __restrict__ base, const uint32_t (&off)[2], char* slot, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[i]),
                                         (__attribute__((address_space(3))) void*)(slot + (wave * 2 + i) * 1024), 16, 0, 0);
}
#define G6_BARRIER() asm volatile("s_barrier" ::: "memory")
#define G6_MFMA_BEGIN() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); } while (0)
#define G6_MFMA_END() do { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } while (0)

// 256 x 256 accumulator tile -> global through the (idle) 128 KB of LDS, one 128-row half at a time: fp32 image [128][256] with the 16-B chunk
// index XOR (row & 15) (fragment-layout writes conflict-free), then every thread owns 8 consecutive columns of a row (epi_row8: 16-B
// coalesced loads / stores, 512 B or 1 KB contiguous per row).  The fragment-layout epilogue (8-B pieces, 16 rows per instruction) cost
// ~30 us per tile here (r01: 828 -> TFLOP/s with the tile loads ablated on the K=2048 forward shape).
template <typename OutT>
__device__ __forceinline__ void epilogue_tile256(const EpiParams& ep, OutT* __restrict__ C, int64_t m0, int64_t n0, int64_t M, int64_t N,
                                                 const f32x4 (&acc)[8][4], char* lds, int tid, int wr, int wc, int lane) {
    if (ep.atomic) {
#pragma clang loop unroll(full)
        for (int i = 0; i < 8; ++i)
#pragma clang loop unroll(full)
            for (int j = 0; j < 4; ++j) {
                const int64_t m = m0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + (lane & 15);
                const int64_t n = n0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + (lane >> 4) * 4;
                if (m < M && n < N) epi_store4_call<OutT>(ep, C, m, n, acc[i][j], N);
            }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wr * 64 + i * 16 + (lane & 15);
                const int chunk = ((j >> 1) * 128 + wc * 32 + (j & 1) * 16 + (lane >> 4) * 4) >> 2;   // 16-B chunk 0..63
                *(f32x4*)(lds + row * 1024 + ((chunk ^ (row & 15)) << 4)) = acc[4 * h + i][j];
            }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + 512 * it;
            const int row = idx >> 5, grp = idx & 31;
            const int64_t m = m0 + h * 128 + row, n = n0 + grp * 8;
            if (m < M && n < N) {
                const f32x4 lo = *(const f32x4*)(lds + row * 1024 + (((2 * grp) ^ (row & 15)) << 4));
                const f32x4 hi = *(const f32x4*)(lds + row * 1024 + (((2 * grp + 1) ^ (row & 15)) << 4));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                epi_row8<OutT>(ep, C, m, n, v, N);
            }
        }
    }
}

// RS: 0 plain, 1 = a_rowsum (sum over k of A, per output row m), 2 = b_rowsum (sum over k of B, per output column n): the wgrad bias
// gradient as one extra MFMA against an all-ones operand, on 1/tiles_n (1/tiles_m) of the K-tiles per block and one fragment per wave.
template <bool A_KC, bool B_KC, typename OutT, int RS>
__global__ __launch_bounds__(512) void gemm_bf16_g6_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                           OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int64_t k_per_split, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x {A0, A1, B0, B1} x 16 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tiles_n = (N + G3_N - 1) / G3_N, tiles_m = (M + G3_M - 1) / G3_M;
    int64_t tm, tn, split = 0;
    if (ep.atomic) {
        splitk_coords(tiles_m, tiles_n, ep.atomic, tm, tn, split);
        if (ep.ws_stride) { C += split * ep.ws_stride; ep.atomic = 0; }
    }
    else tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * G3_M, n0 = tn * G3_N;
    const int64_t kbeg = split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int nk = __builtin_amdgcn_readfirstlane((int)((kend - kbeg) / G3_K));
    const int wr = wave >> 2, wc = wave & 3;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 rsacc[2];
    rsacc[0] = rsacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t one_b = (bf16_t)1.f;
    const bf16x8 ones = {one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b};
    int rs_own = RS == 1 ? (int)((kbeg / G3_K) % tiles_n) : (RS == 2 ? (int)((kbeg / G3_K) % tiles_m) : 0);
    const int rs_me = RS == 1 ? (int)tn : (int)tm, rs_mod = RS == 1 ? (int)tiles_n : (int)tiles_m;

    uint32_t offA[2][2], offB[2][2];
    g6_offsets<A_KC>(lda, m0, M, wave, lane, offA);
    g6_offsets<B_KC>(ldb, n0, N, wave, lane, offB);
    const char* gA = (const char*)A + (A_KC ? kbeg : kbeg * lda) * 2;
    const char* gB = (const char*)B + (B_KC ? kbeg : kbeg * ldb) * 2;
    const int64_t stepA = (A_KC ? (int64_t)G3_K : (int64_t)G3_K * lda) * 2;
    const int64_t stepB = (B_KC ? (int64_t)G3_K : (int64_t)G3_K * ldb) * 2;
#define G6_ISSUE_A(h, t) g6_issue(gA + (int64_t)(t) * stepA, offA[h], smem + ((t) & 1) * G6_STAGE + (h) * G6_HT, wave)
#define G6_ISSUE_B(h, t) g6_issue(gB + (int64_t)(t) * stepB, offB[h], smem + ((t) & 1) * G6_STAGE + (2 + (h)) * G6_HT, wave)
    const bool noload = (ep.ablate & 1) != 0;     // diagnostics: no tile DMA inside the loop
    G6_ISSUE_B(0, 0); G6_ISSUE_A(0, 0); G6_ISSUE_B(1, 0); G6_ISSUE_A(1, 0);
    if (nk > 1) {
        G6_ISSUE_A(0, 1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G6_BARRIER();
    const int grp = (ep.ablate & 4) ? (wave & 1) : (wave >> 2);      // the two waves of a SIMD (w, w + 4) must be in different groups
    if (grp == 1) G6_BARRIER();         // group 1 runs one barrier behind group 0
    bf16x8 fa[4], fb[4][2];
#if G6_PROFILE
    const bool dbg = (ep.ablate & 8) != 0;
    uint64_t dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dprev = dbg ? __builtin_readcyclecounter() : 0;
#define G6_STAMP(slot) do { if (dbg) { const uint64_t c_ = __builtin_readcyclecounter(); dsum[slot] += c_ - dprev; dprev = c_; } } while (0)
#else
#define G6_STAMP(slot) do { } while (0)
#endif
#define G6_READ_B(ks)                                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                                       \
        fb[j][ks] = lfrag2<B_KC, 64>(st + (2 + (j >> 1)) * G6_HT, wc * 32 + (j & 1) * 16, ks, lane)
#define G6_READ_A(h, ks)                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = lfrag2<A_KC, 64>(st + (h) * G6_HT, wr * 64 + i * 16, ks, lane)
#define G6_MMA(h, ks)                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                                   \
            acc[4 * (h) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i], acc[4 * (h) + i][j], 0, 0, 0)
    // bias-gradient MFMAs against the all-ones operand: fragment i = wc of the A half (RS = 1) / fragments j = wr, 2 + wr of B (RS = 2);
    // wave-uniform branches keep the register indices static
#define G6_RS_A(h)                                                                                                                     \
    if (RS == 1 && rs_own == rs_me) {                                                                                                  \
        if (wc == 0) rsacc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[0], rsacc[h], 0, 0, 0);                                \
        else if (wc == 1) rsacc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[1], rsacc[h], 0, 0, 0);                           \
        else if (wc == 2) rsacc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[2], rsacc[h], 0, 0, 0);                           \
        else rsacc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[3], rsacc[h], 0, 0, 0);                                        \
    }
#define G6_RS_B(ks)                                                                                                                    \
    if (RS == 2 && rs_own == rs_me) {                                                                                                  \
        if (wr == 0) {                                                                                                                 \
            rsacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][ks], ones, rsacc[0], 0, 0, 0);                                     \
            rsacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2][ks], ones, rsacc[1], 0, 0, 0);                                     \
        } else {                                                                                                                       \
            rsacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1][ks], ones, rsacc[0], 0, 0, 0);                                     \
            rsacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[3][ks], ones, rsacc[1], 0, 0, 0);                                     \
        }                                                                                                                              \
    }
    for (int t = 0; t < nk; ++t) {
        const char* st = smem + (t & 1) * G6_STAGE;
        const bool more = t + 1 < nk && !noload;
        // ---- P1: k-half 0 of (A0, B0|B1)
        G6_READ_B(0);
        G6_READ_A(0, 0);
        if (more) G6_ISSUE_B(0, t + 1);
        G6_BARRIER();
        G6_STAMP(0);
        G6_MFMA_BEGIN();
        G6_MMA(0, 0);
        G6_RS_A(0);
        G6_RS_B(0);
        G6_MFMA_END();
        G6_BARRIER();
        G6_STAMP(1);
        // ---- P2: k-half 1 of (A0, B0|B1); the wait that makes A1 of this K-tile readable in P3
        G6_READ_B(1);
        G6_READ_A(0, 1);
        if (more) {
            G6_ISSUE_B(1, t + 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // newer than A1(t): A0, B0, B1 of t + 1
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G6_BARRIER();
        G6_STAMP(2);
        G6_MFMA_BEGIN();
        G6_MMA(0, 1);
        G6_RS_A(0);
        G6_RS_B(1);
        G6_MFMA_END();
        G6_BARRIER();
        G6_STAMP(3);
        // ---- P3: k-half 0 of (A1, B0|B1)
        G6_READ_A(1, 0);
        if (more) G6_ISSUE_A(1, t + 1);
        G6_BARRIER();
        G6_STAMP(4);
        G6_MFMA_BEGIN();
        G6_MMA(1, 0);
        G6_RS_A(1);
        G6_MFMA_END();
        G6_BARRIER();
        G6_STAMP(5);
        // ---- P4: k-half 1 of (A1, B0|B1); the wait that makes A0, B0, B1 of K-tile t + 1 readable in its P1
        G6_READ_A(1, 1);
        if (more) {
            if (t + 2 < nk) {
                G6_ISSUE_A(0, t + 2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // newer than B1(t + 1): A1 of t + 1, A0 of t + 2
            } else {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
        }
        G6_BARRIER();
        G6_STAMP(6);
        G6_MFMA_BEGIN();
        G6_MMA(1, 1);
        G6_RS_A(1);
        G6_MFMA_END();
        G6_BARRIER();
        G6_STAMP(7);
        if (RS != 0) { if (++rs_own == rs_mod) rs_own = 0; }
    }
    if (grp == 0) G6_BARRIER();
#undef G6_ISSUE_A
#undef G6_ISSUE_B
#undef G6_READ_A
#undef G6_READ_B
#undef G6_MMA
#undef G6_RS_A
#undef G6_RS_B
    if (RS == 1 && (lane >> 4) == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t m = m0 + h * 128 + wr * 64 + wc * 16 + (lane & 15);
            if (m < M) atomicAdd(ep.a_rowsum + m, rsacc[h][0]);
        }
    }
    if (RS == 2 && (lane & 15) == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = n0 + h * 128 + wc * 32 + wr * 16 + (lane >> 4) * 4 + r;
                if (n < N) atomicAdd(ep.b_rowsum + n, rsacc[h][r]);
            }
    }
    epilogue_tile256<OutT>(ep, C, m0, n0, M, N, acc, smem, tid, wr, wc, lane);
#if G6_PROFILE
    if (dbg && blockIdx.x == 0 && lane == 0 && ep.mul_aux) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ((float*)ep.mul_aux)[wave * 8 + q] = (float)dsum[q] / (float)nk;
        ((float*)ep.mul_aux)[64 + wave] = (float)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3);   // HW_ID.SIMD_ID
    }
#endif
#undef G6_STAMP
}

template <bool A_KC, bool B_KC, typename OutT, int RS>
static void launch_g6(dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t M, int64_t N,
                      int64_t K, int64_t kps, const EpiParams& ep) {
    auto kfn = gemm_bf16_g6_kernel<A_KC, B_KC, OutT, RS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G6_STAGE);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(512), 2 * G6_STAGE, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
}
template <typename OutT>
static void dispatch_g6(bool akc, bool bkc, int rs, dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C,
                        int64_t M, int64_t N, int64_t K, int64_t kps, const EpiParams& ep) {
    if (!akc && !bkc) {                  // the wgrad layout carries the bias-gradient instances
        if (rs == 1) launch_g6<false, false, OutT, 1>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
        else if (rs == 2) launch_g6<false, false, OutT, 2>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
        else launch_g6<false, false, OutT, 0>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    }
    else if (akc && bkc) launch_g6<true, true, OutT, 0>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (akc && !bkc) launch_g6<true, false, OutT, 0>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else launch_g6<false, true, OutT, 0>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
}

// ================================================================================================
// skinny GEMM for the decode step (M <= 32 rows: n streams x 1 token): weight-bandwidth / launch bound.
// One wave per 16 output columns, the whole K loop in registers: weight rows (nn.Linear [N,K]) and the M
// activation rows are fetched as MFMA fragments straight from global memory (16 B per lane, no LDS —
// the guide's rule for M <= 16 GEMV-like shapes), 4 K-steps of loads in flight.
// CW: k width of a wave's chunk (128; 64 for the 16-wave K = 2048 variant so that the 16 tiles fit the LDS)
template <typename OutT, int NW = 4, int CW = 128>
__global__ __launch_bounds__(64 * NW) void gemm_bf16_skinny_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                               OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, EpiParams ep) {
    // block = 16 output columns; its NW waves split the K range (shorter dependent chains, NW x the loads in flight) and
    // combine through LDS; wave 0 runs the fused epilogue.
    //
    // Operand path: an MFMA fragment wants lane -> (row l&15, 16-B k-chunk l>>4), i.e. the 16 lanes of a quarter wave in 16 DIFFERENT rows;
    // read straight from global that is 16 cache-line tag look-ups per quarter wave and the texture path delivers ~16 B/clk: the kernel
    // time was linear in K (r01: 3.8 / 5.1 / 8.1 / 13.6 us at K = 512 / 1024 / 2048 / 4096, whatever N).  Each wave therefore streams its
    // [32 + 16 rows] x 128-k chunk with row-contiguous 16-B loads (a quarter wave = one 256-B row), parks it in a private LDS tile (row
    // stride 144 bf16 = 8 mod 16 dwords: conflict-free ds_read_b128) and takes the fragments from there; the next chunk's loads are in
    // flight during the MFMAs.
    //
    // LayerNorm folding (decode step: a standalone LN launch on 32 rows costs as much as this whole GEMM).  For A' = LN(A) * gamma + beta:
    //   A'.W^T [m][n] = rstd[m] * ( (A.(gamma*W)^T)[m][n] - mean[m] * c1[n] ) + (beta.W^T)[n],   c1[n] = sum_k gamma_k W[n][k]
    // so the kernel multiplies the RAW rows by the gamma-scaled weights (prepared once by the caller, like c1 and the folded bias), gets
    // mean / rstd of its A rows from the fragments it loads anyway, and applies them in the epilogue (ep.ln_c1).  A residual that is itself
    // a LayerNorm output is rebuilt from the raw tensor and the statistics an earlier kernel exported (ep.rln_*, ep.ln_stats_out).
    constexpr int LDT = CW + 16, TROWS = 48;                // tile rows 0..31: A, 32..47: B; row stride = 8 (mod 16) dwords
    constexpr int CPR = CW / 8, RPI = 64 / CPR, NA = 32 / RPI, NB = 16 / RPI;   // 16-B chunks per row, rows per load instruction, loads per lane
    extern __shared__ __attribute__((aligned(16))) char skinny_smem[];
    bf16_t* tile = (bf16_t*)skinny_smem + (threadIdx.x >> 6) * (TROWS * LDT);
    f32x4 (*red)[2][64] = (f32x4 (*)[2][64])(skinny_smem + (size_t)NW * TROWS * LDT * sizeof(bf16_t));
    float (*st)[2][16][2] = (float (*)[2][16][2])((char*)red + sizeof(f32x4) * (NW - 1) * 2 * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int64_t kq = ((K / 32 + NW - 1) / NW) * 32;      // K slice per wave (multiple of 32)
    const int64_t kb = wave * kq;
    int64_t ke = kb + kq;
    if (ke > K) ke = K;
    // loader mapping: item = lane + 64 j -> (row = item / 16, 16-B chunk = item % 16); A: j < 8 (32 rows), B: j < 4 (16 rows)
    const int lrow = lane / CPR, lch = lane % CPR;
    const bf16_t* ga[NA];
    const bf16_t* gb[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int64_t m = lrow + RPI * j;
        if (m > M - 1) m = M - 1;
        ga[j] = A + m * lda + lch * 8;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int64_t nn = n0 + lrow + RPI * j;
        if (nn > N - 1) nn = N - 1;
        gb[j] = B + nn * ldb + lch * 8;
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const bool ln = ep.ln_c1 != nullptr;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;        // sum / sum of squares of this lane's slices of rows (lane&15) / 16 + (lane&15)
    bf16x8 ra[NA], rb[NB];
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    auto fetch = [&](int64_t k) {
        const bool ok = k + lch * 8 < ke;                   // chunk tail (ke - k < CW): the missing k columns read as zeros
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = ok ? *(const bf16x8*)(ga[j] + k) : zero8;
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = ok ? *(const bf16x8*)(gb[j] + k) : zero8;
    };
    if (kb < ke) fetch(kb);
    for (int64_t k = kb; k < ke; k += CW) {
#pragma unroll
        for (int j = 0; j < NA; ++j) *(bf16x8*)(tile + (lrow + RPI * j) * LDT + lch * 8) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *(bf16x8*)(tile + (32 + lrow + RPI * j) * LDT + lch * 8) = rb[j];
        if (k + CW < ke) fetch(k + CW);                     // in flight during this chunk's MFMAs
        __builtin_amdgcn_wave_barrier();                    // wave-private tile: LDS operations of one wave complete in program order
        const int steps = (int)(((ke - k) < CW ? (ke - k) : CW) / 32);
#pragma unroll
        for (int u = 0; u < CW / 32; ++u) {
            if (u < steps) {
                const bf16x8 fb = *(const bf16x8*)(tile + (32 + (lane & 15)) * LDT + u * 32 + (lane >> 4) * 8);
                const bf16x8 fa0 = *(const bf16x8*)(tile + (lane & 15) * LDT + u * 32 + (lane >> 4) * 8);
                const bf16x8 fa1 = *(const bf16x8*)(tile + (16 + (lane & 15)) * LDT + u * 32 + (lane >> 4) * 8);
                if (ln) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (float)fa0[e], y = (float)fa1[e];
                        s0 += x; q0 += x * x; s1 += y; q1 += y * y;
                    }
                }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa1, acc1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (ln) {                                                // the 4 lane groups hold different k slices of the same row
        s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
        q0 += __shfl_xor(q0, 16, 64); q0 += __shfl_xor(q0, 32, 64);
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        q1 += __shfl_xor(q1, 16, 64); q1 += __shfl_xor(q1, 32, 64);
        if (lane < 16) { st[wave][0][lane][0] = s0; st[wave][0][lane][1] = q0; st[wave][1][lane][0] = s1; st[wave][1][lane][1] = q1; }
    }
    if (wave > 0) { red[wave - 1][0][lane] = acc0; red[wave - 1][1][lane] = acc1; }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) { acc0 += red[w][0][lane]; acc1 += red[w][1][lane]; }
    const int64_t n = n0 + (lane >> 4) * 4;
    const int64_t m0 = lane & 15;
    if (ln) {
        const int r = lane & 15;
        const float invK = 1.f / (float)K;
        float su = 0.f, sq = 0.f, tu = 0.f, tq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { su += st[w][0][r][0]; sq += st[w][0][r][1]; tu += st[w][1][r][0]; tq += st[w][1][r][1]; }
        const float mean0 = su * invK, mean1 = tu * invK;
        const float rstd0 = rsqrtf(fmaxf(sq * invK - mean0 * mean0, 0.f) + ep.ln_eps), rstd1 = rsqrtf(fmaxf(tq * invK - mean1 * mean1, 0.f) + ep.ln_eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float c1 = (n + i < N) ? ep.ln_c1[n + i] : 0.f;
            acc0[i] = rstd0 * (acc0[i] - mean0 * c1);
            acc1[i] = rstd1 * (acc1[i] - mean1 * c1);
        }
        if (ep.ln_stats_out && blockIdx.x == 0 && lane < 16) {
            if (m0 < M) { ep.ln_stats_out[m0 * 2] = mean0; ep.ln_stats_out[m0 * 2 + 1] = rstd0; }
            if (m0 + 16 < M) { ep.ln_stats_out[(m0 + 16) * 2] = mean1; ep.ln_stats_out[(m0 + 16) * 2 + 1] = rstd1; }
        }
    }
    if (ep.rln_x) {                                          // residual = LayerNorm(rln_x) rebuilt from exported statistics
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t m = m0 + 16 * h;
            if (m < M) {
                const float mean = ep.rln_stats[m * 2], rstd = ep.rln_stats[m * 2 + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (n + i < N) {
                        const float x = to_f32<OutT>(((const OutT*)ep.rln_x)[m * ep.ldc + n + i]);
                        const float rv = (x - mean) * rstd * ep.rln_gamma[n + i] + ep.rln_beta[n + i];
                        if (h == 0) acc0[i] += rv; else acc1[i] += rv;
                    }
                }
            }
        }
    }
    if (n < N) {
        if (m0 < M) epi_store4<OutT>(ep, C, m0, n, acc0, N);
        if (m0 + 16 < M) epi_store4<OutT>(ep, C, m0 + 16, n, acc1, N);
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
    if constexpr (!A_KC && !B_KC && !SAFE && sizeof(OutT) == 4) {       // the wgrad layout: bias-gradient variants
        if (ep.a_rowsum || ep.b_rowsum) {
            auto k1 = gemm_bf16_kernel<A_KC, B_KC, SAFE, OutT, GB_K, 1>;
            auto k2 = gemm_bf16_kernel<A_KC, B_KC, SAFE, OutT, GB_K, 2>;
            static bool attr_rs = false;
            if (!attr_rs) {
                (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                attr_rs = true;
            }
            if (ep.a_rowsum) hipLaunchKernelGGL(k1, grid, dim3(256), 65536, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
            else hipLaunchKernelGGL(k2, grid, dim3(256), 65536, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
            return;
        }
    }
    auto kfn = gemm_bf16_kernel<A_KC, B_KC, SAFE, OutT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), 65536, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
}

// v2p: the BK=32 / 4-stage ring with SOFTWARE-PIPELINED fragments.  Ablation (EMO_GEMM_ABLATE=1: no tile loads) showed
// the plain ring's inner loop alone reaches only ~43 % MFMA utilisation: every K step does barrier -> 8 ds_read_b128 ->
// lgkmcnt(0) -> 16 MFMA, so the LDS latency is exposed once per step.  Here the fragments of stage k+1 are read into a
// second register set while the MFMAs of stage k run (the ring guarantees stage k+1 has landed one step earlier), at
// the price of one tile less in flight (2 instead of 3).
template <bool A_KC, bool B_KC, typename OutT>
__global__ __launch_bounds__(256) void gemm_bf16_glds_pf_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                                OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int64_t k_per_split,
                                                                EpiParams ep) {
    constexpr int BK = 32, ST = 4, OPB = 128 * BK * 2, STAGE = 2 * OPB, NI = BK / 16, LPT = 2 * NI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tiles_n = (N + GB_N - 1) / GB_N, tiles_m = (M + GB_M - 1) / GB_M;
    int64_t tm, tn, split = 0;
    if (ep.atomic) {
        splitk_coords(tiles_m, tiles_n, ep.atomic, tm, tn, split);
        if (ep.ws_stride) { C += split * ep.ws_stride; ep.atomic = 0; }
    }
    else tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * GB_M, n0 = tn * GB_N;
    const int64_t kbeg = split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int nk = (int)((kend - kbeg) / BK);
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t offA[NI], offB[NI];
    glds_offsets<A_KC, BK>(lda, m0, M, wave, lane, offA);
    glds_offsets<B_KC, BK>(ldb, n0, N, wave, lane, offB);
    const char* gA = (const char*)A + (A_KC ? kbeg : kbeg * lda) * 2;
    const char* gB = (const char*)B + (B_KC ? kbeg : kbeg * ldb) * 2;
    const int64_t stepA = (A_KC ? (int64_t)BK : (int64_t)BK * lda) * 2;
    const int64_t stepB = (B_KC ? (int64_t)BK : (int64_t)BK * ldb) * 2;
    int issued = 0;
    auto issue_next = [&]() {
        if (issued < nk) {
            char* st = smem + (issued % ST) * STAGE;
            glds_issue2<NI>(gA, offA, st, wave);
            glds_issue2<NI>(gB, offB, st + OPB, wave);
            gA += stepA;
            gB += stepB;
            ++issued;
        }
    };
    issue_next(); issue_next(); issue_next();
    // stage 0 must have landed before the first fragment read: at most (issued-1) younger tiles may stay in flight
    if (issued >= 3) wait_vmcnt<2 * LPT>(); else if (issued == 2) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa0[i] = lfrag2<A_KC, BK>(smem, wm * 64 + i * 16, 0, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb0[j] = lfrag2<B_KC, BK>(smem + OPB, wn * 64 + j * 16, 0, lane);

#define EMO_PF_STEP(CURA, CURB, NXTA, NXTB, KT)                                                                                   \
    {                                                                                                                             \
        const int kt_ = (KT);                                                                                                     \
        if (kt_ + 1 < nk) {                                                                                                       \
            /* stage kt+1 must have landed; only stage kt+2 may still be in flight */                                             \
            if (issued > kt_ + 2) wait_vmcnt<LPT>(); else wait_vmcnt<0>();                                                        \
            __builtin_amdgcn_s_barrier();                                                                                         \
            asm volatile("" ::: "memory");                                                                                        \
            issue_next();                                                                                                         \
            const char* la_ = smem + ((kt_ + 1) % ST) * STAGE;                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) NXTA[i] = lfrag2<A_KC, BK>(la_, wm * 64 + i * 16, 0, lane);             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) NXTB[j] = lfrag2<B_KC, BK>(la_ + OPB, wn * 64 + j * 16, 0, lane);       \
        }                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CURB[j], CURA[i], acc[i][j], 0, 0, 0);                        \
    }
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        EMO_PF_STEP(fa0, fb0, fa1, fb1, kt)
        EMO_PF_STEP(fa1, fb1, fa0, fb0, kt + 1)
    }
    if (kt < nk) EMO_PF_STEP(fa0, fb0, fa1, fb1, kt)
#undef EMO_PF_STEP
    epilogue_tile128<OutT>(ep, C, m0, n0, M, N, acc, smem, tid, wm, wn, lane);
}

template <bool A_KC, bool B_KC, typename OutT>
static void launch_glds_pf(dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t M, int64_t N,
                           int64_t K, int64_t kps, const EpiParams& ep) {
    auto kfn = gemm_bf16_glds_pf_kernel<A_KC, B_KC, OutT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), 65536, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
}

// v4: 128 x 256 block tile, 8 waves (2 x 4, wave tile 64 x 64 as in v2), BK = 32, 4-stage LDS-DMA ring (96 KB, one block
// = 2 waves per SIMD).  The K=512 forward GEMMs are bound by the CU's vector-memory front end (ablation: tile loads alone
// take 80 % of the kernel time at ~12-16 B/clk/CU); doubling BN cuts the tile bytes per FLOP by 25 % at unchanged
// MFMA / LDS-read structure.  Both operands K-contiguous or B MN-contiguous (same images as v2).
constexpr int G4_N = 256, G4_STAGE = 24576;   // per stage: 8 KB A + 16 KB B

template <bool B_KC, typename OutT>
__global__ __launch_bounds__(512) void gemm_bf16_g4_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                           OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, EpiParams ep) {
    constexpr int BK = 32, ST = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tiles_n = (N + G4_N - 1) / G4_N, tiles_m = (M + GB_M - 1) / GB_M;
    int64_t tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * GB_M, n0 = tn * G4_N;
    const int nk = (int)(K / BK);
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A tile: 8 wave-instructions (1 per wave); B tile: 16 (2 per wave)
    uint32_t offA, offB[2];
    {
        const int row = wave * 16 + (lane >> 2), c = (lane & 3) ^ swz32(row);
        int64_t gr = m0 + row;
        if (gr > M - 1) gr = M - 1;
        offA = (uint32_t)((gr * lda + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wave * 2 + i;
        if constexpr (B_KC) {
            const int row = j * 16 + (lane >> 2), c = (lane & 3) ^ swz32(row);
            int64_t gr = n0 + row;
            if (gr > N - 1) gr = N - 1;
            offB[i] = (uint32_t)((gr * ldb + c * 8) * 2);
        } else {
            // [32 k][256 rows] = 512 B per k-row: wave-instruction j covers k-rows 2j, 2j+1
            const int k = j * 2 + (lane >> 5), p16 = lane & 31;
            const int g = (p16 >> 1) ^ swz_k(k);
            int64_t gr = n0 + (g * 2 + (p16 & 1)) * 8;
            if (gr > N - 1) gr = ((N - 1) >> 3) << 3;
            offB[i] = (uint32_t)(((int64_t)k * ldb + gr) * 2);
        }
    }
    const char* gA = (const char*)A;
    const char* gB = (const char*)B;
    const int64_t stepA = (int64_t)BK * 2, stepB = (B_KC ? (int64_t)BK : (int64_t)BK * ldb) * 2;
    int issued = 0;
    auto issue_next = [&]() {
        if (issued < nk) {
            char* st = smem + (issued % ST) * G4_STAGE;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + offA),
                                             (__attribute__((address_space(3))) void*)(st + wave * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + offB[i]),
                                                 (__attribute__((address_space(3))) void*)(st + 8192 + (wave * 2 + i) * 1024), 16, 0, 0);
            gA += stepA;
            gB += stepB;
            ++issued;
        }
    };
    issue_next(); issue_next(); issue_next();
    for (int kt = 0; kt < nk; ++kt) {
        const int newer = issued - 1 - kt;          // tiles issued after kt
        if (newer >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_next();
        const char* la = smem + (kt % ST) * G4_STAGE;
        const char* lb = la + 8192;
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = lfrag2<true, BK>(la, wm * 64 + i * 16, 0, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (B_KC) fb[j] = lfrag2<true, BK>(lb, wn * 64 + j * 16, 0, lane);
            else {
                const int rbase = wn * 64 + j * 16, i2 = lane & 15, g = rbase >> 4;
                bf16x8 v;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = (lane >> 4) * 8 + h * 4 + (i2 >> 2);
                    const char* p = lb + k * 512 + ((g ^ swz_k(k)) << 5) + ((i2 & 3) << 3);
                    short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
                    bf16x4 tb = __builtin_bit_cast(bf16x4, t);
                    v[h * 4 + 0] = tb[0]; v[h * 4 + 1] = tb[1]; v[h * 4 + 2] = tb[2]; v[h * 4 + 3] = tb[3];
                }
                fb[j] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    // epilogue: two 128x128 halves (columns [0,128) = waves wn<2, [128,256) = wn>=2) through 64 KB of LDS each
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((wn >> 1) == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * 64 + i * 16 + (lane & 15);
                    const int chunk = ((wn & 1) * 64 + j * 16 + (lane >> 4) * 4) >> 2;
                    *(f32x4*)(smem + row * 512 + ((chunk ^ (row & 7)) << 4)) = acc[i][j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 512 * it;   // 0..2047 : row = idx >> 4, 8-column group = idx & 15
            const int row = idx >> 4, grp = idx & 15;
            const int64_t m = m0 + row, n = n0 + half * 128 + grp * 8;
            if (m < M && n < N) {
                const f32x4 lo = *(const f32x4*)(smem + row * 512 + (((2 * grp) ^ (row & 7)) << 4));
                const f32x4 hi = *(const f32x4*)(smem + row * 512 + (((2 * grp + 1) ^ (row & 7)) << 4));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                epi_row8<OutT>(ep, C, m, n, v, N);
            }
        }
    }
}

template <bool B_KC, typename OutT>
static void launch_g4(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t M, int64_t N, int64_t K,
                      const EpiParams& ep) {
    auto kfn = gemm_bf16_g4_kernel<B_KC, OutT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G4_STAGE);
        attr_set = true;
    }
    const int64_t tiles_m8 = cdiv64(cdiv64(M, GB_M), 8) * 8, tiles_n = cdiv64(N, G4_N);
    hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles_m8 * tiles_n)), dim3(512), 4 * G4_STAGE, st, A, lda, B, ldb, (OutT*)C, M, N, K, ep);
}

// v5: 256 x 128 block tile, FOUR waves (2 x 2, wave tile 128 x 64 = 8 x 4 MFMA tiles), BK = 32, 3-stage LDS-DMA ring
// (72 KB => 2 blocks per CU).  Ablation of the 128^2 kernel at K=512 (EMO_GEMM_ABLATE) showed four ADDITIVE costs of
// similar size — MFMA, fragment reads, tile DMA, and the per-tile fixed part (prologue + epilogue + C store burst);
// the bigger wave tile cuts fragment reads per MFMA by 25 %, the bigger block tile cuts DMA bytes per FLOP by 25 % and
// halves the number of per-tile fixed costs per FLOP.  A operand K-contiguous; B K-contiguous or MN-contiguous.
constexpr int G5_M = 256, G5_STAGE = 24576;   // per stage: 16 KB A + 8 KB B

template <bool B_KC, typename OutT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_g5_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                              OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, EpiParams ep) {
    constexpr int BK = 32, ST = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tiles_n = (N + GB_N - 1) / GB_N, tiles_m = (M + G5_M - 1) / G5_M;
    int64_t tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * G5_M, n0 = tn * GB_N;
    const int nk = (int)(K / BK);
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A tile [256 rows][32 k]: 16 wave-instructions (4 per wave); B tile [128][32]: 8 (2 per wave)
    uint32_t offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 16 + (lane >> 2), c = (lane & 3) ^ swz32(row);
        int64_t gr = m0 + row;
        if (gr > M - 1) gr = M - 1;
        offA[i] = (uint32_t)((gr * lda + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wave * 2 + i;
        if constexpr (B_KC) {
            const int row = j * 16 + (lane >> 2), c = (lane & 3) ^ swz32(row);
            int64_t gr = n0 + row;
            if (gr > N - 1) gr = N - 1;
            offB[i] = (uint32_t)((gr * ldb + c * 8) * 2);
        } else {
            const int k = j * 4 + (lane >> 4), p16 = lane & 15;
            const int g = (p16 >> 1) ^ swz_k(k);
            int64_t gr = n0 + (g * 2 + (p16 & 1)) * 8;
            if (gr > N - 1) gr = ((N - 1) >> 3) << 3;
            offB[i] = (uint32_t)(((int64_t)k * ldb + gr) * 2);
        }
    }
    const char* gA = (const char*)A;
    const char* gB = (const char*)B;
    const int64_t stepA = (int64_t)BK * 2, stepB = (B_KC ? (int64_t)BK : (int64_t)BK * ldb) * 2;
    int issued = 0;
    auto issue_next = [&]() {
        if (issued < nk) {
            char* st = smem + (issued % ST) * G5_STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + offA[i]),
                                                 (__attribute__((address_space(3))) void*)(st + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + offB[i]),
                                                 (__attribute__((address_space(3))) void*)(st + 16384 + (wave * 2 + i) * 1024), 16, 0, 0);
            gA += stepA;
            gB += stepB;
            ++issued;
        }
    };
    issue_next(); issue_next();
    for (int kt = 0; kt < nk; ++kt) {
        const int newer = issued - 1 - kt;          // tiles issued after kt (6 DMA instructions per tile per thread)
        if (newer >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_next();
        const char* la = smem + (kt % ST) * G5_STAGE;
        const char* lb = la + 16384;
        bf16x8 fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = lfrag2<B_KC, BK>(lb, wn * 64 + j * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = lfrag2<true, BK>(la, wm * 128 + i * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    // epilogue: four 64-row slabs through 32 KB of LDS (the ring is idle now)
#pragma clang loop unroll(full)
    for (int pass = 0; pass < 4; ++pass) {
        __syncthreads();
        if (wm == (pass >> 1)) {
#pragma clang loop unroll(full)
            for (int ii = 0; ii < 4; ++ii)
#pragma clang loop unroll(full)
                for (int j = 0; j < 4; ++j) {
                    const int row = ii * 16 + (lane & 15);
                    const int chunk = (wn * 64 + j * 16 + (lane >> 4) * 4) >> 2;
                    *(f32x4*)(smem + row * 512 + ((chunk ^ (row & 7)) << 4)) = acc[(pass & 1) * 4 + ii][j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + 256 * it;   // 0..1023 : row = idx >> 4 (64 rows), 8-column group = idx & 15
            const int row = idx >> 4, grp = idx & 15;
            const int64_t m = m0 + pass * 64 + row, n = n0 + grp * 8;
            if (m < M && n < N) {
                const f32x4 lo = *(const f32x4*)(smem + row * 512 + (((2 * grp) ^ (row & 7)) << 4));
                const f32x4 hi = *(const f32x4*)(smem + row * 512 + (((2 * grp + 1) ^ (row & 7)) << 4));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                epi_row8<OutT>(ep, C, m, n, v, N);
            }
        }
    }
}

template <bool B_KC, typename OutT>
static void launch_g5(hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t M, int64_t N, int64_t K,
                      const EpiParams& ep) {
    auto kfn = gemm_bf16_g5_kernel<B_KC, OutT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * G5_STAGE);
        attr_set = true;
    }
    const int64_t tiles_m8 = cdiv64(cdiv64(M, G5_M), 8) * 8, tiles_n = cdiv64(N, GB_N);
    hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles_m8 * tiles_n)), dim3(256), 3 * G5_STAGE, st, A, lda, B, ldb, (OutT*)C, M, N, K, ep);
}

static int g_glds_bk = -1;   // EMO_GEMM_BK=32|64 forces one LDS-DMA geometry (default: per-shape heuristic)
static int glds_bk() {
    if (g_glds_bk < 0) {
        const char* e = getenv("EMO_GEMM_BK");
        g_glds_bk = !e ? 0 : ((e[0] == '6') ? 64 : 32);
    }
    return g_glds_bk;
}
static bool g_glds_bk_forced() { return glds_bk() != 0; }

template <bool A_KC, bool B_KC, typename OutT, int BK, int ST>
static void launch_glds(dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t M, int64_t N,
                        int64_t K, int64_t kps, const EpiParams& ep) {
    auto kfn = gemm_bf16_glds_kernel<A_KC, B_KC, OutT, BK, ST>;
    constexpr int LDS = ST * 2 * 128 * BK * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), LDS, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
}

template <typename OutT, int BK, int ST>
static void dispatch_glds2(bool akc, bool bkc, dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C,
                           int64_t M, int64_t N, int64_t K, int64_t kps, const EpiParams& ep) {
    if (akc && bkc) launch_glds<true, true, OutT, BK, ST>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (akc && !bkc) launch_glds<true, false, OutT, BK, ST>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (!akc && bkc) launch_glds<false, true, OutT, BK, ST>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else launch_glds<false, false, OutT, BK, ST>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
}
template <typename OutT>
static void dispatch_glds(bool akc, bool bkc, dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C,
                          int64_t M, int64_t N, int64_t K, int64_t kps, const EpiParams& ep) {
    // measured r01 (tools/bench_gemm.py, MI355X): the per-tile fixed cost (prologue latency + epilogue, ~4-5 us) is what
    // limits the K=512 GEMMs, so OCCUPANCY wins over prefetch depth: a 3-stage ring of 32-deep tiles (48 KB LDS, 3 blocks
    // per CU) beats the 4-stage ring (64 KB, 2 blocks) by 13-16 % (qkv 575 vs 509, ffn1 659 vs 568 TFLOP/s); for the
    // long reduction (K=2048) the double buffer of full 128-B lines is best (842 vs 796); for MN-contiguous B (dgrad)
    // the 2-stage ring of 32-deep tiles (32 KB) is best.  EMO_GEMM_ST / EMO_GEMM_BK / EMO_GEMM_PF force one geometry.
    const bool can64 = (kps % 64) == 0 && (K % 64) == 0;
    static const int st_env = getenv("EMO_GEMM_ST") ? atoi(getenv("EMO_GEMM_ST")) : 0;
    static const bool use_pf = getenv("EMO_GEMM_PF") != nullptr;
    if (st_env == 4 || g_glds_bk_forced() || use_pf) {
        const int bk = g_glds_bk_forced() ? glds_bk() : 32;
        if (bk == 64 && can64) dispatch_glds2<OutT, 64, 2>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
        else if (use_pf && akc && bkc) launch_glds_pf<true, true, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
        else if (use_pf && akc && !bkc) launch_glds_pf<true, false, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
        else dispatch_glds2<OutT, 32, 4>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
        return;
    }
    if (st_env == 3) { dispatch_glds2<OutT, 32, 3>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep); return; }
    if (st_env == 2) { dispatch_glds2<OutT, 32, 2>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep); return; }
    if (bkc && K > 1024 && can64) dispatch_glds2<OutT, 64, 2>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (bkc) dispatch_glds2<OutT, 32, 3>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else dispatch_glds2<OutT, 32, 2>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
}

static int g_gemm_variant = -1;   // EMO_GEMM_VARIANT: 1 = register-staged v1, 2 = LDS-DMA ring (default when eligible)
static int gemm_variant() {
    if (g_gemm_variant < 0) {
        const char* e = getenv("EMO_GEMM_VARIANT");
        g_gemm_variant = (e && e[0] >= '1' && e[0] <= '3') ? (e[0] - '0') : 2;
    }
    return g_gemm_variant;
}

template <typename OutT>
static void launch_bf16_tn32(dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t M, int64_t N,
                             int64_t K, int64_t kps, const EpiParams& ep) {
    auto kfn = gemm_bf16_kernel<false, false, false, OutT, 32>;
    hipLaunchKernelGGL(kfn, grid, dim3(256), 32768, st, A, lda, B, ldb, (OutT*)C, M, N, K, kps, ep);
}

template <bool SAFE, typename OutT>
static void dispatch_bf16(bool akc, bool bkc, dim3 grid, hipStream_t st, const bf16_t* A, int64_t lda, const bf16_t* B,
                          int64_t ldb, void* C, int64_t M, int64_t N, int64_t K, int64_t kps, const EpiParams& ep) {
    if (akc && bkc) launch_bf16<true, true, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (akc && !bkc) launch_bf16<true, false, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (!akc && bkc) launch_bf16<false, true, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else launch_bf16<false, false, SAFE, OutT>(grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
}

// ------------------------------------------------------------------------------------------------ split-K
// out (+)= sum over splits of the partial results written by the GEMM (deterministic: fixed summation order, no atomics)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int64_t stride, int splits, float* __restrict__ out, int64_t n4,
                                                            int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    // four independent partial sums: up to 32 loads per thread, 4 in flight instead of a dependent chain (fixed order: still deterministic)
    f32x4 a = accumulate ? ((const f32x4*)out)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 b = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
    const float* p = ws + 4 * i;
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
        const f32x4 x0 = *(const f32x4*)(p + (int64_t)s * stride), x1 = *(const f32x4*)(p + (int64_t)(s + 1) * stride);
        const f32x4 x2 = *(const f32x4*)(p + (int64_t)(s + 2) * stride), x3 = *(const f32x4*)(p + (int64_t)(s + 3) * stride);
        a += x0; b += x1; c += x2; d += x3;
    }
    for (; s < splits; ++s) a += *(const f32x4*)(p + (int64_t)s * stride);
    ((f32x4*)out)[i] = (a + b) + (c + d);
}

// number of K-splits for a plain fp32-output GEMM (wgrad).  max_ws_splits > 0: partials go to a caller workspace (plain stores +
// splitk_reduce_kernel); 0: fp32 atomics into C.
static int64_t choose_splits(int64_t M, int64_t N, int64_t K, bool big, bool has_epi, int dtype_out, int64_t BMt, int64_t BNt, int64_t BKt,
                             int64_t max_ws_splits) {
    const int64_t tiles_m = cdiv64(M, BMt), tiles_n = cdiv64(N, BNt);
    int64_t splits = 1;
    if (dtype_out == EMO_F32 && !has_epi && tiles_m * tiles_n < 256 && K >= 8 * BKt) {
        const int64_t max_splits = K / (4 * BKt);
        if (big) {
            // Cost model fitted to the r01 sweep (tools/bench_wgrad_splits.py, M = 8k/32k/131k tokens): a 128^2 x 64 K-step takes ~1.1 us
            // with one block per CU and ~1.4 us with two; blocks run in rounds of 512 (2 per CU); every split pays ~0.18 us per output
            // tile for its fp32 atomics (device-scope atomics from 8 XCDs resolve beyond the L2s: ~0.3 TB/s), or ~0.03 us per tile for
            // plain stores + its share of the reduce pass when a workspace is available.
            static const int cand[] = {1, 2, 4, 8, 16, 24, 32};
            const double tiles = (double)(tiles_m * tiles_n), ksteps = (double)cdiv64(K, BKt);
            const double per_split = max_ws_splits > 0 ? 0.03 : 0.18;
            double best = 1e30;
            for (int c : cand) {
                if (c > 1 && (c > max_splits || (max_ws_splits > 0 && c > max_ws_splits))) break;
                const double blocks = tiles * c, rounds = (double)cdiv64((int64_t)blocks, 512);
                const double t = rounds * (double)cdiv64((int64_t)ksteps, c) * (blocks > 256 ? 1.4 : 1.1) + c * tiles * per_split;
                if (t < best) { best = t; splits = c; }
            }
        } else {
            splits = cdiv64(512, tiles_m * tiles_n);
            if (splits > max_splits) splits = max_splits;
            if (splits >= 4) {                       // one K-split per XCD round (splitk_coords): keep the 8 XCDs evenly loaded
                const int64_t r8 = cdiv64(splits, 8) * 8;
                splits = r8 <= max_splits ? r8 : ((splits / 8) * 8 > 0 ? (splits / 8) * 8 : splits);
            }
            if (max_ws_splits > 0 && splits > max_ws_splits) splits = max_ws_splits;
        }
        if (splits < 1) splits = 1;
        { const char* fs = getenv("EMO_GEMM_SPLITS"); if (fs && atoi(fs) > 0) splits = atoi(fs) <= max_splits ? atoi(fs) : max_splits; }   // tuning sweeps
    }
    return splits;
}

#define EMO_GEMM_MAX_SPLITS 32
// v6 (256^2 four-phase kernel) dispatch.  OFF by default: in microbenchmarks it beats the 128^2 kernels on the long reductions (FFN wgrad
// 680-705 -> 800-820 TFLOP/s, 8192^3 NT 906 -> 1155), but INSIDE the training step the 128^2 wgrad runs at ~890 TFLOP/s (its dY operand was
// just written and is still on-die) against 864 for v6, and the K = 2048 forward is equal (663 vs 667): 65.6 vs 65.5 ms/step.
// EMO_GEMM_G6=1 forces it for every eligible shape (K % 64 == 0, M and N multiples of 256), =2 enables it for the >= 12-tile wgrads only.
static int g6_mode() {
    static int m = -1;
    if (m < 0) { const char* e = getenv("EMO_GEMM_G6"); m = e ? atoi(e) + 1 : 1; }   // 1 = off (default), 2 = forced, 3 = wgrad heuristic
    return m;
}
static bool g6_shape_ok(int64_t M, int64_t N, int64_t K) { return (M % G3_M) == 0 && (N % G3_N) == 0 && (K % G3_K) == 0 && K >= 2 * G3_K; }
// split count of the v6 wgrad: all tiles of a split run on ONE XCD (32 CUs, one 128-KB block per CU), so a split costs `tiles` CUs
static int64_t g6_wgrad_splits(int64_t M, int64_t N, int64_t K, int64_t max_ws_splits) {
    const int64_t tiles = (M / G3_M) * (N / G3_N);
    int64_t per_xcd = tiles >= 32 ? 1 : 32 / tiles;
    int64_t splits = 8 * per_xcd;
    const int64_t max_by_k = K / (8 * G3_K);
    while (splits > 8 && splits > max_by_k) splits -= 8;
    if (splits > max_by_k) splits = max_by_k > 0 ? max_by_k : 1;
    if (max_ws_splits > 0 && splits > max_ws_splits) splits = max_ws_splits >= 8 ? (max_ws_splits / 8) * 8 : max_ws_splits;
    { const char* fs = getenv("EMO_GEMM_SPLITS"); if (fs && atoi(fs) > 0) splits = atoi(fs); }
    return splits < 1 ? 1 : splits;
}
extern "C" int64_t emo_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype_in, int dtype_out) {
    if (dtype_out != EMO_F32 || M <= 0 || N <= 0 || K <= 0) return 0;
    const bool big = dtype_in == EMO_BF16;
    const int64_t BMt = big ? GB_M : 64, BNt = big ? GB_N : 64, BKt = big ? (gemm_variant() >= 2 ? G2_BK : GB_K) : 16;
    int64_t splits = choose_splits(M, N, K, big, false, dtype_out, BMt, BNt, BKt, EMO_GEMM_MAX_SPLITS);
    if (big && g6_mode() != 1 && g6_shape_ok(M, N, K) && (g6_mode() == 2 ? K >= 4096 : (K >= 32768 && (M / G3_M) * (N / G3_N) >= 12))) {   // layout unknown here: size for both kernels
        const int64_t s6 = g6_wgrad_splits(M, N, K, 0);
        if (s6 > splits) splits = s6;
    }
    return splits > 1 ? splits * M * N * (int64_t)sizeof(float) : 0;
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
    { const char* ab = getenv("EMO_GEMM_ABLATE"); ep.ablate = ab ? atoi(ab) : 0; }
    bool has_epi = false;
    if (e) {
        ep.bias = e->bias; ep.act = e->act; ep.aux_out = e->aux_out; ep.mul_aux = e->mul_aux;
        ep.mul_mode = e->mul_aux ? e->mul_mode : EMO_MUL_NONE; ep.mul_scale = e->mul_scale;
        ep.drop = make_drop(e->p_drop, e->seed, e->offset); ep.residual = e->residual;
        has_epi = e->bias || e->act || e->aux_out || ep.mul_mode || ep.drop.thr16 || e->residual;
        ep.a_rowsum = e->a_rowsum; ep.b_rowsum = e->b_rowsum; ep.mask_out = e->mask_out;
        ep.ln_c1 = e->ln_c1; ep.ln_stats_out = e->ln_stats_out; ep.ln_eps = e->ln_eps;
        ep.rln_x = e->rln_x; ep.rln_stats = e->rln_stats; ep.rln_gamma = e->rln_gamma; ep.rln_beta = e->rln_beta;
        EMO_CHECK(!e->rln_x || (e->rln_stats && e->rln_gamma && e->rln_beta && !e->act && !ep.drop.thr16 && !ep.mul_mode),
                  "emo_gemm: rln_x needs rln_stats/gamma/beta and no activation / dropout / mul epilogue");
        EMO_CHECK(!e->ln_stats_out || e->ln_c1, "emo_gemm: ln_stats_out needs ln_c1");
    }
    const bool ln_fused = e && (e->ln_c1 || e->rln_x);
    EMO_CHECK(!(ep.a_rowsum && ep.b_rowsum), "emo_gemm: a_rowsum and b_rowsum are exclusive");
    bool use_g6 = false;
    if (dtype_in == EMO_BF16 && gemm_variant() >= 2 && !use_safe_tr() && g6_mode() != 1 && g6_shape_ok(M, N, K) && !ln_fused && M > 32 &&
        (lda & 7) == 0 && (ldb & 7) == 0 && getenv("EMO_GEMM_FORCE_G3") == nullptr && getenv("EMO_GEMM_TN32") == nullptr) {
        // measured r01 (tools/bench_g6.py, 131072 tokens): FFN wgrads 725 -> 840-890 TFLOP/s, fused-QKV wgrad 548 -> 581, the 4-tile
        // 512 x 512 wgrad 575 -> 500 (64 splits), K = 1536 / 2048 single-pass GEMMs within noise of the 128^2 kernels -> wgrad only
        const bool wgrad = a_trans && b_trans && dtype_out == EMO_F32 && !has_epi && ldc == N && (M / G3_M) * (N / G3_N) >= 12 && K >= 32768;
        use_g6 = g6_mode() == 2 || (g6_mode() == 3 && wgrad);
    }
    if (ep.b_rowsum) {
        EMO_CHECK(a_trans && b_trans, "emo_gemm: b_rowsum needs a_trans and b_trans (B stored [K, N]: the Conv1D wgrad layout)");
        const bool in_kernel = dtype_in == EMO_BF16 && dtype_out == EMO_F32 && gemm_variant() < 3 && !use_safe_tr() && getenv("EMO_GEMM_TN32") == nullptr &&
                               getenv("EMO_GEMM_FORCE_G3") == nullptr && getenv("EMO_GEMM_NO_ROWSUM_FUSE") == nullptr;   // (v6 carries it too: a_trans && b_trans)
        if (!in_kernel) {
            const int rc = emo_colsum(B, dtype_in, K, N, ldb, ep.b_rowsum, 1, stream);
            if (rc) return rc;
            ep.b_rowsum = nullptr;
        }
    }
    if (ep.a_rowsum) {
        EMO_CHECK(a_trans, "emo_gemm: a_rowsum needs a_trans (A stored [K, M]: the wgrad layout)");
        const bool in_kernel = dtype_in == EMO_BF16 && dtype_out == EMO_F32 && b_trans && gemm_variant() < 3 && !use_safe_tr() &&
                               getenv("EMO_GEMM_TN32") == nullptr && getenv("EMO_GEMM_FORCE_G3") == nullptr && getenv("EMO_GEMM_NO_ROWSUM_FUSE") == nullptr;
        if (!in_kernel) {                                   // other kernels: the plain column-sum launch over A [K, M]
            const int rc = emo_colsum(A, dtype_in, K, M, lda, ep.a_rowsum, 1, stream);
            if (rc) return rc;
            ep.a_rowsum = nullptr;
        }
    }
    const bool big = dtype_in == EMO_BF16;
    const int variant = big ? gemm_variant() : 0;
    if (big && M <= 32 && !a_trans && !b_trans && (K % 32) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 &&
        !accumulate && getenv("EMO_GEMM_NO_SKINNY") == nullptr) {
        dim3 g((unsigned)cdiv64(N, 16));
        const int nw = (K >= 2048 && getenv("EMO_SKINNY_NW8") == nullptr) ? 16 : (K >= 1024 ? 8 : 4);   // 16 waves x two 64-wide chunks at K = 2048
#define SKINNY_LAUNCH(OutT, NWv, CWv)                                                                                                       \
    do {                                                                                                                                    \
        constexpr size_t lds_ = (size_t)NWv * 48 * (CWv + 16) * 2 + sizeof(f32x4) * (NWv - 1) * 2 * 64 + sizeof(float) * NWv * 2 * 16 * 2;   \
        auto kfn = gemm_bf16_skinny_kernel<OutT, NWv, CWv>;                                                                                 \
        static bool attr_ = false;                                                                                                          \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); attr_ = true; }   \
        hipLaunchKernelGGL(kfn, g, dim3(64 * NWv), lds_, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (OutT*)C, M, N, K, ep);          \
    } while (0)
        if (dtype_out == EMO_F32) { if (nw == 16) SKINNY_LAUNCH(float, 16, 64); else if (nw == 8) SKINNY_LAUNCH(float, 8, 128); else SKINNY_LAUNCH(float, 4, 128); }
        else { if (nw == 16) SKINNY_LAUNCH(bf16_t, 16, 64); else if (nw == 8) SKINNY_LAUNCH(bf16_t, 8, 128); else SKINNY_LAUNCH(bf16_t, 4, 128); }
#undef SKINNY_LAUNCH
        EMO_LAUNCH_CHECK();
        return EMO_OK;
    }
    if (big && !a_trans && !b_trans && !ln_fused && !use_safe_tr() &&
        emo_gemm_astat_try((const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, dtype_out, M, N, K, ep, st)) {   // K = 512, A stationary in registers
        EMO_LAUNCH_CHECK();
        return EMO_OK;
    }
    EMO_CHECK(!ep.mask_out && ep.mul_mode != EMO_MUL_BITMASK, "emo_gemm: mask_out / EMO_MUL_BITMASK need bf16 in/out, NT, K = 512, M %% 128 == 0, N %% 64 == 0, N <= 2048");
    EMO_CHECK(!ln_fused, "emo_gemm: the LayerNorm-folded epilogue (ln_c1 / rln_x) exists only on the skinny path (bf16, M <= 32, NT, K %% 32 == 0)");
    // v3 (256^2 tile) when the problem fills the chip with 256^2 tiles and N does not waste a tile
    bool use_g3 = false;
    if (big && variant >= 2 && !use_safe_tr() && (K % G3_K) == 0 && M >= 256) {
        const int64_t t3 = cdiv64(M, G3_M) * cdiv64(N, G3_N);
        const bool n_ok = (N % G3_N) == 0 || N >= 4 * G3_N;
        const bool enough = t3 >= 512 || (dtype_out == EMO_F32 && !has_epi && K >= 64 * G3_K);   // split-K refills the grid
        // measured r01: with 1 block/CU the exposed prologue/epilogue makes the 256^2 kernel SLOWER than the 128^2 ones at
        // K=512 (293 vs 505 TFLOP/s) and no faster at K=2048 -> kept for experiments only (EMO_GEMM_FORCE_G3=1)
        use_g3 = n_ok && enough && false;
    }
    if (big && variant >= 2 && !use_safe_tr() && (K % G3_K) == 0 && M >= 8 && N >= 8 && getenv("EMO_GEMM_FORCE_G3") != nullptr) use_g3 = true;
    if (use_g6) use_g3 = true;                                // same 256 x 256 x 64 tile grid; the kernel is chosen at the launch below
    const int64_t BMt = big ? (use_g3 ? G3_M : GB_M) : 64, BNt = big ? (use_g3 ? G3_N : GB_N) : 64;
    const int64_t BKt = big ? (use_g3 ? G3_K : (variant >= 2 ? G2_BK : GB_K)) : 16;
    const int64_t tiles_m = cdiv64(M, BMt), tiles_n = cdiv64(N, BNt);
    const int64_t tiles_m8 = cdiv64(tiles_m, 8) * 8;
    // split-K: only for plain fp32 outputs (wgrad) when the tile grid cannot fill the chip
    void* ws = e ? e->workspace : nullptr;
    const int64_t ws_bytes = e ? e->workspace_bytes : 0;
    const bool ws_ok = ws && ldc == N && ((M * N) & 3) == 0 && ((uintptr_t)ws & 15) == 0 && getenv("EMO_GEMM_SPLIT_ATOMIC") == nullptr;
    int64_t max_ws_splits = ws_ok ? ws_bytes / (M * N * (int64_t)sizeof(float)) : 0;
    int64_t splits = choose_splits(M, N, K, big, has_epi, dtype_out, BMt, BNt, BKt, max_ws_splits >= 2 ? max_ws_splits : 0);
    if (use_g6) splits = (a_trans && b_trans && dtype_out == EMO_F32 && !has_epi && K >= 4096 && ldc == N) ? g6_wgrad_splits(M, N, K, max_ws_splits >= 2 ? max_ws_splits : 0) : 1;
    if (ldc != N) splits = 1;                                // split-K partials need a contiguous C: a strided output view runs unsplit
    int64_t kps = cdiv64(cdiv64(K, splits), BKt) * BKt;
    splits = cdiv64(K, kps);
    const bool use_ws = splits > 1 && max_ws_splits >= splits;
    const int accumulate_final = accumulate;
    if (splits > 1) {
        EMO_CHECK(ldc == N, "emo_gemm: split-K needs contiguous C");
        if (use_ws) {
            ep.ws_stride = M * N;
            ep.accumulate = 0;
        } else if (!accumulate) {
            hipError_t me = hipMemsetAsync(C, 0, (size_t)(M * N) * sizeof(float), st);
            EMO_CHECK(me == hipSuccess, "emo_gemm: memset failed");
        }
        ep.atomic = (splits == 2 || splits == 4) ? (int)(8 / splits) : 1;
    }
    void* const C_final = C;
    if (use_ws) C = ws;
    dim3 grid((unsigned)(tiles_m8 * tiles_n), 1, 1);
    if (splits > 1)   // see splitk_coords()
        grid.x = ep.atomic > 1 ? (unsigned)(8 * cdiv64(tiles_m * tiles_n, ep.atomic)) : (unsigned)(tiles_m * tiles_n * (cdiv64(splits, 8) * 8));
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
        const int64_t rowsA = a_trans ? (use_g3 ? G3_K : G2_BK) : M, rowsB = b_trans ? (use_g3 ? G3_K : G2_BK) : N;
        const int64_t spanA = (rowsA * lda + (a_trans ? M : 0)) * 2, spanB = (rowsB * ldb + (b_trans ? N : 0)) * 2;
        const bool span_ok = spanA < (int64_t)0xFFFF0000 && spanB < (int64_t)0xFFFF0000;
        const bool glds_ok = !safe && variant >= 2 && (akc || variant >= 3) && (K % G2_BK) == 0 && (kps % G2_BK) == 0 && M >= 8 && N >= 8 && span_ok;
        static const bool g4_on = getenv("EMO_GEMM_G4") != nullptr;   // 128x256 tile: measured no faster than 128x128 (r01) -> opt-in
        const bool use_g4 = g4_on && !safe && variant >= 2 && akc && splits == 1 && !accumulate && (K % G2_BK) == 0 && (N % G4_N) == 0 && span_ok &&
                            cdiv64(M, GB_M) * (N / G4_N) >= 512 && K <= 1024;
        static const bool g5_on = getenv("EMO_GEMM_G5") != nullptr;
        const bool use_g5 = g5_on && !safe && variant >= 2 && akc && splits == 1 && !accumulate && (K % G2_BK) == 0 && span_ok &&
                            cdiv64(M, G5_M) * cdiv64(N, GB_N) >= 512;
        if (use_g5) {
            if (dtype_out == EMO_F32) { if (bkc) launch_g5<true, float>(st, a, lda, b, ldb, C, M, N, K, ep); else launch_g5<false, float>(st, a, lda, b, ldb, C, M, N, K, ep); }
            else { if (bkc) launch_g5<true, bf16_t>(st, a, lda, b, ldb, C, M, N, K, ep); else launch_g5<false, bf16_t>(st, a, lda, b, ldb, C, M, N, K, ep); }
        } else if (use_g4) {
            if (dtype_out == EMO_F32) { if (bkc) launch_g4<true, float>(st, a, lda, b, ldb, C, M, N, K, ep); else launch_g4<false, float>(st, a, lda, b, ldb, C, M, N, K, ep); }
            else { if (bkc) launch_g4<true, bf16_t>(st, a, lda, b, ldb, C, M, N, K, ep); else launch_g4<false, bf16_t>(st, a, lda, b, ldb, C, M, N, K, ep); }
        } else if (use_g6 && span_ok && (kps % G3_K) == 0) {
            const int rs = ep.a_rowsum ? 1 : (ep.b_rowsum ? 2 : 0);
            if (dtype_out == EMO_F32) dispatch_g6<float>(akc, bkc, rs, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_g6<bf16_t>(akc, bkc, rs, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else if (use_g3 && span_ok && (kps % G3_K) == 0) {
            if (dtype_out == EMO_F32) dispatch_g3<float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_g3<bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else if (use_g3) {
            emo_set_error("emo_gemm: internal tile-selection error");
            return EMO_ERR_INVALID;
        } else if (glds_ok) {
            if (dtype_out == EMO_F32) dispatch_glds<float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_glds<bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else if (!akc && !bkc && !safe && getenv("EMO_GEMM_TN32") != nullptr && dtype_out == EMO_F32) {
            launch_bf16_tn32<float>(grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else if (dtype_out == EMO_F32) {   // register-staged v1 (same 128^2 grid; any K, predicated edges)
            if (safe) dispatch_bf16<true, float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_bf16<false, float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else {
            if (safe) dispatch_bf16<true, bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_bf16<false, bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        }
    }
    EMO_LAUNCH_CHECK();
    if (use_ws) {
        const int64_t n4 = (M * N) >> 2;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, st, (const float*)ws, M * N, (int)splits, (float*)C_final, n4,
                           accumulate_final);
        EMO_LAUNCH_CHECK();
    }
    return EMO_OK;
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): HBM-bound stream.  A thread owns 16 B of columns (8 bf16 / 4 f32), a wave a
// 1-KiB row segment, the block's 4 waves take interleaved rows (4 rows in flight per thread); partials are
// combined in LDS and flushed with one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t M, int64_t N, int64_t ld,
                                                     float* __restrict__ out, int64_t rows_per_block) {
    constexpr int VE = 16 / sizeof(T);
    __shared__ float part[4][64 * VE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 64 * VE + lane * VE;
    const int64_t mbeg = (int64_t)blockIdx.y * rows_per_block;
    int64_t mend = mbeg + rows_per_block;
    if (mend > M) mend = M;
    float s[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s[e] = 0.f;
    if (c0 + VE <= N && (ld % VE) == 0 && (((uintptr_t)X) & 15) == 0) {
        int64_t m = mbeg + wave;
        for (; m + 12 < mend; m += 16) {
            T v[4][VE];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (sizeof(T) == 2) *(bf16x8*)v[u] = *(const bf16x8*)(X + (m + 4 * u) * ld + c0);
                else *(f32x4*)v[u] = *(const f32x4*)(X + (m + 4 * u) * ld + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VE; ++e) s[e] += to_f32<T>(v[u][e]);
        }
        for (; m < mend; m += 4) {
            T v[VE];
            if constexpr (sizeof(T) == 2) *(bf16x8*)v = *(const bf16x8*)(X + m * ld + c0);
            else *(f32x4*)v = *(const f32x4*)(X + m * ld + c0);
#pragma unroll
            for (int e = 0; e < VE; ++e) s[e] += to_f32<T>(v[e]);
        }
    } else {
        for (int64_t m = mbeg + wave; m < mend; m += 4)
#pragma unroll
            for (int e = 0; e < VE; ++e)
                if (c0 + e < N) s[e] += to_f32<T>(X[m * ld + c0 + e]);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) part[wave][lane * VE + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * VE; c += 256) {
        const int64_t n = (int64_t)blockIdx.x * 64 * VE + c;
        if (n < N) atomicAdd(out + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
    }
}

extern "C" int emo_colsum(const void* X, int dtype, int64_t M, int64_t N, int64_t ld, float* out, int accumulate,
                          emo_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    EMO_CHECK(X && out && M > 0 && N > 0, "emo_colsum: bad args");
    if (!accumulate) {
        hipError_t me = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), st);
        EMO_CHECK(me == hipSuccess, "emo_colsum: memset failed");
    }
    const int64_t cb = cdiv64(N, dtype == EMO_F32 ? 256 : 512);
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
