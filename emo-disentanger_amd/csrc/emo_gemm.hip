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

// ------------------------------------------------------------------------------------------------
// f32 parity kernel, 128 x 128 x 16 tile (r06).  The 64 x 64 kernel above loads element-wise with arbitrary strides, single-buffered, two workgroup
// barriers per 16-deep tile, four MFMAs per wave and k-step: 25 TFLOP/s = 0.16 of the 157-TFLOP/s exact-f32 MFMA peak, and 93 % of the fp32 parity
// mode's training step (profiles/r06_fp32_step_rocprof_stats.txt).  Same instruction (v_mfma_f32_16x16x4_f32: an fp32 FMA chain over k in ascending
// order — a dot product sums in the SAME order as in the kernel above, so unsplit results are bit-identical), but: a wave owns 64 x 64 outputs (16
// MFMAs per two times four 4-byte operand reads), operand tiles arrive by 16-byte loads along whichever dimension is contiguous (KC: k-contiguous
// rows -> LDS image [row][20]; otherwise m/n-contiguous -> [k][132]; both 16-B aligned row pitches), staged through registers so that the next
// tile's loads are in flight during the MFMAs, two LDS buffers, ONE barrier per tile.  Edges (rows beyond M / N, a K range that is not a multiple
// of 4, unaligned bases or strides) take element-wise predicated loads.  Shapes: the launcher sends M >= 128 and N >= 128 here.
constexpr int F3_BM = 128, F3_BN = 128, F3_BK = 16, F3_LDK = 20, F3_LDM = 132, F3_TILE = 128 * F3_LDK;   // 2560 floats >= 16 * 132 = 2112
template <bool KC>
__device__ __forceinline__ void f3_gload(const float* __restrict__ P, int64_t s_row, int64_t s_k, int64_t row0, int64_t nrows, int64_t k0, int64_t kend,
                                         bool vec_ok, int tid, f32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (KC) {                                   // chunk = 4 consecutive k of one row: row = idx / 4, chunk = idx % 4
            const int64_t gr = row0 + (idx >> 2), gk = k0 + (idx & 3) * 4;
            if (gr < nrows && gk < kend) {
                const float* q = P + gr * s_row + gk;
                if (vec_ok && gk + 3 < kend) v = *(const f32x4*)q;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gk + e < kend) v[e] = q[e];
                }
            }
        } else {                                              // chunk = 4 consecutive rows at one k: k = idx / 32, chunk = idx % 32
            const int64_t gk = k0 + (idx >> 5), gr = row0 + (idx & 31) * 4;
            if (gk < kend && gr < nrows) {
                const float* q = P + gk * s_k + gr;
                if (vec_ok && gr + 3 < nrows) v = *(const f32x4*)q;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gr + e < nrows) v[e] = q[e];
                }
            }
        }
        r[i] = v;
    }
}
template <bool KC>
__device__ __forceinline__ void f3_lstore(float* lds, int tid, const f32x4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        if constexpr (KC) *(f32x4*)(lds + (idx >> 2) * F3_LDK + (idx & 3) * 4) = r[i];
        else *(f32x4*)(lds + (idx >> 5) * F3_LDM + (idx & 31) * 4) = r[i];
    }
}
template <bool KC>
__device__ __forceinline__ float f3_frag(const float* lds, int row, int k) { return KC ? lds[row * F3_LDK + k] : lds[k * F3_LDM + row]; }

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_t128_kernel(const float* __restrict__ A, int64_t sam, int64_t sak, const float* __restrict__ B, int64_t sbn,
                                                               int64_t sbk, float* __restrict__ C, int64_t M, int64_t N, int64_t K, int64_t k_per_split,
                                                               EpiParams ep, int a_vec, int b_vec) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][F3_TILE];           // [buffer][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tiles_n = (N + F3_BN - 1) / F3_BN, tiles_m = (M + F3_BM - 1) / F3_BM;
    int64_t tm, tn, split = 0;
    if (ep.atomic) {
        splitk_coords(tiles_m, tiles_n, ep.atomic, tm, tn, split);
        if (ep.ws_stride) { C += split * ep.ws_stride; ep.atomic = 0; }
    } else tile_coords(tiles_m, tiles_n, tm, tn);
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * F3_BM, n0 = tn * F3_BN;
    const int64_t kbeg = split * k_per_split;
    const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
    if (kbeg >= kend) return;
    const int wm = wave >> 1, wn = wave & 1;                  // 2 x 2 waves, 64 x 64 each
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[2], rb[2];
    f3_gload<A_KC>(A, sam, sak, m0, M, kbeg, kend, a_vec != 0, tid, ra);
    f3_gload<B_KC>(B, sbn, sbk, n0, N, kbeg, kend, b_vec != 0, tid, rb);
    f3_lstore<A_KC>(lds[0][0], tid, ra);
    f3_lstore<B_KC>(lds[0][1], tid, rb);
    __syncthreads();
    const int nk = (int)((kend - kbeg + F3_BK - 1) / F3_BK);
    const int r16 = lane & 15, kq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {                                           // next tile's loads in flight during this tile's MFMAs
            const int64_t k0 = kbeg + (int64_t)(kt + 1) * F3_BK;
            f3_gload<A_KC>(A, sam, sak, m0, M, k0, kend, a_vec != 0, tid, ra);
            f3_gload<B_KC>(B, sbn, sbk, n0, N, k0, kend, b_vec != 0, tid, rb);
        }
        const float* As = lds[cur][0];
        const float* Bs = lds[cur][1];
#pragma unroll
        for (int ks = 0; ks < F3_BK; ks += 4) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = f3_frag<A_KC>(As, wm * 64 + i * 16 + r16, ks + kq);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = f3_frag<B_KC>(Bs, wn * 64 + j * 16 + r16, ks + kq);
            // swapped operands: D'[n][m] so that a lane owns 4 consecutive n
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            f3_lstore<A_KC>(lds[cur ^ 1][0], tid, ra);
            f3_lstore<B_KC>(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + wm * 64 + i * 16 + r16;
            const int64_t n = n0 + wn * 64 + j * 16 + kq * 4;
            if (m < M && n < N) epi_store4_call<float>(ep, C, m, n, acc[i][j], N);
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

    // Two K tiles are in flight in registers (r02: with ONE, every K step ended in `s_waitcnt vmcnt(0)` on loads issued at its own start —
    // PMC: MFMA pipe 32 % busy, LDS 32 %, no bank conflicts, waves waiting 37 % of their cycles): tile t+2 is requested at the start of
    // step t, tile t+1 (requested a whole step earlier) goes to the other LDS buffer at the end of step t.  The loop is unrolled by two so
    // that both register sets and both LDS buffers are addressed statically.
    bf16x8 ra0[BKT / 16], rb0[BKT / 16], ra1[BKT / 16], rb1[BKT / 16];
    int rs_n = (int)((kbeg / BKT) % tiles_n), rs_m = (int)((kbeg / BKT) % tiles_m);   // which block of the tile row / column owns this K step's rowsum
    auto compute = [&](const char* la) {
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
        if (++rs_n == (int)tiles_n) rs_n = 0;
        if (++rs_m == (int)tiles_m) rs_m = 0;
    };
    char* const s0 = smem;
    char* const s1 = smem + 2 * OPB;
    gload_tile<A_KC, BKT>(A, lda, m0, M, kbeg, kend, tid, ra0);
    gload_tile<B_KC, BKT>(B, ldb, n0, N, kbeg, kend, tid, rb0);
    lstore_tile<A_KC, BKT>(s0, tid, ra0);
    lstore_tile<B_KC, BKT>(s0 + OPB, tid, rb0);
    if (kbeg + BKT < kend) {
        gload_tile<A_KC, BKT>(A, lda, m0, M, kbeg + BKT, kend, tid, ra1);
        gload_tile<B_KC, BKT>(B, ldb, n0, N, kbeg + BKT, kend, tid, rb1);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += 2 * BKT) {
        // step t: LDS buffer 0 is current, register set 1 holds tile t+1
        if (k0 + 2 * BKT < kend) {
            gload_tile<A_KC, BKT>(A, lda, m0, M, k0 + 2 * BKT, kend, tid, ra0);
            gload_tile<B_KC, BKT>(B, ldb, n0, N, k0 + 2 * BKT, kend, tid, rb0);
        }
        compute(s0);
        if (k0 + BKT >= kend) break;
        lstore_tile<A_KC, BKT>(s1, tid, ra1);
        lstore_tile<B_KC, BKT>(s1 + OPB, tid, rb1);
        __syncthreads();
        // step t+1: LDS buffer 1 is current, register set 0 holds tile t+2
        if (k0 + 3 * BKT < kend) {
            gload_tile<A_KC, BKT>(A, lda, m0, M, k0 + 3 * BKT, kend, tid, ra1);
            gload_tile<B_KC, BKT>(B, ldb, n0, N, k0 + 3 * BKT, kend, tid, rb1);
        }
        compute(s1);
        if (k0 + 2 * BKT >= kend) break;
        lstore_tile<A_KC, BKT>(s0, tid, ra0);
        lstore_tile<B_KC, BKT>(s0 + OPB, tid, rb0);
        __syncthreads();
    }
    __syncthreads();                                   // the epilogue stages through the same LDS
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
    // Epilogue operands of the decode step (bias, LN fold vector, residual rows / the raw rows + statistics the residual LayerNorm is rebuilt
    // from) are requested FIRST, before the operand tiles: the kernel is a chain of dependent round trips (rocprofv3 r02: 5.4 us per call for
    // 0.4 us of weight traffic), and loading them only in the epilogue added one more trip at the end of the chain.
    const int64_t pn = n0 + (lane >> 4) * 4;
    const bool fast_epi = wave == 0 && !ep.drop.thr16 && ep.mul_mode == EMO_MUL_NONE && !ep.aux_out && !ep.atomic && !ep.accumulate && (ep.ldc & 3) == 0 &&
                          pn + 3 < N && ((lane & 15) + 16) < M;
    f32x4 p_bias = {0.f, 0.f, 0.f, 0.f}, p_c1 = {0.f, 0.f, 0.f, 0.f}, p_gam = {0.f, 0.f, 0.f, 0.f}, p_bet = {0.f, 0.f, 0.f, 0.f};
    float p_res[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, p_st[2][2] = {{0.f, 1.f}, {0.f, 1.f}};
    if (fast_epi) {
        if (ep.bias) p_bias = *(const f32x4*)(ep.bias + pn);
        if (ln) p_c1 = *(const f32x4*)(ep.ln_c1 + pn);
        const void* rsrc = ep.rln_x ? ep.rln_x : ep.residual;
        if (rsrc) {
            Out4<OutT>::load((const OutT*)rsrc + (lane & 15) * ep.ldc + pn, p_res[0]);
            Out4<OutT>::load((const OutT*)rsrc + ((lane & 15) + 16) * ep.ldc + pn, p_res[1]);
        }
        if (ep.rln_x) {
            p_gam = *(const f32x4*)(ep.rln_gamma + pn);
            p_bet = *(const f32x4*)(ep.rln_beta + pn);
            p_st[0][0] = ep.rln_stats[(lane & 15) * 2]; p_st[0][1] = ep.rln_stats[(lane & 15) * 2 + 1];
            p_st[1][0] = ep.rln_stats[((lane & 15) + 16) * 2]; p_st[1][1] = ep.rln_stats[((lane & 15) + 16) * 2 + 1];
        }
    }
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
        s0 = rows4_sum(s0); q0 = rows4_sum(q0); s1 = rows4_sum(s1); q1 = rows4_sum(q1);    // lane swaps, not 8 LDS shuffles (decode chain)
        if (lane < 16) { st[wave][0][lane][0] = s0; st[wave][0][lane][1] = q0; st[wave][1][lane][0] = s1; st[wave][1][lane][1] = q1; }
    }
    if (wave > 0) { red[wave - 1][0][lane] = acc0; red[wave - 1][1][lane] = acc1; }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) { acc0 += red[w][0][lane]; acc1 += red[w][1][lane]; }
    const int64_t n = n0 + (lane >> 4) * 4;
    const int64_t m0 = lane & 15;
    float mean0 = 0.f, mean1 = 0.f, rstd0 = 1.f, rstd1 = 1.f;
    if (ln) {
        const int r = lane & 15;
        const float invK = 1.f / (float)K;
        float su = 0.f, sq = 0.f, tu = 0.f, tq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { su += st[w][0][r][0]; sq += st[w][0][r][1]; tu += st[w][1][r][0]; tq += st[w][1][r][1]; }
        mean0 = su * invK; mean1 = tu * invK;
        rstd0 = rsqrtf(fmaxf(sq * invK - mean0 * mean0, 0.f) + ep.ln_eps); rstd1 = rsqrtf(fmaxf(tq * invK - mean1 * mean1, 0.f) + ep.ln_eps);
        if (ep.ln_stats_out && blockIdx.x == 0 && lane < 16) {
            if (m0 < M) { ep.ln_stats_out[m0 * 2] = mean0; ep.ln_stats_out[m0 * 2 + 1] = rstd0; }
            if (m0 + 16 < M) { ep.ln_stats_out[(m0 + 16) * 2] = mean1; ep.ln_stats_out[(m0 + 16) * 2 + 1] = rstd1; }
        }
    }
    if (fast_epi) {                                          // every operand is already in registers: arithmetic + two stores
        float v[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[0][i] = ln ? rstd0 * (acc0[i] - mean0 * p_c1[i]) : acc0[i];
            v[1][i] = ln ? rstd1 * (acc1[i] - mean1 * p_c1[i]) : acc1[i];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = v[h][i];
                if (ep.rln_x) x += (p_res[h][i] - p_st[h][0]) * p_st[h][1] * p_gam[i] + p_bet[i];     // (same order as the general path: LN'd residual, bias, act)
                x += p_bias[i];
                if (ep.act == EMO_ACT_RELU) x = fmaxf(x, 0.f);
                else if (ep.act == EMO_ACT_GELU_NEW) x = gelu_new_o<OutT>(x);
                else if (ep.act == EMO_ACT_GELU) x = gelu_erf_f(x);
                if (ep.residual) x += p_res[h][i];
                v[h][i] = x;
            }
            Out4<OutT>::store(C + (m0 + 16 * h) * ep.ldc + n, v[h]);
        }
        return;
    }
    if (ln) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float c1 = (n + i < N) ? ep.ln_c1[n + i] : 0.f;
            acc0[i] = rstd0 * (acc0[i] - mean0 * c1);
            acc1[i] = rstd1 * (acc1[i] - mean1 * c1);
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
    // measured r01 (tools/bench_gemm.py of that round — git history — MI355X): the per-tile fixed cost (prologue latency + epilogue, ~4-5 us) is what
    // limits the K=512 GEMMs, so OCCUPANCY wins over prefetch depth: a 3-stage ring of 32-deep tiles (48 KB LDS, 3 blocks
    // per CU) beats the 4-stage ring (64 KB, 2 blocks) by 13-16 % (qkv 575 vs 509, ffn1 659 vs 568 TFLOP/s); for the
    // long reduction (K=2048) the double buffer of full 128-B lines is best (842 vs 796); for MN-contiguous B (dgrad)
    // the 2-stage ring of 32-deep tiles (32 KB, 4 blocks per CU) is best (in-step A/B of the other geometries: DESIGN.md §4.1).
    const bool can64 = (kps % 64) == 0 && (K % 64) == 0;
    // Small grids (at most two tiles per CU: the reference YAML's batch size 4, stage 1): occupancy has nothing to offer, a block is alone on
    // its CU and every K step exposed an L2 / HBM round trip -> the 4-stage ring of 32-deep tiles (three tiles in flight inside the block)
    static const bool small_off = getenv("EMO_GEMM_SMALLGRID") != nullptr && atoi(getenv("EMO_GEMM_SMALLGRID")) == 0;
    // (r03, same box: batch-4 step 8.85 -> 8.79 ms, stage 1 7.64 -> 7.47 ms; a 64-deep double buffer instead was erratic: 8.5 / 9.2 ms)
    if (grid.x <= 512 && !small_off && K >= 128) {
        // 64-deep tiles in the same 4-stage ring = twice the bytes in flight per block (128 KB of LDS, one block per CU): what bounds these products is
        // the bytes one block keeps in flight (r04, same box, two pairs: stage 1 262 / 266 -> 284 / 283 k tokens/s, batch-size-4 steps unchanged);
        // EMO_GEMM_SMALL64=0: the 32-deep ring
        static const bool small64 = !(getenv("EMO_GEMM_SMALL64") != nullptr && atoi(getenv("EMO_GEMM_SMALL64")) == 0);
        if (small64 && can64 && kps >= 512) { dispatch_glds2<OutT, 64, 4>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep); return; }
        dispatch_glds2<OutT, 32, 4>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
        return;
    }
    if (bkc && K > 1024 && can64) dispatch_glds2<OutT, 64, 2>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else if (bkc) dispatch_glds2<OutT, 32, 3>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
    else dispatch_glds2<OutT, 32, 2>(akc, bkc, grid, st, A, lda, B, ldb, C, M, N, K, kps, ep);
}

static int g_gemm_variant = -1;   // EMO_GEMM_VARIANT: 1 = register-staged v1, 2 = LDS-DMA ring for k-contiguous A, 3 = also for the TN (wgrad) layout; unset = per call
static bool g_gemm_variant_auto = true;
static int gemm_variant() {
    if (g_gemm_variant < 0) {
        const char* e = getenv("EMO_GEMM_VARIANT");
        g_gemm_variant_auto = !(e && e[0] >= '1' && e[0] <= '3');
        g_gemm_variant = g_gemm_variant_auto ? 2 : (e[0] - '0');
    }
    return g_gemm_variant;
}
// Per call (r05): the weight-gradient layout (A stored [K, M]) on the LDS-DMA kernel when the reduction is short — stage 1 (2048 tokens per step:
// 48 weight gradients on the register-staged kernel) 6.35 -> 6.13 ms per step; at 8192 tokens it measures the same, at 131072 it loses 1.5 ms
// (those shapes have the 256 x 256 wgrad kernel; what reaches this point then is better off with the in-kernel bias gradient of v1).
static int gemm_variant_for(int a_trans, int64_t K) {
    const int v = gemm_variant();
    return (g_gemm_variant_auto && a_trans && K <= 4096) ? 3 : v;
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

// The same sum followed by the FULL fused epilogue (bias -> aux -> act -> mul -> dropout -> residual, emo_gemm_epi.h) and the store in the output
// dtype: split-K for products WITH an epilogue whose tile grid leaves the chip empty (r04: stage-1 / batch-4 dgrads and FFN2 forwards with
// 64-256 tiles ran at the ~25-32 GB/s one block's 4-stage DMA ring can keep in flight; two to four K-splits put 2 blocks on every CU).
// A thread owns 8 consecutive columns of a row (N % 8 == 0).
template <typename OutT>
__global__ __launch_bounds__(256) void splitk_reduce_epi_kernel(const float* __restrict__ ws, int64_t stride, int splits, OutT* __restrict__ C, int64_t M,
                                                                int64_t N, EpiParams ep) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i * 8 >= M * N) return;
    const float* p = ws + 8 * i;
    f32x4 a0 = *(const f32x4*)p, a1 = *(const f32x4*)(p + 4);
    for (int s = 1; s < splits; ++s) {
        a0 += *(const f32x4*)(p + (int64_t)s * stride);
        a1 += *(const f32x4*)(p + (int64_t)s * stride + 4);
    }
    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const int64_t m = (8 * i) / N, n = (8 * i) - m * N;
    epi_row8<OutT>(ep, C, m, n, v, N);
}

// K-splits of a bf16-output product with an epilogue (0 / 1 = none): a tile grid of at most 256 blocks and a long reduction
static int64_t choose_epi_splits(int64_t M, int64_t N, int64_t K) {
    const char* e = getenv("EMO_GEMM_EPI_SPLIT");                     // (read per call: tests toggle it in-process)
    if ((e && atoi(e) == 0) || (N & 7) || (K % (4 * G2_BK)) != 0) return 1;
    const int64_t tiles = cdiv64(M, GB_M) * cdiv64(N, GB_N);
    // measured r04 (same-box A/B script of that round, rocprofv3 per kernel; profiles/r04_ab_epi_split*): stage 1's K = 2048 dgrads on 64 tiles 48 -> 22 + 6 us with 4 splits; at 256 tiles
    // (batch-size-4 stage-2 dgrads) 2 splits only tie (52 -> 45 + 15 us: the 32 MB of fp32 partials cost what the second resident block gains)
    if (tiles > 128) return 1;
    // (K = 512 products on <= 64 tiles, 4 splits: 18.6 -> 13.4 + 5 us per launch, the step's kernel time unchanged at 89.9 vs 89.8 ms: not split)
    return K >= 1024 ? 4 : 1;
}

// the same pass with a tail of blocks that adds up the bias-gradient partials of the 256 x 256 weight-gradient kernel: rs_out[i] += sum over
// rs_parts vectors of length rs_stride (fixed order)
__global__ __launch_bounds__(256) void splitk_reduce_rs_kernel(const float* __restrict__ ws, int64_t stride, int splits, float* __restrict__ out, int64_t n4,
                                                               int accumulate, const float* __restrict__ rs_ws, int64_t rs_stride, int rs_parts,
                                                               float* __restrict__ rs_out, int64_t main_blocks) {
    if ((int64_t)blockIdx.x >= main_blocks) {
        const int64_t i = ((int64_t)blockIdx.x - main_blocks) * 256 + threadIdx.x;
        if (i * 4 >= rs_stride) return;
        f32x4 a = ((const f32x4*)rs_out)[i], b = {0.f, 0.f, 0.f, 0.f};
        const float* p = rs_ws + 4 * i;
        int s = 0;
        for (; s + 2 <= rs_parts; s += 2) { a += *(const f32x4*)(p + (int64_t)s * rs_stride); b += *(const f32x4*)(p + (int64_t)(s + 1) * rs_stride); }
        if (s < rs_parts) a += *(const f32x4*)(p + (int64_t)s * rs_stride);
        ((f32x4*)rs_out)[i] = a + b;
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
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
void emo_splitk_reduce_rs_launch(const float* ws, int64_t stride, int splits, float* out, int64_t n4, int accumulate, const float* rs_ws, int64_t rs_stride,
                                 int rs_parts, float* rs_out, hipStream_t st) {
    const int64_t mb = cdiv64(n4, 256), rb = cdiv64(rs_stride >> 2, 256);
    hipLaunchKernelGGL(splitk_reduce_rs_kernel, dim3((unsigned)(mb + rb)), dim3(256), 0, st, ws, stride, splits, out, n4, accumulate, rs_ws, rs_stride, rs_parts,
                       rs_out, mb);
}

void emo_splitk_reduce_launch(const float* ws, int64_t stride, int splits, float* out, int64_t n4, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n4, 256)), dim3(256), 0, st, ws, stride, splits, out, n4, accumulate);
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
            // Cost model fitted to the r01 sweep (tools/bench_wgrad_splits.py of that round — git history —, M = 8k/32k/131k tokens): a 128^2 x 64 K-step takes ~1.1 us
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

// which kernel family the calling thread's last emo_gemm went to (diagnostics: bench.py attributes a launch to the kernel that RAN, not to the
// one its shape suggests): 1 skinny (M <= 32), 2 A-stationary K = 512, 3 256 x 256 tile wgrad, 4 256 x 256 tile NT, 5 128 x 128 LDS-DMA ring,
// 6 128 x 128 register-staged, 7 exact-fp32; + 16 split-K through the workspace, + 32 split-K with the reduce-and-epilogue pass
static thread_local int g_last_gemm_kernel = 0;
extern "C" int emo_epilogue_size(void) { return (int)sizeof(emo_epilogue_t); }
extern "C" int emo_gemm_last_kernel(void) { return g_last_gemm_kernel; }

// fp32 products: tile edge of the kernel that serves the shape (a pure function of the shape: the split-K plan and the workspace size follow it).
// EMO_GEMM_F32_TILE=64 keeps every shape on the 64 x 64 kernel (A/B, tests).
static int64_t f32_tile(int64_t M, int64_t N) {
    const char* e = getenv("EMO_GEMM_F32_TILE");                 // (read per call: tests toggle it in-process)
    const bool small_only = e != nullptr && atoi(e) == 64;
    return (!small_only && M >= 128 && N >= 128) ? 128 : 64;
}

#define EMO_GEMM_MAX_SPLITS 32
extern "C" int64_t emo_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int dtype_in, int dtype_out) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (dtype_in == EMO_BF16 && dtype_out == EMO_BF16) {               // split-K + reduce-with-epilogue of the small-grid long-K products
        const int64_t es = choose_epi_splits(M, N, K);
        return es > 1 ? es * M * N * (int64_t)sizeof(float) : 0;
    }
    if (dtype_out != EMO_F32) return 0;
    const bool big = dtype_in == EMO_BF16;
    const int64_t BMt = big ? GB_M : f32_tile(M, N), BNt = big ? GB_N : f32_tile(M, N), BKt = big ? (gemm_variant() >= 2 ? G2_BK : GB_K) : 16;
    int64_t splits = choose_splits(M, N, K, big, false, dtype_out, BMt, BNt, BKt, EMO_GEMM_MAX_SPLITS);
    int64_t rs_floats = 0;
    if (big) {                                                  // (layout unknown here: sized for the wgrad kernel too, with its bias-gradient partials)
        const int64_t s2 = emo_gemm_w128_tn_splits(M, N, K);
        if (s2 > splits) splits = s2;
        if (s2 > 0) rs_floats = emo_gemm_w128_tn_rs_floats(M, N, s2);
    }
    return splits > 1 ? (splits * M * N + rs_floats) * (int64_t)sizeof(float) : 0;
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
#ifdef EMO_DIAG
    { const char* ab = getenv("EMO_GEMM_ABLATE"); ep.ablate = ab ? atoi(ab) : 0; }
#endif
    bool has_epi = false;
    if (e) {
        ep.bias = e->bias; ep.act = e->act; ep.aux_out = e->aux_out; ep.mul_aux = e->mul_aux;
        ep.mul_mode = e->mul_aux ? e->mul_mode : EMO_MUL_NONE; ep.mul_scale = e->mul_scale;
        ep.drop = make_drop(e->p_drop, e->seed, e->offset); ep.residual = e->residual;
        has_epi = e->bias || e->act || e->aux_out || ep.mul_mode || ep.drop.thr16 || e->residual;
        ep.a_rowsum = e->a_rowsum; ep.b_rowsum = e->b_rowsum; ep.mask_out = e->mask_out;
        ep.ln_c1 = e->ln_c1; ep.ln_stats_out = e->ln_stats_out; ep.ln_eps = e->ln_eps;
        ep.rln_x = e->rln_x; ep.rln_stats = e->rln_stats; ep.rln_gamma = e->rln_gamma; ep.rln_beta = e->rln_beta;
        ep.lna_gamma = e->lna_gamma; ep.lna_beta = e->lna_beta; ep.lna_out = e->lna_out; ep.lna_mean = e->lna_mean; ep.lna_rstd = e->lna_rstd;
        EMO_CHECK(!e->lna_gamma || (e->lna_beta && e->lna_out && e->lna_mean && e->lna_rstd && !a_trans && !b_trans && dtype_in == EMO_BF16 && !e->ln_c1 && !e->rln_x),
                  "emo_gemm: lna_gamma needs lna_beta / lna_out / lna_mean / lna_rstd, bf16, NT, and no ln_c1 / rln_x");
        EMO_CHECK(!e->lna_gamma || (K == 512 && (M % 128) == 0 && M >= 4096 && (N % 64) == 0 && N <= 2048 && !accumulate && (((uintptr_t)e->lna_out) & 15) == 0),
                  "emo_gemm: lna_* (LayerNorm of the A operand) exists only on the A-stationary kernel: K = 512, M %% 128 == 0, M >= 4096, N %% 64 == 0, N <= 2048");
        ep.hdiv = e->hdiv; ep.hdiv_T = e->hdiv_T;
        EMO_CHECK(!e->hdiv || (!has_epi && !e->lna_gamma && !e->mask_out && !e->a_rowsum && !e->b_rowsum && !e->ln_c1 && !e->rln_x && !a_trans && !b_trans && !accumulate &&
                               dtype_in == EMO_BF16 && dtype_out == EMO_BF16),
                  "emo_gemm: hdiv needs a plain epilogue, bf16 in / out, NT");
        EMO_CHECK(!e->hdiv || (K == 512 && (M % 128) == 0 && M >= 4096 && (N % 64) == 0 && N <= 2048 && e->hdiv_T > 0 && (e->hdiv_T % 32) == 0 && (M % e->hdiv_T) == 0 &&
                               (((uintptr_t)e->hdiv) & 3) == 0),
                  "emo_gemm: hdiv exists only on the A-stationary kernel (K = 512, M %% 128 == 0, M >= 4096, N %% 64 == 0, N <= 2048) with hdiv_T %% 32 == 0 and M %% hdiv_T == 0");
        EMO_CHECK(!e->rln_x || (e->rln_stats && e->rln_gamma && e->rln_beta && !e->act && !ep.drop.thr16 && !ep.mul_mode),
                  "emo_gemm: rln_x needs rln_stats/gamma/beta and no activation / dropout / mul epilogue");
        EMO_CHECK(!e->ln_stats_out || e->ln_c1, "emo_gemm: ln_stats_out needs ln_c1");
    }
    const bool ln_fused = e && (e->ln_c1 || e->rln_x);
    EMO_CHECK(!(ep.a_rowsum && ep.b_rowsum), "emo_gemm: a_rowsum and b_rowsum are exclusive");
    if (ep.b_rowsum) {
        EMO_CHECK(a_trans && b_trans, "emo_gemm: b_rowsum needs a_trans and b_trans (B stored [K, N]: the Conv1D wgrad layout)");
        const bool in_kernel = dtype_in == EMO_BF16 && dtype_out == EMO_F32 && gemm_variant_for(a_trans, K) < 3 && !use_safe_tr();
        if (!in_kernel) {
            const int rc = emo_colsum(B, dtype_in, K, N, ldb, ep.b_rowsum, 1, stream);
            if (rc) return rc;
            ep.b_rowsum = nullptr;
        }
    }
    if (ep.a_rowsum) {
        EMO_CHECK(a_trans, "emo_gemm: a_rowsum needs a_trans (A stored [K, M]: the wgrad layout)");
        const bool in_kernel = dtype_in == EMO_BF16 && dtype_out == EMO_F32 && b_trans && gemm_variant_for(a_trans, K) < 3 && !use_safe_tr();
        if (!in_kernel) {                                   // other kernels: the plain column-sum launch over A [K, M]
            const int rc = emo_colsum(A, dtype_in, K, M, lda, ep.a_rowsum, 1, stream);
            if (rc) return rc;
            ep.a_rowsum = nullptr;
        }
    }
    const bool big = dtype_in == EMO_BF16;
    const int variant = big ? gemm_variant_for(a_trans, K) : 0;
    if (big && M <= 32 && !a_trans && !b_trans && (K % 32) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 &&
        !accumulate) {
        dim3 g((unsigned)cdiv64(N, 16));
        const int nw = K >= 2048 ? 16 : (K >= 1024 ? 8 : 4);   // 16 waves x two 64-wide chunks at K = 2048
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
        g_last_gemm_kernel = 1;
        return EMO_OK;
    }
#ifdef EMO_EXPERIMENTAL
    if (big && !a_trans && !b_trans && !ln_fused && !accumulate && !use_safe_tr() && !ep.lna_gamma && !ep.hdiv) {
        const int pk = emo_gemm_p256_try((const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, dtype_out, M, N, K, ep, st);   // opt-in persistent tile walks (r05)
        if (pk) {
            EMO_LAUNCH_CHECK();
            g_last_gemm_kernel = pk;                              // 8: 256 x 256 tile, 9: 128 x 512 tile
            return EMO_OK;
        }
    }
#endif
    if (big && !a_trans && !b_trans && !ln_fused && !use_safe_tr() &&
        emo_gemm_astat_try((const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, dtype_out, M, N, K, ep, st)) {   // K = 512, A stationary in registers
        EMO_LAUNCH_CHECK();
        g_last_gemm_kernel = 2;
        return EMO_OK;
    }
    if (big && a_trans && b_trans && dtype_out == EMO_F32 && !has_epi && !ln_fused && !use_safe_tr() && e &&
        emo_gemm_w128_tn_try((const bf16_t*)A, lda, (const bf16_t*)B, ldb, (float*)C, ldc, M, N, K, accumulate, ep.a_rowsum, ep.b_rowsum, e->workspace,
                             e->workspace_bytes, st)) {
        EMO_LAUNCH_CHECK();
        g_last_gemm_kernel = 3;
        return EMO_OK;
    }
    EMO_CHECK(!ep.lna_gamma, "emo_gemm: lna_* (LayerNorm of the A operand) exists only on the A-stationary kernel: bf16, NT, K = 512, M %% 128 == 0, M >= 4096, N %% 64 == 0, N <= 2048");
    EMO_CHECK(!ep.hdiv, "emo_gemm: hdiv exists only on the A-stationary kernel and this call was refused by it (alignment / strides / EMO_GEMM_NO_ASTAT)");
    if (big && !a_trans && !b_trans && !ln_fused && !accumulate &&
        emo_gemm_w128_try((const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, dtype_out, M, N, K, ep, st)) {    // long-K NT products on 256 x 256 tiles (opt-in)
        EMO_LAUNCH_CHECK();
        g_last_gemm_kernel = 4;
        return EMO_OK;
    }
    EMO_CHECK(!ep.mask_out && ep.mul_mode != EMO_MUL_BITMASK, "emo_gemm: mask_out / EMO_MUL_BITMASK need bf16 in/out, NT, K = 512, M %% 128 == 0, M >= 4096, N %% 64 == 0, N <= 2048");
    EMO_CHECK(!ln_fused, "emo_gemm: the LayerNorm-folded epilogue (ln_c1 / rln_x) exists only on the skinny path (bf16, M <= 32, NT, K %% 32 == 0)");
    const int64_t BMt = big ? GB_M : f32_tile(M, N), BNt = big ? GB_N : f32_tile(M, N);
    const int64_t BKt = big ? (variant >= 2 ? G2_BK : GB_K) : 16;
    const int64_t tiles_m = cdiv64(M, BMt), tiles_n = cdiv64(N, BNt);
    const int64_t tiles_m8 = cdiv64(tiles_m, 8) * 8;
    // split-K: only for plain fp32 outputs (wgrad) when the tile grid cannot fill the chip
    void* ws = e ? e->workspace : nullptr;
    const int64_t ws_bytes = e ? e->workspace_bytes : 0;
    const bool ws_ok = ws && ldc == N && ((M * N) & 3) == 0 && ((uintptr_t)ws & 15) == 0 && getenv("EMO_GEMM_SPLIT_ATOMIC") == nullptr;
    int64_t max_ws_splits = ws_ok ? ws_bytes / (M * N * (int64_t)sizeof(float)) : 0;
    int64_t splits = choose_splits(M, N, K, big, has_epi, dtype_out, BMt, BNt, BKt, max_ws_splits >= 2 ? max_ws_splits : 0);
    if (ldc != N) splits = 1;                                // split-K partials need a contiguous C: a strided output view runs unsplit
    // bf16 outputs (with or without an epilogue) on a small tile grid: split-K through the workspace + splitk_reduce_epi_kernel
    EpiParams ep_final = ep;
    bool epi_split = false;
    if (big && dtype_out == EMO_BF16 && !accumulate && ldc == N && ws_ok && variant >= 2 && !use_safe_tr() && (!a_trans || variant >= 3) && !ep.a_rowsum && !ep.b_rowsum) {
        const int64_t es = choose_epi_splits(M, N, K);
        if (es > 1 && max_ws_splits >= es) {
            splits = es;
            epi_split = true;
            ep.bias = nullptr; ep.act = EMO_ACT_NONE; ep.aux_out = nullptr; ep.mul_aux = nullptr; ep.mul_mode = EMO_MUL_NONE; ep.mul_scale = 1.f;
            ep.drop = make_drop(0.f, 0, 0); ep.residual = nullptr;
        }
    }
    const int kdt = epi_split ? EMO_F32 : dtype_out;         // the GEMM pass of an epilogue split writes raw fp32 partial sums
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
    g_last_gemm_kernel = (dtype_in == EMO_F32 ? 7 : 6) + (use_ws ? (epi_split ? 32 : 16) : 0);
    if (dtype_in == EMO_F32) {
        const float* a = (const float*)A;
        const float* b = (const float*)B;
        const int64_t sam = a_trans ? 1 : lda, sak = a_trans ? lda : 1;
        const int64_t sbn = b_trans ? 1 : ldb, sbk = b_trans ? ldb : 1;
        // 128 x 128 tile kernel when one of each operand's two strides is 1 (always, for the layouts emo_gemm takes); vector loads need 16-B aligned
        // bases and non-unit strides that are multiples of 4 elements
        const bool akc = sak == 1, bkc = sbk == 1;
        if (BMt == 128 && (akc || sam == 1) && (bkc || sbn == 1)) {
            const int av = (((uintptr_t)a & 15) == 0 && ((akc ? sam : sak) & 3) == 0 && (kps & 3) == 0) ? 1 : 0;
            const int bv = (((uintptr_t)b & 15) == 0 && ((bkc ? sbn : sbk) & 3) == 0 && (kps & 3) == 0) ? 1 : 0;
#define F3_LAUNCH(AKv, BKv) hipLaunchKernelGGL((gemm_f32_t128_kernel<AKv, BKv>), grid, dim3(256), 0, st, a, sam, sak, b, sbn, sbk, (float*)C, M, N, K, kps, ep, av, bv)
            if (akc && bkc) F3_LAUNCH(true, true); else if (akc) F3_LAUNCH(true, false); else if (bkc) F3_LAUNCH(false, true); else F3_LAUNCH(false, false);
#undef F3_LAUNCH
        } else {
            EMO_CHECK(BMt == 64, "emo_gemm(fp32): operand with two non-unit strides on the 128-tile plan");
            hipLaunchKernelGGL(gemm_f32_kernel<float>, grid, dim3(256), 0, st, a, sam, sak, b, sbn, sbk, (float*)C, M, N, K, kps, ep);
        }
    } else {
        EMO_CHECK((lda & 7) == 0 && (ldb & 7) == 0, "emo_gemm(bf16): lda/ldb must be multiples of 8 (16-B rows)");
        EMO_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "emo_gemm(bf16): A/B must be 16-B aligned");
        const bf16_t* a = (const bf16_t*)A;
        const bf16_t* b = (const bf16_t*)B;
        const bool akc = !a_trans, bkc = !b_trans;
        const bool safe = use_safe_tr();
        const int64_t rowsA = a_trans ? G2_BK : M, rowsB = b_trans ? G2_BK : N;
        const int64_t spanA = (rowsA * lda + (a_trans ? M : 0)) * 2, spanB = (rowsB * ldb + (b_trans ? N : 0)) * 2;
        const bool span_ok = spanA < (int64_t)0xFFFF0000 && spanB < (int64_t)0xFFFF0000;
        const bool glds_ok = !safe && variant >= 2 && (akc || variant >= 3) && (K % G2_BK) == 0 && (kps % G2_BK) == 0 && M >= 8 && N >= 8 && span_ok;
        if (glds_ok) {
            g_last_gemm_kernel += 5 - 6;
            if (kdt == EMO_F32) dispatch_glds<float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_glds<bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else if (kdt == EMO_F32) {   // register-staged v1 (same 128^2 grid; any K, predicated edges)
            if (safe) dispatch_bf16<true, float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_bf16<false, float>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        } else {
            if (safe) dispatch_bf16<true, bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
            else dispatch_bf16<false, bf16_t>(akc, bkc, grid, st, a, lda, b, ldb, C, M, N, K, kps, ep);
        }
    }
    EMO_LAUNCH_CHECK();
    if (use_ws && epi_split) {
        ep_final.atomic = 0; ep_final.ws_stride = 0; ep_final.accumulate = 0; ep_final.ldc = N;
        hipLaunchKernelGGL(splitk_reduce_epi_kernel<bf16_t>, dim3((unsigned)cdiv64(M * N / 8, 256)), dim3(256), 0, st, (const float*)ws, M * N, (int)splits,
                           (bf16_t*)C_final, M, N, ep_final);
        EMO_LAUNCH_CHECK();
    } else if (use_ws) {
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
