// LDS-image MFMA helpers shared by the attention kernels (emo_favor.hip, emo_softmax_attn.hip).
// An "image" is an LDS matrix [rows][ld] whose reduction index k is contiguous; every contraction
// is acc(16x16) += R . C^T over two images (bf16: ds_read_b128 + v_mfma_f32_16x16x32_bf16;
// exact-f32 mode: ds_read_b32 + v_mfma_f32_16x16x4_f32).
#pragma once
#include "emo_common.h"

template <typename CT> struct Img;
template <> struct Img<bf16_t> {
    static constexpr int KMIN = 32, PAD = 8, KSTEP = 32;
    typedef bf16x8 V;
    static __device__ __forceinline__ V load(const bf16_t* img, int ld, int row0, int k, int lane) {
        return *(const bf16x8*)(img + (row0 + (lane & 15)) * ld + k + (lane >> 4) * 8);
    }
    static __device__ __forceinline__ f32x4 mma(V r, V c, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(r, c, acc, 0, 0, 0); }
    static __device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
        bf16x4 t = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
        *(bf16x4*)p = t;
    }
    static __device__ __forceinline__ float ex(float x) { return __expf(x); }
};
template <> struct Img<float> {
    static constexpr int KMIN = 4, PAD = 2, KSTEP = 4;
    typedef float V;
    static __device__ __forceinline__ V load(const float* img, int ld, int row0, int k, int lane) {
        return img[(row0 + (lane & 15)) * ld + k + (lane >> 4)];
    }
    static __device__ __forceinline__ f32x4 mma(V r, V c, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x4f32(r, c, acc, 0, 0, 0); }
    static __device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
    static __device__ __forceinline__ float ex(float x) { return expf(x); }
};

// acc(16x16) += R[rrow0.., :K] . Cc[crow0.., :K]^T.  Lane l, reg r owns (R-row = rrow0 + (l>>4)*4 + r, C-row = crow0 + (l&15)).
// K is a compile-time constant: ALL operand fragments of the product are requested first and the MFMA chain runs afterwards, so the
// LDS round trip is paid once per product (and overlaps across the independent products of a phase) instead of once per k step — with
// the run-time k loop every step was read -> wait -> MFMA in sequence (r02 estimate from the rocprofv3 times: ~11 % MFMA issue occupancy
// in the FAVOR+ backward kernels, ~2300 cycles per barrier-delimited phase for ~900 cycles of MFMA issue per chunk).
template <typename CT, int K>
__device__ __forceinline__ void mm16(f32x4& acc, const CT* R, int ldr, int rrow0, const CT* Cc, int ldc, int crow0, int lane) {
    constexpr int NS = K / Img<CT>::KSTEP;
    typename Img<CT>::V a[NS], b[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        a[s] = Img<CT>::load(R, ldr, rrow0, s * Img<CT>::KSTEP, lane);
        b[s] = Img<CT>::load(Cc, ldc, crow0, s * Img<CT>::KSTEP, lane);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = Img<CT>::mma(a[s], b[s], acc);
}

// The same product with ONE operand's fragments already in registers: the tiles a wave owns inside a phase share an operand row block
// (tile = wave + FW * i with FW a multiple of the tile-grid width), so that operand is read from LDS once per phase instead of once per
// tile — the FAVOR+ kernels are LDS-bound (r01 PMC: LDS array busy 64-86 % of the kernel time) and 28-40 % of their fragment reads
// were such repeats.
template <typename CT, int K> struct Frags {
    static constexpr int NS = K / Img<CT>::KSTEP;
    typename Img<CT>::V f[NS];
    __device__ __forceinline__ void load(const CT* img, int ld, int row0, int lane) {
#pragma unroll
        for (int s = 0; s < NS; ++s) f[s] = Img<CT>::load(img, ld, row0, s * Img<CT>::KSTEP, lane);
    }
};
template <typename CT, int K>   // second operand preloaded
__device__ __forceinline__ void mm16_c(f32x4& acc, const CT* R, int ldr, int rrow0, const Frags<CT, K>& c, int lane) {
    Frags<CT, K> a;
    a.load(R, ldr, rrow0, lane);
#pragma unroll
    for (int s = 0; s < Frags<CT, K>::NS; ++s) acc = Img<CT>::mma(a.f[s], c.f[s], acc);
}
template <typename CT, int K>   // first operand preloaded
__device__ __forceinline__ void mm16_r(f32x4& acc, const Frags<CT, K>& r, const CT* Cc, int ldc, int crow0, int lane) {
    Frags<CT, K> b;
    b.load(Cc, ldc, crow0, lane);
#pragma unroll
    for (int s = 0; s < Frags<CT, K>::NS; ++s) acc = Img<CT>::mma(r.f[s], b.f[s], acc);
}

// fragment of an LDS image [rows][ld] (k contiguous) in the PERMUTED k order of the score registers: a lane that owns S[j][t] for
// j = jt*16 + (l>>4)*4 + r (r = 0..3) feeds P straight from registers into the next MFMA if the other operand is read as
//   bf16: step s (32 k):  e = 0..7  <->  k = (2s + e/4)*16 + (l>>4)*4 + e%4          (two 8-B reads)
//   f32 : step (jt, r)    <->  k = jt*16 + (l>>4)*4 + r                              (one 4-B read)
// (the contraction index order is free as long as both operands agree), so the probabilities never go through LDS.
template <typename CT>
__device__ __forceinline__ typename Img<CT>::V load_perm(const CT* img, int ld, int row0, int step, int lane) {
    const CT* p = img + (row0 + (lane & 15)) * ld + (lane >> 4) * 4;
    if constexpr (sizeof(CT) == 2) {
        const bf16x4 lo = *(const bf16x4*)(p + (2 * step) * 16), hi = *(const bf16x4*)(p + (2 * step + 1) * 16);
        return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else {
        return p[(step >> 2) * 16 + (step & 3)];
    }
}
// bf16 only: the same permuted-k fragment of the TRANSPOSE of a row-major image rm[k][ld] (k = row index, operand rows = columns col0..col0+15),
// fetched with ds_read_b64_tr_b16: per 16-lane group a [4 k][16 columns] block, lane i supplies the address of (row k0 + i/4, columns (i%4)*4..)
// and receives column i.  Replaces the explicitly transposed V^T / K^T / Q^T / dO^T images (2-byte scattered LDS stores + a second prefetch).
__device__ __forceinline__ bf16x8 load_perm_tr(const bf16_t* rm, int ld, int col0, int step, int lane) {
    const int i = lane & 15, kc = lane >> 4;
    bf16x8 v;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bf16_t* p = rm + ((2 * step + h) * 16 + kc * 4 + (i >> 2)) * ld + col0 + (i & 3) * 4;
        const short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
        const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
        v[h * 4 + 0] = tb[0]; v[h * 4 + 1] = tb[1]; v[h * 4 + 2] = tb[2]; v[h * 4 + 3] = tb[3];
    }
    return v;
}
// the matching register operand built from s[jt][r]
template <typename CT>
__device__ __forceinline__ typename Img<CT>::V reg_perm(const float (&s)[4][4], int step) {
    if constexpr (sizeof(CT) == 2) {
        return (bf16x8){(bf16_t)s[2 * step][0], (bf16_t)s[2 * step][1], (bf16_t)s[2 * step][2], (bf16_t)s[2 * step][3],
                        (bf16_t)s[2 * step + 1][0], (bf16_t)s[2 * step + 1][1], (bf16_t)s[2 * step + 1][2], (bf16_t)s[2 * step + 1][3]};
    } else {
        return s[step >> 2][step & 3];
    }
}

template <int A, int B> struct CMax { static constexpr int v = A > B ? A : B; };

// ---- global row tile [rows x NC] -> LDS image [rows][ld] (zero-fill invalid rows and pad columns up to NCP)
template <typename CT, int NC, int NCP, int NTHR = 256>
__device__ __forceinline__ void load_rows(CT* img, int ld, const CT* __restrict__ src, int64_t ld_src, int rows, int valid_rows, int tid) {
    constexpr int VE = 16 / sizeof(CT);  // elements per 16-B vector
    constexpr int CH = NCP / VE;
    for (int it = tid; it < rows * CH; it += NTHR) {
        const int r = it / CH, c = (it % CH) * VE;
        CT tmp[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) tmp[e] = from_f32<CT>(0.f);
        if (r < valid_rows && c < NC) {
            if constexpr (sizeof(CT) == 2) *(bf16x8*)tmp = *(const bf16x8*)(src + (int64_t)r * ld_src + c);
            else *(f32x4*)tmp = *(const f32x4*)(src + (int64_t)r * ld_src + c);
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) img[r * ld + c + e] = tmp[e];
    }
}
// ---- global row tile [rows x NC] -> transposed LDS image [NC][ld] (k = row index contiguous); rows even
template <typename CT, int NC, int NTHR = 256>
__device__ __forceinline__ void load_rows_T(CT* img, int ld, const CT* __restrict__ src, int64_t ld_src, int rows, int valid_rows, int tid) {
    constexpr int VE = 16 / sizeof(CT);
    constexpr int CH = NC / VE;
    const int pairs = rows >> 1;
    for (int it = tid; it < pairs * CH; it += NTHR) {
        const int p = it % pairs, c = (it / pairs) * VE;
        CT a[VE], b[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) a[e] = b[e] = from_f32<CT>(0.f);
        if (2 * p < valid_rows) {
            if constexpr (sizeof(CT) == 2) *(bf16x8*)a = *(const bf16x8*)(src + (int64_t)(2 * p) * ld_src + c);
            else *(f32x4*)a = *(const f32x4*)(src + (int64_t)(2 * p) * ld_src + c);
        }
        if (2 * p + 1 < valid_rows) {
            if constexpr (sizeof(CT) == 2) *(bf16x8*)b = *(const bf16x8*)(src + (int64_t)(2 * p + 1) * ld_src + c);
            else *(f32x4*)b = *(const f32x4*)(src + (int64_t)(2 * p + 1) * ld_src + c);
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            if constexpr (sizeof(CT) == 2) {      // rows 2p, 2p+1 of column c+e are adjacent in the image: one 4-B store
                bf16x2 pr = {a[e], b[e]};
                *(bf16x2*)(img + (c + e) * ld + 2 * p) = pr;
            } else {
                img[(c + e) * ld + 2 * p] = a[e];
                img[(c + e) * ld + 2 * p + 1] = b[e];
            }
        }
    }
}


// sum / dot over N contiguous image elements starting at p (16-B aligned, N multiple of the 16-B vector width)
template <typename CT, int N>
__device__ __forceinline__ float sum_contig(const CT* p) {
    constexpr int VE = 16 / sizeof(CT);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N / VE; ++i) {
        if constexpr (sizeof(CT) == 2) { bf16x8 v = *(const bf16x8*)(p + i * VE);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[e]; }
        else { f32x4 v = *(const f32x4*)(p + i * VE); s += v[0] + v[1] + v[2] + v[3]; }
    }
    return s;
}
template <typename CT, int N>
__device__ __forceinline__ float dot_contig(const CT* p, const float* w) {
    constexpr int VE = 16 / sizeof(CT);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N / VE; ++i) {
        if constexpr (sizeof(CT) == 2) { bf16x8 v = *(const bf16x8*)(p + i * VE);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[e] * w[i * VE + e]; }
        else { f32x4 v = *(const f32x4*)(p + i * VE);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += v[e] * w[i * VE + e]; }
    }
    return s;
}

// Register prefetch of one [ROWS x NC] row tile (one 16-B vector per item, item -> thread round-robin): the global
// loads of chunk c+1 are issued right after chunk c's images are written and stay in flight during chunk c's compute.
// ROWFAST: item -> (row = it % ROWS, chunk = it / ROWS): the lanes of a wave hold 64 different rows of one 16-B column chunk, so the
// transposed image store (store_T) writes 64 consecutive k positions per instruction (conflict-free; the row-major mapping makes the 8 lanes
// of a row hit one bank: 8-way conflict).  Costs less coalesced global loads (prefetched a chunk ahead, latency hidden).
template <typename CT, int NC, int NCP, int ROWS, int NTHR, bool ROWFAST = false>
struct RowPrefetch {
    static constexpr int VE = 16 / sizeof(CT), CH = NC / VE, NI = (ROWS * CH + NTHR - 1) / NTHR;
    CT r[NI][VE];
    __device__ __forceinline__ void load(const CT* __restrict__ src, int64_t ld_src, int valid, int tid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + NTHR * i;
            const int row = ROWFAST ? it % ROWS : it / CH, c = (ROWFAST ? it / ROWS : it % CH) * VE;
#pragma unroll
            for (int e = 0; e < VE; ++e) r[i][e] = from_f32<CT>(0.f);
            if (it < ROWS * CH && row < valid) {
                if constexpr (sizeof(CT) == 2) *(bf16x8*)r[i] = *(const bf16x8*)(src + (int64_t)row * ld_src + c);
                else *(f32x4*)r[i] = *(const f32x4*)(src + (int64_t)row * ld_src + c);
            }
        }
    }
    // row-major image [ROWS][ld] (k = column); pad columns NC..NCP are (re)zeroed
    __device__ __forceinline__ void store_rows(CT* img, int ld, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + NTHR * i;
            if (it < ROWS * CH) {
                const int row = ROWFAST ? it % ROWS : it / CH, c = (ROWFAST ? it / ROWS : it % CH) * VE;
                if constexpr (sizeof(CT) == 2) *(bf16x8*)(img + row * ld + c) = *(const bf16x8*)r[i];
                else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) img[row * ld + c + e] = r[i][e];
                }
            }
        }
        if constexpr (NCP > NC) {
            for (int it = tid; it < ROWS * (NCP - NC); it += NTHR) img[(it / (NCP - NC)) * ld + NC + it % (NCP - NC)] = from_f32<CT>(0.f);
        }
    }
    // store_rows + the FAVOR+ feature offset of every row, off[row] = 0.5 * c2 * |row|^2 + half_ln_f, from the registers (the CH lanes that
    // hold one row are consecutive: xor-shuffle reduce) — the separate LDS pass over the image and its barrier were 8.6 % of the forward kernel
    __device__ __forceinline__ void store_rows_off(CT* img, int ld, int tid, float* off, float c2, float half_ln_f) const {
        static_assert(!ROWFAST && (CH & (CH - 1)) == 0 && CH <= 64 && NTHR % CH == 0, "row pieces must sit in consecutive lanes");
        store_rows(img, ld, tid);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + NTHR * i;
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < VE; ++e) { const float x = to_f32<CT>(r[i][e]); sq += x * x; }
#pragma unroll
            for (int o = CH >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            if (it < ROWS * CH && (it % CH) == 0) off[it / CH] = 0.5f * c2 * sq + half_ln_f;
        }
    }
    // transposed image [NC][ld] (k = row index)
    __device__ __forceinline__ void store_T(CT* img, int ld, int tid) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int it = tid + NTHR * i;
            if (it < ROWS * CH) {
                const int row = ROWFAST ? it % ROWS : it / CH, c = (ROWFAST ? it / ROWS : it % CH) * VE;
#pragma unroll
                for (int e = 0; e < VE; ++e) img[(c + e) * ld + row] = r[i][e];
            }
        }
    }
};
