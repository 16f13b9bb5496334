// Causal softmax attention of the GPT-2 backbone (HF GPT2Attention._attn), bf16, d_head 64, T % 128 == 0 — 32 x 32 x 16 MFMA kernels (r03).
//
// The generic kernels of emo_softmax_attn.hip give a wave 16 query rows on 16 x 16 tiles: a row's 64 scores of a key tile sit in FOUR lanes,
// so every key tile pays cross-lane reductions, and a lane carries only 16 scores between two dependent MFMA stages (r02: latency-bound,
// 0.09 of the MFMA peak, "only more resident waves hide it").  Here a wave owns 32 query rows on 32 x 32 x 16 tiles with the swapped product
// S^T = K Q^T: the accumulator layout is lane <-> query (lane % 32), 16 keys per lane and tile half (key = (i & 3) + 8 (i >> 2) + 4 (lane / 32)),
// i.e. a query's 64 scores of a key tile sit in TWO lanes (l, l + 32): row maximum = 31 v_max + one v_permlane32_swap, the row sum stays a
// per-lane partial until the sweep ends, and 32 independent scores per lane give the softmax VALU work its instruction-level parallelism.
// P goes from the score registers straight into the P V product (O^T = V^T P^T) as B operand in the permuted key order
// k-slot e of lane half hi <-> key 16 u + (e & 3) + 8 (e >> 2) + 4 hi; the A operand V^T is read from the row-major V tile with
// ds_read_b64_tr_b16 in the same order.  K / V tiles (64 keys) arrive by LDS-DMA (inline asm: see emo_favor_fs.hip) into a 2-slot ring, 128-B
// rows with the 16-B pieces XOR-swizzled by a32_sw(row); one workgroup barrier per key tile; 4 waves = 128 query rows per workgroup.
// Numerics = the generic bf16 kernel's: base-2 domain (scores scaled by log2(e)/sqrt(dh)), fp32 statistics, dropout regenerated from
// (seed, offset, ((b H + h) T + t) T + j) with one keyed hash per 4 consecutive keys, lse = m ln 2 + ln(l).
#include "emo_common.h"

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr float A32_LOG2E = 1.4426950408889634f, A32_LN2 = 0.6931471805599453f;
constexpr int A32_ROWB = 128, A32_TILEB = 64 * A32_ROWB;   // one K or V tile: 64 keys x 64 bf16

__device__ __forceinline__ f32x16 mma3216(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void a32_dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t a32_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
// Tile-row swizzle: the 16-B piece p of row r sits at piece p ^ a32_sw(r).  r03 used r & 7: with 128-B rows every 16-B fragment read of 32
// consecutive rows and every transpose read was 2-way bank-conflicted (rows r and r + 8 of a lane group share their banks; r04 PMC: half of
// these kernels' LDS cycles were conflicts, the LDS 27-40 % busy).  This map (bits: row bit 2, row bit 3, row bit 1) is conflict-free for both
// patterns (tools/lds_conflicts.py model, exhaustive search over the linear maps of the row bits).
__device__ __forceinline__ int a32_sw(int row) { return ((row >> 2) & 3) | (((row >> 1) & 1) << 2); }
__device__ __forceinline__ float a32_pair_max(float x) {       // max over the two lanes (l, l ^ 32) that share a query row
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)a[0]), __builtin_bit_cast(float, (uint32_t)a[1]));
}
__device__ __forceinline__ float a32_pair_sum(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
}

// =============================================================================================== forward
// KB: also write the dropout keep decisions, one 32-bit word per lane and key tile: word [bh][key tile kt][hi][query row], bit i + 16 half <->
// key 64 kt + 32 half + (i & 3) + 8 (i >> 2) + 4 hi — the lane's own score registers, so the word is assembled without any cross-lane traffic
// (a query row's 64 decisions of a key tile = the words of lanes l and l + 32).  The backward passes read the bits instead of re-evaluating the
// keyed hash per score (r04: the hash was ~30 % of the dK/dV pass's VALU work).  Only tiles at or below the diagonal are written / read.
template <bool KB>
__global__ __launch_bounds__(256, 3) void sattn32_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ld,
                                                             bf16_t* __restrict__ out, int64_t ld_out, float* __restrict__ lse_g, int64_t T, int64_t H,
                                                             DropCtx drop, uint32_t* __restrict__ keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2 slots][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, ql = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // longest sweeps first over the WHOLE grid (blockIdx.x = (b, h) runs fastest): with the tiles of one (b, h) adjacent in dispatch order the last
    // (b, h)'s long blocks started late and a quarter of the block slots idled through the tail (r04 PMC: 1.0 resident waves per SIMD of 2)
    const int64_t qt = (int64_t)gridDim.y - 1 - blockIdx.y;
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t q0 = qt * 128, qrow = q0 + 32 * w + ql;
    const bf16_t* qb = q + (b * T) * ld + h * 64;
    const bf16_t* kb = k + (b * T) * ld + h * 64;
    const bf16_t* vb = v + (b * T) * ld + h * 64;
    const int nkt = (int)(2 * (qt + 1));                              // key tiles 0 .. (q0 + 127) / 64
    const int64_t wq_max = q0 + 32 * w + 31, wq_min = q0 + 32 * w;    // query range of this wave

    bf16x8 qf[4];                                                     // Q rows as B operand: column = query ql, k = d = 16 s + 8 hi ..
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qb + qrow * ld + 16 * s + 8 * hi);

    const uint32_t ring = __builtin_amdgcn_readfirstlane(a32_lds_addr(smem));
    // DMA: wave w moves key rows 16 w .. 16 w + 15 of K and of V (two 1-KB pieces each); lane: row + lane / 8, physical piece lane % 8
    const int64_t so0 = (int64_t)(16 * w + (lane >> 3)) * ld + (((lane & 7) ^ a32_sw(lane >> 3)) << 3);
    const int64_t so1 = (int64_t)(16 * w + 8 + (lane >> 3)) * ld + (((lane & 7) ^ a32_sw(8 + (lane >> 3))) << 3);   // rows + 8
    auto issue = [&](int kt) {
        const int64_t o = (int64_t)kt * 64 * ld;
        const uint32_t dst = ring + (kt & 1) * 2 * A32_TILEB + w * 2048;
        a32_dma16(kb + o + so0, dst);
        a32_dma16(kb + o + so1, dst + 1024);
        a32_dma16(vb + o + so0, dst + A32_TILEB);
        a32_dma16(vb + o + so1, dst + A32_TILEB + 1024);
    };
    issue(0);
    const float c2 = rsqrtf(64.f) * A32_LOG2E;
    const bool small_idx = (int64_t)gridDim.x * T * T <= ((int64_t)1 << 34);        // every element index of the call below 2^34
    const uint32_t rowq = (uint32_t)(((bh * T + qrow) * T) >> 2);
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    // lane-constant LDS offsets: K fragment (row ql of a 32-key half, piece 2 s + hi), V transposed fragments (see the header)
    const int vg = lane >> 4, vi = lane & 15;                         // 16-lane group / lane inside it for the transpose reads
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of tile kt landed
        __builtin_amdgcn_s_barrier();                                 // ... everyone's did; everyone is done with the other slot
        asm volatile("" ::: "memory");
        if (kt + 1 < nkt) issue(kt + 1);
        const int64_t k0 = (int64_t)kt * 64;
        if (k0 > wq_max) continue;                                    // key tile entirely above this wave's rows (wave-uniform)
        const char* Kt = smem + (kt & 1) * 2 * A32_TILEB;
        const char* Vt = Kt + A32_TILEB;
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ka = *(const bf16x8*)(Kt + ql * A32_ROWB + (((2 * s + hi) ^ a32_sw(ql)) << 4));
            const bf16x8 kc = *(const bf16x8*)(Kt + (32 + ql) * A32_ROWB + (((2 * s + hi) ^ a32_sw(ql)) << 4));
            s0 = mma3216(ka, qf[s], s0);
            s1 = mma3216(kc, qf[s], s1);
        }
        const bool diag = k0 + 63 > wq_min;                           // some key of the tile lies above some row of the wave
        // statistics in the RAW score domain (max commutes with the positive scale): p = exp2(fma(s, c2, -c2 m)) is one FMA + one v_exp per score
        float mx = -INFINITY;
        if (diag) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kl = (i & 3) + 8 * (i >> 2) + 4 * hi;
                if (k0 + kl > qrow) s0[i] = -INFINITY;
                if (k0 + 32 + kl > qrow) s1[i] = -INFINITY;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[i], s1[i]));
        mx = a32_pair_max(mx);
        float m_new = fmaxf(m_run, mx);                               // (key 0 is visible to every row: never -inf)
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
        const float nm = -m_new * c2;
        float psum = 0.f;
        uint32_t kw = 0u;
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            float d0[4] = {1.f, 1.f, 1.f, 1.f}, d1[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thr16) {
                const uint64_t base = (uint64_t)((bh * T + qrow) * T + k0 + 8 * a4 + 4 * hi);
                if (KB && small_idx) {                                // (wave-uniform) 32-bit quad index: ((bh T + qrow) T + k0 + 8 a4 + 4 hi) / 4
                    const uint32_t quad = rowq + (uint32_t)(k0 >> 2) + 2 * a4 + hi;
                    drop_mult4_bits_q(drop, quad, d0, kw, 4 * a4);
                    drop_mult4_bits_q(drop, quad + 8, d1, kw, 16 + 4 * a4);
                } else if (KB) {
                    drop_mult4_bits(drop, base, d0, kw, 4 * a4);
                    drop_mult4_bits(drop, base + 32, d1, kw, 16 + 4 * a4);
                } else {
                    drop_mult4(drop, base, d0);
                    drop_mult4(drop, base + 32, d1);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(s0[4 * a4 + r], c2, nm)), p1 = __builtin_amdgcn_exp2f(fmaf(s1[4 * a4 + r], c2, nm));
                psum += p0 + p1;
                s0[4 * a4 + r] = p0 * d0[r];
                s1[4 * a4 + r] = p1 * d1[r];
            }
        }
        if (KB) keep[((bh * (T >> 6) + kt) * 2 + hi) * T + qrow] = kw;
        l_run = l_run * alpha + psum;                                 // per-lane partial (this lane's 32 of the row's 64 keys)
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        // O^T[d][q] += sum_key V^T[d][key] P^T[key][q]: four 16-key steps, two 32-row d halves
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bf16x8 pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[e] = (bf16_t)((u < 2 ? s0 : s1)[8 * (u & 1) + e]);
#pragma unroll
            for (int dh2 = 0; dh2 < 2; ++dh2) {
                bf16x8 va;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row = 16 * u + 8 * hh + 4 * (vg >> 1) + (vi >> 2), col = 32 * dh2 + 16 * (vg & 1) + 4 * (vi & 3);
                    const char* p = Vt + row * A32_ROWB + (((col >> 3) ^ a32_sw(row)) << 4) + (col & 7) * 2;
                    const short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
                    const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
                    va[4 * hh + 0] = tb[0]; va[4 * hh + 1] = tb[1]; va[4 * hh + 2] = tb[2]; va[4 * hh + 3] = tb[3];
                }
                if (dh2 == 0) o0 = mma3216(va, pb, o0); else o1 = mma3216(va, pb, o1);
            }
        }
    }
    const float l_tot = a32_pair_sum(l_run);
    const float inv = 1.f / l_tot;
    bf16_t* ob = out + (b * T + qrow) * ld_out + h * 64;
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
        const bf16x4 x0 = {(bf16_t)(o0[4 * a4] * inv), (bf16_t)(o0[4 * a4 + 1] * inv), (bf16_t)(o0[4 * a4 + 2] * inv), (bf16_t)(o0[4 * a4 + 3] * inv)};
        const bf16x4 x1 = {(bf16_t)(o1[4 * a4] * inv), (bf16_t)(o1[4 * a4 + 1] * inv), (bf16_t)(o1[4 * a4 + 2] * inv), (bf16_t)(o1[4 * a4 + 3] * inv)};
        *(bf16x4*)(ob + 8 * a4 + 4 * hi) = x0;
        *(bf16x4*)(ob + 32 + 8 * a4 + 4 * hi) = x1;
    }
    if (hi == 0) lse_g[bh * T + qrow] = m_run * c2 * A32_LN2 + logf(l_tot);
}

// =============================================================================================== backward: dQ (and delta)
// The forward's geometry (a wave owns 32 query rows, lane <-> query, K / V tiles of 64 keys through the 2-slot ring) with two score products per
// key tile, S^T = K Q^T and dP^T = V dO^T, both in the layout lane <-> query / registers <-> keys, so dS = P (dP * dropout - delta) is
// element-wise with per-lane row statistics (lse, delta = dO . O of the lane's row: no LDS, no cross-lane traffic beyond one pair sum in the
// prologue) and goes straight back as B operand into dQ^T[d][q] += K^T[d][key] dS^T[key][q] (A operand K^T by ds_read_b64_tr_b16 from the
// row-major K tile: the forward's V^T recipe).  KB: the dropout decisions are the forward's keep word of (key tile, lane) — the lane's own
// 32 scores, bit = register index — fetched one tile ahead; otherwise one keyed hash per 4 consecutive keys as in the forward.
// The generic kernel (16-row tiles, P through LDS) took 304 us per layer at B = 16, T = 2048 against the forward's 271.
template <bool KB>
__global__ __launch_bounds__(256, KB ? 3 : 2) void sattn32_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ld,
                                                            const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, int64_t ld_out,
                                                            const float* __restrict__ lse_g, float* __restrict__ delta_g, bf16_t* __restrict__ dq, int64_t ld_d,
                                                            int64_t T, int64_t H, DropCtx drop, const uint32_t* __restrict__ keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2 slots][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, ql = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // longest sweeps first over the WHOLE grid (blockIdx.x = (b, h) runs fastest): with the tiles of one (b, h) adjacent in dispatch order the last
    // (b, h)'s long blocks started late and a quarter of the block slots idled through the tail (r04 PMC: 1.0 resident waves per SIMD of 2)
    const int64_t qt = (int64_t)gridDim.y - 1 - blockIdx.y;
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t q0 = qt * 128, qrow = q0 + 32 * w + ql;
    const bf16_t* qb = q + (b * T) * ld + h * 64;
    const bf16_t* kb = k + (b * T) * ld + h * 64;
    const bf16_t* vb = v + (b * T) * ld + h * 64;
    const int nkt = (int)(2 * (qt + 1));
    const int64_t wq_max = q0 + 32 * w + 31, wq_min = q0 + 32 * w;

    bf16x8 qf[4], gf[4];                                              // Q / dO rows as B operands: column = query ql, k = d = 16 s + 8 hi ..
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *(const bf16x8*)(qb + qrow * ld + 16 * s + 8 * hi);
        gf[s] = *(const bf16x8*)(dout + (b * T + qrow) * ld_out + h * 64 + 16 * s + 8 * hi);
        const bf16x8 of = *(const bf16x8*)(out + (b * T + qrow) * ld_out + h * 64 + 16 * s + 8 * hi);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)gf[s][e] * (float)of[e];
    }
    dl = a32_pair_sum(dl);                                            // delta = dO . O of the row (both lanes of the pair hold it)
    if (hi == 0) delta_g[bh * T + qrow] = dl;
    const float nl = -lse_g[bh * T + qrow] * A32_LOG2E;
    const uint32_t* kwp = KB ? keep + ((bh * (T >> 6)) * 2 + hi) * T + qrow : nullptr;       // + kt * 2 T per key tile
    uint32_t kw_next = KB ? kwp[0] : 0u;

    const uint32_t ring = __builtin_amdgcn_readfirstlane(a32_lds_addr(smem));
    const int64_t so0 = (int64_t)(16 * w + (lane >> 3)) * ld + (((lane & 7) ^ a32_sw(lane >> 3)) << 3);
    const int64_t so1 = (int64_t)(16 * w + 8 + (lane >> 3)) * ld + (((lane & 7) ^ a32_sw(8 + (lane >> 3))) << 3);
    auto issue = [&](int kt) {
        const int64_t o = (int64_t)kt * 64 * ld;
        const uint32_t dst = ring + (kt & 1) * 2 * A32_TILEB + w * 2048;
        a32_dma16(kb + o + so0, dst);
        a32_dma16(kb + o + so1, dst + 1024);
        a32_dma16(vb + o + so0, dst + A32_TILEB);
        a32_dma16(vb + o + so1, dst + A32_TILEB + 1024);
    };
    issue(0);
    const float c2 = rsqrtf(64.f) * A32_LOG2E;
    const uint32_t scale_bits = __builtin_bit_cast(uint32_t, drop.scale);
    f32x16 o0, o1;                                                    // dQ^T: rows d (0-31 / 32-63), column = query
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    const int vg = lane >> 4, vi = lane & 15;
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const uint32_t kw = kw_next;
        const int64_t k0 = (int64_t)kt * 64;
        if (KB && kt + 1 < nkt && k0 + 64 <= wq_max) kw_next = kwp[(int64_t)(kt + 1) * 2 * T];     // (the next tile's word, if this wave will sweep it)
        if (kt + 1 < nkt) issue(kt + 1);
        if (k0 > wq_max) continue;
        const char* Kt = smem + (kt & 1) * 2 * A32_TILEB;
        const char* Vt = Kt + A32_TILEB;
        f32x16 s0, s1, p0, p1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; p0[i] = 0.f; p1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int oa = ql * A32_ROWB + (((2 * s + hi) ^ a32_sw(ql)) << 4), oc = oa + 32 * A32_ROWB;
            s0 = mma3216(*(const bf16x8*)(Kt + oa), qf[s], s0);
            s1 = mma3216(*(const bf16x8*)(Kt + oc), qf[s], s1);
            p0 = mma3216(*(const bf16x8*)(Vt + oa), gf[s], p0);
            p1 = mma3216(*(const bf16x8*)(Vt + oc), gf[s], p1);
        }
        const int qrel = (int)(qrow - k0) - 4 * hi;                    // key 8 a4 + r (+ 32) of this lane lies above the row when it exceeds qrel
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            float d0[4] = {1.f, 1.f, 1.f, 1.f}, d1[4] = {1.f, 1.f, 1.f, 1.f};
            if (KB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d0[r] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_sbfe((int)kw, 4 * a4 + r, 1) & scale_bits);
                    d1[r] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_sbfe((int)kw, 16 + 4 * a4 + r, 1) & scale_bits);
                }
            } else if (drop.thr16) {
                const uint64_t base = (uint64_t)((bh * T + qrow) * T + k0 + 8 * a4 + 4 * hi);
                drop_mult4(drop, base, d0);
                drop_mult4(drop, base + 32, d1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * a4 + r;
                float e0 = __builtin_amdgcn_exp2f(fmaf(s0[i], c2, nl)), e1 = __builtin_amdgcn_exp2f(fmaf(s1[i], c2, nl));
                // causal mask as ONE 32-bit compare of a compile-time key index with a per-tile lane value (r06; `if (diag) if (k0 + kl > qrow)`
                // compiled into 64-bit compares + mask ANDs for every score of every tile): away from the diagonal qrel >= 63 and nothing is masked
                if (r + 8 * a4 > qrel) e0 = 0.f;
                if (32 + r + 8 * a4 > qrel) e1 = 0.f;
                s0[i] = e0 * fmaf(p0[i], d0[r], -dl);                 // dS (in units of the scaled scores)
                s1[i] = e1 * fmaf(p1[i], d1[r], -dl);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bf16x8 sb;
#pragma unroll
            for (int e = 0; e < 8; ++e) sb[e] = (bf16_t)((u < 2 ? s0 : s1)[8 * (u & 1) + e]);
#pragma unroll
            for (int dh2 = 0; dh2 < 2; ++dh2) {
                bf16x8 ka;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int row = 16 * u + 8 * hh + 4 * (vg >> 1) + (vi >> 2), col = 32 * dh2 + 16 * (vg & 1) + 4 * (vi & 3);
                    const char* p = Kt + row * A32_ROWB + (((col >> 3) ^ a32_sw(row)) << 4) + (col & 7) * 2;
                    const bf16x4 tb = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
                    ka[4 * hh + 0] = tb[0]; ka[4 * hh + 1] = tb[1]; ka[4 * hh + 2] = tb[2]; ka[4 * hh + 3] = tb[3];
                }
                if (dh2 == 0) o0 = mma3216(ka, sb, o0); else o1 = mma3216(ka, sb, o1);
            }
        }
    }
    const float inv = rsqrtf(64.f);
    bf16_t* ob = dq + (b * T + qrow) * ld_d + h * 64;
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
        const bf16x4 x0 = {(bf16_t)(o0[4 * a4] * inv), (bf16_t)(o0[4 * a4 + 1] * inv), (bf16_t)(o0[4 * a4 + 2] * inv), (bf16_t)(o0[4 * a4 + 3] * inv)};
        const bf16x4 x1 = {(bf16_t)(o1[4 * a4] * inv), (bf16_t)(o1[4 * a4 + 1] * inv), (bf16_t)(o1[4 * a4 + 2] * inv), (bf16_t)(o1[4 * a4 + 3] * inv)};
        *(bf16x4*)(ob + 8 * a4 + 4 * hi) = x0;
        *(bf16x4*)(ob + 32 + 8 * a4 + 4 * hi) = x1;
    }
}

// =============================================================================================== backward: dK, dV
// Key-stationary: a wave owns 32 keys (K / V rows as MFMA B operands in registers), the workgroup 128; query tiles of 64 rows (Q and dO, plus
// the rows' lse and delta) stream through a 2-slot LDS ring.  Products S = Q K^T and dP = dO V^T land in the layout lane <-> key, registers <->
// query (row 8 a + 4 (lane / 32) + r of a 32-row half), so Pd = P * dropout and dS = P (dP * dropout - delta) are element-wise with the row
// statistics read from LDS (broadcast reads), and both go straight back into dV^T += dO^T Pd and dK^T += Q^T dS as B operands in the
// permuted row order — the A operands dO^T / Q^T come from the row-major tiles by ds_read_b64_tr_b16 in the same order (the forward's V^T
// recipe).  Dropout: one keyed hash per 4 consecutive keys of a row, shared by the four lanes of a quad (drop_mult_col4).  No cross-lane
// reduction anywhere; dK^T / dV^T stay in registers for the whole sweep.  The generic kernel (16-row query tiles, P and dS through LDS
// transposes, 168 VGPR spills) took 2.2 x the dQ pass.
__device__ __forceinline__ void a32_dma16s(const char* sbase, uint32_t voff, uint32_t lds_dst) {       // wave-uniform base + 32-bit lane offset
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void a32_dma4s(const char* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
constexpr int A32_QSLOT = 2 * A32_TILEB + 512 + 1024;          // Q tile | dO tile | lse[64] | delta[64] | keep words [2 key tiles][2 hi][64 rows]

// KB: the dropout decisions come from the forward's keep words (a lane's key fixes the word plane and the bit, the rows index the words: four
// consecutive rows = one ds_read_b128 of the 1-KB block that waves 2 and 3 bring in with the query tile) instead of one keyed hash per 4 scores.
template <bool KB>
__global__ __launch_bounds__(256, 2) void sattn32_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ld,
                                                             const bf16_t* __restrict__ dout, int64_t ld_out, const float* __restrict__ lse_g,
                                                             const float* __restrict__ delta_g, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t ld_d,
                                                             int64_t T, int64_t H, DropCtx drop, const uint32_t* __restrict__ keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2 slots][Q tile | dO tile | lse | delta | keep words]
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, kl = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ktile = blockIdx.y;                                 // key tile 0 has the longest sweep: all (b, h) of it are dispatched first
    const int64_t bh = blockIdx.x, b = bh / H, h = bh % H;
    const int64_t kw0 = ktile * 128 + 32 * w, key = kw0 + kl;
    const bf16_t* qb = q + (b * T) * ld + h * 64;
    const bf16_t* kb = k + (b * T) * ld + h * 64;
    const bf16_t* vb = v + (b * T) * ld + h * 64;
    const bf16_t* gb = dout + (b * T) * ld_out + h * 64;
    const int qt0 = (int)(2 * ktile), nqt = (int)(T / 64);            // query tiles qt0 .. nqt - 1 see this key tile

    bf16x8 kB[4], vB[4];                                              // K / V rows as B operands: column = key kl, k = d = 16 s + 8 hi ..
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kB[s] = *(const bf16x8*)(kb + key * ld + 16 * s + 8 * hi);
        vB[s] = *(const bf16x8*)(vb + key * ld + 16 * s + 8 * hi);
    }
    const uint32_t ring = __builtin_amdgcn_readfirstlane(a32_lds_addr(smem));
    // DMA sources as (wave-uniform base) + (32-bit lane byte offset): no per-lane 64-bit pointers live across the sweep
    const int p0 = ((lane & 7) ^ a32_sw(lane >> 3)) << 3, p1 = ((lane & 7) ^ a32_sw(8 + (lane >> 3))) << 3;      // rows r and r + 8 of the wave's 16
    const uint32_t so0 = (uint32_t)(((16 * w + (lane >> 3)) * ld + p0) * 2), so1 = (uint32_t)(((16 * w + 8 + (lane >> 3)) * ld + p1) * 2);
    const uint32_t go0 = (uint32_t)(((16 * w + (lane >> 3)) * ld_out + p0) * 2), go1 = (uint32_t)(((16 * w + 8 + (lane >> 3)) * ld_out + p1) * 2);
    const uint32_t lo4 = (uint32_t)(lane * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the plain loads above are the compiler's; keep them out of the DMA count)
    auto issue = [&](int qt) {
        const uint32_t dst = ring + (qt & 1) * A32_QSLOT;
        const char* qs = (const char*)(qb + (int64_t)qt * 64 * ld);
        const char* gs = (const char*)(gb + (int64_t)qt * 64 * ld_out);
        a32_dma16s(qs, so0, dst + w * 2048);
        a32_dma16s(qs, so1, dst + w * 2048 + 1024);
        a32_dma16s(gs, go0, dst + A32_TILEB + w * 2048);
        a32_dma16s(gs, go1, dst + A32_TILEB + w * 2048 + 1024);
        if (w == 0) a32_dma4s((const char*)(lse_g + bh * T + (int64_t)qt * 64), lo4, dst + 2 * A32_TILEB);
        if (w == 1) a32_dma4s((const char*)(delta_g + bh * T + (int64_t)qt * 64), lo4, dst + 2 * A32_TILEB + 256);
        if (KB && w >= 2) {                                           // wave 2: key tile 2 ktile, wave 3: 2 ktile + 1; both hi planes, rows of this query tile
            const uint32_t* kp = keep + ((bh * (T >> 6) + 2 * ktile + (w - 2)) * 2) * T + (int64_t)qt * 64;
            a32_dma4s((const char*)kp, lo4, dst + 2 * A32_TILEB + 512 + (w - 2) * 512);
            a32_dma4s((const char*)(kp + T), lo4, dst + 2 * A32_TILEB + 512 + (w - 2) * 512 + 256);
        }
    };
    // this lane's key inside its 64-key tile: word plane (tile w / 2, hi = bit 2 of the key's position in its 32-key half) and bit
    const int kb_plane = (w >> 1) * 512 + ((kl >> 2) & 1) * 256, kb_bit = (kl & 3) + 4 * (kl >> 3) + 16 * (w & 1);
    const uint32_t scale_bits = __builtin_bit_cast(uint32_t, drop.scale);
    issue(qt0);
    const float c2 = rsqrtf(64.f) * A32_LOG2E;
    f32x16 dv0, dv1, dk0, dk1;                                        // dV^T / dK^T: rows d (0-31 / 32-63), column = key
#pragma unroll
    for (int i = 0; i < 16; ++i) { dv0[i] = 0.f; dv1[i] = 0.f; dk0[i] = 0.f; dk1[i] = 0.f; }
    const int vg = lane >> 4, vi = lane & 15;
    for (int qt = qt0; qt < nqt; ++qt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (qt + 1 < nqt) issue(qt + 1);
        const int64_t q0 = (int64_t)qt * 64;
        if (q0 + 63 < kw0) continue;                                  // every row of the tile lies before this wave's keys (wave-uniform)
        const char* Qt = smem + (qt & 1) * A32_QSLOT;
        const char* Gt = Qt + A32_TILEB;
        const float* lse_l = (const float*)(Gt + A32_TILEB);
        const float* del_l = lse_l + 64;
        const int krel = (int)(key - q0) - 4 * hi;                     // row 32 hq + 8 a4 + r of the tile lies before this lane's key when it is below krel
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            f32x16 sa, pa;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sa[i] = 0.f; pa[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int off = (32 * hq + kl) * A32_ROWB + (((2 * s + hi) ^ a32_sw(kl)) << 4);
                sa = mma3216(*(const bf16x8*)(Qt + off), kB[s], sa);  // S[row][key]
                pa = mma3216(*(const bf16x8*)(Gt + off), vB[s], pa);  // dP[row][key]
            }
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                const int rl = 32 * hq + 8 * a4 + 4 * hi;             // rows rl .. rl + 3 of the tile
                const f32x4 ls = *(const f32x4*)(lse_l + rl), dl = *(const f32x4*)(del_l + rl);
                float dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (KB) {
                    const u32x4 kwv = *(const u32x4*)((const char*)del_l + 256 + kb_plane + rl * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dm[r] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_sbfe((int)kwv[r], kb_bit, 1) & scale_bits);
                } else if (drop.thr16) drop_mult_col4(drop, (uint64_t)((bh * T + q0 + rl + (lane & 3)) * T + (key & ~(int64_t)3)), lane, dm);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * a4 + r;
                    float p = __builtin_amdgcn_exp2f(fmaf(sa[i], c2, -ls[r] * A32_LOG2E));
                    if (32 * hq + 8 * a4 + r < krel) p = 0.f;            // (one 32-bit compare; r06 — was a 64-bit compare + mask AND per score)
                    sa[i] = p * dm[r];                                // Pd
                    pa[i] = p * (pa[i] * dm[r] - dl[r]);              // dS
                }
            }
            // the half's 32 rows as two 16-row B operands (permuted row order), straight into dV^T += dO^T Pd and dK^T += Q^T dS
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                bf16x8 pb, sb;
#pragma unroll
                for (int e = 0; e < 8; ++e) { pb[e] = (bf16_t)sa[8 * u2 + e]; sb[e] = (bf16_t)pa[8 * u2 + e]; }
                const int u = 2 * hq + u2;
#pragma unroll
                for (int dh2 = 0; dh2 < 2; ++dh2) {
                    bf16x8 gt, qt_;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = 16 * u + 8 * hh + 4 * (vg >> 1) + (vi >> 2), col = 32 * dh2 + 16 * (vg & 1) + 4 * (vi & 3);
                        const int o = row * A32_ROWB + (((col >> 3) ^ a32_sw(row)) << 4) + (col & 7) * 2;
                        const bf16x4 tg = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(Gt + o)));
                        const bf16x4 tq = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(Qt + o)));
                        gt[4 * hh + 0] = tg[0]; gt[4 * hh + 1] = tg[1]; gt[4 * hh + 2] = tg[2]; gt[4 * hh + 3] = tg[3];
                        qt_[4 * hh + 0] = tq[0]; qt_[4 * hh + 1] = tq[1]; qt_[4 * hh + 2] = tq[2]; qt_[4 * hh + 3] = tq[3];
                    }
                    if (dh2 == 0) { dv0 = mma3216(gt, pb, dv0); dk0 = mma3216(qt_, sb, dk0); }
                    else { dv1 = mma3216(gt, pb, dv1); dk1 = mma3216(qt_, sb, dk1); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                        // one 32-row half at a time (register peak)
        }
    }
    const float inv = rsqrtf(64.f);
    bf16_t* dkp = dk + (b * T + key) * ld_d + h * 64;
    bf16_t* dvp = dv + (b * T + key) * ld_d + h * 64;
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
        const bf16x4 k0v = {(bf16_t)(dk0[4 * a4] * inv), (bf16_t)(dk0[4 * a4 + 1] * inv), (bf16_t)(dk0[4 * a4 + 2] * inv), (bf16_t)(dk0[4 * a4 + 3] * inv)};
        const bf16x4 k1v = {(bf16_t)(dk1[4 * a4] * inv), (bf16_t)(dk1[4 * a4 + 1] * inv), (bf16_t)(dk1[4 * a4 + 2] * inv), (bf16_t)(dk1[4 * a4 + 3] * inv)};
        const bf16x4 v0v = {(bf16_t)dv0[4 * a4], (bf16_t)dv0[4 * a4 + 1], (bf16_t)dv0[4 * a4 + 2], (bf16_t)dv0[4 * a4 + 3]};
        const bf16x4 v1v = {(bf16_t)dv1[4 * a4], (bf16_t)dv1[4 * a4 + 1], (bf16_t)dv1[4 * a4 + 2], (bf16_t)dv1[4 * a4 + 3]};
        *(bf16x4*)(dkp + 8 * a4 + 4 * hi) = k0v;
        *(bf16x4*)(dkp + 32 + 8 * a4 + 4 * hi) = k1v;
        *(bf16x4*)(dvp + 8 * a4 + 4 * hi) = v0v;
        *(bf16x4*)(dvp + 32 + 8 * a4 + 4 * hi) = v1v;
    }
}
}  // namespace

// which: 0 forward.  Returns false when the call is not covered (the caller then runs the generic kernels).
// bytes of the keep words of one attention call (0: the 32 x 32 kernels do not serve the shape, or dropout is off)
int64_t emo_sattn32_keep_bytes(int64_t B, int64_t T, int64_t H, int64_t dh, float p_drop) {
    const char* e = getenv("EMO_SATTN32");
    if (e && atoi(e) == 0) return 0;
    const char* e2 = getenv("EMO_SATTN32_BWD");
    if (e2 && atoi(e2) == 0) return 0;
    const char* e3 = getenv("EMO_SATTN_KEEP");               // "0": every pass re-evaluates the hash
    if (e3 && atoi(e3) == 0) return 0;
    if (dh != 64 || T < 128 || (T % 128) != 0 || !(p_drop > 0.f)) return 0;
    return B * H * (T / 64) * 2 * T * (int64_t)sizeof(uint32_t);
}

bool emo_sattn32_try(int which, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, bf16_t* out, int64_t ld_out, float* lse, int64_t B, int64_t T,
                     int64_t H, DropCtx drop, uint32_t* keep, hipStream_t st) {
    const char* e = getenv("EMO_SATTN32");                   // "0": generic kernels only (read per call: tests toggle it)
    if (e && atoi(e) == 0) return false;
    if (which != 0 || T < 128 || (T % 128) != 0 || (ld & 7) || (ld_out & 3)) return false;
    // 16-B LDS-DMA pieces and bf16x8 row accesses: an unaligned view (a C-ABI caller's column-offset slice) falls back to the generic kernels
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) || ((uintptr_t)lse & 3)) return false;
    dim3 grid((unsigned)(B * H), (unsigned)(T / 128));
    const size_t lds = 4 * A32_TILEB;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)sattn32_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)sattn32_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (keep && drop.thr16) hipLaunchKernelGGL(sattn32_fwd_kernel<true>, grid, dim3(256), lds, st, q, k, v, ld, out, ld_out, lse, T, H, drop, keep);
    else hipLaunchKernelGGL(sattn32_fwd_kernel<false>, grid, dim3(256), lds, st, q, k, v, ld, out, ld_out, lse, T, H, drop, (uint32_t*)nullptr);
    return true;
}

// dQ pass of the backward (writes delta for the dK / dV pass); false: not covered, the caller runs the generic kernel
bool emo_sattn32_dq_try(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* out, const bf16_t* dout, int64_t ld_out, const float* lse,
                        float* delta, bf16_t* dq, int64_t ld_d, int64_t B, int64_t T, int64_t H, DropCtx drop, const uint32_t* keep, hipStream_t st) {
    const char* e = getenv("EMO_SATTN32");
    if (e && atoi(e) == 0) return false;
    const char* e2 = getenv("EMO_SATTN32_BWD");
    if (e2 && atoi(e2) == 0) return false;
    const char* e3 = getenv("EMO_SATTN32_DQ");                // "0": generic dQ kernel
    if (e3 && atoi(e3) == 0) return false;
    if (T < 128 || (T % 128) != 0 || (ld & 7) || (ld_out & 7) || (ld_d & 3) || !delta) return false;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq) & 15) return false;
    dim3 grid((unsigned)(B * H), (unsigned)(T / 128));
    const size_t lds = 4 * A32_TILEB;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)sattn32_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)sattn32_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (keep && drop.thr16) hipLaunchKernelGGL(sattn32_dq_kernel<true>, grid, dim3(256), lds, st, q, k, v, ld, out, dout, ld_out, lse, delta, dq, ld_d, T, H, drop, keep);
    else hipLaunchKernelGGL(sattn32_dq_kernel<false>, grid, dim3(256), lds, st, q, k, v, ld, out, dout, ld_out, lse, delta, dq, ld_d, T, H, drop, (const uint32_t*)nullptr);
    return true;
}

// dK / dV pass of the backward (after the dQ pass has written delta); false: not covered, the caller runs the generic kernel
bool emo_sattn32_dkv_try(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const bf16_t* dout, int64_t ld_out, const float* lse, const float* delta,
                         bf16_t* dk, bf16_t* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H, DropCtx drop, const uint32_t* keep, hipStream_t st) {
    const char* e = getenv("EMO_SATTN32");
    if (e && atoi(e) == 0) return false;
    const char* e2 = getenv("EMO_SATTN32_BWD");
    if (e2 && atoi(e2) == 0) return false;
    if (T < 128 || (T % 128) != 0 || (ld & 7) || (ld_out & 7) || (ld_d & 3) || !delta) return false;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dk | (uintptr_t)dv) & 15) return false;
    dim3 grid((unsigned)(B * H), (unsigned)(T / 128));
    const size_t lds = 2 * A32_QSLOT;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)sattn32_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)sattn32_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (keep && drop.thr16) hipLaunchKernelGGL(sattn32_dkv_kernel<true>, grid, dim3(256), lds, st, q, k, v, ld, dout, ld_out, lse, delta, dk, dv, ld_d, T, H, drop, keep);
    else hipLaunchKernelGGL(sattn32_dkv_kernel<false>, grid, dim3(256), lds, st, q, k, v, ld, dout, ld_out, lse, delta, dk, dv, ld_d, T, H, drop, (const uint32_t*)nullptr);
    return true;
}
