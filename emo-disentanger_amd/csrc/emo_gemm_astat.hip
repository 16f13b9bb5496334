// A-stationary bf16 GEMM for the K = 512 products of the Performer layer (QKV / out-projection / FFN1 forward, FFN2 and
// out-projection dgrad against transposed weight mirrors):  C[M,N] = epilogue(A[M,512] . B[N,512]^T), M = B*T tokens.
//
// Why another kernel: with K = 512 a 128x128 (or 256x256) output tile has only 8-16 K steps, so the tile prologue (first
// operand tiles from HBM) and its epilogue are a large fixed cost per tile, and the 128^2 tile re-fetches its A rows from L2
// for every 128 output columns in half-line (64-B) pieces (measured r01: 550-640 TFLOP/s in the training step).
// Here the reduction dimension is short enough to keep the A operand STATIONARY IN REGISTERS:
//   * a workgroup = 4 waves owns a 128-row panel; each wave holds its 32 rows x 512 k as MFMA operand fragments
//     (2 x 16 fragments = 128 VGPRs), loaded ONCE from HBM straight into the fragment layout;
//   * the block then sweeps ALL N output columns in 64-column tiles: the weight rows stream through a 4-slot LDS ring
//     (one slot = 64 rows x 128 k = 16 KB, LDS-DMA `global_load_lds`, counted s_waitcnt vmcnt + raw s_barrier, never 0 in the
//     loop), so the main loop is ONE pipeline of 4 * N/64 stages with one barrier per 32 MFMAs per wave and no A traffic
//     at all: per CU only 32 B/clk of L2 reads at the MFMA peak;
//   * the 32 x 64 accumulator tile of a wave leaves straight from registers: the weight rows of the four B fragments are
//     permuted so that a lane owns 8 CONSECUTIVE output columns per fragment pair -> 16-B stores (bf16), 16-B residual /
//     mask loads, half the dropout hashes; bias comes from an LDS copy (no VMEM load in the loop -> the compiler never
//     drains the DMA ring);
//   * two workgroups per CU (72 KB LDS, <= 256 VGPRs): one block's panel load / epilogue stores overlap the other's MFMAs.
// LDS image of a slot: [64 rows][16 x 16-B chunks], physical chunk = logical ^ swz(row), swz(row) = (row & 3) | ((row >> 1) & 12):
// conflict-free for the permuted-row ds_read_b128 pattern (checked against the b128 lane groups of the guide's LDS table).
// The DMA writes LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE address and again on the read.
#include "emo_gemm_epi.h"

namespace {
constexpr int AS_K = 512, AS_BM = 128, AS_BN = 64, AS_KS = 128, AS_SLOT = AS_BN * AS_KS * 2, AS_RING = 4 * AS_SLOT;
constexpr int AS_MIN_BLOCKS = 32;                              // = ops.ASTAT_MIN_ROWS / 128 (ops.gemm_bitmask_ok mirrors it): below one panel per CU the columns are split
constexpr int AS_MAXN = 2048;                                   // bias copy in LDS: 8 KB

// sum over the four 16-lane rows of a wave (lanes l % 16 + 16 j), result in every lane: two VALU lane swaps (permlane32_swap / permlane16_swap)
__device__ __forceinline__ float as_sum_lane_rows(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t w = __builtin_bit_cast(uint32_t, y);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}
__device__ __forceinline__ int as_swz(int row) { return (row & 3) | ((row >> 1) & 12); }
// tile row (= output column inside the 64-column tile) that MFMA fragment f reads for its N-index i:
// the lane with N-indices 4g..4g+3 of fragments 2h and 2h+1 then owns columns 32h + 8g .. +7
__device__ __forceinline__ int as_nrow(int f, int i) { return 32 * (f >> 1) + 8 * (i >> 2) + 4 * (f & 1) + (i & 3); }

template <int N> __device__ __forceinline__ void as_wait();
template <> __device__ __forceinline__ void as_wait<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<5>() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<12>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<9>() { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
template <> __device__ __forceinline__ void as_wait<13>() { asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }

// Epilogue operands (residual / mask rows) are fetched by inline-asm loads half a stage before the epilogue: hipcc does not see
// them, so it cannot answer them with the `s_waitcnt vmcnt(0)` that would drain the DMA ring at every column tile; the counted
// wait that covers them is `as_pin` (the "+v" operands order every use of the registers behind the wait).
template <int IMM> __device__ __forceinline__ u32x4 as_load16(const char* sbase, uint32_t voff);   // wave-uniform base + per-lane 32-bit offset + immediate
template <> __device__ __forceinline__ u32x4 as_load16<0>(const char* sbase, uint32_t voff) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
template <> __device__ __forceinline__ u32x4 as_load16<64>(const char* sbase, uint32_t voff) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
template <int IMM> __device__ __forceinline__ uint32_t as_load1(const char* sbase, uint32_t voff);
#define AS_LOAD1(IMMv)                                                                                                                    \
    template <> __device__ __forceinline__ uint32_t as_load1<IMMv>(const char* sbase, uint32_t voff) {                                   \
        uint32_t r;                                                                                                                       \
        asm volatile("global_load_ubyte %0, %1, %2 offset:" #IMMv : "=v"(r) : "v"(voff), "s"(sbase) : "memory");                         \
        return r;                                                                                                                         \
    }
AS_LOAD1(0) AS_LOAD1(64) AS_LOAD1(128) AS_LOAD1(192)
#undef AS_LOAD1
// (every asm store ends with `s_nop 1`: a VALU write to the data registers of a > 8-byte VMEM store needs one wait state after it, and hipcc's
// hazard recogniser does not look inside an asm statement — without it one row in ~30000 came out with garbage in its first dword.)
// Stores with an SGPR base + 32-bit per-lane offset (hipcc otherwise keeps one 64-bit per-lane pointer per output tensor live across the
// whole loop, which is what pushed this kernel over 256 VGPRs).  boff: BYTE offset of the lane.
__device__ __forceinline__ void as_store16(const void* sbase, uint32_t boff, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(boff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ void as_store16_nt(const void* sbase, uint32_t boff, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(boff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ void as_store16_o16(const void* sbase, uint32_t boff, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:16\n\ts_nop 1" :: "v"(boff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ void as_store1(const void* sbase, uint32_t boff, uint32_t d) {
    asm volatile("global_store_byte %0, %1, %2\n\ts_nop 1" :: "v"(boff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ void as_store4(const void* sbase, uint32_t boff, uint32_t d) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(boff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ uint32_t as_load4(const char* sbase, uint32_t voff) {
    uint32_t r;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
template <int N> __device__ __forceinline__ void as_pinw(uint32_t& r);
template <> __device__ __forceinline__ void as_pinw<4>(uint32_t& r) { asm volatile("s_waitcnt vmcnt(4)" : "+v"(r) :: "memory"); }
template <typename OutT> __device__ __forceinline__ void as_store_row8(const OutT* sbase, uint32_t eoff, const float (&v)[8], bool nt = false) {
    if constexpr (sizeof(OutT) == 4) {
        as_store16(sbase, eoff * 4, (u32x4){__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[2]), __builtin_bit_cast(uint32_t, v[3])});
        as_store16_o16(sbase, eoff * 4, (u32x4){__builtin_bit_cast(uint32_t, v[4]), __builtin_bit_cast(uint32_t, v[5]), __builtin_bit_cast(uint32_t, v[6]), __builtin_bit_cast(uint32_t, v[7])});
    } else {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
        if (nt) as_store16_nt(sbase, eoff * 2, __builtin_bit_cast(u32x4, o));
        else as_store16(sbase, eoff * 2, __builtin_bit_cast(u32x4, o));
    }
}
template <int N> __device__ __forceinline__ void as_pin(u32x4 (&r)[4]);
template <> __device__ __forceinline__ void as_pin<4>(u32x4 (&r)[4]) { asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory"); }
template <int N> __device__ __forceinline__ void as_pin1(uint32_t (&r)[4]);
template <> __device__ __forceinline__ void as_pin1<4>(uint32_t (&r)[4]) { asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory"); }
template <> __device__ __forceinline__ void as_pin1<0>(uint32_t (&r)[4]) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory"); }
template <> __device__ __forceinline__ void as_pin<0>(u32x4 (&r)[4]) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory"); }

__device__ __forceinline__ void as_unpack8(const u32x4& r, float (&t)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[2 * i] = __builtin_bit_cast(float, r[i] << 16);
        t[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xFFFF0000u);
    }
}

// One 8-column group of one row.  Addresses are (wave-uniform 64-bit base) + (per-lane 32-bit element offset) so that every access uses the
// SGPR-base addressing form (no per-lane 64-bit pointers: the kernel lives at the 256-VGPR edge).
//   ub = m0 * ldc + n_tile0 (+32 h)   lo = row_in_wave * ldc + ecol           (C / aux_out / mul_aux / residual, elements)
//   nb = n_tile0 + 32 h               (bias / columns)                           db = m0 * N + nb, dl = row_in_wave * N + ecol (dropout index)
// pre_bits: the mask byte already in a register (inline-asm prefetch)
// Timing ablations of a -DEMO_DIAG build (EMO_GEMM_ABLATE; results are wrong): 1 no output stores, 2 no weight DMA, 3 no barrier, 4 no fragment
// reads, 5 no MFMAs, 6 no epilogue, 7 every block stores to the same 256 rows (output stays in the L2).  r05, 131072 rows, isolated launches:
// FFN2 dgrad (bit mask, N = 2048) 330 us; no MFMAs 316; no epilogue 220; no stores 223; stores kept in the L2 268 — the kernel is paced by
// its epilogue and by draining 537 MB of output (1.6 TB/s while it runs), not by the MFMA pipe: without a single MFMA it is 4 % faster.
// r06 (tools/astat_ablate.py, tools/astat_cycles.py on the full-line-store build; FFN1 forward with ReLU + dropout + mask, isolated launches of the
// diagnostics build): 431 us; no MFMAs 297; no epilogue 275 — the MFMA stages and the epilogues of the two waves of a SIMD hardly overlap (the column tile
// takes a wave 15.3 k cycles, 8.2 k of them in the epilogue whose VALU content is ~1.7 k); starting the second workgroup of a CU 2-12 k cycles out of
// phase changes nothing (362-370 us in the product build).
// The same with the two groups of a CU made one 8-wave workgroup that alternates stages and epilogue quarters at shared barriers
// ("ping-pong", built and measured in r05): 294 -> 314 us (FFN1 forward), 324 -> 349 (FFN2 dgrad): dropped.
// The epilogue is specialised at COMPILE time (FL = feature flags): with run-time `ep.*` tests the column-tile epilogue was ~650 lines of
// branchy code per tile (dead mul / residual / aux paths with their own `s_waitcnt vmcnt(0)`, per-lane parity branches of the dropout
// hash) and took 35-52 % of a wave's cycles (s_memtime, tools/astat_cycles.py); the flag sets the Performer step uses are straight-line.
enum { AF_RELU = 1, AF_DROP = 2, AF_RES = 4, AF_BITS = 8, AF_MASKOUT = 16, AF_GENERIC = 32, AF_DGELU = 64, AF_GELUAUX = 128, AF_HDIV = 256 };   // (DGELU, GELUAUX: GPT-2's MLP; HDIV: emo_hip.h hdiv)

// 8 consecutive elements starting at a multiple of 8 of a 32-bit linear index: two hashes + two xorshifts, no parity branch, no 64-bit
// arithmetic (bit-identical to drop_mult(); the launcher sends outputs of 2^32 elements or more to the generic epilogue)
__device__ __forceinline__ void as_drop8(const DropCtx& d, uint32_t idx0, float (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t h = emo_drop_hash(d, (idx0 >> 2) + q), h2 = emo_xs32(h);
        v[4 * q] *= (h & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 1] *= (h >> 16) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 2] *= (h2 & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 3] *= (h2 >> 16) >= d.thr16 ? d.scale : 0.f;
    }
}

template <typename OutT, int FL>
__device__ __forceinline__ void as_epi8(const EpiParams& ep, OutT* __restrict__ C, int64_t ub, uint32_t lo, int nb, int ecol, int64_t db, uint32_t dl,
                                        int64_t mb, uint32_t ml, float (&v)[8], const float* bias_lds, uint32_t pre_bits, uint32_t& mask_word, int mask_byte,
                                        const u32x4& pre_res, u32x4* defer = nullptr, u32x4* defer_aux = nullptr) {
    // (the bias is already in the accumulators: they START from it)
    constexpr bool G = (FL & AF_GENERIC) != 0;
    if ((FL & AF_GELUAUX) || (G && ep.aux_out)) {                // the pre-activation (gelu backward)
        bool stored = false;
        if constexpr (sizeof(OutT) == 2) {
            if (defer_aux) {
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
                *defer_aux = __builtin_bit_cast(u32x4, o);
                stored = true;
            }
        }
        if (!stored) as_store_row8<OutT>((const OutT*)ep.aux_out + ub, lo, v);
    }
    // ReLU + dropout (+ the 1-bit mask) of the FFN1 forward in ONE select per element (r06): keep = (v > 0) & (dropout field >= threshold) as a 64-bit lane
    // mask (two v_cmp + one SALU and), out = keep ? v * scale : 0 (v_mul + v_cndmask on the mask), mask bit shifted into the tile's word by ONE
    // v_addc_co_u32 (acc = 2 acc + keep): 5 VALU per element where max / cmp / cndmask / mul and cmp / cndmask / or took 7 — an epilogue instruction
    // costs a wave ~20 cycles next to the other wave's MFMA stages (tools/astat_cycles.py, profiles/r06_valu_mfma_overlap.txt).  Same bits as the
    // separate steps for every non-NaN v (v > 0 and kept <=> max(v, 0) * mult != 0).
    constexpr bool FUSED_RD = (FL & (AF_RELU | AF_DROP)) == (AF_RELU | AF_DROP) && !G && !(FL & (AF_HDIV | AF_BITS | AF_DGELU | AF_GELUAUX));      // (dropout alone keeps as_drop8: its packed multiplies are fewer instructions — 331 against 363 for the out-projection forward)
    if constexpr (FUSED_RD) {
        const DropCtx& d = ep.drop;
        const uint32_t idx0 = (uint32_t)db + dl;
        // the two 16-bit fields of a hash word are compared IN PLACE: high field >= thr <=> word >= thr << 16; low field by an SDWA compare of WORD_0
        uint32_t hw[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) { hw[2 * q] = emo_drop_hash(d, (idx0 >> 2) + q); hw[2 * q + 1] = emo_xs32(hw[2 * q]); }
        const uint32_t thr_hi = d.thr16 << 16;
        uint32_t thr_v = d.thr16;
        asm volatile("" : "+v"(thr_v));                               // (SDWA takes its second source from a VGPR)
#pragma unroll
        for (int i = 7; i >= 0; --i) {                               // (descending: the first element of the group ends in the byte's bit 0)
            uint64_t kd;
            if (i & 1) kd = __builtin_amdgcn_uicmp(hw[i >> 1], thr_hi, 35 /* uge */);
            else asm volatile("v_cmp_ge_u32_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:DWORD" : "=s"(kd) : "v"(hw[i >> 1]), "v"(thr_v));
            const uint64_t k = __builtin_amdgcn_fcmpf(v[i], 0.f, 10 /* ugt: a NaN stays a NaN where it is kept */) & kd;      // two v_cmp into SGPR pairs + s_and_b64
            const float sv = v[i] * d.scale;
            asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(v[i]) : "v"(sv), "s"(k));
            if constexpr ((FL & AF_MASKOUT) != 0) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(mask_word) : "s"(k) : "vcc");
        }
    } else if ((FL & AF_RELU) || (G && ep.act == EMO_ACT_RELU)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v[i]), 0));   // one v_max_i32: same bits as fmaxf(v, 0) for every non-NaN v (fmaxf = canonicalise + v_max_f32)
    } else if (FL & AF_GELUAUX) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_new_fast(v[i]);
    } else if (G && ep.act == EMO_ACT_GELU_NEW) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_new_o<OutT>(v[i]);
    }
    if (FL & AF_HDIV) {                                           // pre_bits = the row's divisor for this 64-column block (prefetched like the mask word)
        const float inv = 1.f / __builtin_bit_cast(float, pre_bits);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= inv;
    }
    if (FL & AF_BITS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= ((pre_bits >> i) & 1u) ? ep.mul_scale : 0.f;
    } else if ((FL & AF_DGELU) && !(FL & (AF_GENERIC | AF_RES)) && sizeof(OutT) == 2) {   // *= gelu_new'(pre-activation), the row group prefetched (pre_res)
        float t[8];
        as_unpack8(pre_res, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= dgelu_new_fast(t[i]);
    } else if (FL & AF_DGELU) {                                   // *= gelu_new'(pre-activation): straight-line (the generic instance ran this at 250 TFLOP/s)
        float t0[4], t1[4];
        Out4<OutT>::load((const OutT*)ep.mul_aux + ub + lo, t0);
        Out4<OutT>::load((const OutT*)ep.mul_aux + ub + lo + 4, t1);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] *= dgelu_new_fast(t0[i]); v[4 + i] *= dgelu_new_fast(t1[i]); }
    } else if (G && ep.mul_mode == EMO_MUL_BITMASK) {
        const uint32_t bits = (uint32_t)((const uint8_t*)ep.mul_aux)[mb + 4 * ml + mask_byte];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= ((bits >> i) & 1u) ? ep.mul_scale : 0.f;
    } else if (G && ep.mul_mode != EMO_MUL_NONE) {
        float t0[4], t1[4];
        Out4<OutT>::load((const OutT*)ep.mul_aux + ub + lo, t0);
        Out4<OutT>::load((const OutT*)ep.mul_aux + ub + lo + 4, t1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] *= (ep.mul_mode == EMO_MUL_NONZERO) ? (t0[i] != 0.f ? ep.mul_scale : 0.f) : dgelu_new_o<OutT>(t0[i]);
            v[4 + i] *= (ep.mul_mode == EMO_MUL_NONZERO) ? (t1[i] != 0.f ? ep.mul_scale : 0.f) : dgelu_new_o<OutT>(t1[i]);
        }
    }
    if constexpr (FUSED_RD) {
    } else if (FL & AF_DROP) as_drop8(ep.drop, (uint32_t)db + dl, v);
    else if (G && ep.drop.thr16) {
        float d0[4], d1[4];
        drop_mult4(ep.drop, (uint64_t)db + dl, d0);
        drop_mult4(ep.drop, (uint64_t)db + dl + 4, d1);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] *= d0[i]; v[4 + i] *= d1[i]; }
    }
    if constexpr (FUSED_RD) {                                     // (mask bits already shifted into mask_word: the caller reverses the four bytes)
    } else if ((FL & AF_MASKOUT) || (G && ep.mask_out)) {               // 1 bit per output: (value after act / dropout) != 0
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) bits |= (v[i] != 0.f ? 1u : 0u) << i;
        mask_word |= bits << (8 * mask_byte);                     // the tile's four bytes of this lane leave as ONE dword (caller)
    }
    if ((FL & AF_RES) && !G && sizeof(OutT) == 2) {               // the residual row group already in registers (inline-asm prefetch, r05)
        float t[8];
        as_unpack8(pre_res, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
    } else if ((FL & AF_RES) || (G && ep.residual)) {
        float t0[4], t1[4];
        Out4<OutT>::load((const OutT*)ep.residual + ub + lo, t0);
        Out4<OutT>::load((const OutT*)ep.residual + ub + lo + 4, t1);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] += t0[i]; v[4 + i] += t1[i]; }
    }
#ifdef EMO_DIAG
    if (ep.ablate == 1) return;                                   // diagnostics: no output stores
#endif
    if constexpr (sizeof(OutT) == 2) {
        if (defer) {                                              // full-line mode: the caller stores the two halves of a row pair-wise (see the tile epilogue)
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
            *defer = __builtin_bit_cast(u32x4, o);
            return;
        }
    }
    as_store_row8<OutT>(C + ub, lo, v, (ep.nt_store & 1) != 0);
}

// BITS: the 1-bit mask operand (EMO_MUL_BITMASK, 1 byte per 8 columns) is prefetched by inline-asm loads half a stage before the epilogue.
// RESP: the 16-B residual row groups of the bf16 straight-line instances likewise (r05: their registers are live from the last stage's middle to
// the end of the epilogue only, where one of the four B fragment sets is dead).
template <typename OutT, int FL>
__global__ __launch_bounds__(256, 2) void gemm_astat_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                           OutT* __restrict__ C, int64_t M, int64_t N, EpiParams ep, int tiles_per_block) {
    constexpr bool BITS = (FL & AF_BITS) != 0;
    constexpr bool HDIV = (FL & AF_HDIV) != 0;                    // two divisor words per lane and column tile (rows lane % 16 and + 16), fetched like the mask word
    // residual rows (AF_RES) or pre-activation rows (AF_DGELU) of the column tile prefetched like the mask word
    constexpr bool RESP = ((FL & AF_RES) != 0 || (FL & (AF_DGELU | AF_RES)) == AF_DGELU) && (FL & AF_GENERIC) == 0 && sizeof(OutT) == 2;
    // VMEM operations of a column tile's epilogue when they are all inline-asm stores (no load the compiler would wait for): 4 output stores, the mask
    // word, 4 pre-activation stores
    constexpr bool RELAX = sizeof(OutT) == 2 && (FL & AF_GENERIC) == 0 && (FL & (AF_DGELU | AF_RES)) != (AF_DGELU | AF_RES);
    constexpr int NE = 4 + ((FL & AF_MASKOUT) ? 1 : 0) + ((FL & AF_GELUAUX) ? 4 : 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [4 slots x 16 KB ring][bias: N floats]
    float* bias_lds = (float*)(smem + AS_RING);                   // bias (or zeros): the accumulators of every column tile start from it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * AS_BM + wave * 32;
    // column split (r05, token counts below one panel per CU — the reference YAML's batch size 4): blockIdx.y sweeps only the column tiles
    // nt0 .. nt0 + n_tiles - 1 of its row panel, so that M / 128 panels still give every CU a block (the panel is re-read from the L2)
    const int n_tiles_all = (int)(N / AS_BN), n_tiles = tiles_per_block, nt0 = (int)blockIdx.y * tiles_per_block;
#ifdef EMO_DIAG
    const uint64_t t_entry = __builtin_readcyclecounter();
#endif

    // bias -> registers first (oldest VMEM ops), -> LDS before the loop
    float bv[AS_MAXN / 256];
#pragma unroll
    for (int q = 0; q < AS_MAXN / 256; ++q) bv[q] = (ep.bias && tid + 256 * q < n_tiles * AS_BN) ? ep.bias[nt0 * AS_BN + tid + 256 * q] : 0.f;

    // ---- the wave's 32 x 512 slice of A, directly in MFMA operand layout (lane: row lane%16, 8 consecutive k at 8*(lane/16))
    bf16x8 a[2][16];
    {
        const bf16_t* ap = A + (m0 + (lane & 15)) * lda + (lane >> 4) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                a[i][ks] = ep.a_nt ? __builtin_nontemporal_load((const bf16x8*)(ap + (int64_t)i * 16 * lda + ks * 32)) : *(const bf16x8*)(ap + (int64_t)i * 16 * lda + ks * 32);
    }
    // ---- per-lane constants of the weight stream
    uint32_t src[4];                                              // byte offsets of this lane's four 16-B pieces of a slot (source side, swizzled)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int row = 4 * (wave * 4 + jj) + (lane >> 4);
        src[jj] = (uint32_t)((row * ldb + (((lane & 15) ^ as_swz(row)) << 3)) * 2);
    }
    uint32_t rd[4];                                               // byte offset inside a slot of fragment f at k-step 0; k-step ks: ^ (ks << 6)
#pragma unroll
    for (int f = 0; f < 4; ++f) {                                 // ((4 ks + c) ^ z) << 4 == (((c ^ z) << 4)) ^ (ks << 6): the k step only flips bits 6-7
        const int row = as_nrow(f, lane & 15);
        rd[f] = (uint32_t)(row * 256 + (((lane >> 4) ^ as_swz(row)) << 4));
    }
    const char* gB = (const char*)(B + (int64_t)nt0 * AS_BN * ldb);   // wave-uniform: start of the NEXT stage to issue
    const int64_t tile_step = (int64_t)AS_BN * ldb * 2 - 3 * AS_KS * 2;
    // LDS-DMA as inline asm with the SGPR-base addressing form (wave-uniform stage pointer + the lane's 32-bit offset): the builtin takes a
    // per-lane 64-bit pointer, which cost two v_lshl_add_u64 per piece and kept the four offsets zero-extended in eight VGPRs (r05: the first
    // thing hipcc spilled when the residual prefetch was added — and a reload is a scratch_load + s_waitcnt vmcnt(0) inside the ring).
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem) + wave * 4096;
    auto issue = [&](int slot) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(src[jj]), "s"(gB), "s"(ring_lds + slot * AS_SLOT + jj * 1024) : "memory");
    };
    auto frags = [&](int slot, int ks, bf16x8 (&bf)[4]) {
#ifdef EMO_DIAG
        if (ep.ablate == 4) return;                               // diagnostics: no fragment reads (timing only)
#endif
#pragma unroll
        for (int f = 0; f < 4; ++f) bf[f] = *(const bf16x8*)(smem + slot * AS_SLOT + (rd[f] ^ (uint32_t)(ks << 6)));
    };
    // prologue: stages 0..2 (always exist: T >= 4)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        issue(s);
        gB += AS_KS * 2;
    }
    int issued = 3;                                               // stages issued so far; stage s lives in slot s & 3 (4 stages per column tile)
#pragma unroll
    for (int q = 0; q < AS_MAXN / 256; ++q)
        if (tid + 256 * q < n_tiles * AS_BN) bias_lds[tid + 256 * q] = bv[q];
    float* lna_gb = (float*)(smem + 3 * AS_SLOT);                 // gamma [512] | beta [512]: the ring's last slot is not filled before the loop's first barrier
    if (ep.lna_gamma) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { lna_gb[tid + 256 * q] = ep.lna_gamma[tid + 256 * q]; lna_gb[AS_K + tid + 256 * q] = ep.lna_beta[tid + 256 * q]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    as_wait<8>();                                                 // A fragments + stage 0 landed (stages 1, 2 may be in flight)
    // (hipcc does not see the asm DMA or this wait: without the pin below it answers the FIRST use of every A fragment inside the unrolled
    // loop body with its own countdown `s_waitcnt vmcnt(15) .. vmcnt(0)` — executed again in every column tile, draining the ring each time)
#pragma unroll
    for (int i = 0; i < 2; ++i)
        asm volatile("" : "+v"(a[i][0]), "+v"(a[i][1]), "+v"(a[i][2]), "+v"(a[i][3]), "+v"(a[i][4]), "+v"(a[i][5]), "+v"(a[i][6]), "+v"(a[i][7]), "+v"(a[i][8]),
                     "+v"(a[i][9]), "+v"(a[i][10]), "+v"(a[i][11]), "+v"(a[i][12]), "+v"(a[i][13]), "+v"(a[i][14]), "+v"(a[i][15]));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ep.lna_gamma) {
        // LayerNorm of the A rows in place (emo_hip.h: lna_*): the wave's two row fragments hold complete 512-wide rows — row lane % 16 (+ 16 i), its
        // 512 elements in the four lanes lane % 16 + 16 j.  The normalised rows and the statistics leave from column block 0 only.
        // Statistics on the MFMA pipe (the VALU form — unpack + add, unpack + centre + square over 256 values per lane — was two thirds of this block's
        // 10 k cycles): row sums = ones . A^T (every lane of column `row` gets the sum), sums of squares = the diagonal of the Gram matrix A A^T
        // (lane group row / 4, register row % 4).  bf16 x bf16 products are exact in fp32; variance = E[x^2] - mean^2, clamped at 0.
        float mean_[2], rstd_[2];
        {
            const bf16_t one_b = (bf16_t)1.f;
            bf16x8 ones = {one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b};
            asm volatile("" : "+v"(ones));
            const int dr = lane & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, gacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i][ks], sacc, 0, 0, 0);
                    gacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], a[i][ks], gacc, 0, 0, 0);
                }
                const float dg = dr == 0 ? gacc[0] : dr == 1 ? gacc[1] : dr == 2 ? gacc[2] : gacc[3];
                const float sq = as_sum_lane_rows((lane >> 4) == ((lane & 15) >> 2) ? dg : 0.f);
                mean_[i] = sacc[0] * (1.f / AS_K);
                rstd_[i] = rsqrtf(fmaxf(sq * (1.f / AS_K) - mean_[i] * mean_[i], 0.f) + ep.ln_eps);
            }
        }
        const bool writer = blockIdx.y == 0;
        bf16_t* lo = (bf16_t*)ep.lna_out + (m0 + (lane & 15)) * AS_K + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f32x4 g0 = *(const f32x4*)(lna_gb + ks * 32 + (lane >> 4) * 8), g1 = *(const f32x4*)(lna_gb + ks * 32 + (lane >> 4) * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(lna_gb + AS_K + ks * 32 + (lane >> 4) * 8), b1 = *(const f32x4*)(lna_gb + AS_K + ks * 32 + (lane >> 4) * 8 + 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    y[e] = (bf16_t)(((float)a[i][ks][e] - mean_[i]) * rstd_[i] * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]));
                a[i][ks] = y;
                if (writer) *(bf16x8*)(lo + (int64_t)i * 16 * AS_K + ks * 32) = y;
            }
            __builtin_amdgcn_sched_barrier(0);                    // one k step at a time: the gamma / beta reads of later steps are not hoisted into the register peak
        }
        if (writer && lane < 16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { ep.lna_mean[m0 + 16 * i + lane] = mean_[i]; ep.lna_rstd[m0 + 16 * i + lane] = rstd_[i]; }
        }
    }
    // B fragments: 4 rotating register sets (step g of a column tile computes with set g & 3 and prefetches step g + 2 into set (g + 2) & 3,
    // so three sets are live: the LDS round trip is covered by 16 MFMAs of the wave itself, not only by the partner wave)
    bf16x8 bq[4][4];
    frags(0, 0, bq[0]);
    frags(0, 1, bq[1]);
    // Loop invariant at the top of stage s: stage s has landed for every wave and its first fragments are in `bf`; stages s+1, s+2
    // are in flight.  The wait + barrier + refill for stage s+1 sits in the MIDDLE of stage s (between its 2nd and 3rd k step), so no
    // LDS-latency bubble follows a barrier, and the first fragments of stage s+1 are read under the last MFMAs of stage s.
    // The body is branch-free: past the last stage the refill re-fetches the last column tile (harmless, drained before exit) so that
    // every wait count is a constant.
    const int ecol = 8 * (lane >> 4);
    // 1-bit mask (mask_out / EMO_MUL_BITMASK), TILED layout private to this kernel: the 256 mask bytes of a wave's 32-row x 64-column tile are
    // contiguous — block ((m / 32) * (N / 64) + n / 64) * 256, byte 4 lane + (2 i + h) for row 16 i + (lane & 15), columns 32 h + 8 (lane >> 4) .. +7
    // — so a lane's four bytes of a tile are ONE dword and the wave writes / reads the tile's 256 bytes with ONE store / load instruction
    // (r05; r03-r04: byte (2 i + h) * 64 + lane, four byte stores / loads of 64-byte runs per tile.  The r05 counters show the same write
    // traffic as before — 727 MB against 570 MB algorithmic — so the excess the r04 counters showed is NOT the byte stores: it comes with the
    // non-temporal output stores this instance uses, 1.31 x on the 537-MB output, see the launcher).  ops.bitmask_rows() converts to the row-major view for tests.
    const int64_t mtile0 = ((m0 >> 5) * (int64_t)n_tiles_all + nt0) * 256;
    // per-lane offsets of row (lane & 15) only: the 16-row step of the second row fragment goes into the wave-uniform (scalar) part of the
    // address — two loop-invariant VGPRs fewer (at 256 VGPRs a spilled one comes back as scratch_load + s_waitcnt vmcnt(0) = a drained DMA ring)
    const uint32_t eoff0 = (uint32_t)((lane & 15) * ep.ldc + ecol);                                                              // elements
    const uint32_t doff0 = (uint32_t)((lane & 15) * N + ecol);
    const uint32_t floff0 = (uint32_t)((lane & 7) * ep.ldc + ecol + ((lane >> 3) & 1) * 32);      // full-line mode: row lane % 8 (+ 8), half (lane / 8) % 2
    // hdiv (emo_hip.h): divisor of row m and column block j at ((m / T) * (N / 64) + j) * T + m % T; a wave's 32 rows share m / T (T % 32 == 0)
    const int64_t hbase = HDIV ? ((m0 / ep.hdiv_T) * n_tiles_all + nt0) * ep.hdiv_T + (m0 % ep.hdiv_T) : 0;
#ifdef EMO_DIAG
    uint64_t t_wait = 0, t_epi = 0, t_loop0 = __builtin_readcyclecounter();
#endif
    for (int nt = 0; nt < n_tiles; ++nt) {
        f32x4 acc[2][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {                             // accumulators start from the bias of their 4 columns (zeros in LDS without a bias)
            const f32x4 b4 = *(const f32x4*)(bias_lds + nt * AS_BN + 32 * (f >> 1) + 4 * (f & 1) + ecol);
            acc[0][f] = b4;
            acc[1][f] = b4;
        }
        uint32_t prew = 0, preh[2] = {0, 0};
        u32x4 pres[4] = {};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks == 2) {
#ifdef EMO_DIAG
                    const uint64_t tw0 = __builtin_readcyclecounter();
#endif
                    if (HDIV && kc == 3) {                        // the two divisors of this column tile (= head nt0 + nt): youngest VMEM ops at the wait below
                        const char* hp = (const char*)(ep.hdiv + hbase + (int64_t)nt * ep.hdiv_T);   // wave-uniform: the 32 rows' divisors of block (m / T, head nt0 + nt)
                        preh[0] = as_load4(hp, (uint32_t)(lane & 15) * 4);
                        preh[1] = as_load4(hp + 64, (uint32_t)(lane & 15) * 4);
                        as_wait<6>();
                    } else if (BITS && kc == 3) {                        // the mask word of this column tile: youngest VMEM op at the wait below
                        const char* op = (const char*)ep.mul_aux + mtile0 + nt * 256;                          // wave-uniform
                        prew = as_load4(op, (uint32_t)lane * 4);
                        as_wait<5>();
                    } else if (RESP && kc == 2) {                 // the residual rows of this column tile (4 x 16 B per lane), 2.5 stages before their use
                        const char* rp = (const char*)((FL & AF_RES) ? ep.residual : ep.mul_aux) + ((m0 * ep.ldc + (int64_t)(nt0 + nt) * AS_BN) << 1);   // (wave-uniform).  In the epilogue each of
                        const uint32_t vo = eoff0 * 2;            // the four plain loads was a full HBM round trip (tools/astat_cycles.py: 13.7 k cycles per
                        pres[0] = as_load16<0>(rp, vo);           // column tile against 2 k of MFMA issue).  vmcnt retires in order, so the lead cannot
                        pres[1] = as_load16<64>(rp, vo);          // exceed the ring: the stage issued behind these loads is needed 2 stages later
                        pres[2] = as_load16<0>(rp + ((16 * ep.ldc) << 1), vo);
                        pres[3] = as_load16<64>(rp + ((16 * ep.ldc) << 1), vo);
                        as_wait<8>();
                    } else if (RESP && kc == 3) {
                        as_wait<8>();                             // stage s+1 landed; behind it: the residual rows and stage s+2
                    } else if (RELAX && kc < 2 && nt > 0) {
                        // the two waits behind an epilogue: its NE output stores are YOUNGER than the stage these waits are for and may stay in flight too.
                        // (vmcnt retires in order: with vmcnt(4) here the first wait behind an epilogue also waited for the stage issued half a stage
                        // before it and — with a mask or second output — for the first stores, the second one for every store: the stores had half a
                        // stage to one stage to be acknowledged; now two and a half)
                        as_wait<4 + NE>();
                    } else {
#ifdef EMO_DIAG
                        if (ep.ablate != 2)
#endif
                        as_wait<4>();                             // stage s+1 landed; stage s+2 (4 DMA ops) may stay in flight
                    }
#ifdef EMO_DIAG
                    if (ep.ablate != 3)
#endif
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#ifdef EMO_DIAG
                    t_wait += __builtin_readcyclecounter() - tw0;
                    if (ep.ablate != 2)
#endif
                    issue((kc + 3) & 3);                          // refill the slot of stage s-1: every wave is past it
                    ++issued;
                    gB += ((issued & 3) != 0) ? (int64_t)(AS_KS * 2) : ((issued >> 2) < n_tiles ? tile_step : (int64_t)(-3 * AS_KS * 2));
                }
                {
                    constexpr int dummy = 0; (void)dummy;
                    const int g2 = kc * 4 + ks + 2;              // step to prefetch (compile-time after unrolling); steps 16, 17 = the next column tile's 0, 1
                    frags((g2 >> 2) & 3, g2 & 3, bq[g2 & 3]);
                }
#ifdef EMO_DIAG
                if (ep.ablate != 5)                               // diagnostics: no MFMAs
#endif
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks][f], a[i][kc * 4 + ks], acc[i][f], 0, 0, 0);
            }
        }
        // ---- the wave's 32 x 64 outputs of this column tile, straight from the accumulators
#ifdef EMO_DIAG
        const uint64_t te0 = __builtin_readcyclecounter();
#endif
#ifdef EMO_DIAG
        if (ep.ablate == 6) continue;                             // diagnostics: no epilogue at all
#endif
        if (BITS) as_pinw<4>(prew);                               // one refill (4 DMA ops) was issued after the mask word
        if (HDIV) { as_pinw<4>(preh[0]); as_pinw<4>(preh[1]); }
        if (RESP) as_pin<4>(pres);
        uint32_t mask_word = 0;
        constexpr bool full_line = sizeof(OutT) == 2 && !(FL & AF_GENERIC);      // bf16 straight-line instances: full-line output stores (below)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 piece[2], piece_aux[2];
            constexpr bool full_aux = full_line && (FL & AF_GELUAUX) != 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8] = {acc[i][2 * h][0], acc[i][2 * h][1], acc[i][2 * h][2], acc[i][2 * h][3],
                              acc[i][2 * h + 1][0], acc[i][2 * h + 1][1], acc[i][2 * h + 1][2], acc[i][2 * h + 1][3]};
                const int nb = (nt0 + nt) * AS_BN + 32 * h;
                uint32_t lo = eoff0;
                asm volatile("" : "+v"(lo));                     // opaque per tile: keeps (base + lane offset) out of loop-invariant 64-bit VGPR pointers
#ifdef EMO_DIAG
                const int64_t m0s = ep.ablate == 7 ? (m0 & 255) : m0;       // diagnostics: every block writes the same 256 rows (the output stays in the L2)
#else
                const int64_t m0s = m0;
#endif
                as_epi8<OutT, FL>(ep, C, (m0s + 16 * i) * ep.ldc + nb, lo, nb, ecol, (m0 + 16 * i) * N + nb, doff0, mtile0 + nt * 256, (uint32_t)lane, v, bias_lds,
                                    HDIV ? preh[i] : (prew >> (8 * (i * 2 + h))) & 0xFFu, mask_word, i * 2 + h, pres[i * 2 + h], full_line ? &piece[h] : nullptr, full_aux ? &piece_aux[h] : nullptr);
                __builtin_amdgcn_sched_barrier(0);               // one 8-column group at a time: keeps the epilogue's temporaries out of the register peak
            }
            if constexpr (sizeof(OutT) == 2) {
#ifdef EMO_DIAG
                if (ep.ablate == 1) continue;                     // diagnostics: no output stores
#endif
                if constexpr (full_line) {
                    // FULL-LINE stores (r06): a lane's two 16-B pieces of a row (columns 8 g .. and 32 + 8 g ..) sit 64 B apart, so each of the two store
                    // instructions of a row tile wrote 16 HALF lines.  Lanes r and r + 8 of a 16-lane row (rows r, r + 8 of the tile) swap one piece (DPP
                    // row_ror:8): the first instruction then writes rows 0-7 complete (8 lanes x 16 B = the 128-B line of the tile row), the second rows
                    // 8-15.  Same-box in the training step (EMO_ASTAT_FULL 0 / 1 of the run-time-switched build): K = 512 class 13.78 -> 13.32 ms,
                    // step 44.67 -> 44.38 ms.
                    const bool lowr = (lane & 8) == 0;
                    u32x4 X, R, SA, SB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        X[e] = lowr ? piece[1][e] : piece[0][e];
                        R[e] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[e], 0x128, 0xF, 0xF, true);
                        SA[e] = lowr ? piece[0][e] : R[e];
                        SB[e] = lowr ? R[e] : piece[1][e];
                    }
                    uint32_t la = floff0;
                    asm volatile("" : "+v"(la));
                    const OutT* cb = C + (m0 + 16 * i) * ep.ldc + (int64_t)(nt0 + nt) * AS_BN;
                    if (ep.nt_store & 1) { as_store16_nt(cb, la * 2, SA); as_store16_nt(cb, (la + 8 * (uint32_t)ep.ldc) * 2, SB); }
                    else { as_store16(cb, la * 2, SA); as_store16(cb, (la + 8 * (uint32_t)ep.ldc) * 2, SB); }
                    if constexpr (full_aux) {                     // the pre-activation rows of the GPT-2 MLP likewise
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            X[e] = lowr ? piece_aux[1][e] : piece_aux[0][e];
                            R[e] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[e], 0x128, 0xF, 0xF, true);
                            SA[e] = lowr ? piece_aux[0][e] : R[e];
                            SB[e] = lowr ? R[e] : piece_aux[1][e];
                        }
                        const OutT* ab = (const OutT*)ep.aux_out + (m0 + 16 * i) * ep.ldc + (int64_t)(nt0 + nt) * AS_BN;
                        as_store16(ab, la * 2, SA);
                        as_store16(ab, (la + 8 * (uint32_t)ep.ldc) * 2, SB);
                    }
                }
            }
        }
        if constexpr ((FL & (AF_RELU | AF_DROP | AF_MASKOUT)) == (AF_RELU | AF_DROP | AF_MASKOUT) && !(FL & (AF_GENERIC | AF_HDIV | AF_BITS | AF_DGELU | AF_GELUAUX)))
            mask_word = __builtin_bswap32(mask_word);           // the fused ReLU + dropout path shifted the groups in: group 0 sits in the top byte
        if ((FL & AF_MASKOUT) || ((FL & AF_GENERIC) && ep.mask_out)) as_store4(ep.mask_out + mtile0 + nt * 256, (uint32_t)lane * 4, mask_word);
#ifdef EMO_DIAG
        t_epi += __builtin_readcyclecounter() - te0;
#endif
    }
#ifdef EMO_DIAG
    if (ep.ablate == 8 && ep.rln_stats && lane == 0) {
        unsigned long long* dg = (unsigned long long*)ep.rln_stats;   // (diagnostics: the otherwise unused rln_stats pointer carries the counter buffer)
        atomicAdd(dg + 0, (unsigned long long)t_wait);
        atomicAdd(dg + 1, (unsigned long long)t_epi);
        atomicAdd(dg + 2, (unsigned long long)(__builtin_readcyclecounter() - t_loop0));
        atomicAdd(dg + 3, 1ull);
        atomicAdd(dg + 4, (unsigned long long)(t_loop0 - t_entry));       // panel load + prologue
    }
#endif
    as_wait<0>();                                                 // the run-ahead refills past the last stage must land before the LDS is released
}
}  // namespace

bool emo_gemm_astat_try(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int dtype_out, int64_t M, int64_t N, int64_t K,
                        const EpiParams& ep_in, hipStream_t st) {
    EpiParams ep = ep_in;
    const bool off = getenv("EMO_GEMM_NO_ASTAT") != nullptr;      // (read per call: the parity test toggles it in-process)
    // one block owns a 128-row panel and sweeps all of N: the grid is M / 128 blocks.  Below one panel per CU (the reference's batch_size 4:
    // 64 panels) the column tiles of a panel are split over blockIdx.y, every block keeping at least two column tiles (r05; until then these
    // shapes ran on the 128 x 128 tiling at 0.15 of the MFMA peak)
    int64_t min_rows = (int64_t)AS_BM * AS_MIN_BLOCKS;
    { const char* e = getenv("EMO_ASTAT_MIN_ROWS"); if (e && atoll(e) > 0) min_rows = atoll(e); }      // (A/B of the column split; ops.py reads the same variable)
    if (off || K != AS_K || (M % AS_BM) != 0 || M < min_rows || (N % AS_BN) != 0 || N > AS_MAXN || N < AS_BN) return false;
    if (ep.atomic || ep.accumulate || ep.ws_stride || ep.a_rowsum || ep.b_rowsum || ep.ln_c1 || ep.rln_x) return false;
    if (ep.hdiv && dtype_out != EMO_BF16) return false;
    if ((ep.mask_out || ep.mul_mode == EMO_MUL_BITMASK) && dtype_out != EMO_BF16) return false;
    if (ep.act == EMO_ACT_GELU || ep.mul_mode == EMO_MUL_DGELU) return false;          // (exact erf GELU: the generic tiled epilogue only)
    if ((lda & 7) || (ldb & 7) || (ep.ldc & 7) || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15)) return false;
    if ((uint64_t)(AS_BN * ldb + AS_K) * 2 >= 0xFFFF0000ull) return false;
    const size_t lds = AS_RING + AS_MAXN * sizeof(float);
    const int n_tiles_all = (int)(N / AS_BN);
    int split = 1;
    // the smallest split that gives every CU one block — two for the long column sweeps (N >= 2048), where a lone block per CU leaves the MFMA pipe
    // without a partner wave for 32 column tiles (tools/bench_astat_split.py at 8192 / 16384 / 32768 rows; GPT-2 step at 32768 tokens, whose
    // products ran one block per CU: 19.0 -> 18.4 ms with two)
    const int64_t want_blocks = N >= 2048 ? 512 : 256;
    if (M / AS_BM < want_blocks)
        for (int sp = 2; sp <= n_tiles_all / 2; ++sp)
            if (n_tiles_all % sp == 0) {
                split = sp;
                if ((M / AS_BM) * sp >= want_blocks) break;
            }
    { const char* e = getenv("EMO_ASTAT_SPLIT"); if (e && atoi(e) > 0 && n_tiles_all % atoi(e) == 0) split = atoi(e); }      // (tests / experiments)
    const int tiles_per_block = n_tiles_all / split;
    dim3 grid((unsigned)(M / AS_BM), (unsigned)split);
    // Non-temporal output stores only for the FFN1 forward (the mask-out instance: 537 MB + the mask, read back once by the FFN2 forward).  r02
    // streamed every output beyond the 256-MB MALL; r03 per-instance sweep inside the step (EMO_ASTAT_NT_MASK, same box, 3 alternations):
    // all three large outputs 46.40 ms/step, FFN1 only 46.15, none 46.45 (the A-stationary kernels then run 0.9 ms faster and the kernels that
    // read the outputs 0.7 ms slower).  PMC: with nt stores WRITE_SIZE is 1.31 x the algorithmic bytes.
    ep.nt_store = (M * N * 2 >= (int64_t)256 << 20 && ep.mask_out) ? 1 : 0;
    { const char* e = getenv("EMO_ASTAT_NT"); if (e) ep.nt_store = atoi(e); }
    { const char* e = getenv("EMO_ASTAT_A_NT"); ep.a_nt = e ? atoi(e) : 0; }
    { const char* e = getenv("EMO_ASTAT_NT_MASK");                 // diagnostics: 1 = mask-out instance, 2 = bit-mask instance, 4 = the others (outputs >= 256 MB)
      if (e && M * N * 2 >= (int64_t)256 << 20) { const int m = atoi(e); ep.nt_store = (m & (ep.mask_out ? 1 : (ep.mul_mode == EMO_MUL_BITMASK ? 2 : 4))) ? 1 : 0; } }
#define AS_LAUNCH(OutT, FLv)                                                                                                              \
    do {                                                                                                                                  \
        auto k = gemm_astat_kernel<OutT, FLv>;                                                                                            \
        static bool attr = false;                                                                                                         \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }      \
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, A, lda, B, ldb, (OutT*)C, M, N, ep, tiles_per_block);                             \
    } while (0)
    // feature flags of this launch; the sets the Performer layer uses have their own straight-line instantiation, the rest runs the generic one
    const bool gelu_aux = ep.act == EMO_ACT_GELU_NEW && ep.aux_out;
    int fl = (ep.act == EMO_ACT_RELU ? AF_RELU : 0) | (ep.drop.thr16 ? AF_DROP : 0) | (ep.residual ? AF_RES : 0) | (ep.mul_mode == EMO_MUL_BITMASK ? AF_BITS : 0) |
             (ep.mask_out ? AF_MASKOUT : 0) | (ep.mul_mode == EMO_MUL_DGELU_NEW ? AF_DGELU : 0) | (gelu_aux ? AF_GELUAUX : 0);
    const bool other = (!gelu_aux && (ep.aux_out || ep.act == EMO_ACT_GELU_NEW)) || ep.mul_mode == EMO_MUL_NONZERO ||
                       (ep.drop.thr16 && (uint64_t)M * (uint64_t)N >= (1ull << 32));
    if (ep.hdiv) AS_LAUNCH(bf16_t, AF_HDIV);                       // (emo_gemm admits hdiv only with a plain epilogue)
    else if (dtype_out == EMO_F32) AS_LAUNCH(float, AF_GENERIC);
    else if (other) AS_LAUNCH(bf16_t, AF_GENERIC);
    else if (fl == 0) AS_LAUNCH(bf16_t, 0);
    else if (fl == AF_RELU) AS_LAUNCH(bf16_t, AF_RELU);
    // (relu + dropout WITHOUT the mask output — no caller in the training or decoding paths — runs the generic instance: its straight-line
    // instantiation no longer fits 256 VGPRs without an in-loop reload, which drains the ring)
    else if (fl == (AF_RELU | AF_DROP | AF_MASKOUT)) AS_LAUNCH(bf16_t, AF_RELU | AF_DROP | AF_MASKOUT);
    else if (fl == AF_RES) AS_LAUNCH(bf16_t, AF_RES);
    else if (fl == (AF_DROP | AF_RES)) AS_LAUNCH(bf16_t, AF_DROP | AF_RES);
    else if (fl == AF_BITS) AS_LAUNCH(bf16_t, AF_BITS);
    else if (fl == AF_DGELU) AS_LAUNCH(bf16_t, AF_DGELU);
    else if (fl == AF_GELUAUX) AS_LAUNCH(bf16_t, AF_GELUAUX);
    else AS_LAUNCH(bf16_t, AF_GENERIC);
#undef AS_LAUNCH
    return true;
}
