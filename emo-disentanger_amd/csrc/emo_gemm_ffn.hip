// Fused feed-forward block of the post-LN encoder layer (r06; r05 verdict item 1):
//   h1 = LayerNorm1(x1);  f = dropout(relu(h1 W1^T + b1));  x2 = h1 + dropout(f W2^T + b2)
// in ONE kernel — replaces `norm1 -> linear1 -> activation -> dropout -> linear2 -> dropout -> residual` of fast-transformers'
// TransformerEncoderLayer.forward as called from stage2_accompaniment/model/fast_transformer_decoder.py:45-51, i.e. the A-stationary FFN1 launch
// (gemm_astat_kernel<bf16,19>) AND the 256 x 256-tile FFN2 launch (gemm_w128_kernel) of the training step.  The hidden activation f is still WRITTEN
// (the FFN2 weight gradient and the 1-bit mask of the FFN2 dgrad need it) but never re-read in the forward, and the residual h1 never leaves the
// registers: per layer 671 MB less HBM traffic (FFN2's read of f and of h1).
//
// Shape of the kernel (the only one that keeps the A-stationary kernel's LDS traffic per flop: every weight fragment a wave reads feeds two MFMAs):
//   * a workgroup = 4 waves owns 128 rows, ONE workgroup per CU (one wave per SIMD, 512 registers per lane): a wave holds its 32 rows x 512 k of
//     h1 as MFMA operand fragments (128 VGPRs, normalised in place exactly as gemm_astat_kernel's lna path) AND its 32 x 512 fp32 output tile
//     (64 fragments = 256 accumulation registers, pinned with "+a" as in emo_gemm_w128.hip);
//   * the hidden dimension is swept in 32 chunks of 64 columns.  Phase 1 of a chunk = the A-stationary product against 64 rows of W1 (4 stages of
//     128 k); its 32 x 64 fp32 result gets bias (accumulators start from it), ReLU, dropout, the 1-bit mask, one rounding to bf16, the store of f —
//     and IS, register for register, the B operand of phase 2 (a lane owns 8 consecutive hidden columns = one 16-byte k-chunk of a 32-deep
//     step): out[32 x 512] += f_chunk[32 x 64] . W2[:, chunk]^T against 4 stages of 128 output columns x 64 k;
//   * both weight streams go through ONE 8-slot LDS ring of 16-KB stages (LDS-DMA as inline asm, counted vmcnt, one raw s_barrier per stage placed
//     between its 2nd and 3rd sub-step, refill 7 stages ahead); a chunk is exactly 8 stages, so every slot index is a compile-time constant;
//     every stage is 4 sub-steps of 4 fragment reads + 8 MFMAs, fragments rotate through 4 register sets two sub-steps ahead across stage AND
//     phase boundaries;
//   * the final epilogue adds b2, the output dropout and the residual straight from the h1 fragments (a lane's 8 consecutive output columns of
//     column tile t, half h are the 8 k-values of its fragment 2 t + h).
// Dropout masks, mask-bit layout and statistics are those of the unfused kernels (same (seed, offset, m * N + n) hashes): the backward is unchanged.
#include "emo_gemm_epi.h"

namespace {
constexpr int FF_D = 512, FF_H = 2048, FF_BM = 128, FF_STAGE = 16384, FF_SLOTS = 8, FF_RING = FF_SLOTS * FF_STAGE;
constexpr int FF_NCH = FF_H / 64;                              // 32 hidden chunks
#ifndef FF_ABL
#define FF_ABL 0                                               // timing ablations (wrong results): 1 no chunk-epilogue math / stores, 2 no phase-2 MFMAs, 3 no MFMAs, 4 no f / mask stores
#endif
#ifndef FF_PF
#define FF_PF 2                                                // fragment prefetch distance in sub-steps
#endif

__device__ __forceinline__ int ff_swz1(int row) { return (row & 3) | ((row >> 1) & 12); }          // phase-1 image (256-B rows): gemm_astat_kernel's map
__device__ __forceinline__ int ff_nrow(int f, int i) { return 32 * (f >> 1) + 8 * (i >> 2) + 4 * (f & 1) + (i & 3); }
__device__ __forceinline__ int ff_swz2(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }   // phase-2 image (128-B rows): gemm_w128_kernel's B map
__device__ __forceinline__ float ff_sum_lane_rows(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t w = __builtin_bit_cast(uint32_t, y);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}
template <int N> __device__ __forceinline__ void ff_wait();
template <> __device__ __forceinline__ void ff_wait<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void ff_wait<20>() { asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
template <> __device__ __forceinline__ void ff_wait<24>() { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
// accumulators: phase 1 in ordinary VGPRs, phase 2 pinned to the accumulation registers (see emo_gemm_w128.hip: w_mma / w_mma_v)
__device__ __forceinline__ void ff_mma_v(f32x4& c, const bf16x8& a, const bf16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void ff_mma_a(f32x4& c, const bf16x8& a, const bf16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void ff_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
__device__ __forceinline__ void ff_store16(const void* sbase, uint32_t boff, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(boff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ void ff_store16_nt(const void* sbase, uint32_t boff, u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(boff), "v"(d), "s"(sbase) : "memory");
}
__device__ __forceinline__ void ff_store4(const void* sbase, uint32_t boff, uint32_t d) { asm volatile("global_store_dword %0, %1, %2" ::"v"(boff), "v"(d), "s"(sbase) : "memory"); }
// 8 consecutive elements starting at a multiple of 8 of a 32-bit linear index (bit-identical to drop_mult(); gemm_astat_kernel's as_drop8)
__device__ __forceinline__ void ff_drop8(const DropCtx& d, uint32_t idx0, float (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t h = emo_drop_hash(d, (idx0 >> 2) + q), h2 = emo_xs32(h);
        v[4 * q] *= (h & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 1] *= (h >> 16) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 2] *= (h2 & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 3] *= (h2 >> 16) >= d.thr16 ? d.scale : 0.f;
    }
}

#ifdef EMO_DIAG
__device__ unsigned long long emo_ffn_diag[16];            // diagnostics build only (tools/ffn_cycles.py): cycle sums over all waves
#define FFD(k) do { const uint64_t tn_ = __builtin_readcyclecounter(); tc_[k] += tn_ - ts_; ts_ = tn_; } while (0)
#else
#define FFD(k) do {} while (0)
#endif

struct FfnArgs {
    const bf16_t* x1; const float* gamma; const float* beta; float ln_eps;
    const bf16_t* W1; const float* b1; const bf16_t* W2; const float* b2;
    bf16_t* h1; float* mean; float* rstd; bf16_t* f; uint8_t* mask; bf16_t* x2;
    int64_t M; DropCtx drop_f, drop_y; int nt_store;
};

__global__ __launch_bounds__(256, 1) void ffn_fused_fwd_kernel(FfnArgs a_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [8 x 16 KB ring][b1: 2048 floats][b2: 512 floats]
    float* b1_lds = (float*)(smem + FF_RING);
    float* b2_lds = b1_lds + FF_H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * FF_BM + wave * 32;
    const int g = lane >> 4, r16 = lane & 15;
#ifdef EMO_DIAG
    uint64_t tc_[8] = {}, ts_ = __builtin_readcyclecounter(), tstart_ = ts_;
#endif

    // ---- the wave's 32 x 512 slice of x1 in MFMA operand layout (lane: row lane % 16 (+ 16 i), 8 consecutive k at 32 ks + 8 (lane / 16))
    bf16x8 a[2][16];
    {
        const bf16_t* ap = a_.x1 + (m0 + r16) * FF_D + g * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) a[i][ks] = *(const bf16x8*)(ap + (int64_t)i * 16 * FF_D + ks * 32);
    }
    // ---- per-lane constants of the two weight streams
    uint32_t src1[4], src2[4];                                    // byte offsets of this lane's four 16-B pieces of a stage (source side, swizzled)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int row1 = 4 * (wave * 4 + jj) + g;                 // phase 1: 64 rows x 16 chunks
        src1[jj] = (uint32_t)((row1 * FF_D + ((r16 ^ ff_swz1(row1)) << 3)) * 2);
        const int p = (wave * 4 + jj) * 64 + lane, row2 = p >> 3;   // phase 2: 128 rows x 8 chunks
        src2[jj] = (uint32_t)((row2 * FF_H + (((p & 7) ^ ff_swz2(row2)) << 3)) * 2);
    }
    // Fragment addresses = ONE lane register per (phase, k-step) + a compile-time offset (slot, fragment): both swizzles depend on the lane only
    // (ff_swz1 looks at row bits 0, 1, 3, 4 and ff_swz2 at bits 1, 3, 4; the fragment index moves bits 2 and 5, the tile bit 6), and the k-step
    // only flips bits 6-7 of the byte offset.  (Written as rd[f] ^ (u << 6) per read, hipcc hoists all 18 combinations out of the chunk loop into
    // registers: the first build spilled an A fragment and reloaded it — scratch_load + s_waitcnt vmcnt(0) = a drained DMA ring — once per chunk.)
    // Phase 1 lives in slots 0-3 and phase 2 in slots 4-7, so every literal stays below the 64-KB reach of a ds_read offset.
    // (The k-step XOR is applied per sub-step to an opaque copy of the lane register: one VALU instruction, one transient register, instead of six
    // loop-invariant ones — the kernel sits at the 256-VGPR edge like the A-stationary kernel.)
    const int row1_ = ff_nrow(0, r16);
    const uint32_t rd1 = (uint32_t)(row1_ * 256 + ((g ^ ff_swz1(row1_)) << 4));
    const uint32_t rd2 = (uint32_t)(4 * FF_STAGE + (8 * (r16 >> 2) + (r16 & 3)) * 128 + ((g ^ (((r16 >> 1) & 1) | (((r16 >> 2) & 3) << 1))) << 4));
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem) + wave * 4096;
    const char* W1b = (const char*)a_.W1;
    const char* W2b = (const char*)a_.W2;
    // stage j (0..7) of chunk cc: j < 4: W1 rows 64 cc .. +63, k 128 j .. +127;  j >= 4: W2 rows 128 (j - 4) .. +127, columns 64 cc .. +63
    auto issue = [&](int cc, int j) {
        if (cc > FF_NCH - 1) cc = FF_NCH - 1;                     // past the end: a harmless re-fetch (every wait count stays a constant)
        const uint32_t dst = ring_lds + j * FF_STAGE;
        if (j < 4) {
            const char* gp = W1b + ((int64_t)cc * 64 * FF_D + j * 128) * 2;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(src1[jj]), "s"(gp), "s"(dst + jj * 1024) : "memory");
        } else {
            const char* gp = W2b + ((int64_t)(j - 4) * 128 * FF_H + cc * 64) * 2;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(src2[jj]), "s"(gp), "s"(dst + jj * 1024) : "memory");
        }
    };
    auto frags = [&](int j, int u, bf16x8 (&bf)[4]) {             // the 4 fragments of sub-step u of stage j (slot j)
        uint32_t base = j < 4 ? rd1 : rd2;
        asm volatile("" : "+v"(base));                            // opaque: the XOR below is redone per sub-step, not hoisted into registers
        if (j < 4) {
            const char* pp = smem + (base ^ (uint32_t)(u << 6));
#pragma unroll
            for (int f = 0; f < 4; ++f) bf[f] = *(const bf16x8*)(pp + j * FF_STAGE + (32 * (f >> 1) + 4 * (f & 1)) * 256);
        } else {
            const int t = u >> 1, s = u & 1;
            const char* pp = smem + (base ^ (uint32_t)(s << 6));
#pragma unroll
            for (int f = 0; f < 4; ++f) bf[f] = *(const bf16x8*)(pp + (j - 4) * FF_STAGE + (64 * t + 32 * (f >> 1) + 4 * (f & 1)) * 128);
        }
    };
    // prologue: stages 0..6 of chunk 0
#pragma unroll
    for (int j = 0; j < 7; ++j) issue(0, j);
    for (int q = tid; q < FF_H; q += 256) b1_lds[q] = a_.b1[q];
    for (int q = tid; q < FF_D; q += 256) b2_lds[q] = a_.b2[q];
    float* lna_gb = (float*)(smem + 7 * FF_STAGE);                // gamma | beta: slot 7 is not filled before the loop's first barrier
    for (int q = tid; q < FF_D; q += 256) { lna_gb[q] = a_.gamma[q]; lna_gb[FF_D + q] = a_.beta[q]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ff_wait<24>();                                                // the x1 fragments + stage 0 landed (stages 1..6 may be in flight)
#pragma unroll
    for (int i = 0; i < 2; ++i)
        asm volatile("" : "+v"(a[i][0]), "+v"(a[i][1]), "+v"(a[i][2]), "+v"(a[i][3]), "+v"(a[i][4]), "+v"(a[i][5]), "+v"(a[i][6]), "+v"(a[i][7]), "+v"(a[i][8]),
                     "+v"(a[i][9]), "+v"(a[i][10]), "+v"(a[i][11]), "+v"(a[i][12]), "+v"(a[i][13]), "+v"(a[i][14]), "+v"(a[i][15]));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        // LayerNorm1 of the rows in place (gemm_astat_kernel's lna path: statistics from MFMAs — row sums against a ones operand, sums of squares from
        // the Gram diagonal); the normalised rows (= the residual, the weight gradient's operand) and mean / rstd leave here
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
                const float sq = ff_sum_lane_rows(g == (r16 >> 2) ? dg : 0.f);
                mean_[i] = sacc[0] * (1.f / FF_D);
                rstd_[i] = rsqrtf(fmaxf(sq * (1.f / FF_D) - mean_[i] * mean_[i], 0.f) + a_.ln_eps);
            }
        }
        bf16_t* lo = a_.h1 + (m0 + r16) * FF_D + g * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f32x4 g0 = *(const f32x4*)(lna_gb + ks * 32 + g * 8), g1 = *(const f32x4*)(lna_gb + ks * 32 + g * 8 + 4);
            const f32x4 be0 = *(const f32x4*)(lna_gb + FF_D + ks * 32 + g * 8), be1 = *(const f32x4*)(lna_gb + FF_D + ks * 32 + g * 8 + 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    y[e] = (bf16_t)(((float)a[i][ks][e] - mean_[i]) * rstd_[i] * (e < 4 ? g0[e] : g1[e - 4]) + (e < 4 ? be0[e] : be1[e - 4]));
                a[i][ks] = y;
                *(bf16x8*)(lo + (int64_t)i * 16 * FF_D + ks * 32) = y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { a_.mean[m0 + 16 * i + lane] = mean_[i]; a_.rstd[m0 + 16 * i + lane] = rstd_[i]; }
        }
    }
    // (gamma / beta in slot 7 are dead from here; the first refill of slot 7 comes behind the first barrier of the loop)
    FFD(0);                                                       // prologue: A panel, LayerNorm, first stages
    f32x4 acc2[2][32];                                            // the 32 x 512 output tile: column tile t = j / 4 ... see the final epilogue
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 32; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 bq[4][4];                                              // fragment sets: sub-step gs of the chunk computes with set gs & 3, prefetches gs + 2
    bf16x8 hcb[2][2];                                             // the chunk's 32 x 64 hidden tile as phase-2 operand: [row fragment][k-step]
    frags(0, 0, bq[0]);
    frags(0, 1, bq[1]);
    if (FF_PF == 3) frags(0, 2, bq[2]);
    const int ecol = 8 * g;
    const uint32_t foff0 = (uint32_t)(r16 * FF_H + ecol);          // element offset of (row lane % 16, column 8 g) in an [.., 2048] row-major tensor
    for (int c = 0; c < FF_NCH; ++c) {
        f32x4 acc1[2][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4 b4 = *(const f32x4*)(b1_lds + c * 64 + 32 * (f >> 1) + 4 * (f & 1) + ecol);
            acc1[0][f] = b4;
            acc1[1][f] = b4;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j == 4) {
                FFD(1);                                           // phase 1 of the chunk (incl. its waits)
                // ---- the chunk's hidden tile: ReLU, dropout, mask bits, one rounding to bf16, store — and hand-over to phase 2
                ff_drain();
                uint32_t mask_word = 0;
                const bf16_t* fb = a_.f + m0 * FF_H + c * 64;          // wave-uniform
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float v[8] = {acc1[i][2 * h][0], acc1[i][2 * h][1], acc1[i][2 * h][2], acc1[i][2 * h][3],
                                      acc1[i][2 * h + 1][0], acc1[i][2 * h + 1][1], acc1[i][2 * h + 1][2], acc1[i][2 * h + 1][3]};
                        if (FF_ABL != 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v[e]), 0));
                        if (a_.drop_f.thr16) ff_drop8(a_.drop_f, (uint32_t)((m0 + 16 * i) * FF_H + c * 64 + 32 * h) + foff0, v);
                        uint32_t bits = 0;
#pragma unroll
                        for (int e = 0; e < 8; ++e) bits |= (v[e] != 0.f ? 1u : 0u) << e;
                        mask_word |= bits << (8 * (2 * i + h));
                        }
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                        hcb[i][h] = o;
                        const uint32_t bo = (foff0 + (uint32_t)(16 * i * FF_H + 32 * h)) * 2;
                        if (FF_ABL != 1 && FF_ABL != 4) {
                        if (a_.nt_store) ff_store16_nt(fb, bo, __builtin_bit_cast(u32x4, o));
                        else ff_store16(fb, bo, __builtin_bit_cast(u32x4, o));
                        }
                    }
                if (FF_ABL != 1 && FF_ABL != 4) ff_store4(a_.mask + ((m0 >> 5) * (int64_t)FF_NCH + c) * 256, (uint32_t)lane * 4, mask_word);
                frags(4, 0, bq[0]);
                frags(4, 1, bq[1]);
                if (FF_PF == 3) frags(4, 2, bq[2]);
                asm volatile("s_nop 4" ::: "memory");             // VALU-written operands ahead of an MFMA the hazard recogniser cannot see
                FFD(2);                                           // chunk epilogue
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u == 2) {
#ifdef EMO_DIAG
                    const uint64_t tw0_ = __builtin_readcyclecounter();
#endif
                    ff_wait<20>();                                // stage s + 1 landed; stages s + 2 .. s + 6 (20 DMA operations) may stay in flight
#ifdef EMO_DIAG
                    const uint64_t tw1_ = __builtin_readcyclecounter();
                    tc_[4] += tw1_ - tw0_;
#endif
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#ifdef EMO_DIAG
                    tc_[5] += __builtin_readcyclecounter() - tw1_;
#endif
                    if (j == 0) issue(c, 7); else issue(c + 1, j - 1);   // refill the slot of stage s - 1 with stage s + 7
                }
                if (!(j == 3 && u + FF_PF >= 4)) {                // (nothing is prefetched across the chunk epilogue: register peak)
                    const int g2 = j * 4 + u + FF_PF;             // sub-step to prefetch (32 .. = the next chunk's 0 ..)
                    frags((g2 >> 2) & 7, g2 & 3, bq[g2 & 3]);
                }
                if (j < 4) {
                    if (FF_ABL != 3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int f = 0; f < 4; ++f) ff_mma_v(acc1[i][f], bq[u][f], a[i][j * 4 + u]);
                    } else asm volatile("" :: "v"(bq[u][0]), "v"(bq[u][1]), "v"(bq[u][2]), "v"(bq[u][3]));
                } else {
                    if (FF_ABL != 2 && FF_ABL != 3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int f = 0; f < 4; ++f) ff_mma_a(acc2[i][(j - 4) * 8 + (u >> 1) * 4 + f], bq[u][f], hcb[i][u & 1]);
                    } else asm volatile("" :: "v"(bq[u][0]), "v"(bq[u][1]), "v"(bq[u][2]), "v"(bq[u][3]), "v"(hcb[0][0]), "v"(hcb[0][1]), "v"(hcb[1][0]), "v"(hcb[1][1]));
                }
            }
        }
        FFD(3);                                                   // phase 2 of the chunk
    }
    ff_drain();
    ff_wait<0>();                                                 // the run-ahead refills past the last stage must land before the LDS is released
    // ---- x2 = h1 + dropout(out + b2): acc2[i][8 q + 4 t + f] holds, for row 16 i + lane % 16, output columns 128 q + 64 t + ff_nrow(f, 4 g + r)
    //      = column tile T = 2 q + t (64 wide), half h = f >> 1: columns 64 T + 32 h + 8 g + 4 (f & 1) + r — and the lane's h1 fragment 2 T + h holds
    //      exactly h1[row][64 T + 32 h + 8 g .. + 7]
    const uint32_t xoff0 = (uint32_t)(r16 * FF_D + ecol);
    const bf16_t* xb = a_.x2 + m0 * FF_D;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 lo4 = acc2[i][(T >> 1) * 8 + (T & 1) * 4 + 2 * h], hi4 = acc2[i][(T >> 1) * 8 + (T & 1) * 4 + 2 * h + 1];
                const f32x4 bl = *(const f32x4*)(b2_lds + 64 * T + 32 * h + ecol), bh = *(const f32x4*)(b2_lds + 64 * T + 32 * h + ecol + 4);
                float v[8] = {lo4[0] + bl[0], lo4[1] + bl[1], lo4[2] + bl[2], lo4[3] + bl[3], hi4[0] + bh[0], hi4[1] + bh[1], hi4[2] + bh[2], hi4[3] + bh[3]};
                if (a_.drop_y.thr16) ff_drop8(a_.drop_y, (uint32_t)((m0 + 16 * i) * FF_D + 64 * T + 32 * h) + xoff0, v);
                const bf16x8 res = a[i][2 * T + h];
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] + (float)res[e]);
                ff_store16(xb, (xoff0 + (uint32_t)(16 * i * FF_D + 64 * T + 32 * h)) * 2, __builtin_bit_cast(u32x4, o));
            }
#ifdef EMO_DIAG
    FFD(6);                                                       // final epilogue
    if (lane == 0) {
        for (int k_ = 0; k_ < 7; ++k_) atomicAdd(&emo_ffn_diag[k_], (unsigned long long)tc_[k_]);
        atomicAdd(&emo_ffn_diag[14], (unsigned long long)(__builtin_readcyclecounter() - tstart_));
        atomicAdd(&emo_ffn_diag[15], 1ull);
    }
#endif
}
}  // namespace

#ifdef EMO_DIAG
extern "C" int emo_ffn_diag_fetch(unsigned long long* host, int reset) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(emo_ffn_diag), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(emo_ffn_diag), z, sizeof(z)); }
    return 0;
}
#endif

extern "C" int emo_ffn_fwd_supported(int dtype, int64_t M, int64_t d_model, int64_t d_ff) {
    const char* e = getenv("EMO_FFN_FUSED");
    if (e && atoi(e) == 0) return 0;
    return (dtype == EMO_BF16 && d_model == FF_D && d_ff == FF_H && M >= 32768 && (M % FF_BM) == 0 && M * FF_H < ((int64_t)1 << 32)) ? 1 : 0;
}

extern "C" int emo_ffn_fwd(const void* x1, const float* gamma, const float* beta, float ln_eps, const void* W1, const float* b1, const void* W2, const float* b2,
                           void* h1_out, float* mean_out, float* rstd_out, void* f_out, uint8_t* mask_out, void* x2_out, int64_t M, int64_t d_model, int64_t d_ff,
                           int dtype, float p_drop, uint64_t seed, uint64_t offset_f, uint64_t offset_y, emo_stream_t stream) {
    EMO_CHECK(x1 && gamma && beta && W1 && b1 && W2 && b2 && h1_out && mean_out && rstd_out && f_out && mask_out && x2_out, "emo_ffn_fwd: null pointer");
    EMO_CHECK(emo_ffn_fwd_supported(dtype, M, d_model, d_ff), "emo_ffn_fwd: needs bf16, d_model 512, d_ff 2048, M %% 128 == 0, M >= 32768 (emo_ffn_fwd_supported)");
    EMO_CHECK((((uintptr_t)x1 | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)h1_out | (uintptr_t)f_out | (uintptr_t)x2_out | (uintptr_t)mask_out) & 15) == 0,
              "emo_ffn_fwd: pointers must be 16-B aligned");
    FfnArgs a;
    a.x1 = (const bf16_t*)x1; a.gamma = gamma; a.beta = beta; a.ln_eps = ln_eps;
    a.W1 = (const bf16_t*)W1; a.b1 = b1; a.W2 = (const bf16_t*)W2; a.b2 = b2;
    a.h1 = (bf16_t*)h1_out; a.mean = mean_out; a.rstd = rstd_out; a.f = (bf16_t*)f_out; a.mask = mask_out; a.x2 = (bf16_t*)x2_out;
    a.M = M; a.drop_f = make_drop(p_drop, seed, offset_f); a.drop_y = make_drop(p_drop, seed, offset_y);
    a.nt_store = 1;
    { const char* e = getenv("EMO_FFN_NT"); if (e) a.nt_store = atoi(e); }
    const size_t lds = FF_RING + (FF_H + FF_D) * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)ffn_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(ffn_fused_fwd_kernel, dim3((unsigned)(M / FF_BM)), dim3(256), lds, (hipStream_t)stream, a);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
