// 256 x 256 x 64 bf16 GEMM tile with ONE wave per SIMD: 4 waves, each owning a 128 x 128 quadrant (64 accumulator fragments = 256 registers,
// which hipcc places in the accumulation VGPRs at this occupancy), for the long-K products of the layer (K >= 1024: FFN2 forward, QKV / FFN1
// dgrad).  Why this shape: the 128 x 128 tile with 64 x 64 wave quadrants spends twice the LDS-read bytes and twice the L2 -> LDS bytes per flop
// of this one (0.031 / 0.0156 B per flop against 0.0156 / 0.0078).  (r03 gave the 1400-W package limit as the reason; r04 measured 2.18-2.43 GHz
// per kernel class inside the training step, i.e. no cap there; r05: what couples these products to the clock is their HBM stream — the same
// kernel runs at 2.10 GHz with the A rows in L2 and at 1.79 GHz with A from HBM, profiles/r05_gemm_isa_diff.txt.)
//   * operands: K-contiguous rows (A [M,K]; B [N,K] = nn.Linear weights), LDS image [256 rows][8 x 16-B chunks], chunk ^= row & 7, filled by
//     LDS-DMA issued as inline asm (SGPR base + 32-bit lane offset; behind the builtin hipcc answers later fragment reads with full waits);
//     the B rows a fragment reads are permuted (as in the A-stationary kernel) so that a lane owns 8 CONSECUTIVE output columns per fragment
//     pair -> 16-B stores / residual loads straight from the accumulators; B's chunk swizzle is ((row >> 1) & 1) | ((row >> 3) & 3) << 1,
//     the one that is conflict-free for those permuted rows (A: row & 7);
//   * rings: A 3 slots, B 2 slots (160 KB): at the sync point in the middle of K-tile t (between its two 32-deep steps) a wave waits for
//     tile t+1 with a counted vmcnt (tile t+2's A stays in flight), one s_barrier, then re-stages B(t+2) / A(t+3) into the slots tile t no
//     longer reads (its second-step fragments are already in registers);
//   * fragments are double-buffered in registers: the reads of the NEXT 32-deep step are spread over the first six MFMA groups of the
//     current one (pinned with sched_barrier: left alone the scheduler sinks them next to their uses), so neither an LDS round trip nor a
//     DMA wait sits in front of an MFMA in steady state: per K-tile 128 MFMAs per wave, one barrier.
// Where a block's cycles go (-DEMO_DIAG, tools/w128_cycles.py, FFN2-forward shape): first DMA round trip 7 %, K loop 68 % (the mid-tile sync is 6 %
// of it), epilogue 25 % (output stores of all CUs at once = an HBM write burst: 10 %; dropout hashes 5 %; residual loads 8 % cold).
// Tried and dropped (r03, same-box A/B of the training step): (i) residual rows by LDS-DMA into the idle operand ring under the last 64 MFMAs
// (wave-private 32-KB region, swizzled): epilogue 38 -> 26 thousand cycles on cold operands, but inside the step the residual is L2 / MALL-hot
// and nothing changed (8.72 vs 8.79 ms per step for the class); (ii) one block per CU walking several output tiles so that a tile's first DMA
// round trip overlaps the previous tile's store drain: 1142 -> 1086 TFLOP/s sustained, 1038 -> 981 in the step (loop-invariant lane offsets
// spill across the epilogue, their scratch reloads wait for the output stores; a contiguous run of tiles per block was worse still, 871: every
// row panel re-read from beyond the L2).
#include "emo_gemm_epi.h"

namespace {
constexpr int W_BM = 256, W_BN = 256, W_BK = 64;
constexpr int W_TILE = 256 * W_BK * 2;                        // one operand tile: 32 KB
constexpr int W_NA = 3, W_NB = 2;
constexpr int W_LDS = (W_NA + W_NB) * W_TILE;                 // 160 KB

__device__ __forceinline__ uint32_t w_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

// two 1-KB LDS-DMA pieces (wave-uniform 64-bit base + per-lane 32-bit byte offsets) to lds_dst, lds_dst + 1024
__device__ __forceinline__ void w_dma2(const char* sbase, uint32_t o0, uint32_t o1, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "v"(o1), "s"(sbase), "s"(lds_dst) : "memory", "scc");
}
__device__ __forceinline__ void w_dma2_nt(const char* sbase, uint32_t o0, uint32_t o1, uint32_t lds_dst) {      // the same with the streaming (nt) cache hint
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "v"(o1), "s"(sbase), "s"(lds_dst) : "memory", "scc");
}
// MFMA with the accumulator PINNED to the accumulation registers: with the builtin hipcc splits the 256 accumulator registers of a wave between
// both register files and rotates them through copies on the loop back-edge (264 v_accvgpr moves per K-tile).  The statement is opaque to the
// hazard recogniser: the same accumulator is never touched again within 63 MFMAs, and the epilogue waits out the last results (w_mma_drain).
__device__ __forceinline__ void w_mma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// the same with the accumulator in ordinary VGPRs (the bias-gradient sums): as a builtin, hipcc also wants these in the accumulation registers,
// evicts accumulators of w_mma to make room and reads them back right behind an MFMA it cannot see (no wait states): wrong sums
__device__ __forceinline__ void w_mma_v(f32x4& c, const bf16x8& a, const bf16x8& b) {
    // (s_nop: hipcc rebuilds the constant all-ones operand with v_mov right in front of the statement, and a VALU write needs wait states
    // before an MFMA reads the register — the hazard recogniser does not look inside an asm statement)
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void w_mma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
// dropout multipliers of 8 consecutive elements whose linear index is a multiple of 8: two hashes, bit-identical to drop_mult()
__device__ __forceinline__ void w_drop8(const DropCtx& d, uint64_t idx0, float (&v)[8]) {
    const uint32_t lo = (uint32_t)(idx0 >> 2), hi = (uint32_t)(idx0 >> 34) * 0x9E3779B1u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t h = emo_drop_hash(d, (lo + q) ^ hi), h2 = emo_xs32(h);
        v[4 * q] *= (h & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 1] *= (h >> 16) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 2] *= (h2 & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
        v[4 * q + 3] *= (h2 >> 16) >= d.thr16 ? d.scale : 0.f;
    }
}
template <int N> __device__ __forceinline__ void w_wait();
template <> __device__ __forceinline__ void w_wait<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void w_wait<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void w_wait<24>() { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }

template <typename OutT>
__global__ __launch_bounds__(256, 1) void gemm_w128_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                          OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef EMO_DIAG
    const uint64_t dg_t0 = __builtin_readcyclecounter();
    uint64_t dg_sync = 0, dg_loop0 = 0, dg_loop1 = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t tiles_n = N / W_BN, tiles_m = M / W_BM;
    const int64_t bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
    const int64_t tn = local % tiles_n, tm = (local / tiles_n) * 8 + xcd;      // the column tiles of one row panel back-to-back on one XCD
    if (tm >= tiles_m) return;
    const int64_t m0 = tm * W_BM, n0 = tn * W_BN;
    const int nk = (int)(K / W_BK);

    uint32_t offA[8], offB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (wave * 8 + i) * 8 + (lane >> 3);
        offA[i] = (uint32_t)((r * lda + ((lane & 7) ^ (r & 7)) * 8) * 2);
        offB[i] = (uint32_t)((r * ldb + ((lane & 7) ^ (((r >> 1) & 1) | (((r >> 3) & 3) << 1))) * 8) * 2);
    }
    const char* gA = (const char*)(A + m0 * lda);              // wave-uniform; K-tile t at + t * 128 bytes
    const char* gB = (const char*)(B + n0 * ldb);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(w_lds_addr(smem));
    const uint32_t dstw = lds0 + wave * 8192;                   // this wave's 8 KB of a tile slot
    // fragment addressing: row = rbase + (lane & 15) (rbase % 16 == 0 -> row & 7 == lane & 7), 32-deep step ks: chunk (4 ks + lane / 16) ^ (lane & 7)
    // B: fragment j, N-index i reads tile row 32 (j >> 1) + 8 (i >> 2) + 4 (j & 1) + (i & 3); its swizzle depends on the lane only
    const uint32_t foA = (uint32_t)((lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4)) + wr * 128 * 128;     // step 1: ^ 64
    const uint32_t foB = (uint32_t)((8 * ((lane & 15) >> 2) + (lane & 3)) * 128 + (((lane >> 4) ^ (((lane >> 1) & 1) | (((lane >> 2) & 3) << 1))) << 4)) +
                         W_NA * W_TILE + wc * 128 * 128;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int sa_issue = 0, sb_issue = 0;                             // ring slots of the next A / B tile to issue
    int t_issueA = 0, t_issueB = 0;
    const bool a_nt = ep.a_nt != 0;                            // streaming hint on the A tiles
    auto issueA_part = [&](int g) {
        if (a_nt) w_dma2_nt(gA + (int64_t)t_issueA * (W_BK * 2), offA[2 * g], offA[2 * g + 1], dstw + sa_issue * W_TILE + g * 2048);
        else w_dma2(gA + (int64_t)t_issueA * (W_BK * 2), offA[2 * g], offA[2 * g + 1], dstw + sa_issue * W_TILE + g * 2048);
    };
    auto issueB_part = [&](int g) { w_dma2(gB + (int64_t)t_issueB * (W_BK * 2), offB[2 * g], offB[2 * g + 1], dstw + (W_NA + sb_issue) * W_TILE + g * 2048); };
    auto doneA = [&]() { ++t_issueA; sa_issue = sa_issue == W_NA - 1 ? 0 : sa_issue + 1; };
    auto doneB = [&]() { ++t_issueB; sb_issue ^= 1; };
    // prologue: A0 B0 A1 B1 A2 (tiles past the end re-fetch the last one: every count below is a constant; drained before exit)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const bool isA = (s & 1) == 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) { if (isA) issueA_part(g); else issueB_part(g); }
        if (isA) { doneA(); if (t_issueA > nk - 1) t_issueA = nk - 1; } else { doneB(); if (t_issueB > nk - 1) t_issueB = nk - 1; }
    }
    w_wait<24>();                                               // A0, B0 landed (A1, B1, A2 may be in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 fa[2][8], fb[2][8];
    int sa = 0, sb = 0;                                         // slots of the K-tile being computed
    const char* ldsb = smem;
    // read number q (0..15) of a fragment set: B fragments first (the first MFMA group needs all eight), then A
    // (a_off / b_off: lane offsets with the step's chunk flip already applied: the fragment index is then an immediate offset)
    auto rd = [&](int set, int q, uint32_t a_off, uint32_t b_off) {
        if (q < 8) fb[set][q] = *(const bf16x8*)(ldsb + b_off + (32 * (q >> 1) + 4 * (q & 1)) * 128);
        else fa[set][q - 8] = *(const bf16x8*)(ldsb + a_off + (q - 8) * 2048);
    };
    {
        const uint32_t a_off = foA, b_off = foB;
#pragma unroll
        for (int q = 0; q < 16; ++q) rd(0, q, a_off, b_off);
    }
    constexpr int RQ[9] = {0, 3, 6, 9, 12, 14, 16, 16, 16};      // reads issued before MFMA group i: RQ[i] .. RQ[i+1]-1
#ifdef EMO_DIAG
    dg_loop0 = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < nk; ++t) {
        const uint32_t a_off = (foA + sa * W_TILE) ^ 64u, b_off = (foB + sb * W_TILE) ^ 64u;
        // ---- step 0 of K-tile t (set 0), reading step 1 (set 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int q = RQ[i]; q < RQ[i + 1]; ++q) rd(1, q, a_off, b_off);
#pragma unroll
            for (int j = 0; j < 8; ++j) w_mma(acc[i][j], fb[0][j], fa[0][i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sync point: K-tile t+1 landed for every wave; every wave's reads of tile t are in registers
#ifdef EMO_DIAG
        const uint64_t dg_s0 = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        w_wait<8>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef EMO_DIAG
        dg_sync += __builtin_readcyclecounter() - dg_s0;
#endif
        sa = sa == W_NA - 1 ? 0 : sa + 1;
        sb ^= 1;
        const uint32_t a_nx = foA + sa * W_TILE, b_nx = foB + sb * W_TILE;
        // ---- step 1 of K-tile t (set 1), reading step 0 of tile t+1 (set 0), re-staging B(t+2) then A(t+3)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int q = RQ[i]; q < RQ[i + 1]; ++q) rd(0, q, a_nx, b_nx);
            if (i < 4) issueB_part(i); else issueA_part(i - 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) w_mma(acc[i][j], fb[1][j], fa[1][i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        doneB(); if (t_issueB > nk - 1) t_issueB = nk - 1;
        doneA(); if (t_issueA > nk - 1) t_issueA = nk - 1;
    }
#ifdef EMO_DIAG
    dg_loop1 = __builtin_readcyclecounter();
#endif
    w_wait<0>();
    w_mma_drain();
    // ---- epilogue straight from the accumulators: lane = row .. + (lane & 15), columns 32 h + 8 (lane / 16) .. + 7 of fragment pair h
    const int ecol = 8 * (lane >> 4);
    {                                                             // (the launcher admits bias / dropout / residual epilogues only)
        f32x4 bv[4][2];
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) bv[h][q] = ep.bias ? *(const f32x4*)(ep.bias + n0 + wc * 128 + 32 * h + ecol + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const OutT* rp = (const OutT*)ep.residual;
#pragma clang loop unroll(full)
        for (int i2 = 0; i2 < 4; ++i2) {                          // two row fragments at a time: their eight residual pieces are requested together
            bf16x8 res[2][4];
            if (rp) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        res[u][h] = *(const bf16x8*)(rp + (m0 + wr * 128 + (2 * i2 + u) * 16 + (lane & 15)) * ep.ldc + n0 + wc * 128 + 32 * h + ecol);
            }
#pragma clang loop unroll(full)
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * i2 + u;
                const int64_t m = m0 + wr * 128 + i * 16 + (lane & 15);
#pragma clang loop unroll(full)
                for (int h = 0; h < 4; ++h) {
                    const int64_t n = n0 + wc * 128 + 32 * h + ecol;
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * h][r] + bv[h][0][r]; v[4 + r] = acc[i][2 * h + 1][r] + bv[h][1][r]; }
                    if (ep.drop.thr16) w_drop8(ep.drop, (uint64_t)(m * N + n), v);
                    if (rp) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += (float)res[u][h][r];
                    }
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (bf16_t)v[r];
                    if (ep.nt_store) __builtin_nontemporal_store(o, (bf16x8*)(C + m * ep.ldc + n));
                    else *(bf16x8*)(C + m * ep.ldc + n) = o;
                }
            }
        }
    }
#ifdef EMO_DIAG
    if (ep.ablate == 8 && ep.rln_stats && lane == 0) {            // (diagnostics, tools/w128_cycles.py: the otherwise unused rln_stats pointer carries the counter buffer)
        unsigned long long* dg = (unsigned long long*)ep.rln_stats;
        const uint64_t t1 = __builtin_readcyclecounter();
        atomicAdd(dg + 0, (unsigned long long)(dg_loop0 - dg_t0));        // prologue
        atomicAdd(dg + 1, (unsigned long long)(dg_loop1 - dg_loop0));     // K loop
        atomicAdd(dg + 2, (unsigned long long)dg_sync);                   // of it: lgkmcnt + vmcnt + barrier at the mid-tile sync
        atomicAdd(dg + 3, (unsigned long long)(t1 - dg_loop1));           // epilogue
        atomicAdd(dg + 4, 1ull);
    }
#endif
}

// ================================================================================================ TN: dW[M,N] (+)= A[K,M]^T B[K,N]  (wgrad)
// Both operands are stored token-major (K = the B*T tokens, rows of dY / X), so a K-tile of an operand is 64 rows x 256 columns: LDS image =
// two halves of [64 k][128 columns] with the 32-B granule swizzle of the 128 x 128 wgrad kernel, fragments by `ds_read_b64_tr_b16` (two per
// fragment).  Split-K over the tokens: all output tiles of one split run on ONE XCD (its token range is fetched from HBM once into that L2),
// the fp32 partial tiles go to the caller's workspace with plain 16-B stores and splitk_reduce_kernel sums them in a fixed order.
// RS = 1: a_rowsum[m] += sum_k A[k][m] (the bias gradient) as MFMAs against an all-ones operand: the two waves that hold the same A rows
// take alternate fragments, and of the tiles_n blocks that read the same A columns block tn takes the K-tiles with index % tiles_n == tn.
// RS = 2: b_rowsum[n] += sum_k B[k][n] (HF Conv1D layout, where dY is the B operand): the same with the roles of m and n exchanged.
__device__ __forceinline__ int w_swz_k(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <int RS>
__global__ __launch_bounds__(256, 1) void gemm_w128_tn_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                             float* __restrict__ Cws, int64_t M, int64_t N, int64_t K, int64_t kps, float* __restrict__ a_rowsum, float* __restrict__ b_rowsum,
                                                             int full, int extra, int nt_mask, float* __restrict__ rs_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t tiles_n = N / W_BN, ntile = (M / W_BM) * tiles_n;
    // 32 blocks per XCD: `full` complete K-splits (all their output tiles: the split's token range is fetched once into that L2), and the
    // 32 - full * ntile slots left over on every XCD are filled with the tiles of `extra` more splits, each spread over as few XCDs as
    // possible (fused QKV: 12 tiles -> 2 full splits per XCD + 5 shared ones = 21 splits on 252 CUs instead of 16 on 192)
    const int64_t bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
    int64_t split, t_id;
    if (q < (int64_t)full * ntile) { split = xcd * full + q / ntile; t_id = q % ntile; }
    else {
        const int64_t left = 32 - (int64_t)full * ntile, slot = xcd * left + (q - (int64_t)full * ntile);
        if (slot >= (int64_t)extra * ntile) return;
        split = 8 * full + slot / ntile;
        t_id = slot % ntile;
    }
    const int64_t tm = t_id / tiles_n, tn = t_id % tiles_n;
    const int64_t m0 = tm * W_BM, n0 = tn * W_BN;
    const int64_t kbeg = split * kps;
    const int64_t kend = (kbeg + kps < K) ? kbeg + kps : K;
    float* C = Cws + split * M * N;
    const int nk = kbeg < kend ? (int)((kend - kbeg) / W_BK) : 0;
    if (nk == 0) {                                              // (a split past the end: its partial tile is zeros)
#pragma unroll 4
        for (int e = tid; e < 256 * 64; e += 256) *(f32x4*)(C + (m0 + (e >> 6)) * N + n0 + (e & 63) * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (RS == 1 && rs_ws) rs_ws[(split * tiles_n + tn) * M + m0 + tid] = 0.f;
        if (RS == 2 && rs_ws) rs_ws[(split * (M / W_BM) + tm) * N + n0 + tid] = 0.f;
        return;
    }
    uint32_t offA[8], offB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int J = wave * 8 + i, half = J >> 4, j = J & 15;
        const int k = j * 4 + (lane >> 4), p16 = lane & 15, g = (p16 >> 1) ^ w_swz_k(k);
        offA[i] = (uint32_t)((k * lda + half * 128 + (g * 2 + (p16 & 1)) * 8) * 2);
        offB[i] = (uint32_t)((k * ldb + half * 128 + (g * 2 + (p16 & 1)) * 8) * 2);
    }
    const char* gA = (const char*)(A + kbeg * lda + m0);         // wave-uniform; K-tile t at + t * 64 rows
    const char* gB = (const char*)(B + kbeg * ldb + n0);
    const int64_t stepA = 64 * lda * 2, stepB = 64 * ldb * 2;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(w_lds_addr(smem));
    const uint32_t dstw = lds0 + wave * 8192;
    // fragment addressing (see lfrag2<false, 64> of the 128 x 128 kernel): k = 32 ks + 8 (lane / 16) + 4 h + (lane & 15) / 4, swizzle = lane constant
    const int li = lane & 15;
    const uint32_t fo = (uint32_t)((((lane >> 4) * 8 + (li >> 2)) * 256) | ((((li >> 2) & 3) | (((lane >> 4) & 1) << 2)) << 5) | ((li & 3) << 3));
    const uint32_t foA = fo + wr * 16384, foB = fo + W_NA * W_TILE + wc * 16384;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 rs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rs[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t one_b = (bf16_t)1.f;
    bf16x8 ones = {one_b, one_b, one_b, one_b, one_b, one_b, one_b, one_b};
    asm volatile("" : "+v"(ones));                              // opaque: one live register tuple instead of a rematerialised constant
    const int rs_mod = RS == 2 ? (int)(M / W_BM) : (int)tiles_n, rs_me = RS == 2 ? (int)tm : (int)tn;
    int rs_phase = RS ? (int)((kbeg / W_BK) % rs_mod) : 0;      // K-tile index modulo the number of blocks that share the summed operand

    int sa_issue = 0, sb_issue = 0, t_issueA = 0, t_issueB = 0;
    const bool a_nt = (nt_mask & 1) != 0, b_nt = (nt_mask & 2) != 0;     // streaming hint on the operand tiles
    auto issueA_part = [&](int g) {
        if (a_nt) w_dma2_nt(gA + (int64_t)t_issueA * stepA, offA[2 * g], offA[2 * g + 1], dstw + sa_issue * W_TILE + g * 2048);
        else w_dma2(gA + (int64_t)t_issueA * stepA, offA[2 * g], offA[2 * g + 1], dstw + sa_issue * W_TILE + g * 2048);
    };
    auto issueB_part = [&](int g) {
        if (b_nt) w_dma2_nt(gB + (int64_t)t_issueB * stepB, offB[2 * g], offB[2 * g + 1], dstw + (W_NA + sb_issue) * W_TILE + g * 2048);
        else w_dma2(gB + (int64_t)t_issueB * stepB, offB[2 * g], offB[2 * g + 1], dstw + (W_NA + sb_issue) * W_TILE + g * 2048);
    };
    auto doneA = [&]() { ++t_issueA; sa_issue = sa_issue == W_NA - 1 ? 0 : sa_issue + 1; if (t_issueA > nk - 1) t_issueA = nk - 1; };
    auto doneB = [&]() { ++t_issueB; sb_issue ^= 1; if (t_issueB > nk - 1) t_issueB = nk - 1; };
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const bool isA = (s & 1) == 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) { if (isA) issueA_part(g); else issueB_part(g); }
        if (isA) doneA(); else doneB();
    }
    w_wait<24>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bf16x8 fa[2][8], fb[2][8];
    int sa = 0, sb = 0;
    const char* ldsb = smem;
    auto rd1 = [&](uint32_t off) -> bf16x8 {                     // off: lane offset with slot / half / fragment / step already applied
        bf16x8 v;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(ldsb + off + h * 1024));
            const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
            v[4 * h] = tb[0]; v[4 * h + 1] = tb[1]; v[4 * h + 2] = tb[2]; v[4 * h + 3] = tb[3];
        }
        return v;
    };
    // read number q (0..15) of a fragment set (B first); ks8 = 8192 * step
    auto rd = [&](int set, int q2, uint32_t a_off, uint32_t b_off, int ks8) {
        if (q2 < 8) fb[set][q2] = rd1((b_off ^ (uint32_t)(q2 << 5)) + ks8);
        else fa[set][q2 - 8] = rd1((a_off ^ (uint32_t)((q2 - 8) << 5)) + ks8);
    };
#pragma unroll
    for (int q2 = 0; q2 < 16; ++q2) rd(0, q2, foA, foB, 0);
    constexpr int RQ[9] = {0, 3, 6, 9, 12, 14, 16, 16, 16};
    for (int t = 0; t < nk; ++t) {
        const uint32_t a_off = foA + sa * W_TILE, b_off = foB + sb * W_TILE;
        const bool rs_on = RS && rs_phase == rs_me;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int q2 = RQ[i]; q2 < RQ[i + 1]; ++q2) rd(1, q2, a_off, b_off, 8192);
#pragma unroll
            for (int j = 0; j < 8; ++j) w_mma(acc[i][j], fb[0][j], fa[0][i]);
            if (RS == 1 && rs_on && (i & 1) == wc) w_mma_v(rs[i >> 1], ones, fa[0][i]);
            if (RS == 2 && rs_on && (i & 1) == wr) w_mma_v(rs[i >> 1], fb[0][i], ones);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        w_wait<8>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        sa = sa == W_NA - 1 ? 0 : sa + 1;
        sb ^= 1;
        const uint32_t a_nx = foA + sa * W_TILE, b_nx = foB + sb * W_TILE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int q2 = RQ[i]; q2 < RQ[i + 1]; ++q2) rd(0, q2, a_nx, b_nx, 0);
            if (i < 4) issueB_part(i); else issueA_part(i - 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) w_mma(acc[i][j], fb[1][j], fa[1][i]);
            if (RS == 1 && rs_on && (i & 1) == wc) w_mma_v(rs[i >> 1], ones, fa[1][i]);
            if (RS == 2 && rs_on && (i & 1) == wr) w_mma_v(rs[i >> 1], fb[1][i], ones);
            __builtin_amdgcn_sched_barrier(0);
        }
        doneB();
        doneA();
        if (RS) { if (++rs_phase == rs_mod) rs_phase = 0; }
    }
    w_wait<0>();
    w_mma_drain();
    // bias-gradient partials: every (split, tile) block owns its 256 rows (columns) of ONE partial vector in the workspace behind the weight
    // partials — plain stores, summed in a fixed order by the reduce pass (r04: deterministic; without the extra workspace: fp32 atomics)
    if (RS == 1 && (lane >> 4) == 0) {
        float* dst = rs_ws ? rs_ws + (split * tiles_n + tn) * M : nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t m = m0 + wr * 128 + (2 * u + wc) * 16 + lane;
            if (dst) dst[m] = rs[u][0]; else atomicAdd(a_rowsum + m, rs[u][0]);
        }
    }
    if (RS == 2 && (lane & 15) == 0) {                            // D[n][*]: a lane holds the sums of n = 4 (lane / 16) .. + 3 of its fragment
        float* dst = rs_ws ? rs_ws + (split * (M / W_BM) + tm) * N : nullptr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t n = n0 + wc * 128 + (2 * u + wr) * 16 + (lane >> 4) * 4;
            if (dst) *(f32x4*)(dst + n) = rs[u];
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(b_rowsum + n + r, rs[u][r]);
            }
        }
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < 8; ++i) {
        float* crow = C + (m0 + wr * 128 + i * 16 + (lane & 15)) * N + n0 + wc * 128 + (lane >> 4) * 4;
#pragma clang loop unroll(full)
        for (int j = 0; j < 8; ++j) *(f32x4*)(crow + j * 16) = acc[i][j];
    }
}
}  // namespace

// NT bf16 product on 256 x 256 tiles; true when the shape is eligible and the launch was queued.  Default: long reductions (K >= 1024) with at
// least one tile per CU; EMO_GEMM_W128=0 switches it off, =1 admits every eligible shape (tests).
bool emo_gemm_w128_try(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int dtype_out, int64_t M, int64_t N, int64_t K,
                       const EpiParams& ep, hipStream_t st) {
    const char* e = getenv("EMO_GEMM_W128");                   // (read per call: tests toggle it in-process)
    const int mode = e ? atoi(e) : -1;
    if (mode == 0) return false;
    if ((M % W_BM) || (N % W_BN) || (K % W_BK) || K < 4 * W_BK || dtype_out != EMO_BF16) return false;
    if (mode < 0 && (K < 1024 || (M / W_BM) * (N / W_BN) < 256)) return false;
    if (ep.atomic || ep.accumulate || ep.ws_stride || ep.a_rowsum || ep.b_rowsum || ep.ln_c1 || ep.rln_x || ep.mask_out) return false;
    if (ep.aux_out || ep.mul_mode != EMO_MUL_NONE || ep.act != EMO_ACT_NONE) return false;       // register epilogue: bias, dropout, residual
    if ((lda & 7) || (ldb & 7) || (ep.ldc & 7) || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15)) return false;
    if ((uint64_t)(256 * (lda > ldb ? lda : ldb) + 64) * 2 >= 0xFFFF0000ull) return false;
    const int64_t tiles_m = M / W_BM, tiles_n = N / W_BN;
    dim3 grid((unsigned)(((tiles_m + 7) / 8) * 8 * tiles_n));
    EpiParams ep2 = ep;
    { const char* e4 = getenv("EMO_W128_NT_STORE"); ep2.nt_store = e4 ? atoi(e4) : 0; }
    { const char* e3 = getenv("EMO_W128_A_NT"); ep2.a_nt = e3 ? atoi(e3) : 1; }     // default on (r03: -0.4 .. -0.55 ms/step; the wgrad kernel loses with it)
    auto k = gemm_w128_kernel<bf16_t>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS); attr = true; }
    hipLaunchKernelGGL(k, grid, dim3(256), W_LDS, st, A, lda, B, ldb, (bf16_t*)C, M, N, K, ep2);
    return true;
}

// K-splits of the TN (wgrad) product on 256 x 256 tiles (at most one block per CU, see the kernel's block mapping); 0 = shape not taken.
static int64_t w_tn_plan(int64_t M, int64_t N, int64_t K, int& full, int& extra) {
    full = extra = 0;
    const char* e = getenv("EMO_GEMM_W128");
    if (e && atoi(e) == 0) return 0;
    const char* e2 = getenv("EMO_GEMM_W128_TN");
    if (e2 && atoi(e2) == 0) return 0;
    if ((M % W_BM) || (N % W_BN) || (K % W_BK)) return 0;
    const int64_t ntile = (M / W_BM) * (N / W_BN), nkt = K / W_BK;
    if (ntile > 32) return 0;
    int64_t f = 32 / ntile;
    if (f > 8) f = 8;                                          // at most 64 splits
    int64_t x = f < 8 ? (8 * (32 - f * ntile)) / ntile : 0;
    if (e2 && atoi(e2) == 2) x = 0;                            // (diagnostics: whole splits per XCD only)
    int64_t min_kt = 16;                                       // K-tiles per split (r04: 8 let batch-size-4 steps — 512 tokens per split — onto this tile, where
    { const char* e3 = getenv("EMO_W128_TN_MINKT"); if (e3 && atoi(e3) > 0) min_kt = atoi(e3); }      // prologue + epilogue dominate: 8.89 vs 8.39 ms per step)
    while (nkt / (8 * f + x) < min_kt) {
        if (x) x = 0;
        else if (f > 1) --f;
        else return 0;
    }
    int64_t min_blk = 96;                                      // (r04: 160 -> 96 lets the batch-size-4 FFN / QKV weight gradients on: 8.30 -> 7.97 ms per step)
    { const char* e4 = getenv("EMO_W128_TN_MINBLK"); if (e4 && atoi(e4) > 0) min_blk = atoi(e4); }
    if (ntile * (8 * f + x) < min_blk) return 0;                   // (too few blocks for the chip: the 128 x 128 kernel's finer tiles win)
    full = (int)f;
    extra = (int)x;
    return 8 * f + x;
}
int64_t emo_gemm_w128_tn_splits(int64_t M, int64_t N, int64_t K) {
    int full, extra;
    return w_tn_plan(M, N, K, full, extra);
}

void emo_splitk_reduce_launch(const float* ws, int64_t stride, int splits, float* out, int64_t n4, int accumulate, hipStream_t st);
void emo_splitk_reduce_rs_launch(const float* ws, int64_t stride, int splits, float* out, int64_t n4, int accumulate, const float* rs_ws, int64_t rs_stride,
                                 int rs_parts, float* rs_out, hipStream_t st);
// floats of bias-gradient partials behind the weight partials (one vector per (split, tile of the other dimension))
int64_t emo_gemm_w128_tn_rs_floats(int64_t M, int64_t N, int64_t splits) {
    const int64_t a = (N / W_BN) * M, b = (M / W_BM) * N;
    return splits * (a > b ? a : b);
}

// dW[M,N] (+)= A[K,M]^T B[K,N], fp32 out, contiguous C, workspace of >= splits * M * N floats; optional a_rowsum[M] += column sums of A or
// b_rowsum[N] += column sums of B (one of them)
bool emo_gemm_w128_tn_try(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int accumulate,
                          float* a_rowsum, float* b_rowsum, void* ws, int64_t ws_bytes, hipStream_t st) {
    int full, extra;
    const int64_t splits0 = w_tn_plan(M, N, K, full, extra);
    if (!splits0 || ldc != N || !ws || ((uintptr_t)ws & 15) || ws_bytes < splits0 * M * N * (int64_t)sizeof(float)) return false;
    if ((lda & 7) || (ldb & 7) || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15)) return false;
    if ((uint64_t)(64 * (lda > ldb ? lda : ldb) + 256) * 2 >= 0xFFFF0000ull) return false;
    const int64_t kps = ((K / W_BK + splits0 - 1) / splits0) * W_BK;
    const int64_t ntile = (M / W_BM) * (N / W_BN);
    dim3 grid(256);                                            // 32 block slots per XCD (w_tn_plan)
    int nt_mask = 0;
    { const char* e3 = getenv("EMO_W128_TN_NT"); if (e3) nt_mask = atoi(e3); }
    float* rs_ws = nullptr;                                     // bias-gradient partials through the workspace when it has room for them
    if ((a_rowsum || b_rowsum) && ws_bytes >= (splits0 * M * N + emo_gemm_w128_tn_rs_floats(M, N, splits0)) * (int64_t)sizeof(float) &&
        !(((uintptr_t)a_rowsum | (uintptr_t)b_rowsum) & 15))
        rs_ws = (float*)ws + splits0 * M * N;
#define W_TN_LAUNCH(RSv)                                                                                                                     \
    do {                                                                                                                                     \
        auto k = gemm_w128_tn_kernel<RSv>;                                                                                                   \
        static bool attr = false;                                                                                                            \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS); attr = true; }            \
        hipLaunchKernelGGL(k, grid, dim3(256), W_LDS, st, A, lda, B, ldb, (float*)ws, M, N, K, kps, a_rowsum, b_rowsum, full, extra, nt_mask, rs_ws); \
    } while (0)
    if (a_rowsum) W_TN_LAUNCH(1); else if (b_rowsum) W_TN_LAUNCH(2); else W_TN_LAUNCH(0);
#undef W_TN_LAUNCH
    if (rs_ws)
        emo_splitk_reduce_rs_launch((const float*)ws, M * N, (int)splits0, C, (M * N) >> 2, accumulate, rs_ws, a_rowsum ? M : N,
                                    (int)(splits0 * (a_rowsum ? N / W_BN : M / W_BM)), a_rowsum ? a_rowsum : b_rowsum, st);
    else
        emo_splitk_reduce_launch((const float*)ws, M * N, (int)splits0, C, (M * N) >> 2, accumulate, st);
    return true;
}
