// FAVOR+ causal linear attention, bf16, d_head 64, 128 features — "slice" kernels (r03).
//
// Why a second set of kernels: the generic kernels of emo_favor.hip give one (b, h) scan to an 8-wave workgroup and run every contraction of
// a 64-token chunk as a barrier-delimited phase over LDS images (7-8 barriers per chunk; r02 PMC: ~2000-2500 cycles per phase for ~200
// cycles of MFMA issue, 42-49 % of the LDS cycles bank conflicts, forward 0.28 / backward 0.13 of the HBM roofline).  Here a (b, h) scan
// belongs to FOUR waves that each own a SLICE of the problem and keep their whole chain in registers:
//   * the feature map is split by feature index m (wave w: m in [16w, 16w+16), i.e. features f = m and 64 + m);
//   * every product is arranged so that an MFMA accumulator tile is directly the operand of the next MFMA: a 16x16 accumulator has
//     lane l <-> column l%16 and rows 4*(l/16) .. +3, which IS the A (or B) operand layout for row (column) l%16 with the contraction
//     index running over the accumulator rows — provided the OTHER operand is read in the same permuted k order
//     (k-step s, element e <-> k = 32 s + 16 (e/4) + 4 (l/16) + e%4; emo_lds_mma.h load_perm / load_perm_tr);
//   * what the waves must exchange is only bf16 operand data (feature rows, dU rows: 16-32 KB per 32-token chunk), never fp32 partial
//     sums; the running state is split by output column d (forward, dV) or by feature f (dq, dk) so that no state is ever reduced
//     across waves;
//   * the normaliser (row sums of the masked A matrix, q'.z) and the z / r vectors come out of the same MFMAs through a 65th "ones"
//     column of V (an extra 16-column tile whose column 0 is 1): no VALU reductions, no atomics;
//   * ONE workgroup barrier per 32-token chunk (the LDS images are double buffered), 45-66 KB of LDS => 2-3 workgroups per CU, each
//     wave runs 40-75 MFMAs per chunk between barriers.
// Numerics = the generic bf16 kernels' (bf16 operands, fp32 accumulation and state, features exp2((c log2e) u - off)).
// Requirements: bf16, d_head 64, n_feat 128, T % 32 == 0, single-segment scan; everything else stays on emo_favor.hip.
#include "emo_common.h"

#include "emo_lds_mma.h"

#ifdef EMO_DIAG
__device__ unsigned long long emo_fs_diag[64];             // diagnostics build only: per-phase cycle sums (tools/fs_cycles.py)
extern "C" int emo_diag_fetch(unsigned long long* host, int n, int reset) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(emo_fs_diag), sizeof(unsigned long long) * n) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(emo_fs_diag), z, sizeof(z)); }
    return 0;
}
#define FSD_BEGIN uint64_t tc_[16] = {}, ts_ = __builtin_readcyclecounter(), tstart_ = ts_;
#define FSD(k) do { const uint64_t tn_ = __builtin_readcyclecounter(); tc_[k] += tn_ - ts_; ts_ = tn_; } while (0)
#define FSD_END(base, nk) do { if (lane == 0) { for (int k_ = 0; k_ < nk; ++k_) atomicAdd(&emo_fs_diag[base + k_], (unsigned long long)tc_[k_]); \
    atomicAdd(&emo_fs_diag[base + 14], (unsigned long long)(__builtin_readcyclecounter() - tstart_)); atomicAdd(&emo_fs_diag[base + 15], 1ull); } } while (0)
#else
#define FSD_BEGIN
#define FSD(k) do {} while (0)
#define FSD_END(base, nk) do {} while (0)
#endif

namespace {
constexpr int FS_NT = 256;                     // 4 waves
constexpr int FS_C = 32;                       // tokens per chunk
constexpr int FS_LDF = 136;                    // feature image row stride (128 f + 8): perm 8-B reads conflict-free
constexpr float FS_LOG2E = 1.4426950408889634f;

__device__ __forceinline__ f32x4 mma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
    return (bf16x8){(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
}
__device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
    const bf16x4 t = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *(bf16x4*)p = t;
}

// sum over the four 16-lane rows of a wave, result in every lane: two VALU lane swaps instead of two LDS round trips (ds_bpermute):
// permlane32_swap(u, u) = {[lo, lo], [hi, hi]}, permlane16_swap(u, u) = {[r0, r0, r2, r2], [r1, r1, r3, r3]}
__device__ __forceinline__ float fs_sum_rows(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t w = __builtin_bit_cast(uint32_t, y);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}
// |x_t|^2 of the 16 tokens of a tile, in every lane of column t: the diagonal of the Gram matrix X X^T from two MFMAs (the tile's
// operand registers serve as A AND B operand), instead of 16 unpack + 16 FMA VALU instructions per lane — r03 PMC: the first version of
// the forward kernel issued 456 VALU instructions per 32-token chunk and wave, 160 of them for these sums, and VALU issue was 33 % of
// every wave's cycles (82 % of a SIMD's issue time with two waves).  Accumulator (g, c) register r = G[4 g + r][c]: the diagonal element
// of column c sits in row group g = c / 4, register c % 4; fs_sum_rows then spreads it over the four row groups.
__device__ __forceinline__ float fs_sumsq_gram(const bf16x8& x0, const bf16x8& x1, int g, int c) {
    f32x4 gm = mma32(x0, x0, zero4());
    gm = mma32(x1, x1, gm);
    const int r = c & 3;
    const float d = r == 0 ? gm[0] : r == 1 ? gm[1] : r == 2 ? gm[2] : gm[3];
    return fs_sum_rows(g == (c >> 2) ? d : 0.f);
}
// FAVOR+ features of NT 16-token tiles for the wave's 16 projections, all tiles side by side (independent chains: the first version ran
// one tile after the other — operand reads -> 2 MFMAs -> 16-deep |x|^2 chain -> two LDS shuffles -> 8 exps -> stores — and a chunk's four
// tiles took 3000 cycles for ~900 cycles of issue).  Lane (g = l/16, c = l%16) gets, for token c of tile j, the features of
// m = 16 w + 4 g + r: p[j][r] = exp(u - off), n[j][r] = exp(-u - off), u = c_s x.w_m, off = c_s^2 |x|^2 / 2 + ln(F) / 2.
// x[j][s]: the token's row as B operand (lane: 8 consecutive d at 32 s + 8 g), wop[s]: omega^T rows of the slice as A operand.
template <int NT>
__device__ __forceinline__ void fs_features(const bf16x8 (&wop)[2], const bf16x8 (&x)[NT][2], float cs2, float c2h, float hl, int g, int c,
                                            float (&p)[NT][4], float (&n)[NT][4]) {
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = mma32(wop[0], x[j][0], zero4());
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = mma32(wop[1], x[j][1], acc[j]);
    float o[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) o[j] = c2h * fs_sumsq_gram(x[j][0], x[j][1], g, c) + hl;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u = cs2 * acc[j][r];
            p[j][r] = __builtin_amdgcn_exp2f(u - o[j]);
            n[j][r] = __builtin_amdgcn_exp2f(-u - o[j]);
        }
}

// =============================================================================================== forward
// Operand stream: the chunk's q / k / v rows arrive by LDS-DMA (`global_load_lds`, 1 KB per wave instruction, three per wave and chunk)
// into rings — q, k: 3 slots, v: 4 slots (v is still read while the q / k slot of the same chunk is being refilled) — issued THREE
// chunks ahead, so that two chunks (24 KB per workgroup) are in flight at any time: with 512 scans on 256 CUs the kernel is a latency
// problem (the first register-prefetch version waited one full HBM round trip + the store acknowledgements per 32-token chunk and ran
// at the generic kernel's 0.27 ms).  Ring rows are 128 B with the 16-B pieces XOR-swizzled by fs_sw(row) on the DMA *source* address
// and on the read (fragment reads of 16 rows x one piece would otherwise be 8-way bank conflicted).  The output rows of chunk i are
// stored one iteration later, right after the counted wait, so that a `s_waitcnt vmcnt(3)` never has young stores in front of it.
// Phase A (per chunk): the wave computes the features of its 16 projections for the chunk's q and k rows and writes them into the
// shared row-major images QF / KF [32][128] (double buffered).  Barrier.  Phase B: the wave owns the output columns d in [16w, 16w+16)
// and the state slice S[all f][16w..] (+ the "ones" tile S[f][64] = z[f]): A^T = masked Kf Qf^T (full f), num^T = V^T A^T + S^T Qf^T,
// S += Kf^T V.
constexpr int FS_ROWB = 128;                           // ring row: 64 bf16
constexpr int FS_TILEB = FS_C * FS_ROWB;               // one tensor of one chunk: 4 KB

template <int N> __device__ __forceinline__ void fs_wait();
template <> __device__ __forceinline__ void fs_wait<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void fs_wait<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
template <> __device__ __forceinline__ void fs_wait<5>() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
template <> __device__ __forceinline__ void fs_wait<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void fs_wait<10>() { asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
__device__ __forceinline__ void fs_barrier() {         // LDS writes of this wave visible, then the workgroup barrier; DMA stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// LDS-DMA as INLINE ASM (guide 5.7): with the builtin, hipcc answers every later transpose read (ds_read_b64_tr_b16 builtin) of ANY LDS
// address with `s_waitcnt vmcnt(0)` — it cannot prove that the read does not alias the LDS-DMA it has seen issued — which drained the
// operand ring once per chunk.  The compiler never sees this DMA; its completion is counted by hand (fs_wait).  lds_dst: wave-uniform
// LDS byte address of the 1 KB the wave's 64 lanes fill; M0 is saved / restored inside the statement.
__device__ __forceinline__ void fs_dma16(const void* gsrc, uint32_t lds_dst) {
#ifdef FS_NO_DMA
    return;                                            // timing ablation only: the kernels then compute on whatever the LDS holds
#endif
#ifdef FS_NO_M0SAVE
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
    return;
#endif
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// load_perm (emo_lds_mma.h) for the feature images of these kernels, with the two 8-B reads kept APART: from one base register with constant
// offsets hipcc fuses them into ds_read2_b64 (or ds_read2st64_b64 across k-steps), which is serviced at half the rate of two ds_read_b64 and
// banked modulo 32 — there rows c and c + 8 of the 136-element stride collide (r04: tools/ubench/lds_patterns + PMC: 8.9 LDS cycles and 4.0
// conflict cycles per LDS instruction in the forward kernel, LDS 50 % busy, SQ_WAIT_INST_LDS on a third of the cycles).  The k-step offset goes
// through an opaque SGPR, so every read has its own address register (one v_add each) and stays a plain ds_read_b64: 2 cycles, conflict-free.
__device__ __forceinline__ int fs_opaque(int x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ bf16x8 fs_load_perm(const bf16_t* img, int row0, int step, int lane) {
    const bf16_t* p = img + (row0 + (lane & 15)) * FS_LDF + (lane >> 4) * 4;
    const bf16x4 lo = *(const bf16x4*)(p + fs_opaque((2 * step) * 16)), hi = *(const bf16x4*)(p + fs_opaque((2 * step + 1) * 16));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// Ring-row swizzle: the 16-B piece p of row r sits at piece p ^ fs_sw(r).  r03 used r & 7, which is conflict-free for the 16-B fragment reads and
// the transpose reads but puts rows c and c + 8 of an 8-B permuted read (fs_ring_perm, xown: 32 lanes = 16 rows x 2 halves = one full
// 256-B bank row) on the same banks: 2-way on every such read.  This map — bit 0 = row bit 2, bit 1 = row bit 1, bit 2 = row bits 2 ^ 3 —
// is conflict-free for all three read patterns (tools/lds_conflicts.py: the guide's lane-group model, exhaustive search over the linear
// maps of the row bits; confirmed per pattern with SQ_LDS_BANK_CONFLICT in tools/ubench/lds_patterns.hip).
__device__ __forceinline__ int fs_sw(int row) { return ((row >> 2) & 1) | (row & 2) | ((((row >> 2) ^ (row >> 3)) & 1) << 2); }
// B (or A) operand fragment of a ring tile: row = row0 + c, the 8 elements at 32 s + 8 g
__device__ __forceinline__ bf16x8 fs_ring_frag(const char* tile, int row0, int s, int g, int c) {
    return *(const bf16x8*)(tile + (row0 + c) * FS_ROWB + (((4 * s + g) ^ fs_sw(row0 + c)) << 4));
}
// load_perm_tr (emo_lds_mma.h) on a swizzled ring tile: permuted-k fragment of the transpose, operand rows = columns col0 .. col0 + 15
__device__ __forceinline__ bf16x8 fs_ring_perm_tr(const char* tile, int col0, int lane) {
    const int i = lane & 15, kc = lane >> 4;
    bf16x8 v;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = h * 16 + kc * 4 + (i >> 2), col = col0 + (i & 3) * 4;
        const char* p = tile + row * FS_ROWB + ((((col >> 3)) ^ fs_sw(row)) << 4) + (col & 7) * 2;
        const short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
        const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
        v[h * 4 + 0] = tb[0]; v[h * 4 + 1] = tb[1]; v[h * 4 + 2] = tb[2]; v[h * 4 + 3] = tb[3];
    }
    return v;
}
__device__ __forceinline__ uint32_t fs_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

// ---- 8-byte LDS reads as INLINE ASM with a literal offset (r06; r05 verdict item 2 (i)).  fs_load_perm / fs_ring_perm above keep hipcc from fusing
// neighbouring reads into ds_read2_b64 by routing every offset through an opaque register — one v_add per read: 218 of the dk / dv loop's 672 VALU
// instructions were 32-bit address arithmetic (profiles/r05_isa_census.txt).  A literal `offset:` can neither be fused nor does it need an add: the
// per-lane part of an address pattern lives in ONE register per XOR class of the swizzle and every read of the pattern is that register + a constant.
// hipcc does not see these reads: their results are only used behind fs_pin<N>, a counted `s_waitcnt lgkmcnt(N)` that takes the registers as "+v"
// operands (N = asm reads issued after the last one the pin covers; LDS operations retire in order, so reads / writes the compiler issues in between
// only make the wait stricter).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 fs_rd8(uint32_t addr, int off) {
    u32x2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(off));
    return r;
}
__device__ __forceinline__ bf16x8 fs_join(u32x2 lo, u32x2 hi) { return __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]}); }
template <int N> __device__ __forceinline__ void fs_pin(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
    static_assert(N == 0 || N == 4 || N == 8 || N == 12 || N >= 15, "");
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if constexpr (N == 12) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));      // (4-bit counter: a smaller count than the true one only waits longer)
}
// feature image [32][FS_LDF] (fs_load_perm's fragment): lane part of the address = (c * FS_LDF + 4 g) * 2 + image base; row0 / k-step in the literal
__device__ __forceinline__ uint32_t fs_lp_lane(int g, int c) { return (uint32_t)((c * FS_LDF + 4 * g) * 2); }
__device__ __forceinline__ bf16x8 fs_lp_asm(uint32_t base, int img_off, int row0, int step) {
    const int o = img_off + (row0 * FS_LDF + 32 * step) * 2;
    return fs_join(fs_rd8(base, o), fs_rd8(base, o + 32));
}
// swizzled ring tile (fs_ring_perm's fragment): piece index (4 s + (g >> 1) (+ 2)) ^ fs_sw(row) — row0 is a multiple of 16 and fs_sw only looks at row
// bits 1-3, so with P = (g >> 1) ^ fs_sw(c) the four (k-step, half) combinations are the XOR classes P ^ {0, 2, 4, 6} of ONE lane constant:
// rp[s][h] = slot base + c * 128 + (g & 1) * 8 + ((P ^ (4 s + 2 h)) << 4); tensor and row block go into the literal
__device__ __forceinline__ void fs_rp_lane(int g, int c, uint32_t (&l)[2][2]) {
    const int P = (g >> 1) ^ fs_sw(c);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int h = 0; h < 2; ++h) l[s_][h] = (uint32_t)(c * FS_ROWB + (g & 1) * 8 + ((P ^ (4 * s_ + 2 * h)) << 4));
}
__device__ __forceinline__ bf16x8 fs_rp_asm(const uint32_t (&rp)[2][2], int tensor, int row0, int s_) {
    const int o = tensor * FS_TILEB + row0 * FS_ROWB;
    return fs_join(fs_rd8(rp[s_][0], o), fs_rd8(rp[s_][1], o));
}

__global__ __launch_bounds__(FS_NT, 2) void favor_fs_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                int64_t ld, const float* __restrict__ omega, bf16_t* __restrict__ out, int64_t ld_out,
                                                                float* __restrict__ den_g, float* __restrict__ state_S, float* __restrict__ state_z,
                                                                int64_t Tfull, int64_t H, float eps, const float* __restrict__ S_ws,
                                                                const float* __restrict__ z_ws, int P, int64_t Ts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* QKr = smem;                                  // [3 slots][q, k][32 rows][128 B]
    char* Vr = QKr + 3 * 2 * FS_TILEB;                 // [4 slots][32 rows][128 B]
    bf16_t* QFb = (bf16_t*)(Vr + 4 * FS_TILEB);        // [2][32][LDF]
    bf16_t* KFb = QFb + 2 * FS_C * FS_LDF;             // [2][32][LDF]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // segment-parallel scan (P > 1, B*H < 256): block = (b, h, segment); the segment starts from the sum of the state increments of the
    // segments before it (workspace written by the generic state-only pass, emo_favor.hip) and is otherwise a scan of T = its own length
    const int64_t bh = blockIdx.x / P, b = bh / H, h = bh % H;
    const int p_seg = (int)(blockIdx.x % P);
    const int64_t tbeg = (int64_t)p_seg * Ts, T = (tbeg + Ts < Tfull) ? Ts : Tfull - tbeg;
    const bf16_t* qb = q + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* kb = k + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* vb = v + (b * Tfull + tbeg) * ld + h * 64;
    bf16_t* ob = out + (b * Tfull + tbeg) * ld_out + h * 64;
    float* dg = den_g + bh * Tfull + tbeg;
    const float cs = rsqrtf(sqrtf(64.f));
    const float cs2 = cs * FS_LOG2E, c2h = 0.5f * cs * cs * FS_LOG2E, hl = 0.5f * logf(128.f) * FS_LOG2E;

    bf16x8 wop[2];                                     // omega^T rows m = 16 w + c, k = d
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) wop[s][e] = (bf16_t)omega[(32 * s + 8 * g + e) * 64 + 16 * w + c];
    bf16x8 oneop;                                      // the "ones" tile of V' (columns 64..79, column 64 = 1) as A / B operand
#pragma unroll
    for (int e = 0; e < 8; ++e) oneop[e] = (bf16_t)(c == 0 ? 1.f : 0.f);

    f32x4 S[8][2];                                     // S[ft][0]: rows f = 16 ft + 4 g + r, column d = 16 w + c;  S[ft][1]: the ones tile (c == 0: z[f])
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) { S[ft][0] = zero4(); S[ft][1] = zero4(); }
    for (int pp = 0; pp < p_seg; ++pp) {               // carried-in state
        const float* Sp = S_ws + (bh * P + pp) * (int64_t)(128 * 64);
        const float* zp = z_ws + (bh * P + pp) * (int64_t)128;
        // (all loads of an increment first, no lane-dependent branch between them: with the `if (c == 0)` inside the loop every load was
        // followed by its own wait — 8 us per increment, 56 of the segmented forward's 90 us at eight segments, r04)
        float ts[8][4];
        f32x4 tz[8];
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ts[ft][r] = Sp[(16 * ft + 4 * g + r) * 64 + 16 * w + c];
            tz[ft] = *(const f32x4*)(zp + 16 * ft + 4 * g);
        }
        const float zm = c == 0 ? 1.f : 0.f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) { S[ft][0][r] += ts[ft][r]; S[ft][1][r] += zm * tz[ft][r]; }
    }

    const int nch = (int)(T / FS_C);
    const uint32_t qk_lds = __builtin_amdgcn_readfirstlane(fs_lds_addr(QKr)), vr_lds = __builtin_amdgcn_readfirstlane(fs_lds_addr(Vr));
    // DMA: this wave moves rows 8 w .. 8 w + 7 of q, k and v (lane: row 8 w + lane / 8, physical piece lane % 8 <- logical piece ^ fs_sw(row))
    const int64_t src_off = (int64_t)(8 * w + (lane >> 3)) * ld + (((lane & 7) ^ fs_sw(8 * w + (lane >> 3))) << 3);
    auto issue = [&](int n) {
        const int64_t o = (int64_t)n * FS_C * ld + src_off;
        const uint32_t qk = qk_lds + (n % 3) * 2 * FS_TILEB + w * 1024;
        fs_dma16(qb + o, qk);
        fs_dma16(kb + o, qk + FS_TILEB);
        fs_dma16(vb + o, vr_lds + (n & 3) * FS_TILEB + w * 1024);
    };
    issue(0);
    if (nch > 1) issue(1);
    if (nch > 2) issue(2);
    if (nch > 2) fs_wait<6>(); else if (nch > 1) fs_wait<3>(); else fs_wait<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x4 o_prev[2] = {};                             // chunk i - 1's output rows, stored during iteration i
    float d_prev[2] = {0.f, 0.f};
    const uint32_t lpb0 = fs_lds_addr(QFb) + fs_lp_lane(g, c);       // lane part of the asm feature-image reads (fs_lp_asm)
    constexpr int KFO = 2 * FS_C * FS_LDF * 2;         // KF images behind the two QF images
#ifdef EMO_DIAG
    uint64_t tc[8] = {}, ts = __builtin_readcyclecounter(), tstart = ts;
#define FS_STAMP(k) do { const uint64_t tn_ = __builtin_readcyclecounter(); tc[k] += tn_ - ts; ts = tn_; } while (0)
#else
#define FS_STAMP(k) do {} while (0)
#endif
    for (int i = 0; i < nch; ++i) {
        const int64_t t0 = (int64_t)i * FS_C;
        bf16_t* QF = QFb + (i & 1) * FS_C * FS_LDF;
        bf16_t* KF = KFb + (i & 1) * FS_C * FS_LDF;
        const char* Xq = QKr + (i % 3) * 2 * FS_TILEB;
        const char* Xk = Xq + FS_TILEB;
        const char* VB = Vr + (i & 3) * FS_TILEB;
        // ---------------- phase A: tiles 0, 1 = q rows 0-15, 16-31; tiles 2, 3 = k rows
        {
            bf16x8 x[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) x[j][s] = fs_ring_frag(j < 2 ? Xq : Xk, 16 * (j & 1), s, g, c);
            float p[4][4], n[4][4];
            fs_features<4>(wop, x, cs2, c2h, hl, g, c, p, n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16_t* dst = (j < 2 ? QF : KF) + (16 * (j & 1) + c) * FS_LDF + 16 * w + 4 * g;
                st4(dst, p[j][0], p[j][1], p[j][2], p[j][3]);
                st4(dst + 64, n[j][0], n[j][1], n[j][2], n[j][3]);
            }
        }
        FS_STAMP(0);
        // chunk i + 1 landed (this wave's pieces; the barrier below covers the other waves'); chunk i + 2 stays in flight
        if (i + 2 < nch) fs_wait<3>(); else fs_wait<0>();
        FS_STAMP(1);
        if (i > 0) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                *(bf16x4*)(ob + (t0 - FS_C + 16 * tt + c) * ld_out + 16 * w + 4 * g) = o_prev[tt];
                if (w == 0 && g == 0) dg[t0 - FS_C + 16 * tt + c] = d_prev[tt];
            }
        }
        fs_barrier();
        FS_STAMP(2);
        if (i + 3 < nch) issue(i + 3);                     // q / k slot of chunk i and v slot of chunk i - 1 are free
#ifdef EMO_DIAG
        FS_STAMP(6);
#endif
        // ---------------- phase B
        bf16x8 qf[2][4], kf[2][4];
        const uint32_t lpb = lpb0 + (i & 1) * (FS_C * FS_LDF * 2);
#pragma unroll
        for (int hs = 0; hs < 2; ++hs)                 // k-steps 0, 1 of every fragment first (16 reads), then 2, 3
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) { qf[tt][2 * hs + u] = fs_lp_asm(lpb, 0, 16 * tt, 2 * hs + u); kf[tt][2 * hs + u] = fs_lp_asm(lpb, KFO, 16 * tt, 2 * hs + u); }
        const bf16x8 vop = fs_ring_perm_tr(VB, 16 * w, lane);
        fs_pin<24>(qf[0][0], kf[0][0], qf[1][0], kf[1][0]);
        fs_pin<16>(qf[0][1], kf[0][1], qf[1][1], kf[1][1]);
#ifdef EMO_DIAG
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(vop), "v"(qf[0][0]), "v"(qf[1][3]), "v"(kf[0][0]), "v"(kf[1][3]));
        FS_STAMP(7);
#endif
        // A^T(jt, tt): rows j = 16 jt + 4 g + r, column t = 16 tt + c.  Six independent accumulator chains (even / odd k steps of the three tiles)
        f32x4 aa[3][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            aa[0][u] = mma32(kf[0][u], qf[0][u], zero4());
            aa[1][u] = mma32(kf[0][u], qf[1][u], zero4());
            aa[2][u] = mma32(kf[1][u], qf[1][u], zero4());
        }
        fs_pin<8>(qf[0][2], kf[0][2], qf[1][2], kf[1][2]);
        fs_pin<0>(qf[0][3], kf[0][3], qf[1][3], kf[1][3]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            aa[0][u] = mma32(kf[0][2 + u], qf[0][2 + u], aa[0][u]);
            aa[1][u] = mma32(kf[0][2 + u], qf[1][2 + u], aa[1][u]);
            aa[2][u] = mma32(kf[1][2 + u], qf[1][2 + u], aa[2][u]);
        }
        f32x4 a00 = aa[0][0] + aa[0][1], a11 = aa[2][0] + aa[2][1];
        const f32x4 a01 = aa[1][0] + aa[1][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                      // causal mask inside the diagonal tiles
            const bool keep = (4 * g + r) <= c;
            a00[r] = keep ? a00[r] : 0.f;
            a11[r] = keep ? a11[r] : 0.f;
        }
        bf16x8 at[2];
        at[0] = pack8(a00, zero4());
        at[1] = pack8(a01, a11);
#ifdef EMO_DIAG
        asm volatile("" :: "v"(at[0]), "v"(at[1]));
#endif
        FS_STAMP(3);
        // num^T = V^T A^T + S^T Qf^T for the wave's d columns (nm[tt][0]) and the ones tile (nm[tt][1]): four independent chains
        f32x4 nm[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { nm[tt][0] = mma32(vop, at[tt], zero4()); nm[tt][1] = mma32(oneop, at[tt], zero4()); }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 s0 = pack8(S[2 * s][0], S[2 * s + 1][0]), s1 = pack8(S[2 * s][1], S[2 * s + 1][1]);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { nm[tt][0] = mma32(s0, qf[tt][s], nm[tt][0]); nm[tt][1] = mma32(s1, qf[tt][s], nm[tt][1]); }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const float dn = __shfl(nm[tt][1][0], c, 64) + eps;   // row d' = 64 of the ones tile lives in lanes 0..15, register 0
            const float inv = 1.f / dn;
            o_prev[tt] = (bf16x4){(bf16_t)(nm[tt][0][0] * inv), (bf16_t)(nm[tt][0][1] * inv), (bf16_t)(nm[tt][0][2] * inv), (bf16_t)(nm[tt][0][3] * inv)};
            d_prev[tt] = dn;
        }
#ifdef EMO_DIAG
        asm volatile("" :: "v"(o_prev[0]), "v"(o_prev[1]));
#endif
        FS_STAMP(4);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const bf16x8 kT = load_perm_tr(KF, FS_LDF, 16 * ft, 0, lane);
            S[ft][0] = mma32(kT, vop, S[ft][0]);
            S[ft][1] = mma32(kT, oneop, S[ft][1]);
        }
#ifdef EMO_DIAG
        asm volatile("" :: "v"(S[0][0]), "v"(S[7][1]), "v"(S[3][0]), "v"(S[5][1]));
#endif
        FS_STAMP(5);
    }
#ifdef EMO_DIAG
    if (state_S && lane == 0) {                        // diagnostics: the state pointer receives the cycle counters instead of the state
        unsigned long long* dgp = (unsigned long long*)state_S;
        for (int k_ = 0; k_ < 8; ++k_) atomicAdd(dgp + k_, (unsigned long long)tc[k_]);
        atomicAdd(dgp + 8, (unsigned long long)(__builtin_readcyclecounter() - tstart));
        atomicAdd(dgp + 9, 1ull);
    }
    if (state_S) return;
#endif
    {
        const int64_t t0 = (int64_t)nch * FS_C;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            *(bf16x4*)(ob + (t0 - FS_C + 16 * tt + c) * ld_out + 16 * w + 4 * g) = o_prev[tt];
            if (w == 0 && g == 0) dg[t0 - FS_C + 16 * tt + c] = d_prev[tt];
        }
    }
    if (state_S && p_seg == P - 1) {
        float* So = state_S + bh * (int64_t)(128 * 64);
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) So[(16 * ft + 4 * g + r) * 64 + 16 * w + c] = S[ft][0][r];
        if (w == 0 && c == 0) {
            float* zo = state_z + bh * 128;
#pragma unroll
            for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) zo[16 * ft + 4 * g + r] = S[ft][1][r];
        }
    }
}

// =============================================================================================== backward: dq (forward sweep)
// Per (b, h): dPhi_q[t][f] = sum_{j<=t} P[t][j] Kf[j][f] + sum_d' G'[t][d'] S'[f][d'],  P[t][j] = dN_t.v_j + dD_t,  G' = [dN | dD],
// S' = running sum of Kf_j (x) [v_j | 1];  dU = dPhi+ Phi+ - dPhi- Phi-;  dq = c (W dU - c q sum_f dPhi Phi).
// Wave w owns the feature slice f in {16w..16w+15} u {64+16w..} for everything up to dU (state S'^T[all d'][slice] in accumulators, its
// own K features transposed through a PRIVATE 2-KB LDS image, its own Q features in registers for the Jacobian); the waves then exchange
// dU (bf16, [32 t][64 m]) and their partial row sums, and each computes the dq columns d in [16w, 16w+16).  The P matrix and dN / dD are
// computed by every wave (6 + 4 MFMAs: cheaper than another barrier).  dD_t = -(dout_t.out_t)/den_t is the diagonal of an MFMA Gram
// product of the out and dout fragments; z.dD enters through the ones column of V' (16x16x16 MFMA against the z tile of the state).
// q, k, v, dout, out rows and den arrive through a 3-slot LDS-DMA ring (6 DMA instructions per wave and chunk).  One barrier per chunk.
constexpr int FS_SLOTB = 5 * FS_TILEB;                 // q, k, v, dout, out of one chunk
template <> __device__ __forceinline__ void fs_wait<12>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
__device__ __forceinline__ void fs_dma4(const void* gsrc, uint32_t lds_dst) {   // 4 B per lane (den): 256 B per wave instruction
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ f32x4 mma16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, a), __builtin_bit_cast(short4v, b), c, 0, 0, 0);
}
// permuted-k fragment (k-step s: elements 32 s + 4 g .. +3 and 32 s + 16 + 4 g .. +3) of row row0 + c of a swizzled ring tile
__device__ __forceinline__ bf16x8 fs_ring_perm(const char* tile, int row0, int s, int g, int c) {
    // (row0 goes through an opaque SGPR: with a constant the reads of rows c and c + 16 — same lane offsets, 2048 B apart — are fused into
    // ds_read2st64_b64, half the rate of two ds_read_b64 and banked modulo 32; see fs_load_perm)
    const int row = row0 + c, ch = 4 * s + (g >> 1), sw = fs_sw(row);
    const char* base = tile + fs_opaque(row0 * FS_ROWB) + c * FS_ROWB + (g & 1) * 8;
    const bf16x4 lo = *(const bf16x4*)(base + ((ch ^ sw) << 4)), hi = *(const bf16x4*)(base + (((ch + 2) ^ sw) << 4));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ float fs_diag_sum_rows(const f32x4& gm, int g, int c) {   // diagonal element of column c, in every row group
    const int r = c & 3;
    const float d = r == 0 ? gm[0] : r == 1 ? gm[1] : r == 2 ? gm[2] : gm[3];
    return fs_sum_rows(g == (c >> 2) ? d : 0.f);
}
__device__ __forceinline__ bf16x8 fs_scale8(const bf16x8& x, float a) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)((float)x[e] * a);
    return r;
}

// PRE (r06): `dout` already holds dN = dout / den (the out-projection dgrad divided it in its epilogue, emo_hip.h: hdiv / emo_favor_attn_bwd_dn): no
// normaliser stream (5 DMA instructions per chunk instead of 6), no reciprocal, no rescaled operand copies — dD = -(dN . out) comes straight from
// the Gram diagonal.  r05 measured the same arithmetic removal with a timing-only build: -10 % / -17 % VALU in dq / dk-dv, -4.4 % time.
template <bool PRE>
__global__ __launch_bounds__(FS_NT, 2) void favor_fs_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                               int64_t ld, const float* __restrict__ omega, const bf16_t* __restrict__ out,
                                                               const bf16_t* __restrict__ dout, int64_t ld_out, const float* __restrict__ den_g,
                                                               bf16_t* __restrict__ dq, int64_t ld_d, int64_t Tfull, int64_t H,
                                                               const float* __restrict__ S_ws, const float* __restrict__ z_ws, int P, int64_t Ts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* RING = smem;                                 // [3 slots][q, k, v, dout, out][32 rows][128 B]
    char* DEN = RING + 3 * FS_SLOTB;                   // [3 slots][64 floats]
    char* KP = DEN + 3 * 256;                          // [4 waves][32 rows][64 B]   private K-feature images (8-B pieces XOR-swizzled)
    char* DU = KP + 4 * 2048;                          // [2][32 rows][128 B]        dU rows (16-B pieces XOR-swizzled)
    float* SA = (float*)(DU + 2 * FS_TILEB);           // [2][4 waves][32]           partial sum_f dPhi Phi
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bh = blockIdx.x / P, b = bh / H, h = bh % H;   // (b, h, segment): see the forward kernel
    const int p_seg = (int)(blockIdx.x % P);
    const int64_t tbeg = (int64_t)p_seg * Ts, T = (tbeg + Ts < Tfull) ? Ts : Tfull - tbeg;
    const bf16_t* qb = q + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* kb = k + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* vb = v + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* ob = out + (b * Tfull + tbeg) * ld_out + h * 64;
    const bf16_t* gb = dout + (b * Tfull + tbeg) * ld_out + h * 64;
    bf16_t* dqb = dq + (b * Tfull + tbeg) * ld_d + h * 64;
    const float* dnb = den_g + bh * Tfull + tbeg;
    const float cs = rsqrtf(sqrtf(64.f));
    const float cs2 = cs * FS_LOG2E, c2h = 0.5f * cs * cs * FS_LOG2E, hl = 0.5f * logf(128.f) * FS_LOG2E;

    bf16x8 wop[2], wrow[2];                            // omega^T rows m = 16 w + c (k = d);  omega rows d = 16 w + c (k = m)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wop[s][e] = (bf16_t)omega[(32 * s + 8 * g + e) * 64 + 16 * w + c];
            wrow[s][e] = (bf16_t)omega[(16 * w + c) * 64 + 32 * s + 8 * g + e];
        }
    bf16x8 oneop;
#pragma unroll
    for (int e = 0; e < 8; ++e) oneop[e] = (bf16_t)(c == 0 ? 1.f : 0.f);

    f32x4 ST[5][2];                                    // S'^T tiles: rows d' = 16 dl + 4 g + r (dl = 4: row 64 = z), column f = slice element c (plus / minus)
#pragma unroll
    for (int dl = 0; dl < 5; ++dl) { ST[dl][0] = zero4(); ST[dl][1] = zero4(); }
    for (int pp = 0; pp < p_seg; ++pp) {               // K-state carried in from the earlier segments: S'^T[d'][f] (+ row 64 = z)
        const float* Sp = S_ws + (bh * P + pp) * (int64_t)(128 * 64);
        const float* zp = z_ws + (bh * P + pp) * (int64_t)128;
        f32x4 t4[2][4];                                // (loads first, no lane-dependent branch between them: see the forward kernel)
        float tz[2];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int f = 64 * ph + 16 * w + c;
#pragma unroll
            for (int dl = 0; dl < 4; ++dl) t4[ph][dl] = *(const f32x4*)(Sp + f * 64 + 16 * dl + 4 * g);
            tz[ph] = zp[f];
        }
        const float gm = g == 0 ? 1.f : 0.f;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int dl = 0; dl < 4; ++dl) ST[dl][ph] += t4[ph][dl];
            ST[4][ph][0] += gm * tz[ph];
        }
    }

    const int nch = (int)(T / FS_C);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(fs_lds_addr(RING)), den_lds = __builtin_amdgcn_readfirstlane(fs_lds_addr(DEN));
    const int srow = 8 * w + (lane >> 3), spc = ((lane & 7) ^ fs_sw(srow)) << 3;
    const int64_t so_qkv = (int64_t)srow * ld + spc, so_o = (int64_t)srow * ld_out + spc;
    // One DMA instruction of chunk n (part 0..5: q, k, v, dout, out, den).  The six parts of a chunk are issued at six different points of
    // the following iteration instead of back to back: right after a barrier all eight waves of the CU used to push their six 1-KB requests
    // at once and the burst cost ~1300 cycles per wave and chunk (15 % of the kernel, s_memtime stamps).
    auto issue_part = [&](int n, int part) {
        if (n >= nch) return;
        const int64_t t0n = (int64_t)n * FS_C;
        const uint32_t dst = ring_lds + (n % 3) * FS_SLOTB + w * 1024;
        if (part == 0) fs_dma16(qb + t0n * ld + so_qkv, dst);
        else if (part == 1) fs_dma16(kb + t0n * ld + so_qkv, dst + FS_TILEB);
        else if (part == 2) fs_dma16(vb + t0n * ld + so_qkv, dst + 2 * FS_TILEB);
        else if (part == 3) fs_dma16(gb + t0n * ld_out + so_o, dst + 3 * FS_TILEB);
        else if (part == 4) fs_dma16(ob + t0n * ld_out + so_o, dst + 4 * FS_TILEB);
        else if (!PRE) {
            const int64_t tl = t0n + lane;
            fs_dma4(dnb + (tl < T ? tl : T - 1), den_lds + (n % 3) * 256);  // (every wave writes the same 256 B: keeps the DMA count uniform)
        }
    };
    constexpr int NPART = PRE ? 5 : 6;                 // DMA instructions per wave and chunk
    auto issue = [&](int n) {
#pragma unroll
        for (int part = 0; part < 6; ++part) issue_part(n, part);
    };
    issue(0);
    if (nch > 1) issue(1);
    if (nch > 2) issue(2);
    if (nch > 2) fs_wait<2 * NPART>(); else if (nch > 1) fs_wait<NPART>(); else fs_wait<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    char* KPw = KP + w * 2048;
    bf16x4 o_prev[2] = {};
    uint32_t rpl[2][2];                                // lane parts of the asm ring reads (fs_rp_asm)
    fs_rp_lane(g, c, rpl);
    FSD_BEGIN
    for (int i = 0; i < nch; ++i) {
        const int64_t t0 = (int64_t)i * FS_C;
        const char* Xq = RING + (i % 3) * FS_SLOTB;
        const char* Xk = Xq + FS_TILEB;
        const char* Xv = Xq + 2 * FS_TILEB;
        const char* Xg = Xq + 3 * FS_TILEB;
        const char* Xo = Xq + 4 * FS_TILEB;
        const float* dens = (const float*)(DEN + (i % 3) * 256);
        char* DUb = DU + (i & 1) * FS_TILEB;
        float* SAb = SA + (i & 1) * 128;
        // dN / dD operands first (ring reads + the Gram MFMAs), so that their latency runs under the feature phase's exponentials
        bf16x8 dfr[2][2], ofr[2][2], vr[2][2];
        f32x4 gmm[2];
        uint32_t rp[2][2];
        {
            const uint32_t slot = ring_lds + (i % 3) * FS_SLOTB;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int h_ = 0; h_ < 2; ++h_) rp[s_][h_] = rpl[s_][h_] + slot;
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) { dfr[tt][s_] = fs_rp_asm(rp, 3, 16 * tt, s_); ofr[tt][s_] = fs_rp_asm(rp, 4, 16 * tt, s_); }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)                 // the V rows of the P product: landed long before they are used
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) vr[jt][s_] = fs_rp_asm(rp, 2, 16 * jt, s_);
        fs_pin<16>(dfr[0][0], ofr[0][0], dfr[0][1], ofr[0][1]);
        fs_pin<8>(dfr[1][0], ofr[1][0], dfr[1][1], ofr[1][1]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            gmm[tt] = mma32(ofr[tt][0], dfr[tt][0], zero4());
            gmm[tt] = mma32(ofr[tt][1], dfr[tt][1], gmm[tt]);
        }
        // ---------------- phase A: features of the slice
        float pq[2][4], nq[2][4];
        {
            bf16x8 x[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) x[j][s] = fs_ring_frag(j < 2 ? Xq : Xk, 16 * (j & 1), s, g, c);
            float p[4][4], n[4][4];
            fs_features<4>(wop, x, cs2, c2h, hl, g, c, p, n);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pq[tt][r] = p[tt][r]; nq[tt][r] = n[tt][r]; }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {           // K features -> private image row j: pieces g (plus) and 4 + g (minus), swizzled by bit 2 of the row
                const int row = 16 * jt + c, sw = ((row >> 2) & 1) << 2;
                st4((bf16_t*)(KPw + row * 64 + ((g ^ sw) << 3)), p[2 + jt][0], p[2 + jt][1], p[2 + jt][2], p[2 + jt][3]);
                st4((bf16_t*)(KPw + row * 64 + (((4 + g) ^ sw) << 3)), n[2 + jt][0], n[2 + jt][1], n[2 + jt][2], n[2 + jt][3]);
            }
        }
        FSD(0);
        if (i > 0) issue_part(i + 2, 3);
        bf16x4 xown[2];                                // q[t][16 w + 4 g ..] for the last line of dq
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int row = 16 * tt + c;
            xown[tt] = *(const bf16x4*)(Xq + row * FS_ROWB + (((2 * w + (g >> 1)) ^ fs_sw(row)) << 4) + (g & 1) * 8);
        }
        // dN = dout / den (B operands, permuted k), dD = -(dout . out) / den
        bf16x8 gop[2][2];
        float dD[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if constexpr (PRE) {
                dD[tt] = -fs_diag_sum_rows(gmm[tt], g, c);
                gop[tt][0] = dfr[tt][0];
                gop[tt][1] = dfr[tt][1];
            } else {
                const float inv = 1.f / dens[16 * tt + c];
                dD[tt] = -fs_diag_sum_rows(gmm[tt], g, c) * inv;
                gop[tt][0] = fs_scale8(dfr[tt][0], inv);
                gop[tt][1] = fs_scale8(dfr[tt][1], inv);
            }
        }
        FSD(1);
        if (i > 0) issue_part(i + 2, 4);
        // P^T(jt, tt) = V G^T + dD, masked j <= t
        bf16x8 at[2];
        {
            fs_pin<0>(vr[0][0], vr[0][1], vr[1][0], vr[1][1]);
            f32x4 p00 = mma32(vr[0][0], gop[0][0], zero4()), p01 = mma32(vr[0][0], gop[1][0], zero4()), p11 = mma32(vr[1][0], gop[1][0], zero4());
            p00 = mma32(vr[0][1], gop[0][1], p00);
            p01 = mma32(vr[0][1], gop[1][1], p01);
            p11 = mma32(vr[1][1], gop[1][1], p11);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = (4 * g + r) <= c;
                p00[r] = keep ? p00[r] + dD[0] : 0.f;
                p01[r] = p01[r] + dD[1];
                p11[r] = keep ? p11[r] + dD[1] : 0.f;
            }
            at[0] = pack8(p00, zero4());
            at[1] = pack8(p01, p11);
        }
        FSD(2);
        if (i > 0) issue_part(i + 2, 5);
        // K features of the slice, transposed (rows f, permuted k = j)
        bf16x8 kfT[2];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int row = hh * 16 + g * 4 + (c >> 2), pc = (4 * ph + (c & 3)) ^ (((row >> 2) & 1) << 2);
                const short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(KPw + row * 64 + (pc << 3)));
                const bf16x4 tb = __builtin_bit_cast(bf16x4, t);
                kfT[ph][hh * 4 + 0] = tb[0]; kfT[ph][hh * 4 + 1] = tb[1]; kfT[ph][hh * 4 + 2] = tb[2]; kfT[ph][hh * 4 + 3] = tb[3];
            }
        }
        // dPhi_q^T(ph, tt) = Kf^T P^T + S' G'^T (state BEFORE this chunk), then the Jacobian
        {
            bf16x8 sop[2][2];
            bf16x4 szop[2];
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                sop[ph][0] = pack8(ST[0][ph], ST[1][ph]);
                sop[ph][1] = pack8(ST[2][ph], ST[3][ph]);
                szop[ph] = (bf16x4){(bf16_t)ST[4][ph][0], (bf16_t)ST[4][ph][1], (bf16_t)ST[4][ph][2], (bf16_t)ST[4][ph][3]};
            }
            f32x4 dph[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const bf16x4 ddop = {(bf16_t)(g == 0 ? dD[tt] : 0.f), (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    f32x4 a = mma32(kfT[ph], at[tt], zero4());
                    a = mma32(sop[ph][0], gop[tt][0], a);
                    a = mma32(sop[ph][1], gop[tt][1], a);
                    dph[ph][tt] = mma16(szop[ph], ddop, a);
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float du[4], sa = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ap = dph[0][tt][r] * pq[tt][r], am = dph[1][tt][r] * nq[tt][r];
                    du[r] = ap - am;
                    sa += ap + am;
                }
                sa = fs_sum_rows(sa);
                const int row = 16 * tt + c;
                st4((bf16_t*)(DUb + row * FS_ROWB + (((2 * w + (g >> 1)) ^ fs_sw(row)) << 4) + (g & 1) * 8), du[0], du[1], du[2], du[3]);
                if (g == 0) SAb[w * 32 + row] = sa;
            }
        }
        FSD(3);
        // state: S'^T[d'][f] += sum_j V'^T[d'][j] Kf[j][f]
#pragma unroll
        for (int dl = 0; dl < 4; ++dl) {
            const bf16x8 vT = fs_ring_perm_tr(Xv, 16 * dl, lane);
            ST[dl][0] = mma32(vT, kfT[0], ST[dl][0]);
            ST[dl][1] = mma32(vT, kfT[1], ST[dl][1]);
        }
        ST[4][0] = mma32(oneop, kfT[0], ST[4][0]);
        ST[4][1] = mma32(oneop, kfT[1], ST[4][1]);
        FSD(4);
        if (i + 2 < nch) fs_wait<NPART>(); else fs_wait<0>();
        FSD(5);
        if (i > 0) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) *(bf16x4*)(dqb + (t0 - FS_C + 16 * tt + c) * ld_d + 16 * w + 4 * g) = o_prev[tt];
        }
        fs_barrier();
        FSD(6);
        issue_part(i + 3, 0);                              // ring slot of chunk i is free; parts 3..5 follow in the next iteration's phase A
        FSD(7);
        // ---------------- phase C: dq columns d in [16 w, 16 w + 16)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            issue_part(i + 3, 1 + tt);
            const int row = 16 * tt + c;
            const bf16x8 u0 = *(const bf16x8*)(DUb + row * FS_ROWB + (((g) ^ fs_sw(row)) << 4)), u1 = *(const bf16x8*)(DUb + row * FS_ROWB + (((4 + g) ^ fs_sw(row)) << 4));
            f32x4 dx = mma32(wrow[0], u0, zero4());
            dx = mma32(wrow[1], u1, dx);
            const float sa = (SAb[row] + SAb[32 + row]) + (SAb[64 + row] + SAb[96 + row]);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = cs * (dx[r] - cs * (float)xown[tt][r] * sa);
            o_prev[tt] = (bf16x4){(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        }
        FSD(8);
    }
    FSD_END(16, 9);
    {
        const int64_t t0 = (int64_t)nch * FS_C;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) *(bf16x4*)(dqb + (t0 - FS_C + 16 * tt + c) * ld_d + 16 * w + 4 * g) = o_prev[tt];
    }
}

// =============================================================================================== backward: dk, dv (reverse sweep)
// Chunks are walked from the last to the first; state = sums over the LATER tokens t of Qf_t (x) G'_t, kept twice:
//   RT  = R'^T[all d' (+ the dD column)][feature slice of the wave]   -> dPhi_k (split by feature, like dq)
//   RD  = R[all f][d in 16w..16w+15]                                  -> dV      (split by output column, like the forward)
// dPhi_k[j][f] = sum_{t>=j} P[t][j] Qf[t][f] + sum_d' V'[j][d'] R'[f][d'];  dV[j][d] = sum_{t>=j} A[t][j] dN[t][d] + sum_f Kf[j][f] R[f][d].
// Two barriers per chunk: after the features went to the shared QF / KF images (the A matrix and RD need every feature), and after dU / the
// row sums were published (dk columns need every m).  1 / den is folded into the OPERAND that is indexed by t (Qf^T for RT, the wave's own
// dN^T fragment for RD / dV), so the dout rows are used raw from the ring (transpose reads) and no scaled dN image is built.
// q, k, v, dout, out, den: 2-slot LDS-DMA ring (an iteration is long enough to cover the HBM round trip of the next-but-one chunk).
template <bool PRE>                                    // PRE: dout = dN already (see favor_fs_dq_kernel)
__global__ __launch_bounds__(FS_NT, 2) void favor_fs_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                int64_t ld, const float* __restrict__ omega, const bf16_t* __restrict__ out,
                                                                const bf16_t* __restrict__ dout, int64_t ld_out, const float* __restrict__ den_g,
                                                                bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t ld_d, int64_t Tfull, int64_t H,
                                                                const float* __restrict__ R_ws, const float* __restrict__ r_ws, int P, int64_t Ts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* RING = smem;                                 // [2 slots][q, k, v, dout, out][32 rows][128 B]
    char* DEN = RING + 2 * FS_SLOTB;                   // [2 slots][64 floats]
    bf16_t* QF = (bf16_t*)(DEN + 2 * 256);             // [32][LDF]  shared feature images (single buffer: two barriers per chunk)
    bf16_t* KF = QF + FS_C * FS_LDF;
    char* DU = (char*)(KF + FS_C * FS_LDF);            // [32 rows][128 B]  dU rows (16-B pieces XOR-swizzled)
    float* SA = (float*)(DU + FS_TILEB);               // [4 waves][32]
    float* PV = SA + 128;                              // [4 waves][dD 32 | -dot 32 | 1/den 32]   private
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t bh = blockIdx.x / P, b = bh / H, h = bh % H;   // (b, h, segment): the R state comes from the LATER segments
    const int p_seg = (int)(blockIdx.x % P);
    const int64_t tbeg = (int64_t)p_seg * Ts, T = (tbeg + Ts < Tfull) ? Ts : Tfull - tbeg;
    const bf16_t* qb = q + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* kb = k + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* vb = v + (b * Tfull + tbeg) * ld + h * 64;
    const bf16_t* ob = out + (b * Tfull + tbeg) * ld_out + h * 64;
    const bf16_t* gb = dout + (b * Tfull + tbeg) * ld_out + h * 64;
    bf16_t* dkb = dk + (b * Tfull + tbeg) * ld_d + h * 64;
    bf16_t* dvb = dv + (b * Tfull + tbeg) * ld_d + h * 64;
    const float* dnb = den_g + bh * Tfull + tbeg;
    const float cs = rsqrtf(sqrtf(64.f));
    const float cs2 = cs * FS_LOG2E, c2h = 0.5f * cs * cs * FS_LOG2E, hl = 0.5f * logf(128.f) * FS_LOG2E;

    bf16x8 wop[2], wrow[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wop[s][e] = (bf16_t)omega[(32 * s + 8 * g + e) * 64 + 16 * w + c];
            wrow[s][e] = (bf16_t)omega[(16 * w + c) * 64 + 32 * s + 8 * g + e];
        }
    const bf16x4 onecol = {(bf16_t)(g == 0 ? 1.f : 0.f), (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};   // V' column 64 = 1 as 16x16x16 B operand

    f32x4 RT[5][2], RD[8];
#pragma unroll
    for (int dl = 0; dl < 5; ++dl) { RT[dl][0] = zero4(); RT[dl][1] = zero4(); }
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) RD[ft] = zero4();
    for (int pp = p_seg + 1; pp < P; ++pp) {
        const float* Rp = R_ws + (bh * P + pp) * (int64_t)(128 * 64);
        const float* rp = r_ws + (bh * P + pp) * (int64_t)128;
        // (loads first, no lane-dependent branch between them: see the forward kernel)
        f32x4 t4[2][4];
        float tr[2], td[8][4];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int f = 64 * ph + 16 * w + c;
#pragma unroll
            for (int dl = 0; dl < 4; ++dl) t4[ph][dl] = *(const f32x4*)(Rp + f * 64 + 16 * dl + 4 * g);
            tr[ph] = rp[f];
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) td[ft][r] = Rp[(16 * ft + 4 * g + r) * 64 + 16 * w + c];
        const float gm = g == 0 ? 1.f : 0.f;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int dl = 0; dl < 4; ++dl) RT[dl][ph] += t4[ph][dl];
            RT[4][ph][0] += gm * tr[ph];
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) RD[ft][r] += td[ft][r];
    }

    const int nch = (int)(T / FS_C);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(fs_lds_addr(RING)), den_lds = __builtin_amdgcn_readfirstlane(fs_lds_addr(DEN));
    const int srow = 8 * w + (lane >> 3), spc = ((lane & 7) ^ fs_sw(srow)) << 3;
    const int64_t so_qkv = (int64_t)srow * ld + spc, so_o = (int64_t)srow * ld_out + spc;
    auto issue = [&](int n) {                          // n-th chunk of the reverse walk = chunk nch - 1 - n
        const int64_t t0n = (int64_t)(nch - 1 - n) * FS_C;
        const uint32_t dst = ring_lds + (n & 1) * FS_SLOTB + w * 1024;
        fs_dma16(qb + t0n * ld + so_qkv, dst);
        fs_dma16(kb + t0n * ld + so_qkv, dst + FS_TILEB);
        fs_dma16(vb + t0n * ld + so_qkv, dst + 2 * FS_TILEB);
        fs_dma16(gb + t0n * ld_out + so_o, dst + 3 * FS_TILEB);
        fs_dma16(ob + t0n * ld_out + so_o, dst + 4 * FS_TILEB);
        if constexpr (!PRE) {
            const int64_t tl = t0n + lane;
            fs_dma4(dnb + (tl < T ? tl : T - 1), den_lds + (n & 1) * 256);
        }
    };
    issue(0);
    if (nch > 1) issue(1);
    if (nch > 1) fs_wait<PRE ? 5 : 6>(); else fs_wait<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* PVw = PV + w * 96;
    bf16x4 dk_prev[2] = {}, dv_prev[2] = {};
    uint32_t rpl[2][2];                                // lane parts of the asm LDS reads (fs_rp_asm / fs_lp_asm)
    fs_rp_lane(g, c, rpl);
    const uint32_t lpb = fs_lds_addr(QF) + fs_lp_lane(g, c);
    constexpr int KFO = FS_C * FS_LDF * 2;             // KF image behind QF
    FSD_BEGIN
    for (int n = 0; n < nch; ++n) {
        const int64_t t0 = (int64_t)(nch - 1 - n) * FS_C;
        const char* Xq = RING + (n & 1) * FS_SLOTB;
        const char* Xk = Xq + FS_TILEB;
        const char* Xv = Xq + 2 * FS_TILEB;
        const char* Xg = Xq + 3 * FS_TILEB;
        const char* Xo = Xq + 4 * FS_TILEB;
        const float* dens = (const float*)(DEN + (n & 1) * 256);
        // ---------------- phase A1: features -> shared images; dN, dD; P
        {
            bf16x8 x[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) x[j][s] = fs_ring_frag(j < 2 ? Xq : Xk, 16 * (j & 1), s, g, c);
            float p[4][4], nn[4][4];
            fs_features<4>(wop, x, cs2, c2h, hl, g, c, p, nn);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16_t* dst = (j < 2 ? QF : KF) + (16 * (j & 1) + c) * FS_LDF + 16 * w + 4 * g;
                st4(dst, p[j][0], p[j][1], p[j][2], p[j][3]);
                st4(dst + 64, nn[j][0], nn[j][1], nn[j][2], nn[j][3]);
            }
        }
        FSD(0);
        bf16x4 xown[2];                                // k[j][16 w + 4 g ..]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int row = 16 * jt + c;
            xown[jt] = *(const bf16x4*)(Xk + row * FS_ROWB + (((2 * w + (g >> 1)) ^ fs_sw(row)) << 4) + (g & 1) * 8);
        }
        bf16x8 gA[2][2];                               // dN rows t as A operand (permuted k = d)
        uint32_t rp[2][2];
        {
            const uint32_t slot = ring_lds + (n & 1) * FS_SLOTB;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int h_ = 0; h_ < 2; ++h_) rp[s_][h_] = rpl[s_][h_] + slot;
        }
        bf16x8 dfr[2][2], ofr[2][2], vB1[2][2];       // 24 asm reads in flight: dout / out rows of both tiles, then the V rows of the P product
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) { dfr[tt][s_] = fs_rp_asm(rp, 3, 16 * tt, s_); ofr[tt][s_] = fs_rp_asm(rp, 4, 16 * tt, s_); }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) vB1[jt][s_] = fs_rp_asm(rp, 2, 16 * jt, s_);
        fs_pin<16>(dfr[0][0], ofr[0][0], dfr[0][1], ofr[0][1]);
        fs_pin<8>(dfr[1][0], ofr[1][0], dfr[1][1], ofr[1][1]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const bf16x8 d0 = dfr[tt][0], d1 = dfr[tt][1];
            const bf16x8 o0 = ofr[tt][0], o1 = ofr[tt][1];
            f32x4 gm = mma32(o0, d0, zero4());
            gm = mma32(o1, d1, gm);
            const float dot = fs_diag_sum_rows(gm, g, c);
            if constexpr (PRE) {
                if (g == 0) PVw[16 * tt + c] = -dot;   // dD_t = -(dN_t . out_t)
                gA[tt][0] = d0;
                gA[tt][1] = d1;
            } else {
                const float inv = 1.f / dens[16 * tt + c];
                if (g == 0) { PVw[16 * tt + c] = -dot * inv; PVw[32 + 16 * tt + c] = -dot; PVw[64 + 16 * tt + c] = inv; }
                gA[tt][0] = fs_scale8(d0, inv);
                gA[tt][1] = fs_scale8(d1, inv);
            }
        }
        FSD(1);
        bf16x8 pb[2];                                  // P[t][j] = dN_t.v_j + dD_t, t >= j: rows t = 16 tt + 4 g + r, column j
        {
            fs_pin<0>(vB1[0][0], vB1[0][1], vB1[1][0], vB1[1][1]);
            bf16x8 (&vB)[2][2] = vB1;                  // V rows j as B operand (permuted k = d)
            f32x4 p00 = mma32(gA[0][0], vB[0][0], zero4()), p10 = mma32(gA[1][0], vB[0][0], zero4()), p11 = mma32(gA[1][0], vB[1][0], zero4());
            p00 = mma32(gA[0][1], vB[0][1], p00);
            p10 = mma32(gA[1][1], vB[0][1], p10);
            p11 = mma32(gA[1][1], vB[1][1], p11);
            const f32x4 dd0 = *(const f32x4*)(PVw + 4 * g), dd1 = *(const f32x4*)(PVw + 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = (4 * g + r) >= c;
                p00[r] = keep ? p00[r] + dd0[r] : 0.f;
                p10[r] = p10[r] + dd1[r];
                p11[r] = keep ? p11[r] + dd1[r] : 0.f;
            }
            pb[0] = pack8(p00, p10);
            pb[1] = pack8(zero4(), p11);
        }
        FSD(2);
        fs_barrier();                                  // X: the shared feature images are complete
        FSD(3);
        // ---------------- phase B (operands are re-read from LDS where that shortens a live range: the kernel sits at the 256-register edge)
        bf16x8 ab[2];                                  // A[t][j] = Qf_t.Kf_j, t >= j
        {
            bf16x8 qfA[2][4], kfB[2][4];
#pragma unroll
            for (int hs = 0; hs < 2; ++hs)             // k-steps 0, 1 of every fragment first (16 reads), then 2, 3
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) { qfA[tt][2 * hs + u] = fs_lp_asm(lpb, 0, 16 * tt, 2 * hs + u); kfB[tt][2 * hs + u] = fs_lp_asm(lpb, KFO, 16 * tt, 2 * hs + u); }
            fs_pin<24>(qfA[0][0], kfB[0][0], qfA[1][0], kfB[1][0]);
            fs_pin<16>(qfA[0][1], kfB[0][1], qfA[1][1], kfB[1][1]);
            f32x4 aa[3][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                aa[0][u] = mma32(qfA[0][u], kfB[0][u], zero4());
                aa[1][u] = mma32(qfA[1][u], kfB[0][u], zero4());
                aa[2][u] = mma32(qfA[1][u], kfB[1][u], zero4());
            }
            fs_pin<8>(qfA[0][2], kfB[0][2], qfA[1][2], kfB[1][2]);
            fs_pin<0>(qfA[0][3], kfB[0][3], qfA[1][3], kfB[1][3]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                aa[0][u] = mma32(qfA[0][2 + u], kfB[0][2 + u], aa[0][u]);
                aa[1][u] = mma32(qfA[1][2 + u], kfB[0][2 + u], aa[1][u]);
                aa[2][u] = mma32(qfA[1][2 + u], kfB[1][2 + u], aa[2][u]);
            }
            f32x4 a00 = aa[0][0] + aa[0][1], a11 = aa[2][0] + aa[2][1];
            const f32x4 a10 = aa[1][0] + aa[1][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = (4 * g + r) >= c;
                a00[r] = keep ? a00[r] : 0.f;
                a11[r] = keep ? a11[r] : 0.f;
            }
            ab[0] = pack8(a00, a10);
            ab[1] = pack8(zero4(), a11);
        }
        FSD(4);
        // 1 / den of the lane's eight t positions (permuted k = t), the slice's Qf^T plain and scaled
        f32x4 iv0 = {}, iv1 = {};
        if constexpr (!PRE) { iv0 = *(const f32x4*)(PVw + 64 + 4 * g); iv1 = *(const f32x4*)(PVw + 64 + 16 + 4 * g); }
        auto scale_t = [&](const bf16x8& x) {
            if constexpr (PRE) return x;               // dN carries 1 / den already
            else return (bf16x8){(bf16_t)((float)x[0] * iv0[0]), (bf16_t)((float)x[1] * iv0[1]), (bf16_t)((float)x[2] * iv0[2]), (bf16_t)((float)x[3] * iv0[3]),
                            (bf16_t)((float)x[4] * iv1[0]), (bf16_t)((float)x[5] * iv1[1]), (bf16_t)((float)x[6] * iv1[2]), (bf16_t)((float)x[7] * iv1[3])};
        };
        bf16x8 qfTs[2];
        // dPhi_k^T(ph, jt) = Qf^T P + R' V'^T (state of the LATER chunks): four independent chains, then the Jacobian with the slice's K features
        {
            bf16x8 qfT[2], vB[2][2];
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                qfT[ph] = load_perm_tr(QF, FS_LDF, 16 * w + 64 * ph, 0, lane);
                qfTs[ph] = scale_t(qfT[ph]);
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int s = 0; s < 2; ++s) vB[jt][s] = fs_rp_asm(rp, 2, 16 * jt, s);
            f32x4 dph[2][2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) dph[jt][ph] = mma32(qfT[ph], pb[jt], zero4());
            fs_pin<0>(vB[0][0], vB[0][1], vB[1][0], vB[1][1]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const bf16x8 rop = pack8(RT[2 * s][ph], RT[2 * s + 1][ph]);
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) dph[jt][ph] = mma32(rop, vB[jt][s], dph[jt][ph]);
                }
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const bf16x4 rzop = {(bf16_t)RT[4][ph][0], (bf16_t)RT[4][ph][1], (bf16_t)RT[4][ph][2], (bf16_t)RT[4][ph][3]};
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) dph[jt][ph] = mma16(rzop, onecol, dph[jt][ph]);
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int row = 16 * jt + c;
                const bf16x4 kp = *(const bf16x4*)(KF + row * FS_LDF + 16 * w + 4 * g), kn = *(const bf16x4*)(KF + row * FS_LDF + 64 + 16 * w + 4 * g);
                float du[4], sa = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ap = dph[jt][0][r] * (float)kp[r], am = dph[jt][1][r] * (float)kn[r];
                    du[r] = ap - am;
                    sa += ap + am;
                }
                sa = fs_sum_rows(sa);
                st4((bf16_t*)(DU + row * FS_ROWB + (((2 * w + (g >> 1)) ^ fs_sw(row)) << 4) + (g & 1) * 8), du[0], du[1], du[2], du[3]);
                if (g == 0) SA[w * 32 + row] = sa;
            }
        }
        FSD(5);
        // dV^T for the wave's columns: dN^T A + R^T Kf^T  (two independent chains)
        const bf16x8 gTs = scale_t(fs_ring_perm_tr(Xg, 16 * w, lane));       // dN^T rows d = 16 w + i, permuted k = t
        {
            bf16x8 kp0[4], kp1[4];                     // K features of both tiles, permuted k = f (16 asm reads)
#pragma unroll
            for (int s = 0; s < 4; ++s) { kp0[s] = fs_lp_asm(lpb, KFO, 0, s); kp1[s] = fs_lp_asm(lpb, KFO, 16, s); }
            f32x4 a0 = mma32(gTs, ab[0], zero4()), a1 = mma32(gTs, ab[1], zero4());
            fs_pin<8>(kp0[0], kp1[0], kp0[1], kp1[1]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s == 2) fs_pin<0>(kp0[2], kp1[2], kp0[3], kp1[3]);
                const bf16x8 rd = pack8(RD[2 * s], RD[2 * s + 1]);
                a0 = mma32(rd, kp0[s], a0);
                a1 = mma32(rd, kp1[s], a1);
            }
            dv_prev[0] = (bf16x4){(bf16_t)a0[0], (bf16_t)a0[1], (bf16_t)a0[2], (bf16_t)a0[3]};
            dv_prev[1] = (bf16x4){(bf16_t)a1[0], (bf16_t)a1[1], (bf16_t)a1[2], (bf16_t)a1[3]};
        }
        FSD(6);
        // states
#pragma unroll
        for (int dl = 0; dl < 4; ++dl) {
            const bf16x8 gT = fs_ring_perm_tr(Xg, 16 * dl, lane);             // raw dout^T: 1 / den sits in qfTs
            RT[dl][0] = mma32(gT, qfTs[0], RT[dl][0]);
            RT[dl][1] = mma32(gT, qfTs[1], RT[dl][1]);
        }
        {
            // (PRE: the dD row itself — the 1 / den that the other form keeps in qfTs is already inside dN and dD)
            const f32x4 nd0 = *(const f32x4*)(PVw + (PRE ? 0 : 32) + 4 * g), nd1 = *(const f32x4*)(PVw + (PRE ? 0 : 32) + 16 + 4 * g);
            bf16x8 ndop = pack8(nd0, nd1);
#pragma unroll
            for (int e = 0; e < 8; ++e) ndop[e] = c == 0 ? ndop[e] : (bf16_t)0.f;  // row d' = 64 of G'^T: dD_t = -dot_t / den_t, 1 / den again in qfTs
            RT[4][0] = mma32(ndop, qfTs[0], RT[4][0]);
            RT[4][1] = mma32(ndop, qfTs[1], RT[4][1]);
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) RD[ft] = mma32(load_perm_tr(QF, FS_LDF, 16 * ft, 0, lane), gTs, RD[ft]);
        FSD(7);
        fs_wait<0>();                                   // chunk n + 1 landed (issued one iteration ago); older stores drained
        FSD(8);
        if (n > 0) {
            const int64_t tp = t0 + FS_C;               // the previous iteration's chunk
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) *(bf16x4*)(dkb + (tp + 16 * jt + c) * ld_d + 16 * w + 4 * g) = dk_prev[jt];
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) *(bf16x4*)(dvb + (t0 + 16 * jt + c) * ld_d + 16 * w + 4 * g) = dv_prev[jt];
        fs_barrier();                                  // Y: dU / row sums published, ring slot and feature images free
        FSD(9);
        if (n + 2 < nch) issue(n + 2);
        FSD(10);
        // ---------------- phase C: dk columns d in [16 w, 16 w + 16)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int row = 16 * jt + c;
            const bf16x8 u0 = *(const bf16x8*)(DU + row * FS_ROWB + ((g ^ fs_sw(row)) << 4)), u1 = *(const bf16x8*)(DU + row * FS_ROWB + (((4 + g) ^ fs_sw(row)) << 4));
            f32x4 dx = mma32(wrow[0], u0, zero4());
            dx = mma32(wrow[1], u1, dx);
            const float sa = (SA[row] + SA[32 + row]) + (SA[64 + row] + SA[96 + row]);
            dk_prev[jt] = (bf16x4){(bf16_t)(cs * (dx[0] - cs * (float)xown[jt][0] * sa)), (bf16_t)(cs * (dx[1] - cs * (float)xown[jt][1] * sa)),
                                   (bf16_t)(cs * (dx[2] - cs * (float)xown[jt][2] * sa)), (bf16_t)(cs * (dx[3] - cs * (float)xown[jt][3] * sa))};
        }
        FSD(11);
    }
    FSD_END(32, 12);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) *(bf16x4*)(dkb + (16 * jt + c) * ld_d + 16 * w + 4 * g) = dk_prev[jt];     // chunk 0 was the last one
}
}  // namespace

// which: 0 forward, 1 backward main passes (P > 1: the caller has run the generic state-only pass into S_ws / z_ws — for the backward the
// K-state increments before `stage` 1 (dq) and the R-state increments before `stage` 2 (dk, dv)).  stage: 0 = whole call (P == 1 only),
// 1 = dq, 2 = dk / dv.  Returns 0 when the shape / mode is not covered (the caller then runs the generic kernels), 1 when it was served.
// den == nullptr in a backward call: `dout` is dN = dout / den (emo_favor_attn_bwd_dn) — the PRE instances.
int emo_favor_fs_try(int which, int stage, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ld, const float* omega, bf16_t* out, int64_t ld_out,
                     float* den, float* sS, float* sz, const bf16_t* dout, bf16_t* dq, bf16_t* dk, bf16_t* dv, int64_t ld_d, int64_t B, int64_t T, int64_t H,
                     float eps, const float* ws_S, const float* ws_z, int P, int64_t Ts, hipStream_t st) {
    const char* e = getenv("EMO_FAVOR_FS");                // (read per call: tests toggle it in-process)  "0": generic kernels only
    if (e && atoi(e) == 0) return 0;
    if (T < FS_C || (T % FS_C) != 0 || B * H <= 0 || P < 1 || (P > 1 && (Ts % FS_C) != 0)) return 0;
    if ((ld & 7) || (ld_out & 7) || (ld_d & 3)) return 0;
    // 16-B LDS-DMA pieces of q / k / v / out / dout rows and 8-/16-B output stores: unaligned views fall back to the generic kernels
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)dout) & 15) || (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15)) return 0;
    if (P == 1) Ts = T;
    // short segments (B*H < 256; r03: the generic forward was faster there, 66 vs 89 us per layer at B=4 x T=2048 in 8 segments, because of the
    // carried-in state loop below)
    // (r04: with the carried-in state loaded without a branch between the loads the segmented forward is 44 us against the generic 66 — default on)
    dim3 grid((unsigned)(B * H * P));
    if (which == 0) {
        const size_t lds = (size_t)(3 * 2 + 4) * FS_TILEB + sizeof(bf16_t) * (size_t)(4 * FS_C * FS_LDF);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)favor_fs_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL(favor_fs_fwd_kernel, grid, dim3(FS_NT), lds, st, q, k, v, ld, omega, out, ld_out, den, sS, sz, T, H, eps, ws_S, ws_z, P, Ts);
        return 1;
    }
    const char* e2 = getenv("EMO_FAVOR_FS_BWD");               // "0": generic backward kernels
    if (e2 && atoi(e2) == 0) return 0;
    const bool pre = den == nullptr;
    if (stage == 0 || stage == 1) {
        const size_t lds = (size_t)3 * FS_SLOTB + 3 * 256 + 4 * 2048 + 2 * FS_TILEB + 2 * 4 * 32 * sizeof(float);
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)favor_fs_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)favor_fs_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        if (pre)
            hipLaunchKernelGGL(favor_fs_dq_kernel<true>, grid, dim3(FS_NT), lds, st, q, k, v, ld, omega, (const bf16_t*)out, dout, ld_out, (const float*)nullptr, dq, ld_d, T, H,
                               ws_S, ws_z, P, Ts);
        else
            hipLaunchKernelGGL(favor_fs_dq_kernel<false>, grid, dim3(FS_NT), lds, st, q, k, v, ld, omega, (const bf16_t*)out, dout, ld_out, (const float*)den, dq, ld_d, T, H,
                               ws_S, ws_z, P, Ts);
    }
    if (stage == 0 || stage == 2) {
        const size_t lds2 = (size_t)2 * FS_SLOTB + 2 * 256 + sizeof(bf16_t) * (size_t)(2 * FS_C * FS_LDF) + FS_TILEB + sizeof(float) * (128 + 4 * 96);
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void*)favor_fs_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            (void)hipFuncSetAttribute((const void*)favor_fs_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            attr2 = true;
        }
        if (pre)
            hipLaunchKernelGGL(favor_fs_dkv_kernel<true>, grid, dim3(FS_NT), lds2, st, q, k, v, ld, omega, (const bf16_t*)out, dout, ld_out, (const float*)nullptr, dk, dv, ld_d, T, H,
                               ws_S, ws_z, P, Ts);
        else
            hipLaunchKernelGGL(favor_fs_dkv_kernel<false>, grid, dim3(FS_NT), lds2, st, q, k, v, ld, omega, (const bf16_t*)out, dout, ld_out, (const float*)den, dk, dv, ld_d, T, H,
                               ws_S, ws_z, P, Ts);
    }
    return 1;
}
