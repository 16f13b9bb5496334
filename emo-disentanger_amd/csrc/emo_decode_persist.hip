// One-launch Performer decode step (BASELINE configs[3]: 32 streams x 2048 tokens): embedding -> 12 post-LN FAVOR+ layers -> logits for every
// stream in ONE persistent kernel.  Replaces the per-token chain of ~62 dependent launches of inference.py (reference loop:
// stage2_accompaniment/inference.py:250-277 -> MusicPerformer.forward with keep_last_only, music_performer.py:50-70), whose every launch sat
// on the ~5-6 us launch floor with < 2 MB of work (r03: 0.383 ms per token step = 0.09 of the HBM roofline).
//
// Structure (DESIGN.md "persistent decode"): 256 workgroups = 8 GROUPS x 32 members; group g = blockIdx % 8 — the XCD the dispatcher puts the
// block on, so a group's traffic normally stays inside one XCD, but NOTHING depends on that placement — owns streams 4g .. 4g+3 for the whole
// token step and never talks to another group.  Inside a group every GEMM of the layer is split by OUTPUT COLUMN over the 32 members (each
// member streams 1/32 of every weight matrix: 197 KB per layer, pre-packed on the host in MFMA fragment order so that a wave's load
// instruction is one contiguous KB that goes straight into the B-operand registers — no LDS staging, each weight byte is used once), the 4
// streams are rows 0-3 of a 16-row MFMA A operand, and the five dependent products of a layer are separated by five all-gather EDGES:
//   P1 q/k/v columns of head h (member = (h, j): dims 16j..16j+15 of q_h, k_h, v_h)        -> E2 (gathered by the 4 members of head h)
//   P2 FAVOR+ recurrent step of (head h, stream j): S += phi(k) (x) v, out = phi(q)^T S / ..  -> E3
//   P3 out-projection + bias + residual (pre-LN row)                                        -> E4
//   P4 LayerNorm1 (every member normalises the gathered rows itself) + FFN1 + ReLU         -> E5
//   P5 FFN2 + bias + residual (pre-LN row)                                                  -> E1 (next layer's P1 applies LayerNorm2)
// An edge is a buffer of 8-byte GRANULES {tag = epoch, value = 2 bf16} written by single agent-scope (sc1) stores and polled with agent-scope
// loads: the data is the flag, no fences, correct for any workgroup -> XCD placement (MI355X guide, Guideline 16 form R2).  Epochs count
// launches (a per-group counter the group's member 0 bumps when it is done) x phases, so nothing is zeroed per launch and hipGraph replay
// works.  A buffer is rewritten one layer later; between two uses lies at least one all-to-all edge, so every reader of the old contents has
// finished.  Every poll is bounded: a group that cannot make progress (a member not resident) writes an error code and leaves.
//
// Same-XCD fast path (measured r04, tools/pd_diag.py: a write-through granule costs the reader a fabric round trip of ~1.5 us per poll, two per
// edge): every launch starts with a CENSUS — each member publishes its XCC id through the placement-independent form, gathers the 32 ids of
// its group, and only if all are equal the group's producers switch to PLAIN stores, which stay in that XCD's L2 where the agent-scope (L1
// bypassing) polls of the other members hit.  A group that is spread over several XCDs keeps the write-through stores: speed depends on the
// placement, the result never does.
// Latency plan: a wave's loads return in order, so a poll cannot overtake an older weight load; everything a phase needs from HBM (its weight
// fragments, the recurrent-state slice, biases, LayerNorm parameters) is therefore requested TWO PHASES AHEAD (right after the gather of phase p - 2:
// measured r04, an HBM fetch takes 2-2.5 us here, a phase 1-1.5), and workgroup barriers are raw s_barrier + lgkmcnt (no vmcnt(0) fence).
//
// What bounds it (r04, tools/pd_diag.py, profiles/r04_pd_diag_v*.txt): 0.244 ms per token step of which ~0.21 in this kernel = 60 phases x ~3.3 us.
// A phase's own work (barrier -> MFMAs -> partial sums -> publish) is 0.3-0.9 us and an L2-local poll 0.2 us; the rest is the CU's memory queue:
// every XCD streams ALL weights for its 4 streams (8 x 76 MB + 200 MB of state = 0.8 GB per token step, ~4 TB/s), a phase's 16-64 KB weight
// burst takes 2-3 us to drain at ~25 GB/s per CU, and polls — from ANY wave of the CU — queue behind it.  A 12-wave variant with dedicated
// poller waves and two compute halves that prefetch 2-3 phases ahead (git history: "Persistent decode v6") moved the waiting from the compute
// side to the pollers' first poll (2.0 us behind a burst vs 0.2 us without) and measured 0.251 ms: not kept.  Next step: stop
// re-streaming the weights — layer-pipelined groups that keep 1/8 of the model resident in registers / LDS across tokens.
//
// Arithmetic mirrors the launch path's bf16 mode (emo_gemm skinny kernel, favor_decode_fast_kernel, layernorm_fwd_bf16_d512_kernel): bf16
// activations between products, fp32 accumulation, fp32 FAVOR+ state, LayerNorm statistics in fp32 from the bf16 row.
#include "emo_common.h"

namespace {
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned long long gu64;      // every word another workgroup reads: GLOBAL address space, never flat
constexpr int PD_D = 512, PD_H = 8, PD_DH = 64, PD_MF = 64, PD_F = 128, PD_FF = 2048;
constexpr int PD_GS = 4, PD_GM = 32, PD_NG = 8, PD_NT = 512, PD_NW = 8;
constexpr int PD_XS = PD_D + 8, PD_FS = PD_FF + 8;              // LDS row strides (bf16 elements): +16 B shifts the rows' banks
constexpr int OFF_CNT = 0, OFF_E1 = 8, OFF_E2 = OFF_E1 + PD_GS * PD_D / 2, OFF_E3 = OFF_E2 + PD_H * PD_GS * 96, OFF_E4 = OFF_E3 + PD_GS * PD_D / 2,
              OFF_E5 = OFF_E4 + PD_GS * PD_D / 2, OFF_CEN = OFF_E5 + PD_GS * PD_FF / 2, PD_GSTRIDE = OFF_CEN + PD_GM;
constexpr int PD_WS_WORDS = PD_NG * PD_GSTRIDE + 8;             // last 8 words: [0] = error code
constexpr int PD_MAX_LAYERS = 15;                               // epoch = launch * 128 + layer * 8 + phase (census: + 127)
constexpr long long PD_TIMEOUT = 5000000;                       // wall_clock64 ticks (100 MHz): 50 ms for the whole launch

struct PdLayer {            // one row of the caller's device table: 16 pointers
    const bf16_t* wqkv; const float* bqkv; const bf16_t* wo; const float* bo; const float* g1; const float* be1; const bf16_t* w1; const float* b1;
    const bf16_t* w2; const float* b2; const float* g2; const float* be2; const float* omega; float* S; float* z; void* pad;
};
struct PdArgs {
    const PdLayer* layers; int n_layers;
    const int64_t* tok; const int64_t* seg; const float* E; const float* Sg; const float* pe; float emb_scale; int64_t pos0; const int64_t* pos_ids;
    const bf16_t* wout; const float* bout; int n_token; float* logits; int n_streams; u64* sync; float eps, ln_eps;
    int flags;      // bit 0: non-temporal weight loads
    u64* diag;      // optional [32 members][16 layers][8 phases][4]: {t_start, t_gathered, t_published, failed poll passes} of GROUP 0, 10-ns ticks (tools/pd_diag.py)
};
struct PdCtx { int tid, lane, wave; long long t0; gu64* err; bool local; };
#define PD_NTW ((a.flags & 1) != 0)      // weight loads non-temporal (EMO_PD_NT=1) or default policy: the 8 groups read the same 76 MB within microseconds

// LDS carve (bytes from the dynamic base; device functions reach the error flag as an LDS address, not through a generic pointer kept in a struct)
constexpr int LDS_XIN = 0, LDS_XA = LDS_XIN + PD_GS * PD_XS * 2, LDS_X1 = LDS_XA + PD_GS * PD_XS * 2, LDS_FH = LDS_X1 + PD_GS * PD_XS * 2,
              LDS_PART = LDS_FH + PD_GS * PD_FS * 2, LDS_ATT = LDS_PART + PD_NW * 4 * 64 * 4,
              LDS_MISC = LDS_ATT + (3 * PD_DH + 2 * PD_F + 8 + PD_NW * PD_DH + 4 * 2 * 64 + 8) * 4, LDS_OM = LDS_MISC + 16, LDS_TOTAL = LDS_OM + PD_DH * PD_MF * 4;
extern __shared__ __attribute__((aligned(16))) char pd_smem[];
#define PD_SERR (*(int*)(pd_smem + LDS_MISC))

#define PD_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PD_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// a granule: write-through (sc1) unless the census found the whole group on one XCD — then a plain store, which stays in that XCD's L2
#define PD_PUBLISH(p, v)                                                                   \
    do {                                                                                   \
        if (c.local) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        else PD_STORE((p), (v));                                                           \
    } while (0)

// wave-uniform: true = give up (the launch is over its time budget, or another workgroup already reported a failure)
__device__ __forceinline__ bool pd_spin_fail(unsigned& spins, const PdCtx& c, unsigned code) {
    ++spins;
    if ((spins & 63) == 0) {
        bool bad = (long long)wall_clock64() - c.t0 > PD_TIMEOUT;
        if (!bad && (spins & 1023) == 0) bad = PD_LOAD(c.err) != 0;
        if (bad) {
            if (c.lane == 0) {
                if (PD_LOAD(c.err) == 0) PD_STORE(c.err, (u64)code);       // (first reporter wins, approximately: the code is a diagnostic)
                PD_SERR = 1;
            }
            return true;
        }
    }
    __builtin_amdgcn_s_sleep(1);
    return false;
}

// NP x 16 B = NP pairs of granules per thread, agent scope (sc1: past the L1), ONE asm statement with its own wait: hipcc must not touch a
// destination register before the data is there.  (vmcnt(0) also covers the phase-ahead loads issued before: in-order return anyway.)
template <int NP> __device__ __forceinline__ void pd_poll(u32x4 (&v)[NP], const gu64* p);
template <> __device__ __forceinline__ void pd_poll<1>(u32x4 (&v)[1], const gu64* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void pd_poll<4>(u32x4 (&v)[4], const gu64* p) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory");
}

// All-gather of GS rows of W bf16 values (W / 2 granules per row) into LDS rows of `stride` elements.  Thread t owns the NP consecutive pairs
// NP t .. of the buffer (a pair = 2 granules = 16 B = 4 values).  Returns the number of failed passes.
template <int W>
__device__ __forceinline__ unsigned pd_gather_rows(const gu64* buf, unsigned ep, bf16_t* dst, int stride, const PdCtx& c, unsigned code) {
    constexpr int NP = PD_GS * W / 4 / PD_NT;
    static_assert(NP == 1 || NP == 4, "pairs per thread");
    u32x4 g[NP];
    unsigned spins = 0;
    for (;;) {
        pd_poll<NP>(g, buf + 2 * NP * c.tid);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NP; ++i) ok = ok && g[i][1] == ep && g[i][3] == ep;
        if (__all(ok)) break;
        if (pd_spin_fail(spins, c, code)) return spins;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = NP * c.tid + i, s = p / (W / 4), c4 = p % (W / 4);
        *(u32x2*)(dst + s * stride + 4 * c4) = (u32x2){g[i][0], g[i][2]};
    }
    return spins;
}

// The wave's share of a member's packed weights: T column tiles x KPW k-steps, one KB (64 lanes x 8 bf16) per fragment, straight into registers.
template <int T, int KPW>
__device__ __forceinline__ void pd_load_w(bf16x8 (&w)[T][KPW], const bf16_t* member_base, const PdCtx& c, bool nt) {
    const bf16_t* p = member_base + (size_t)c.wave * (T * KPW * 512) + c.lane * 8;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int ks = 0; ks < KPW; ++ks) {
            const bf16x8* q = (const bf16x8*)(p + (t * KPW + ks) * 512);
            w[t][ks] = nt ? __builtin_nontemporal_load(q) : *q;
        }
}

// part[wave][t][stream][col] = x[stream, k-slice of the wave] . W[col, k-slice]   (x rows = rows 0..3 of the MFMA A operand, the rest zero)
template <int T, int KPW>
__device__ __forceinline__ void pd_gemv(const bf16x8 (&w)[T][KPW], const bf16_t* xs, int stride, float* part, const PdCtx& c) {
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r = c.lane & 15, kg = c.lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KPW; ++ks) {
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16_t)0.f;
        if (r < PD_GS) a = *(const bf16x8*)(xs + r * stride + (c.wave * KPW + ks) * 32 + kg * 8);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w[t][ks], acc[t], 0, 0, 0);
    }
    if (c.lane < 16) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[((c.wave * T + t) * 4 + i) * 16 + c.lane] = acc[t][i];
    }
}
template <int T>
__device__ __forceinline__ float pd_part_sum(const float* part, int t, int s, int col) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < PD_NW; ++w) v += part[((w * T + t) * 4 + s) * 16 + col];
    return v;
}
// Two neighbouring columns (even lane + the next lane) -> one granule, stored by the even lane.  Executed by whole waves.
__device__ __forceinline__ void pd_publish_pair(gu64* buf, int granule, unsigned ep, float v, int col, const PdCtx& c) {
    const bf16_t b = (bf16_t)v;
    const unsigned mine = (unsigned)__builtin_bit_cast(unsigned short, b);
    const unsigned next = (unsigned)__shfl_down((int)mine, 1, 64);
    if ((col & 1) == 0) PD_PUBLISH(buf + granule, ((u64)ep << 32) | (u64)(mine | (next << 16)));
}
// Sum over the 64 lanes, result in every lane: 4 DPP steps inside the 16-lane rows + the two lane swaps of rows4_sum — no LDS crossbar
// (__shfl_xor = ds_bpermute: 12 dependent ~100-cycle round trips per LayerNorm row were 0.5 us of each LayerNorm phase).
__device__ __forceinline__ float pd_dpp_add(float v, const int ctrl_id) {
    const int x = __builtin_bit_cast(int, v);
    int y;
    switch (ctrl_id) {
        case 0: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;      // quad_perm [1,0,3,2]
        case 1: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;      // quad_perm [2,3,0,1]
        case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); break;     // row_half_mirror
        default: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); break;    // row_mirror
    }
    return v + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float pd_wave_sum(float v) {
    v = pd_dpp_add(v, 0);
    v = pd_dpp_add(v, 1);
    v = pd_dpp_add(v, 2);
    v = pd_dpp_add(v, 3);
    return rows4_sum(v);
}
__device__ __forceinline__ void pd_ln_load(float (&g)[8], float (&b)[8], const float* gamma, const float* beta, const PdCtx& c) {
    if (c.wave < PD_GS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { g[i] = gamma[c.lane * 8 + i]; b[i] = beta[c.lane * 8 + i]; }
    }
}
// LayerNorm of LDS row `wave` (waves 0..3), in place, the arithmetic of layernorm_fwd_bf16_d512_kernel
__device__ __forceinline__ void pd_ln_rows(bf16_t* xs, const float (&g)[8], const float (&b)[8], float eps, const PdCtx& c) {
    if (c.wave < PD_GS) {
        bf16_t* row = xs + c.wave * PD_XS + c.lane * 8;
        const bf16x8 a = *(const bf16x8*)row;
        float v[8], s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = (float)a[i]; s += v[i]; }
        const float mu = pd_wave_sum(s) * (1.f / 512.f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[i] - mu; q += d * d; }
        const float rs = rsqrtf(pd_wave_sum(q) * (1.f / 512.f) + eps);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)((v[i] - mu) * rs * g[i] + b[i]);
        *(bf16x8*)row = o;
    }
}

// A per-phase copy of the context whose thread index the optimiser cannot see through: without it every phase's per-thread addresses are
// hoisted out of the layer loop and kept in ~100 registers, and the phase-ahead weight registers get spilled (= waited for) right after the load.
__device__ __forceinline__ PdCtx pd_fresh(const PdCtx& c) {
    PdCtx r = c;
    asm volatile("" : "+v"(r.tid));
    r.lane = r.tid & 63;
    return r;
}

// Workgroup barrier for LDS traffic ONLY: __syncthreads() carries a workgroup-scope fence, which on gfx950 is s_waitcnt vmcnt(0) — it would
// wait for the phase-ahead HBM loads at every barrier (measured r04: 2-3 us per phase).  Cross-wave data here lives in LDS (lgkmcnt).
#define PD_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PD_SYNC_OR_LEAVE()          \
    do {                            \
        PD_BARRIER();               \
        if (PD_SERR) return;        \
    } while (0)

#define PD_DIAG(l, ph, k, val)                                                                          \
    do {                                                                                               \
        if (a.diag && g == 0 && c.tid == 0) a.diag[(((size_t)m * 16 + (l)) * 8 + (ph)) * 4 + (k)] = (u64)(val); \
    } while (0)
#define PD_NOW() ((u64)wall_clock64())

__global__ __launch_bounds__(PD_NT, 2) void pd_step_kernel(PdArgs a) {
    bf16_t* xin = (bf16_t*)(pd_smem + LDS_XIN);        // layer input (post-LN2 / embedding), kept for the out-projection's residual
    bf16_t* xa = (bf16_t*)(pd_smem + LDS_XA);          // attention output rows
    bf16_t* x1 = (bf16_t*)(pd_smem + LDS_X1);          // post-LN1 rows, kept for the FFN2 residual
    bf16_t* fh = (bf16_t*)(pd_smem + LDS_FH);          // FFN hidden rows
    float* part = (float*)(pd_smem + LDS_PART);        // [8 waves][<= 4 tiles][4 streams][16 columns]
    float* xq = (float*)(pd_smem + LDS_ATT);           // attention scratch: q | k | v rows of (head, stream) as fp32
    float* xk = xq + PD_DH;
    float* xv = xk + PD_DH;
    float* fq = xv + PD_DH;                            // phi(q), phi(k)
    float* fk = fq + PD_F;
    float* dpart = fk + PD_F;                          // [8]
    float* num = dpart + 8;                            // [8 waves][64]
    float* upart = num + PD_NW * PD_DH;                // [4 d-quarters][q | k][64 projections]
    float* npart = upart + 4 * 2 * 64;                 // [4][2] partial |x|^2
    int* s_misc = (int*)(pd_smem + LDS_MISC);          // [0] error flag, [1] launch counter, [2] census: group on one XCD
    float* oml = (float*)(pd_smem + LDS_OM);           // omega of the layer [64][64]

    PdCtx c;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
    c.t0 = (long long)wall_clock64();
    c.err = (gu64*)a.sync + (size_t)PD_NG * PD_GSTRIDE;
    c.local = false;
    const int g = blockIdx.x % PD_NG, m = blockIdx.x / PD_NG;
    if (g * PD_GS >= a.n_streams) return;
    gu64* gs = (gu64*)a.sync + (size_t)g * PD_GSTRIDE;
    if (c.tid == 0) { s_misc[0] = 0; s_misc[1] = (int)(unsigned)PD_LOAD(gs + OFF_CNT); s_misc[2] = 0; }
    PD_BARRIER();
    const unsigned lc = (unsigned)s_misc[1], ep0 = lc * 128u;
    const int hm = m >> 2, jm = m & 3;                                // P1: head / 16-dim slice; P2: head / stream
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15;
    if (c.tid == 0) PD_STORE(gs + OFF_CEN + m, ((u64)(ep0 + 127u) << 32) | (u64)(unsigned)xcc);       // census entry: always write-through
    PD_DIAG(15, 0, 0, c.t0);
    PD_DIAG(15, 0, 1, xcc);
    PD_DIAG(15, 1, 0, (u64)clock64());                               // shader-clock cycles: with the 100-MHz stamps = the effective clock

    // ---------------------------------------------------------------- phase-ahead loads of layer 0's first two phases
    const PdLayer* LY = a.layers;
    bf16x8 wq[3][2], wo[1][2], w1[4][2], w2[1][8];
    f32x4 st[4], om0, om1;
    float zold = 0.f, bq = 0.f, bo = 0.f, b1 = 0.f, b2 = 0.f, lg[8], lb[8];
    const int64_t sh = ((int64_t)g * PD_GS + jm) * PD_H + hm;         // P2: (stream, head) of this member
    pd_load_w<3, 2>(wq, LY[0].wqkv + (size_t)m * (PD_NW * 3 * 2 * 512), c, PD_NTW);
    if (c.tid < 192) bq = LY[0].bqkv[(c.tid >> 6) * PD_D + hm * PD_DH + jm * 16 + (c.tid & 15)];
#define PD_LOAD_STATE(Lp, cx)     /* P2 state mapping: 16 threads per state row, 32 rows per pass, 4 passes */ \
    do {                                                                                                   \
        const int d4 = ((cx).tid & 15) * 4, fg = (cx).tid >> 4;                                            \
        const float* Sb_ = (Lp).S + sh * (PD_F * PD_DH);                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) st[i] = *(const f32x4*)(Sb_ + (fg + 32 * i) * PD_DH + d4); \
        om0 = *(const f32x4*)((Lp).omega + (cx).tid * 8);                                                  \
        om1 = *(const f32x4*)((Lp).omega + (cx).tid * 8 + 4);                                              \
        if ((cx).tid < PD_F) zold = (Lp).z[sh * PD_F + (cx).tid];                                          \
    } while (0)
    PD_LOAD_STATE(LY[0], c);
    pd_load_w<1, 2>(wo, LY[0].wo + (size_t)m * (PD_NW * 1 * 2 * 512), c, PD_NTW);
    if (c.tid < 64) bo = LY[0].bo[m * 16 + (c.tid & 15)];

    // ---------------------------------------------------------------- embedding (every member builds its group's 4 rows itself)
    {
        const int s = c.tid >> 7, c4 = (c.tid & 127) * 4;
        const int64_t stream = (int64_t)g * PD_GS + s;
        const int64_t tk = a.tok[stream], sg = a.seg ? a.seg[stream] : 0, pos = a.pos0 + (a.pos_ids ? a.pos_ids[stream] : 0);
        const f32x4 e = *(const f32x4*)(a.E + tk * PD_D + c4);
        f32x4 sv = {0.f, 0.f, 0.f, 0.f};
        if (a.seg) sv = *(const f32x4*)(a.Sg + sg * PD_D + c4);
        const f32x4 p = *(const f32x4*)(a.pe + pos * PD_D + c4);
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = e[i] * a.emb_scale;                             // reference order: emb.mul_(scale); emb += seg.mul_(scale); + pe
            v += sv[i] * a.emb_scale;
            v += p[i];
            o[i] = (bf16_t)v;
        }
        *(bf16x4*)(xin + s * PD_XS + c4) = o;
    }
    // ---------------------------------------------------------------- census: is the whole group on one XCD?
    if (c.wave == 0) {
        u64 v = 0;
        unsigned spins = 0;
        bool got = true;
        for (;;) {
            bool ok = true;
            if (c.lane < PD_GM) { v = PD_LOAD(gs + OFF_CEN + c.lane); ok = (unsigned)(v >> 32) == ep0 + 127u; }
            if (__all(ok)) break;
            if (pd_spin_fail(spins, c, 0x700u)) { got = false; break; }
        }
        const bool same = c.lane >= PD_GM || (unsigned)v == (unsigned)xcc;
        if (c.lane == 0) s_misc[2] = (got && __all(same)) ? 1 : 0;
    }
    PD_SYNC_OR_LEAVE();
    c.local = s_misc[2] != 0;
    PD_DIAG(15, 0, 3, c.local ? 1 : 0);

    for (int l = 0; l < a.n_layers; ++l) {
        const PdLayer L = LY[l];
        const bool last = l + 1 == a.n_layers;
        const unsigned ep = ep0 + (unsigned)l * 8u;
        // ============================================================ P1: q / k / v columns of (head hm, dims 16 jm ..)
        {
            const PdCtx cc = pd_fresh(c);
            PD_DIAG(l, 1, 0, PD_NOW());
            if (l > 0) {
                const unsigned sp = pd_gather_rows<PD_D>(gs + OFF_E1, ep - 8u + 5u, xin, PD_XS, cc, 0x100u + l);
                PD_SYNC_OR_LEAVE();
                PD_DIAG(l, 1, 3, sp);
                pd_load_w<1, 2>(wo, L.wo + (size_t)m * (PD_NW * 1 * 2 * 512), cc, PD_NTW);      // two phases ahead: P3's weights and bias
                if (cc.tid < 64) bo = L.bo[m * 16 + (cc.tid & 15)];
                pd_ln_rows(xin, lg, lb, a.ln_eps, cc);
                PD_BARRIER();
            }
            PD_DIAG(l, 1, 1, PD_NOW());
            pd_gemv<3, 2>(wq, xin, PD_XS, part, cc);
            PD_BARRIER();
            if (cc.tid < 192) {
                const int t = cc.tid >> 6, s = (cc.tid >> 4) & 3, col = cc.tid & 15;
                const float v = pd_part_sum<3>(part, t, s, col) + bq;
                pd_publish_pair(gs + OFF_E2, ((hm * PD_GS + s) * 3 + t) * 32 + ((jm * 16 + col) >> 1), ep + 1u, v, col, cc);
            }
            PD_DIAG(l, 1, 2, PD_NOW());
        }
        // ============================================================ P2: FAVOR+ recurrent step of (head hm, stream jm)
        {
            const PdCtx cc = pd_fresh(c);
            PD_DIAG(l, 2, 0, PD_NOW());
            float* Sb = L.S + sh * (PD_F * PD_DH);
            const int d4 = (cc.tid & 15) * 4, fg = cc.tid >> 4;
            unsigned sp = 0;
            if (cc.wave == 0) {                                            // 96 granules = 48 pairs: q_h | k_h | v_h of the stream
                const gu64* buf = gs + OFF_E2 + (hm * PD_GS + jm) * 96;
                u32x4 gq[1];
                gq[0] = (u32x4){0u, ep + 1u, 0u, ep + 1u};
                for (;;) {
                    if (cc.lane < 48) pd_poll<1>(gq, buf + 2 * cc.lane);
                    if (__all(gq[0][1] == ep + 1u && gq[0][3] == ep + 1u)) break;
                    if (pd_spin_fail(sp, cc, 0x200u + l)) break;
                }
                if (cc.lane < 48) {
                    float* dst = xq + (cc.lane >> 4) * PD_DH + (cc.lane & 15) * 4;      // xq, xk, xv are contiguous
                    dst[0] = __builtin_bit_cast(float, gq[0][0] << 16);
                    dst[1] = __builtin_bit_cast(float, gq[0][0] & 0xffff0000u);
                    dst[2] = __builtin_bit_cast(float, gq[0][2] << 16);
                    dst[3] = __builtin_bit_cast(float, gq[0][2] & 0xffff0000u);
                }
            }
            *(f32x4*)(oml + cc.tid * 8) = om0;                             // omega [64 d][64 m] of the layer -> LDS
            *(f32x4*)(oml + cc.tid * 8 + 4) = om1;
            PD_SYNC_OR_LEAVE();
            PD_DIAG(l, 2, 3, sp);
            PD_DIAG(l, 2, 1, PD_NOW());
            pd_load_w<4, 2>(w1, L.w1 + (size_t)m * (PD_NW * 4 * 2 * 512), cc, PD_NTW);      // two phases ahead: P4
            if (cc.tid < 256) b1 = L.b1[m * 64 + (cc.tid >> 6) * 16 + (cc.tid & 15)];
            pd_ln_load(lg, lb, L.g1, L.be1, cc);
            // projections: thread = (d-quarter, q | k, projection): 16 of the 64 terms each
            {
                const int col = cc.tid & 63, which = (cc.tid >> 6) & 1, qd = cc.tid >> 7;
                const float* xx = which ? xk : xq;
                float u = 0.f, nn = 0.f;
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const float xv_ = xx[qd * 16 + d];
                    u += xv_ * oml[(qd * 16 + d) * PD_MF + col];
                    nn += xv_ * xv_;
                }
                upart[(qd * 2 + which) * 64 + col] = u;
                if (col == 0) npart[qd * 2 + which] = nn;
            }
            PD_BARRIER();
            const float cs = rsqrtf(sqrtf((float)PD_DH)), half_ln_f = 0.5f * logf((float)PD_F);
            float dn = 0.f;
            if (cc.tid < PD_F) {
                const int col = cc.tid & (PD_MF - 1);
                const float sgn = cc.tid < PD_MF ? 1.f : -1.f;
                const float uq = (upart[col] + upart[128 + col]) + (upart[256 + col] + upart[384 + col]);
                const float uk = (upart[64 + col] + upart[192 + col]) + (upart[320 + col] + upart[448 + col]);
                const float nq = (npart[0] + npart[2]) + (npart[4] + npart[6]), nk = (npart[1] + npart[3]) + (npart[5] + npart[7]);
                const float pq = __expf(sgn * cs * uq - (0.5f * cs * cs * nq + half_ln_f));
                const float pk = __expf(sgn * cs * uk - (0.5f * cs * cs * nk + half_ln_f));
                fq[cc.tid] = pq;
                fk[cc.tid] = pk;
                const float z = zold + pk;
                L.z[sh * PD_F + cc.tid] = z;
                dn = pq * z;
            }
            dn = pd_wave_sum(dn);
            if (cc.lane == 0) dpart[cc.wave] = dn;
            PD_BARRIER();
            const f32x4 vd = {xv[d4], xv[d4 + 1], xv[d4 + 2], xv[d4 + 3]};
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = fg + 32 * i;
                const f32x4 sv = st[i] + fk[f] * vd;
                *(f32x4*)(Sb + f * PD_DH + d4) = sv;
                acc += fq[f] * sv;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = rows4_sum(acc[i]);       // the wave's 4 rows per pass sit in lanes l, l^16, l^32, l^48
            if (cc.lane < 16) *(f32x4*)(num + cc.wave * PD_DH + d4) = acc;
            PD_BARRIER();
            if (cc.tid < PD_DH) {
                float o = 0.f;
#pragma unroll
                for (int w = 0; w < PD_NW; ++w) o += num[w * PD_DH + cc.tid];
                o = o / (dpart[0] + dpart[1] + a.eps);                    // (waves 2..7 hold no features)
                pd_publish_pair(gs + OFF_E3, jm * (PD_D / 2) + ((hm * PD_DH + cc.tid) >> 1), ep + 2u, o, cc.tid, cc);
            }
            PD_DIAG(l, 2, 2, PD_NOW());
        }
        // ============================================================ P3: out-projection columns 16 m .. (+ bias + residual)
        {
            const PdCtx cc = pd_fresh(c);
            PD_DIAG(l, 3, 0, PD_NOW());
            const unsigned sp = pd_gather_rows<PD_D>(gs + OFF_E3, ep + 2u, xa, PD_XS, cc, 0x300u + l);
            PD_SYNC_OR_LEAVE();
            PD_DIAG(l, 3, 3, sp);
            PD_DIAG(l, 3, 1, PD_NOW());
            pd_load_w<1, 8>(w2, L.w2 + (size_t)m * (PD_NW * 1 * 8 * 512), cc, PD_NTW);      // two phases ahead: P5
            if (cc.tid < 64) b2 = L.b2[m * 16 + (cc.tid & 15)];
            pd_gemv<1, 2>(wo, xa, PD_XS, part, cc);
            PD_BARRIER();
            if (cc.tid < 64) {
                const int s = cc.tid >> 4, col = cc.tid & 15, gc = m * 16 + col;
                const float v = pd_part_sum<1>(part, 0, s, col) + bo + (float)xin[s * PD_XS + gc];
                pd_publish_pair(gs + OFF_E4, s * (PD_D / 2) + (gc >> 1), ep + 3u, v, col, cc);
            }
            PD_DIAG(l, 3, 2, PD_NOW());
        }
        // ============================================================ P4: LayerNorm1 + FFN1 columns 64 m .. + ReLU
        {
            const PdCtx cc = pd_fresh(c);
            PD_DIAG(l, 4, 0, PD_NOW());
            const unsigned sp = pd_gather_rows<PD_D>(gs + OFF_E4, ep + 3u, x1, PD_XS, cc, 0x400u + l);
            PD_SYNC_OR_LEAVE();
            PD_DIAG(l, 4, 3, sp);
            PD_DIAG(l, 4, 1, PD_NOW());
            pd_ln_rows(x1, lg, lb, a.ln_eps, cc);
            // two phases ahead: the next layer's P1 (its q / k / v weights, LayerNorm2 of THIS layer), or the logits tile after the last layer
            pd_ln_load(lg, lb, L.g2, L.be2, cc);
            if (!last) {
                pd_load_w<3, 2>(wq, LY[l + 1].wqkv + (size_t)m * (PD_NW * 3 * 2 * 512), cc, PD_NTW);
                if (cc.tid < 192) bq = LY[l + 1].bqkv[(cc.tid >> 6) * PD_D + hm * PD_DH + jm * 16 + (cc.tid & 15)];
            } else if (m < (a.n_token + 15) / 16) {
                pd_load_w<1, 2>(wo, a.wout + (size_t)m * (PD_NW * 1 * 2 * 512), cc, PD_NTW);
                if (cc.tid < 64) bo = (m * 16 + (cc.tid & 15)) < a.n_token ? a.bout[m * 16 + (cc.tid & 15)] : 0.f;
            }
            PD_BARRIER();
            pd_gemv<4, 2>(w1, x1, PD_XS, part, cc);
            PD_BARRIER();
            if (cc.tid < 256) {
                const int t = cc.tid >> 6, s = (cc.tid >> 4) & 3, col = cc.tid & 15, gc = m * 64 + t * 16 + col;
                const float v = fmaxf(pd_part_sum<4>(part, t, s, col) + b1, 0.f);
                pd_publish_pair(gs + OFF_E5, s * (PD_FF / 2) + (gc >> 1), ep + 4u, v, col, cc);
            }
            PD_DIAG(l, 4, 2, PD_NOW());
        }
        // ============================================================ P5: FFN2 columns 16 m .. (+ bias + residual)
        {
            const PdCtx cc = pd_fresh(c);
            PD_DIAG(l, 5, 0, PD_NOW());
            const unsigned sp = pd_gather_rows<PD_FF>(gs + OFF_E5, ep + 4u, fh, PD_FS, cc, 0x500u + l);
            PD_SYNC_OR_LEAVE();
            PD_DIAG(l, 5, 3, sp);
            PD_DIAG(l, 5, 1, PD_NOW());
            if (!last) PD_LOAD_STATE(LY[l + 1], cc);                   // two phases ahead: the next layer's P2 (state slice, omega, z)
            pd_gemv<1, 8>(w2, fh, PD_FS, part, cc);
            PD_BARRIER();
            if (cc.tid < 64) {
                const int s = cc.tid >> 4, col = cc.tid & 15, gc = m * 16 + col;
                const float v = pd_part_sum<1>(part, 0, s, col) + b2 + (float)x1[s * PD_XS + gc];
                pd_publish_pair(gs + OFF_E1, s * (PD_D / 2) + (gc >> 1), ep + 5u, v, col, cc);
            }
            PD_DIAG(l, 5, 2, PD_NOW());
        }
    }
    // ---------------------------------------------------------------- LayerNorm2 of the last layer + logits tile m (21 tiles of 16 columns)
    if (m < (a.n_token + 15) / 16) {                                      // (m is uniform over the workgroup; the other members are done)
        const PdCtx cc = pd_fresh(c);
        pd_gather_rows<PD_D>(gs + OFF_E1, ep0 + (unsigned)(a.n_layers - 1) * 8u + 5u, xin, PD_XS, cc, 0x600u);
        PD_SYNC_OR_LEAVE();
        pd_ln_rows(xin, lg, lb, a.ln_eps, cc);
        PD_BARRIER();
        pd_gemv<1, 2>(wo, xin, PD_XS, part, cc);
        PD_BARRIER();
        if (cc.tid < 64) {
            const int s = cc.tid >> 4, col = cc.tid & 15, gc = m * 16 + col;
            if (gc < a.n_token) a.logits[((int64_t)g * PD_GS + s) * a.n_token + gc] = pd_part_sum<1>(part, 0, s, col) + bo;
        }
        // member 0 gathered the last edge from EVERY member of the group, so all of them have long read the counter
        if (m == 0 && cc.tid == 0) PD_STORE(gs + OFF_CNT, (u64)(lc + 1u));
        PD_DIAG(15, 0, 2, PD_NOW());
        PD_DIAG(15, 1, 1, (u64)clock64());
    }
}
}  // namespace
extern "C" int64_t emo_performer_decode_step_workspace_bytes(void) { return (int64_t)PD_WS_WORDS * 8; }

extern "C" int emo_performer_decode_step(const void* layer_table, int64_t n_layers, const int64_t* tok, const int64_t* seg, const float* E, const float* Sg,
                                         const float* pe, float emb_scale, int64_t pos0, const int64_t* pos_ids, const void* wout_packed,
                                         const float* bout, int64_t n_token, float* logits, int64_t n_streams, int64_t d_model, int64_t n_head,
                                         int64_t n_feat, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes, float eps, float ln_eps,
                                         int64_t* diag, emo_stream_t stream) {
    EMO_CHECK(layer_table && tok && E && pe && wout_packed && bout && logits && sync_ws, "emo_performer_decode_step: null pointer");
    EMO_CHECK(d_model == PD_D && n_head == PD_H && n_feat == PD_F && d_ff == PD_FF,
              "emo_performer_decode_step: built for d_model 512 / 8 heads / 128 features / d_ff 2048 (got %lld / %lld / %lld / %lld)", (long long)d_model,
              (long long)n_head, (long long)n_feat, (long long)d_ff);
    EMO_CHECK(n_layers >= 1 && n_layers <= PD_MAX_LAYERS, "emo_performer_decode_step: 1 <= n_layers <= %d", PD_MAX_LAYERS);
    EMO_CHECK(n_streams >= PD_GS && n_streams <= PD_GS * PD_NG && n_streams % PD_GS == 0, "emo_performer_decode_step: n_streams must be a multiple of 4, <= 32");
    EMO_CHECK(n_token >= 1 && n_token <= 16 * PD_GM, "emo_performer_decode_step: n_token <= 512");
    EMO_CHECK(!(seg && !Sg), "emo_performer_decode_step: seg ids without a segment table");
    EMO_CHECK(sync_ws_bytes >= (int64_t)PD_WS_WORDS * 8 && ((uintptr_t)sync_ws & 15) == 0, "emo_performer_decode_step: workspace too small / unaligned");
    PdArgs a;
    a.layers = (const PdLayer*)layer_table; a.n_layers = (int)n_layers;
    a.tok = tok; a.seg = seg; a.E = E; a.Sg = Sg; a.pe = pe; a.emb_scale = emb_scale; a.pos0 = pos0; a.pos_ids = pos_ids;
    a.wout = (const bf16_t*)wout_packed; a.bout = bout; a.n_token = (int)n_token; a.logits = logits; a.n_streams = (int)n_streams;
    a.sync = (u64*)sync_ws; a.eps = eps; a.ln_eps = ln_eps; a.diag = (u64*)diag;
    { const char* e = getenv("EMO_PD_NT"); a.flags = (e && atoi(e) == 1) ? 1 : 0; }
    static_assert(LDS_TOTAL <= 96 * 1024, "LDS carve");
    const size_t lds = 96 * 1024;                                         // > half of the CU's LDS: one workgroup per CU
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)pd_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(pd_step_kernel, dim3(PD_NG * PD_GM), dim3(PD_NT), lds, (hipStream_t)stream, a);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}
