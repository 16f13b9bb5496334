// One-launch Performer decode step (BASELINE configs[3]: 32 streams x 2048 tokens): embedding -> 12 post-LN FAVOR+ layers -> logits for every
// stream in ONE persistent kernel.  Replaces the per-token chain of ~62 dependent launches of inference.py (reference loop:
// stage2_accompaniment/inference.py:250-277 -> MusicPerformer.forward with keep_last_only, music_performer.py:50-70), whose every launch sat
// on the ~5-6 us launch floor with < 2 MB of work (r03: 0.383 ms per token step = 0.09 of the HBM roofline).
//
// Structure: 256 workgroups = 8 GROUPS x 32 members; group g = blockIdx % 8 — the XCD the dispatcher puts the block on, so a group's traffic
// normally stays inside one XCD, but NOTHING depends on that placement — owns streams 4g .. 4g+3 for the whole token step and never talks to
// another group.  Inside a group every GEMM of the layer is split by OUTPUT COLUMN over the 32 members (each member streams 1/32 of every
// weight matrix: 197 KB per layer, pre-packed on the host in MFMA fragment order so that a wave's load instruction is one contiguous KB that
// goes straight into the B-operand registers — no LDS staging, each weight byte is used once), the 4 streams are rows 0-3 of a 16-row MFMA A
// operand, and the five dependent products of a layer are separated by five all-gather EDGES:
//   P1 q/k/v columns of head h (member = (h, j): dims 16j..16j+15 of q_h, k_h, v_h)        -> E2 (gathered by the 4 members of head h)
//   P2 FAVOR+ recurrent step of (head h, stream j): S += phi(k) (x) v, out = phi(q)^T S / ..  -> E3
//   P3 out-projection + bias + residual (pre-LN row)                                        -> E4
//   P4 LayerNorm1 (every member normalises the gathered rows itself) + FFN1 + ReLU         -> E5
//   P5 FFN2 + bias + residual (pre-LN row)                                                  -> E1 (next layer's P1 applies LayerNorm2)
// An edge is a buffer of 8-byte GRANULES {tag = epoch, value = 2 bf16} (E5, the one large edge, since r06: {3 bf16 | 16-bit epoch}, see
// pd_publish_triple) written by single agent-scope (sc1) stores and polled with agent-scope
// loads: the data is the flag, no fences, correct for any workgroup -> XCD placement (MI355X guide, Guideline 16 form R2).  Epochs count
// launches (a per-group counter the group's member 0 bumps when it is done) x phases, so nothing is zeroed per launch and hipGraph replay
// works.  A buffer is rewritten one layer later; between two uses lies at least one all-to-all edge, so every reader of the old contents has
// finished.  Every poll is bounded: a group that cannot make progress (a member not resident) writes an error code and leaves.
//
// Same-XCD fast path (measured r04, tools/pd_diag.py: a write-through granule costs the reader a fabric round trip of ~1.5 us per poll, two per
// edge): every launch starts with a CENSUS — each member publishes its XCC id through the placement-independent form, gathers the 32 ids of
// its group, and only if all are equal the group's producers switch to PLAIN stores, which stay in that XCD's L2 where the agent-scope (L1
// bypassing) polls of the other members hit (~0.5 us per edge).  A group that is spread over several XCDs keeps the write-through stores:
// speed depends on the placement, the result never does.
//
// Wave roles (12 waves per workgroup).  A wave's loads return IN ORDER and hipcc waits for its own loads with s_waitcnt vmcnt(0), so a poll — or
// the first use of anything — waits for every older AND every newer outstanding load of that wave: in the first versions (all waves did
// everything) each gather waited ~2 us for the HBM weight loads issued just before it, however far ahead they were requested (r04
// diagnostics: 0.26 -> 0.24 ms per token with 1- and 2-phase-ahead loads).  Therefore:
//   * waves 8-11 = POLLERS: gather the edges into LDS (poller wave s owns row s, so it also applies the LayerNorm without a barrier), sum the
//     partial products, add the residual, publish.  They issue NO other global loads: a poll never queues behind HBM.
//   * waves 0-3 = compute half A: the P1 and P4 products;  waves 4-7 = half B: the attention step P2 and the P3 and P5 products.  A half
//     requests its NEXT operand set (weight fragments + bias; B: state slice; A: omega and the LayerNorm parameters, which it hands to B / the
//     pollers through LDS) after it has published the current product, in SLICES of 8-21 KB — one after each barrier it passes while idle — and
//     has nothing to wait for until its next turn, two to three phases later.  (Requested as one 48-64 KB burst the set sat in the CU's memory
//     queue in front of the pollers' loads: first poll 2.0 us instead of 0.2, and the burst's issue stalled its own wave for ~2 us.)
// Measured r04 (tools/pd_diag.py, profiles/r04_pd_diag_v*.txt; bench `gen`): chain of launches 0.383 ms per token step; all-waves-do-everything
// persistent kernel 0.325 (write-through granules) -> 0.257 (census + L2-local granules) -> 0.244 (raw barriers, loads two phases ahead);
// (timing ablation EMO_PD_NT=2 — every weight fragment read from one hot KB, results wrong — : kernel 181 -> 155 us, so the weight stream is NOT
// the bulk of it: 60 phases x 2.6 us remain as publish -> visible -> polled latency (1.3-1.8 failed poll passes of 0.3-0.5 us per edge, the 32-KB E5
// gather 1.6 us on its own), the HBM round trip of the state slice in P2 and 0.6-1.5 us of work; a design that keeps the weights resident would
// gain ~15 %, not 2 x)
// wave roles with burst loads 0.251-0.283; a slice schedule that packs the loads into the poll-free windows (1a, 2a-2c, 4a) 0.233 (the 32-KB E5
// gather stays at 2.4-2.8 us whatever precedes it: it is the CU's own L2 -> CU rate); wave roles with evenly sliced loads 0.221 (kernel 181 us = 60 phases x 3.0 us: 0.5-0.9 us of work each, the
// rest is the wait for the edge behind the CU's weight stream: every XCD streams ALL weights for its 4 streams, 0.8 GB per token step).
// Workgroup barriers are raw s_barrier + lgkmcnt(0): __syncthreads() carries a vmcnt(0) fence and would drain the HBM loads at every barrier.
//
// r06 (tools/pd_diag.py again, profiles/r06_pd_diag.txt, profiles/r06_gpt2_persistent.txt): a poll pass costs time in proportion to its 16-byte sc1
// loads per thread and queues behind any weight slice requested just before it -> dense E5 granules, the E5 gather shared between the pollers and
// half A (first granule pair polled alone), no slice between a publish and the barrier behind the next gather: kernel 197.6 -> 164 us per token step;
// with the weight stream ablated 130 us.  The same kernel as a template also runs the GPT-2 block (KV-cache attention in P2): see the end of the file.
//
// Arithmetic mirrors the launch path's bf16 mode (emo_gemm skinny kernel, favor_decode_fast_kernel, layernorm_fwd_bf16_d512_kernel): bf16
// activations between products, fp32 accumulation, fp32 FAVOR+ state, LayerNorm statistics in fp32 from the bf16 row.
#include "emo_common.h"
#include "emo_nucleus.h"

namespace {
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned long long gu64;      // every word another workgroup reads: GLOBAL address space, never flat
constexpr int PD_D = 512, PD_H = 8, PD_DH = 64, PD_MF = 64, PD_F = 128, PD_FF = 2048;
constexpr int PD_GS = 4, PD_GM = 32, PD_NG = 8, PD_NT = 768, PD_HW = 4, PD_HT = 256;    // threads; waves / threads per role (A, B, pollers)
constexpr int PD_XS = PD_D + 8, PD_FS = PD_FF + 8;              // LDS row strides (bf16 elements): +16 B shifts the rows' banks
constexpr int OFF_CNT = 0, OFF_E1 = 8, OFF_E2 = OFF_E1 + PD_GS * PD_D / 2, OFF_E3 = OFF_E2 + PD_H * PD_GS * 96, OFF_E4 = OFF_E3 + PD_GS * PD_D / 2,
              OFF_E5 = OFF_E4 + PD_GS * PD_D / 2, OFF_CEN = OFF_E5 + PD_GS * PD_FF / 2, OFF_TOK = OFF_CEN + PD_GM,
              PD_GSTRIDE = OFF_TOK + 8;
constexpr int PD_WS_WORDS = PD_NG * PD_GSTRIDE + 8;             // last 8 words: [0] = error code
constexpr int PD_MAX_LAYERS = 15;                               // epoch = launch * 128 + layer * 8 + phase (census: + 127)
constexpr long long PD_TIMEOUT = 5000000;                       // wall_clock64 ticks (100 MHz): 50 ms for the whole launch

// Everything the table points to is addressed as GLOBAL memory (address space 1), never through flat pointers: a flat load counts on lgkmcnt as well
// as vmcnt, so every LDS barrier (lgkmcnt(0)) of the wave would wait for its outstanding weight / state / cache loads (r06: found in the ISA — the
// table's pointers are loaded from memory and hipcc does not infer their address space).
typedef __attribute__((address_space(1))) bf16_t gbf16_t;
typedef __attribute__((address_space(1))) bf16x8 gbf16x8;
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) f32x4 gf32x4;
struct PdLayer {            // one row of the caller's device table: 16 pointers
    const gbf16_t* wqkv; const gfloat* bqkv; const gbf16_t* wo; const gfloat* bo; const gfloat* g1; const gfloat* be1; const gbf16_t* w1; const gfloat* b1;
    const gbf16_t* w2; const gfloat* b2; const gfloat* g2; const gfloat* be2; const gfloat* omega; gfloat* S; gfloat* z; void* pad;
};
struct PdArgs {
    const PdLayer* layers; int n_layers;
    const int64_t* tok; const int64_t* seg; const float* E; const float* Sg; const float* pe; float emb_scale; int64_t pos0; const int64_t* pos_ids;
    const bf16_t* wout; const float* bout; int n_token; float* logits; int n_streams; u64* sync; float eps, ln_eps;
    // in-kernel nucleus draw (samp_mode 1): member s < 4 of a group draws the next token of stream 4 g + s from logits_in (the previous step's
    // logits) exactly as emo_sample_nucleus_step does, writes it to tok_out / seq and hands it to the group through 4 granules; tok is ignored
    int samp_mode; float temp, top_p; const float* u_steps; int64_t* step; int64_t* seq; int64_t ld_seq, col0; int64_t* tok_out; const float* logits_in; int n_real;
    int flags;      // bit 0: non-temporal weight loads
    // GPT-2 form (pd_step_kernel<true>): [gamma | beta] of layer 0's ln_1, rows per (stream, head) of the head-major KV caches the table's S / z slots point to
    const float* ln0; int64_t kv_tmax;
    u64* diag;      // optional [32 members][16 layers][8 phases][4]: {t_start, t_gathered, t_published, failed poll passes} of GROUP 0, 10-ns ticks (tools/pd_diag.py)
};
// t = thread index INSIDE the role (0..255), hw = wave inside the role (0..3)
struct PdCtx { int t, lane, hw; long long t0; gu64* err; bool local; };
#define PD_NTW ((a.flags & 1) != 0)      // weight loads non-temporal (EMO_PD_NT=1) or default policy (r04: no measurable difference)

// LDS carve (bytes from the dynamic base; device functions reach the error flag as an LDS address, not through a generic pointer kept in a struct)
constexpr int LDS_XIN = 0, LDS_XA = LDS_XIN + PD_GS * PD_XS * 2, LDS_X1 = LDS_XA + PD_GS * PD_XS * 2, LDS_FH = LDS_X1 + PD_GS * PD_XS * 2,
              LDS_PART = LDS_FH + PD_GS * PD_FS * 2, LDS_ATT = LDS_PART + PD_HW * 4 * 64 * 4,
              LDS_LN = LDS_ATT + (3 * PD_DH + 2 * PD_F + 8 + PD_HW * PD_DH + 2 * 2 * 64 + 8) * 4, LDS_MISC = LDS_LN + 2 * 2 * PD_D * 4, LDS_OM = LDS_MISC + 64,
              LDS_SAMP = LDS_OM + PD_DH * PD_MF * 4, LDS_TOTAL = LDS_SAMP + EMO_NUCLEUS_LDS,
              LDS_XRAW = LDS_SAMP, LDS_HRAW = LDS_XRAW + PD_GS * PD_XS * 2;      // GPT-2 form: pre-LN rows (the residuals); the draw's scratch is free after barrier Bt
static_assert(LDS_HRAW + PD_GS * PD_XS * 2 <= LDS_TOTAL, "LDS carve");
extern __shared__ __attribute__((aligned(16))) char pd_smem[];
#define PD_SERR (*(int*)(pd_smem + LDS_MISC))

#define PD_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PD_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// a granule: write-through (sc1) unless the census found the whole group on one XCD — then a workgroup-scope (sc0) store, which stays in that XCD's L2
#define PD_PUBLISH(p, v)                                                                   \
    do {                                                                                   \
        if (c.local) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        else PD_STORE((p), (v));                                                           \
    } while (0)
// Workgroup barrier for LDS traffic ONLY (see the header): all 12 waves execute the same sequence of these.
#define PD_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PD_SYNC_OR_LEAVE()          \
    do {                            \
        PD_BARRIER();               \
        if (PD_SERR) return;        \
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
// destination register before the data is there.
template <int NP> __device__ __forceinline__ void pd_poll(u32x4 (&v)[NP], const gu64* p);
template <> __device__ __forceinline__ void pd_poll<1>(u32x4 (&v)[1], const gu64* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void pd_poll<2>(u32x4 (&v)[2], const gu64* p) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void pd_poll<3>(u32x4 (&v)[3], const gu64* p) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:32 sc1\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void pd_poll<6>(u32x4 (&v)[6], const gu64* p) {
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %6, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %6, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %6, off offset:80 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]) : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void pd_poll<8>(u32x4 (&v)[8], const gu64* p) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:96 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:112 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p) : "memory");
}

// POLLERS: all-gather of GS rows of W bf16 values (W / 2 granules per row) into LDS rows of `stride` elements.  Poller thread t owns the NP
// consecutive pairs NP t .. of the buffer (a pair = 2 granules = 16 B = 4 values); for W = 512 poller wave s owns exactly row s.
template <int W>
__device__ __forceinline__ unsigned pd_gather_rows(const gu64* buf, unsigned ep, bf16_t* dst, int stride, const PdCtx& c, unsigned code, u64* first_poll = nullptr) {
    constexpr int NP = PD_GS * W / 4 / PD_HT;
    static_assert(NP == 2 || NP == 8, "pairs per thread");
    u32x4 g[NP];
    unsigned spins = 0;
    for (;;) {
        const u64 tp0 = first_poll && spins == 0 ? (u64)wall_clock64() : 0;
        pd_poll<NP>(g, buf + 2 * NP * c.t);
        if (first_poll && spins == 0) *first_poll = (u64)wall_clock64() - tp0;
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NP; ++i) ok = ok && g[i][1] == ep && g[i][3] == ep;
        if (__all(ok)) break;
        if (pd_spin_fail(spins, c, code)) return spins;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = NP * c.t + i, s = p / (W / 4), c4 = p % (W / 4);
        *(u32x2*)(dst + s * stride + 4 * c4) = (u32x2){g[i][0], g[i][2]};
    }
    return spins;
}

// E5 (the FFN hidden rows: 4 streams x 2048 values, the one edge whose PAYLOAD matters — r06 diagnostics: first poll pass 3.2 us for 32 KB of
// {epoch, 2 bf16} granules, ~10 KB / us into a CU) uses DENSE granules {3 bf16 | 16-bit epoch}: a producer wave's 16 columns of a stream are 6
// granules (5 triples + a single), a member's 64 columns 24, the edge 4 x 32 x 24 = 3072 granules = 24 KB = exactly 6 pairs per poller thread,
// (pairs of 16-byte loads; see pd_gather_e5).  The 16-bit epoch wraps every 512 launches; a
// granule is rewritten in every layer of every launch, so a stale value can never carry the expected tag.
constexpr int PD_E5_GPT = 6;                                  // granules per (member, stream, column tile); 24 per (member, stream)
__device__ __forceinline__ void pd_publish_triple(gu64* buf, unsigned ep, float v, int col, const PdCtx& c) {
    const bf16_t b = (bf16_t)v;
    const unsigned mine = (unsigned)__builtin_bit_cast(unsigned short, b);
    const unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x101, 0xF, 0xF, true);      // row_shl:1 / :2 — column col + 1, col + 2 of the
    const unsigned n2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x102, 0xF, 0xF, true);      // same 16-lane row (0 past column 15)
    if (col % 3 == 0) PD_PUBLISH(buf + col / 3, ((u64)(((ep & 0xffffu) << 16) | n2) << 32) | (u64)(mine | (n1 << 16)));
}
// The gather is shared by the pollers AND half A (512 threads, u = 0 .. 511; half A has published its product and requests its next operand set only
// after this gather, so nothing of its own is in flight in front of the polls): thread u takes the 6 granules of (stream u / 128, member (u % 128) / 4,
// column tile u % 4) = 16 consecutive hidden values.  The time of a poll pass grows with the loads per thread (8 -> 6 -> 3: 3.2 -> 2.4 -> .. us).
__device__ __forceinline__ unsigned pd_gather_e5(const gu64* buf, unsigned ep, bf16_t* fh, int u, const PdCtx& c, unsigned code, u64* first_poll = nullptr) {
    // A thread's 6 granules were written by ONE store instruction of one producer wave: the first pair is polled alone (a pass costs time in
    // proportion to its loads) and the other two are fetched once it carries the epoch — and re-polled in the rare case they do not yet.
    u32x4 g[3];
    const unsigned tag = ep & 0xffffu;
    unsigned spins = 0;
    for (;;) {
        const u64 tp0 = first_poll && spins == 0 ? (u64)wall_clock64() : 0;
        u32x4 g0[1];
        pd_poll<1>(g0, buf + 6 * u);
        if (first_poll && spins == 0) *first_poll = (u64)wall_clock64() - tp0;
        g[0] = g0[0];
        if (__all((g[0][1] >> 16) == tag && (g[0][3] >> 16) == tag)) break;
        if (pd_spin_fail(spins, c, code)) return spins;
    }
    for (;;) {
        u32x4 g12[2];
        pd_poll<2>(g12, buf + 6 * u + 2);
        g[1] = g12[0];
        g[2] = g12[1];
        if (__all((g[1][1] >> 16) == tag && (g[1][3] >> 16) == tag && (g[2][1] >> 16) == tag && (g[2][3] >> 16) == tag)) break;
        if (pd_spin_fail(spins, c, code)) return spins;
    }
    unsigned short h[16];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const unsigned lo = g[k >> 1][(k & 1) * 2], hi = g[k >> 1][(k & 1) * 2 + 1];
        h[3 * k] = (unsigned short)(lo & 0xffffu);
        if (k < 5) { h[3 * k + 1] = (unsigned short)(lo >> 16); h[3 * k + 2] = (unsigned short)(hi & 0xffffu); }
    }
    bf16_t* dst = fh + (u >> 7) * PD_FS + ((u & 127) >> 2) * 64 + (u & 3) * 16;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (unsigned)h[8 * q + 2 * e] | ((unsigned)h[8 * q + 2 * e + 1] << 16);
        *(u32x4*)(dst + 8 * q) = o;
    }
    return spins;
}

// COMPUTE: the wave's share of a member's packed weights: T column tiles x KPW k-steps, one KB (64 lanes x 8 bf16) per fragment, into registers.
template <int T, int KPW>
__device__ __forceinline__ void pd_load_w(bf16x8 (&w)[T][KPW], const gbf16_t* member_base, const PdCtx& c, int nt) {
    const gbf16_t* p = member_base + (size_t)c.hw * (T * KPW * 512) + c.lane * 8;
    if (nt == 2) p = member_base + c.lane * 8;              // TIMING ABLATION ONLY (EMO_PD_NT=2): every fragment from one hot KB — no weight stream, wrong results
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int ks = 0; ks < KPW; ++ks) {
            const gbf16x8* q = (const gbf16x8*)(p + (nt == 2 ? 0 : (t * KPW + ks) * 512));
            w[t][ks] = nt == 1 ? __builtin_nontemporal_load(q) : *q;
        }
}
// fragments [I0, I1) of the same set (flat index t * KPW + ks): an operand set is requested in SLICES, one after each barrier the idle half passes,
// so that no 48-64 KB burst sits in the CU's memory queue in front of the pollers' loads (r04 diagnostics: first poll 2.0 us behind a burst, 0.2 us
// without one; the burst's issue itself stalled its wave for ~2 us)
template <int T, int KPW, int I0, int I1>
__device__ __forceinline__ void pd_load_w_part(bf16x8 (&w)[T][KPW], const gbf16_t* member_base, const PdCtx& c, int nt) {
    const gbf16_t* p = member_base + (size_t)c.hw * (T * KPW * 512) + c.lane * 8;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const gbf16x8* q = (const gbf16x8*)(p + (nt == 2 ? 0 : i * 512));
        w[i / KPW][i % KPW] = nt == 1 ? __builtin_nontemporal_load(q) : *q;
    }
}
// part[hw][t][stream][col] = x[stream, k-slice of the wave] . W[col, k-slice] (+ bias[t] from the role's wave 0): x rows = rows 0..3 of the
// MFMA A operand, the rest zero; lane l < 16 ends up with column l of the 4 streams
template <int T, int KPW>
__device__ __forceinline__ void pd_gemv(const bf16x8 (&w)[T][KPW], const float (&bias)[T], const bf16_t* xs, int stride, float* part, const PdCtx& c) {
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float b = c.hw == 0 ? bias[t] : 0.f;
        acc[t] = (f32x4){b, b, b, b};
    }
    const int r = c.lane & 15, kg = c.lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KPW; ++ks) {
        if (ks > 0 && (ks & 3) == 0) __builtin_amdgcn_sched_barrier(0);       // at most 4 A fragments in flight (16 hoisted reads = 64 registers)
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16_t)0.f;
        if (r < PD_GS) a = *(const bf16x8*)(xs + r * stride + (c.hw * KPW + ks) * 32 + kg * 8);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w[t][ks], acc[t], 0, 0, 0);
    }
    if (c.lane < 16) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[((c.hw * T + t) * 4 + i) * 16 + c.lane] = acc[t][i];
    }
}
template <int T>
__device__ __forceinline__ float pd_part_sum(const float* part, int t, int s, int col) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < PD_HW; ++w) v += part[((w * T + t) * 4 + s) * 16 + col];
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
// (__shfl_xor = ds_bpermute: 12 dependent ~100-cycle round trips per LayerNorm row).
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
// POLLERS: LayerNorm of LDS row `hw` of src (the row this poller wave gathered itself) into row `hw` of xs (Performer: in place), gamma / beta from LDS (ln = [gamma 512 | beta 512]);
// the arithmetic of layernorm_fwd_bf16_d512_kernel
__device__ __forceinline__ void pd_ln_row(const bf16_t* src, bf16_t* xs, const float* ln, float eps, const PdCtx& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the wave's own row is in LDS
    bf16_t* row = xs + c.hw * PD_XS + c.lane * 8;
    const bf16x8 a = *(const bf16x8*)(src + c.hw * PD_XS + c.lane * 8);
    float v[8], s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (float)a[i]; s += v[i]; }
    const float mu = pd_wave_sum(s) * (1.f / 512.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(pd_wave_sum(q) * (1.f / 512.f) + eps);
    const f32x4 g0 = *(const f32x4*)(ln + c.lane * 8), g1 = *(const f32x4*)(ln + c.lane * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(ln + PD_D + c.lane * 8), b1 = *(const f32x4*)(ln + PD_D + c.lane * 8 + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i] = (bf16_t)((v[i] - mu) * rs * g0[i] + b0[i]);
        o[4 + i] = (bf16_t)((v[4 + i] - mu) * rs * g1[i] + b1[i]);
    }
    *(bf16x8*)row = o;
}

// A per-iteration copy of the context whose thread index the optimiser cannot see through: otherwise every per-thread address of the layer loop
// is hoisted out of it and held in (spilled) registers.
__device__ __forceinline__ PdCtx pd_fresh(const PdCtx& c) {
    PdCtx r = c;
    asm volatile("" : "+v"(r.t));
    r.lane = r.t & 63;
    return r;
}

// keep the next operand set's loads BEHIND the MFMAs that free its registers (hoisted above them, both sets are live and hipcc spills)
#define PD_SCHED_FENCE()                         \
    do {                                         \
        asm volatile("" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

#define PD_DIAG(l, ph, k, val)                                                                          \
    do {                                                                                               \
        if (a.diag && g == 0 && c.t == 0) a.diag[(((size_t)m * 16 + (l)) * 8 + (ph)) * 4 + (k)] = (u64)(val); \
    } while (0)
#define PD_NOW() ((u64)wall_clock64())

// out of line: nothing is live at the kernel's start, and inlined the draw's rank loop costs the layer loop a spilled weight fragment
__device__ __noinline__ int64_t pd_draw(const float* l, int n_token, float temp, float top_p, float u, int tid) {
    return emo_nucleus_draw(l, n_token, temp, top_p, u, pd_smem + LDS_SAMP, tid, [] { PD_BARRIER(); });
}

// G2 = false: the Performer layer described above.  G2 = true: the GPT-2 block on the same skeleton (see the GPT-2 notes before the entry points).
template <bool G2>
__global__ __launch_bounds__(PD_NT, 3) void pd_step_kernel(PdArgs a) {
    bf16_t* xin = (bf16_t*)(pd_smem + LDS_XIN);        // layer input (post-LN2 / embedding), kept for the out-projection's residual
    bf16_t* xa = (bf16_t*)(pd_smem + LDS_XA);          // attention output rows
    bf16_t* x1 = (bf16_t*)(pd_smem + LDS_X1);          // post-LN1 rows, kept for the FFN2 residual
    bf16_t* fh = (bf16_t*)(pd_smem + LDS_FH);          // FFN hidden rows
    float* part = (float*)(pd_smem + LDS_PART);        // [4 waves][<= 4 tiles][4 streams][16 columns]
    float* xq = (float*)(pd_smem + LDS_ATT);           // attention scratch: q | k | v rows of (head, stream) as fp32
    float* xk = xq + PD_DH;
    float* xv = xk + PD_DH;
    float* fq = xv + PD_DH;                            // phi(q), phi(k)
    float* fk = fq + PD_F;
    float* dpart = fk + PD_F;                          // [8]
    float* num = dpart + 8;                            // [4 waves][64]
    float* upart = num + PD_HW * PD_DH;                // [2 d-halves][q | k][64 projections]
    float* npart = upart + 2 * 2 * 64;                 // [2][2] partial |x|^2
    float* ln1 = (float*)(pd_smem + LDS_LN);           // [gamma | beta] of norm1 of the layer (written by half A before P2, read by the pollers in P4)
    float* ln2 = ln1 + 2 * PD_D;                       // norm2 (written by half A in P4, read by the pollers in the next P1 / before the logits)
    int* s_misc = (int*)(pd_smem + LDS_MISC);          // [0] error flag, [1] launch counter, [2] census: group on one XCD, [4..7] the group's tokens
    float* oml = (float*)(pd_smem + LDS_OM);           // omega of the layer [64][64]   (GPT-2: scores [<= 2048] | V partial sums [32][64])
    bf16_t* xraw = (bf16_t*)(pd_smem + LDS_XRAW);      // GPT-2: block input before ln_1 (the attention residual)
    bf16_t* hraw = (bf16_t*)(pd_smem + LDS_HRAW);      // GPT-2: attention output + residual before ln_2 (the MLP residual)

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), role = wave >> 2;      // 0 = A, 1 = B, 2 = pollers
    PdCtx c;
    c.t = tid & (PD_HT - 1); c.lane = tid & 63; c.hw = wave & 3;
    c.t0 = (long long)wall_clock64();
    c.err = (gu64*)a.sync + (size_t)PD_NG * PD_GSTRIDE;
    c.local = false;
    const int g = blockIdx.x % PD_NG, m = blockIdx.x / PD_NG;
    if (g * PD_GS >= a.n_streams) return;
    gu64* gs = (gu64*)a.sync + (size_t)g * PD_GSTRIDE;
    const int hm = m >> 2, jm = m & 3;                                // P1: head / 16-dim slice; P2: head / stream
    const int L_ = a.n_layers;
    const PdLayer* LY = a.layers;
    const bool has_logits = m < (a.n_token + 15) / 16;                // member m owns logits tile m (uniform over the workgroup)
    const int nt = a.flags & 3;                                      // 0: default cache policy, 1: non-temporal weight loads, 2: timing ablation

    // ---------------------------------------------------------------- in-kernel nucleus draw: member s < 4 of a group draws stream 4 g + s
    // (the arithmetic of emo_sample_nucleus_step: same header).  The 512 threads of the two compute halves draw, the pollers keep the barrier count.
    const bool sampling = a.samp_mode == 1;
    if (sampling && m < PD_GS) {                                      // (uniform over the workgroup)
        const int64_t r = (int64_t)g * PD_GS + m;
        gu64* tokg = gs + OFF_TOK + m;
        const unsigned ept = (unsigned)PD_LOAD(gs + OFF_CNT) * 128u + 126u;
        if (r < a.n_real) {
            if (role < 2) {
                const int64_t kstep = a.step[r];
                const int64_t tk = pd_draw(a.logits_in + r * a.n_token, a.n_token, a.temp, a.top_p, a.u_steps[kstep * a.n_real + r], tid);
                if (tid == 0) {
                    a.tok_out[r] = tk;
                    if (a.seq) a.seq[r * a.ld_seq + a.col0 + kstep] = tk;
                    PD_STORE(tokg, ((u64)ept << 32) | (u64)(unsigned)tk);      // always write-through: the census has not run yet
                }
            } else {
#pragma unroll
                for (int i = 0; i < EMO_NUCLEUS_BARRIERS; ++i) PD_BARRIER();
            }
        } else if (tid == 0) {
            PD_STORE(tokg, (u64)ept << 32);                             // idle padding stream: token 0
        }
    }

    if (role == 2) {
        // ========================================================================================== POLLERS (waves 8-11)
        if (c.t == 0) { s_misc[0] = 0; s_misc[2] = 0; }
        const unsigned ep0 = (unsigned)PD_LOAD(gs + OFF_CNT) * 128u;                // (every thread reads the counter itself: member 0 bumps it after the last phase)
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15;
        if (c.t == 0) PD_STORE(gs + OFF_CEN + m, ((u64)(ep0 + 127u) << 32) | (u64)(unsigned)xcc);       // census entry: always write-through
        PD_DIAG(15, 0, 0, c.t0);
        PD_DIAG(15, 0, 1, xcc);
        PD_DIAG(15, 1, 0, (u64)clock64());                           // shader-clock cycles: with the 100-MHz stamps = the effective clock
        if (sampling && c.hw == 0) {                                  // the group's 4 drawn tokens
            u64 v = 0;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (c.lane < PD_GS) { v = PD_LOAD(gs + OFF_TOK + c.lane); ok = (unsigned)(v >> 32) == ep0 + 126u; }
                if (__all(ok)) break;
                if (pd_spin_fail(spins, c, 0x800u)) break;
            }
            if (c.lane < PD_GS) s_misc[4 + c.lane] = (int)(unsigned)v;
        }
        if (sampling) PD_SYNC_OR_LEAVE();                             // Bt (every role): s_misc[4..7] = tokens
        {   // embedding: the group's 4 rows, 8 columns per poller thread (row s = poller wave s)
            const int s = c.hw, c8 = c.lane * 8;
            const int64_t stream = (int64_t)g * PD_GS + s;
            int64_t tk, pos;
            if (sampling) {                                           // position = pos0 + (step counter AFTER the draw); the counter is bumped at the kernel's end
                tk = s_misc[4 + s];
                pos = stream < a.n_real ? a.pos0 + a.step[stream] + 1 : 0;
            } else {
                tk = a.tok[stream];
                pos = a.pos0 + (a.pos_ids ? a.pos_ids[stream] : 0);
            }
            const int64_t sg = a.seg ? a.seg[stream] : 0;
            bf16x8 o;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x4 e = *(const f32x4*)(a.E + tk * PD_D + c8 + 4 * h2);
                f32x4 sv = {0.f, 0.f, 0.f, 0.f};
                if (a.seg) sv = *(const f32x4*)(a.Sg + sg * PD_D + c8 + 4 * h2);
                const f32x4 p = *(const f32x4*)(a.pe + pos * PD_D + c8 + 4 * h2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = e[i] * a.emb_scale;                     // reference order: emb.mul_(scale); emb += seg.mul_(scale); + pe
                    v += sv[i] * a.emb_scale;
                    v += p[i];
                    o[4 * h2 + i] = (bf16_t)v;
                }
            }
            *(bf16x8*)((G2 ? xraw : xin) + s * PD_XS + c8) = o;
            if constexpr (G2) {
                if (c.lane == 0) s_misc[8 + s] = (int)pos;            // keys of the stream before this token (the row it appends)
                pd_ln_row(xraw, xin, a.ln0, a.ln_eps, c);
            }
        }
        if (c.hw == 0) {                                              // census: is the whole group on one XCD?
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
            const bool all_same = __all(same);
            if (c.lane == 0) s_misc[2] = (got && all_same) ? 1 : 0;
        }
        PD_SYNC_OR_LEAVE();                                           // B0
        c.local = s_misc[2] != 0;
        PD_DIAG(15, 0, 3, c.local ? 1 : 0);
        for (int l = 0; l < L_; ++l) {
            const unsigned ep = ep0 + (unsigned)l * 8u;
            // ---- P1
            PD_DIAG(l, 1, 0, PD_NOW());
            if (l > 0) {
                const unsigned sp = pd_gather_rows<PD_D>(gs + OFF_E1, ep - 8u + 5u, G2 ? xraw : xin, PD_XS, c, 0x100u + l);
                if (!PD_SERR) pd_ln_row(G2 ? xraw : xin, xin, ln2, a.ln_eps, c);
                PD_DIAG(l, 1, 3, sp);
            }
            PD_SYNC_OR_LEAVE();                                       // 1a: xin ready
            PD_DIAG(l, 1, 1, PD_NOW());
            PD_BARRIER();                                             // 1b: partial products ready (half A sums and publishes them)
            // ---- P2: q_h | k_h | v_h of (head hm, stream jm): 96 granules = 48 pairs
            PD_DIAG(l, 2, 0, PD_NOW());
            if (c.hw == 0) {
                const gu64* buf = gs + OFF_E2 + (hm * PD_GS + jm) * 96;
                u32x4 gq[1];
                gq[0] = (u32x4){0u, ep + 1u, 0u, ep + 1u};
                unsigned sp = 0;
                for (;;) {
                    if (c.lane < 48) pd_poll<1>(gq, buf + 2 * c.lane);
                    if (__all(gq[0][1] == ep + 1u && gq[0][3] == ep + 1u)) break;
                    if (pd_spin_fail(sp, c, 0x200u + l)) break;
                }
                if (c.lane < 48) {
                    float* dst = xq + (c.lane >> 4) * PD_DH + (c.lane & 15) * 4;      // xq, xk, xv are contiguous
                    dst[0] = __builtin_bit_cast(float, gq[0][0] << 16);
                    dst[1] = __builtin_bit_cast(float, gq[0][0] & 0xffff0000u);
                    dst[2] = __builtin_bit_cast(float, gq[0][2] << 16);
                    dst[3] = __builtin_bit_cast(float, gq[0][2] & 0xffff0000u);
                }
                PD_DIAG(l, 2, 3, sp);
            }
            PD_SYNC_OR_LEAVE();                                       // 2a
            PD_DIAG(l, 2, 1, PD_NOW());
            PD_BARRIER();                                             // 2b
            PD_BARRIER();                                             // 2c
            PD_BARRIER();                                             // 2d
            PD_DIAG(l, 2, 2, PD_NOW());
            // ---- P3
            PD_DIAG(l, 3, 0, PD_NOW());
            {
                u64 fp = 0;
                const unsigned sp = pd_gather_rows<PD_D>(gs + OFF_E3, ep + 2u, xa, PD_XS, c, 0x300u + l, a.diag ? &fp : nullptr);
                PD_DIAG(l, 6, 2, fp);
                PD_DIAG(l, 3, 3, sp);
            }
            PD_SYNC_OR_LEAVE();                                       // 3a
            PD_DIAG(l, 3, 1, PD_NOW());
            PD_BARRIER();                                             // 3b
            // ---- P4
            PD_DIAG(l, 4, 0, PD_NOW());
            {
                u64 fp = 0;
                const unsigned sp = pd_gather_rows<PD_D>(gs + OFF_E4, ep + 3u, G2 ? hraw : x1, PD_XS, c, 0x400u + l, a.diag ? &fp : nullptr);
                PD_DIAG(l, 6, 1, fp);
                if (!PD_SERR) pd_ln_row(G2 ? hraw : x1, x1, ln1, a.ln_eps, c);
                PD_DIAG(l, 4, 3, sp);
            }
            PD_SYNC_OR_LEAVE();                                       // 4a
            PD_DIAG(l, 4, 1, PD_NOW());
            PD_BARRIER();                                             // 4b
            // ---- P5
            PD_DIAG(l, 5, 0, PD_NOW());
            {
                u64 fp = 0;
                const unsigned sp = pd_gather_e5(gs + OFF_E5, ep + 4u, fh, c.t, c, 0x500u + l, a.diag ? &fp : nullptr);
                PD_DIAG(l, 6, 0, fp);
                PD_DIAG(l, 5, 3, sp);
            }
            PD_SYNC_OR_LEAVE();                                       // 5a
            PD_DIAG(l, 5, 1, PD_NOW());
            PD_BARRIER();                                             // 5b
        }
        if (!has_logits) return;
        pd_gather_rows<PD_D>(gs + OFF_E1, ep0 + (unsigned)(L_ - 1) * 8u + 5u, xin, PD_XS, c, 0x600u);
        if constexpr (!G2) {                                          // (this GPT-2 has no ln_f: the logits take the last block's output)
            if (!PD_SERR) pd_ln_row(xin, xin, ln2, a.ln_eps, c);
        }
        PD_SYNC_OR_LEAVE();                                           // Fa
        PD_BARRIER();                                                 // Fb
        PD_DIAG(15, 0, 2, PD_NOW());
        PD_DIAG(15, 1, 1, (u64)clock64());
        return;
    }

    if (role == 0) {
        // ========================================================================================== HALF A (waves 0-3): P1 and P4 products
        bf16x8 wq[3][4], w1[4][4];
        float bq[3], b1[4];
        f32x4 lnv, ln1v, om[4] = {};
        // operand set of P1 (+ what half A hands to the others through LDS before P2: omega for half B, LayerNorm1's parameters for the pollers)
#define PD_LOAD_A0(Lp)                                                                                     \
    do {                                                                                                   \
        pd_load_w_part<3, 4, 0, 4>(wq, (Lp).wqkv + (size_t)m * (PD_HW * 3 * 4 * 512), cc, nt);               \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) bq[t] = (Lp).bqkv[t * PD_D + hm * PD_DH + jm * 16 + lc16]; \
    } while (0)
#define PD_LOAD_A1(Lp)                                                                                     \
    do {                                                                                                   \
        pd_load_w_part<3, 4, 4, 8>(wq, (Lp).wqkv + (size_t)m * (PD_HW * 3 * 4 * 512), cc, nt);               \
        if constexpr (!G2) { _Pragma("unroll") for (int i = 0; i < 4; ++i) om[i] = *(const gf32x4*)((Lp).omega + cc.t * 16 + 4 * i); } \
    } while (0)
#define PD_LOAD_A2(Lp)                                                                                     \
    do {                                                                                                   \
        pd_load_w_part<3, 4, 8, 12>(wq, (Lp).wqkv + (size_t)m * (PD_HW * 3 * 4 * 512), cc, nt);              \
        ln1v = *(const gf32x4*)((cc.t < 128 ? (Lp).g1 : (Lp).be1) + (cc.t & 127) * 4);                     \
    } while (0)
#define PD_LOAD_A(Lp) do { PD_LOAD_A0(Lp); PD_LOAD_A1(Lp); PD_LOAD_A2(Lp); } while (0)
        {
            const PdCtx cc = c;
            const int lc16 = cc.lane & 15;
            PD_LOAD_A(LY[0]);
        }
        if (sampling) PD_SYNC_OR_LEAVE();                             // Bt
        PD_SYNC_OR_LEAVE();                                           // B0
        c.local = s_misc[2] != 0;
        const unsigned lc = (unsigned)PD_LOAD(gs + OFF_CNT), ep0 = lc * 128u;
        for (int l = 0; l < L_; ++l) {
            const PdLayer L = LY[l];
            const bool last = l + 1 == L_;
            const unsigned ep = ep0 + (unsigned)l * 8u;
            PdCtx cc = pd_fresh(c);
            cc.local = c.local;
            const int lc16 = cc.lane & 15;
            PD_SYNC_OR_LEAVE();                                       // 1a
            const u64 tg1 = a.diag ? PD_NOW() : 0;
            pd_gemv<3, 4>(wq, bq, xin, PD_XS, part, cc);
            PD_SCHED_FENCE();
            if (a.diag) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PD_DIAG(l, 7, 0, PD_NOW() - tg1); }
            PD_BARRIER();                                             // 1b
            // the half that computed a product sums its partials and publishes: the pollers' next poll then never waits for a store's ~2-us
            // acknowledgement (r04 diagnostics: first poll 0.19 us after a phase the pollers did not publish in, 1.8-2.0 us after one they did)
            if (cc.t < 192) {
                const int t = cc.t >> 6, s = (cc.t >> 4) & 3, col = cc.t & 15;
                pd_publish_pair(gs + OFF_E2, ((hm * PD_GS + s) * 3 + t) * 32 + ((jm * 16 + col) >> 1), ep + 1u, pd_part_sum<3>(part, t, s, col), col, cc);
            }
            PD_DIAG(l, 1, 2, PD_NOW());
            if constexpr (!G2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(oml + cc.t * 16 + 4 * i) = om[i];   // omega [64 d][64 m] of the layer for half B
            }
            *(f32x4*)(ln1 + cc.t * 4) = ln1v;                          // [gamma | beta]: threads 0..127 gamma, 128..255 beta
            PD_SCHED_FENCE();
            // (requested AFTER the publish: a 64-KB burst in front of it held the partial sums back by ~2 us — the issue itself stalls on the full queue)
            // next operand set: P4's FFN1 fragments + bias, and LayerNorm2's parameters for the pollers (handed over through LDS in P4)
            const gbf16_t* w1p = L.w1 + (size_t)m * (PD_HW * 4 * 4 * 512);
            pd_load_w_part<4, 4, 0, 3>(w1, w1p, cc, nt);
#pragma unroll
            for (int t = 0; t < 4; ++t) b1[t] = L.b1[m * 64 + t * 16 + lc16];
            lnv = *(const gf32x4*)((cc.t < 128 ? L.g2 : L.be2) + (cc.t & 127) * 4);
            PD_SYNC_OR_LEAVE();                                       // 2a
            pd_load_w_part<4, 4, 3, 6>(w1, w1p, cc, nt);
            PD_BARRIER();                                             // 2b
            pd_load_w_part<4, 4, 6, 9>(w1, w1p, cc, nt);
            PD_BARRIER();                                             // 2c
            pd_load_w_part<4, 4, 9, 12>(w1, w1p, cc, nt);
            PD_BARRIER();                                             // 2d
            pd_load_w_part<4, 4, 12, 14>(w1, w1p, cc, nt);
            PD_SYNC_OR_LEAVE();                                       // 3a
            pd_load_w_part<4, 4, 14, 16>(w1, w1p, cc, nt);
            PD_BARRIER();                                             // 3b
            // (no slice behind 3b: the E4 polls start there)
            *(f32x4*)(ln2 + cc.t * 4) = lnv;                           // [gamma | beta]: threads 0..127 gamma, 128..255 beta
            PD_SYNC_OR_LEAVE();                                       // 4a
            const u64 tg4 = a.diag ? PD_NOW() : 0;
            pd_gemv<4, 4>(w1, b1, x1, PD_XS, part, cc);
            PD_SCHED_FENCE();
            if (a.diag) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PD_DIAG(l, 7, 1, PD_NOW() - tg4); }
            PD_BARRIER();                                             // 4b
            {
                const int t = cc.t >> 6, s = (cc.t >> 4) & 3, col = cc.t & 15;      // hidden column m * 64 + t * 16 + col
                const float hv = pd_part_sum<4>(part, t, s, col);
                pd_publish_triple(gs + OFF_E5 + ((s * PD_GM + m) * 4 + t) * PD_E5_GPT, ep + 4u, G2 ? gelu_new_fast(hv) : fmaxf(hv, 0.f), col, cc);
            }
            PD_DIAG(l, 4, 2, PD_NOW());
            PD_SCHED_FENCE();
            pd_gather_e5(gs + OFF_E5, ep + 4u, fh, PD_HT + cc.t, cc, 0x580u + l);      // half A's share of the E5 gather (see pd_gather_e5)
            PD_SCHED_FENCE();
            if (!last) {
                PD_LOAD_A0(LY[l + 1]);
            } else {                                                  // (every register of the set is overwritten on both paths: the old set is dead above)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    bq[t] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int i = 0; i < 8; ++i) wq[t][ks][i] = (bf16_t)0.f;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) om[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                ln1v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (has_logits) {                                     // the logits tile reuses the first q/k/v fragment set
                    bf16x8 wl[1][4];
                    pd_load_w<1, 4>(wl, (const gbf16_t*)a.wout + (size_t)m * (PD_HW * 1 * 4 * 512), cc, nt);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) wq[0][ks] = wl[0][ks];
                    bq[0] = (m * 16 + lc16) < a.n_token ? a.bout[m * 16 + lc16] : 0.f;
                }
            }
            PD_SYNC_OR_LEAVE();                                       // 5a
            if (!last) PD_LOAD_A1(LY[l + 1]);
            PD_BARRIER();                                             // 5b
            if (!last) PD_LOAD_A2(LY[l + 1]);
        }
        // the drawn stream's step counter: every member of the group read it for the token's position before its first edge, long ago
        if (sampling && m < PD_GS && c.t == 0 && (int64_t)g * PD_GS + m < a.n_real) a.step[(int64_t)g * PD_GS + m] += 1;
        if (!has_logits) return;
        PD_SYNC_OR_LEAVE();                                           // Fa
        {
            bf16x8 wl[1][4];
            float bl[1] = {bq[0]};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wl[0][ks] = wq[0][ks];
            pd_gemv<1, 4>(wl, bl, xin, PD_XS, part, c);
        }
        PD_BARRIER();                                                 // Fb
        if (c.t < 64) {
            const int s = c.t >> 4, col = c.t & 15, gc = m * 16 + col;
            if (gc < a.n_token) a.logits[((int64_t)g * PD_GS + s) * a.n_token + gc] = pd_part_sum<1>(part, 0, s, col);
        }
        // member 0 gathered the last edge from EVERY member of the group, so all of them have long read the counter
        if (m == 0 && c.t == 0) PD_STORE(gs + OFF_CNT, (u64)(lc + 1u));
        return;
    }

    // ============================================================================================== HALF B (waves 4-7): P2, P3, P5
    {
        bf16x8 wo[1][4], w2[1][16];
        float bo[1], b2[1];
        f32x4 st[8] = {};
        float zold = 0.f;
        const int64_t sh = ((int64_t)g * PD_GS + jm) * PD_H + hm;     // (stream, head) of this member
#define PD_LOAD_B(Lp)                                                                                      \
    do {                                                                                                   \
        if constexpr (!G2) {                                                                               \
            pd_load_w<1, 4>(wo, (Lp).wo + (size_t)m * (PD_HW * 1 * 4 * 512), cc, nt);                       \
            bo[0] = (Lp).bo[m * 16 + lc16];                                                                \
            const gfloat* Sb_ = (Lp).S + sh * (PD_F * PD_DH);                                              \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) st[i] = *(const gf32x4*)(Sb_ + (fg + 16 * i) * PD_DH + d4); \
            if (cc.t < PD_F) zold = (Lp).z[sh * PD_F + cc.t];                                              \
        }                                                                                                  \
    } while (0)
        {
            const PdCtx cc = c;
            const int lc16 = cc.lane & 15, d4 = (cc.t & 15) * 4, fg = cc.t >> 4;
            PD_LOAD_B(LY[0]);
        }
        if (sampling) PD_SYNC_OR_LEAVE();                             // Bt
        PD_SYNC_OR_LEAVE();                                           // B0
        c.local = s_misc[2] != 0;
        const unsigned ep0 = (unsigned)PD_LOAD(gs + OFF_CNT) * 128u;  // (member 0 bumps the counter only after every member's last phase)
        for (int l = 0; l < L_; ++l) {
            const PdLayer L = LY[l];
            const bool last = l + 1 == L_;
            const unsigned ep = ep0 + (unsigned)l * 8u;
            gfloat* Sb = L.S + sh * (PD_F * PD_DH);
            PdCtx cc = pd_fresh(c);
            cc.local = c.local;
            const int lc16 = cc.lane & 15, d4 = (cc.t & 15) * 4, fg = cc.t >> 4;      // state mapping: 16 threads per state row, 16 rows per pass, 8 passes
            PD_SYNC_OR_LEAVE();                                       // 1a
            if (!G2 && l > 0) {                                       // (layer 0's slice was requested before the loop)
#pragma unroll
                for (int i = 0; i < 4; ++i) st[i] = *(const gf32x4*)(Sb + (fg + 16 * i) * PD_DH + d4);
            }
            PD_BARRIER();                                             // 1b
            if (!G2 && l > 0) {
#pragma unroll
                for (int i = 4; i < 8; ++i) st[i] = *(const gf32x4*)(Sb + (fg + 16 * i) * PD_DH + d4);
                if (cc.t < PD_F) zold = L.z[sh * PD_F + cc.t];
            }
            // ---- P2: the recurrent step  (GPT-2: attention over the KV cache)
            // GPT-2: thread = (key row rl of a 32-row pass, dims c8 .. c8 + 7); a SWEEP = 8 passes = 256 rows = 8 x 16-B loads per thread, two sweeps (A, B)
            // in flight.  The first two key sweeps are requested BEFORE the q row arrives (they do not depend on it), the first two value sweeps before the
            // softmax's two barriers.
            const int kprev = G2 ? min(s_misc[8 + jm], (int)a.kv_tmax - 1) : 0;      // cached rows; this token's row gets index kprev
            const int last_c = kprev > 0 ? kprev - 1 : 0;             // rows past the cache: any valid address, the value is replaced / masked
            const int rl = cc.t >> 3, c8 = (cc.t & 7) * 8;
            gbf16_t* Kc = (gbf16_t*)L.S + (sh * a.kv_tmax) * PD_DH + c8;
            gbf16_t* Vc = (gbf16_t*)L.z + (sh * a.kv_tmax) * PD_DH + c8;
            bf16x8 kA[8], kB[8];
#define PD_KV_LOAD(dst, base, J0)                                                                          \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
        const int j_ = (J0) + u * 32 + rl;                                                                 \
        dst[u] = *(const gbf16x8*)((base) + (int64_t)(j_ < kprev ? j_ : last_c) * PD_DH);                  \
    }
            if constexpr (G2) {
                PD_KV_LOAD(kA, Kc, 0);
                PD_KV_LOAD(kB, Kc, 256);      // (unconditional: rows past the cache read one hot row — a conditionally defined array stays live across the layer loop)
            }
            PD_SYNC_OR_LEAVE();                                       // 2a: q | k | v rows (poller wave 0) and omega (half A) staged
            if constexpr (G2) {
                // softmax attention of (head hm, stream jm) over the stream's cached keys + the row this token appends (the arithmetic of
                // sattn_decode_kernel); scores and the 32 partial V sums live in the omega region
                float* sc = oml;
                float* vp = oml + 2048;
                bf16x8 kn, vn;
                float qv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { qv[e] = xq[c8 + e]; kn[e] = (bf16_t)xk[c8 + e]; vn[e] = (bf16_t)xv[c8 + e]; }
                if (cc.t < 8) {
                    *(gbf16x8*)(Kc + (int64_t)kprev * PD_DH) = kn;
                    *(gbf16x8*)(Vc + (int64_t)kprev * PD_DH) = vn;
                }
                float mx = -3.0e38f;
#define PD_KV_SCORE(src, J0)                                                                               \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
        const int j_ = (J0) + u * 32 + rl;                                                                 \
        const bf16x8 kk = j_ == kprev ? kn : src[u];                                                       \
        float d = 0.f;                                                                                     \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) d += qv[e] * (float)kk[e];                           \
        d = pd_dpp_add(d, 0);                                                                              \
        d = pd_dpp_add(d, 1);                                                                              \
        d = pd_dpp_add(d, 2);                                                                              \
        d *= 0.125f;                                                                                       \
        if (j_ <= kprev) {                                                                                 \
            mx = fmaxf(mx, d);                                                                             \
            if ((cc.t & 7) == 0) sc[j_] = d;                                                               \
        }                                                                                                  \
    }
                for (int j0 = 0; j0 <= kprev; j0 += 512) {
                    PD_KV_SCORE(kA, j0);
                    if (j0 + 512 <= kprev) { PD_KV_LOAD(kA, Kc, j0 + 512); }
                    if (j0 + 256 <= kprev) {
                        PD_KV_SCORE(kB, j0 + 256);
                        if (j0 + 768 <= kprev) { PD_KV_LOAD(kB, Kc, j0 + 768); }
                    }
                }
                PD_SCHED_FENCE();
                bf16x8 vA[8], vB[8];
                PD_KV_LOAD(vA, Vc, 0);
                PD_KV_LOAD(vB, Vc, 256);
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                if (cc.lane == 0) dpart[cc.hw] = mx;
                PD_BARRIER();                                         // 2b
                mx = fmaxf(fmaxf(dpart[0], dpart[1]), fmaxf(dpart[2], dpart[3]));
                float sum = 0.f;
                for (int j = cc.t; j <= kprev; j += PD_HT) {
                    const float pj = __expf(sc[j] - mx);
                    sc[j] = pj;
                    sum += pj;
                }
                sum = pd_wave_sum(sum);
                if (cc.lane == 0) npart[cc.hw] = sum;
                PD_BARRIER();                                         // 2c
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#define PD_KV_ACC(src, J0)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
        const int j_ = (J0) + u * 32 + rl;                                                                 \
        const bf16x8 vv = j_ == kprev ? vn : src[u];                                                       \
        const float pj = j_ <= kprev ? sc[j_] : 0.f;                                                       \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) acc[e] += pj * (float)vv[e];                         \
    }
                for (int j0 = 0; j0 <= kprev; j0 += 512) {
                    PD_KV_ACC(vA, j0);
                    if (j0 + 512 <= kprev) { PD_KV_LOAD(vA, Vc, j0 + 512); }
                    if (j0 + 256 <= kprev) {
                        PD_KV_ACC(vB, j0 + 256);
                        if (j0 + 768 <= kprev) { PD_KV_LOAD(vB, Vc, j0 + 768); }
                    }
                }
                *(f32x4*)(vp + rl * PD_DH + c8) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
                *(f32x4*)(vp + rl * PD_DH + c8 + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
                PD_BARRIER();                                         // 2d
                if (cc.t < PD_DH) {
                    float o = 0.f;
#pragma unroll 8
                    for (int r = 0; r < 32; ++r) o += vp[r * PD_DH + cc.t];
                    o = o / ((npart[0] + npart[1]) + (npart[2] + npart[3]));
                    pd_publish_pair(gs + OFF_E3, jm * (PD_D / 2) + ((hm * PD_DH + cc.t) >> 1), ep + 2u, o, c.t, cc);
                }
                // P3's operand set is requested HERE and lands behind the E3 edge (held from the end of the previous layer, or requested between the
                // sweeps, hipcc spilled it: each spill is a vmcnt(0) in the middle of the sweeps)
                PD_SCHED_FENCE();
                pd_load_w<1, 4>(wo, L.wo + (size_t)m * (PD_HW * 1 * 4 * 512), cc, nt);
                bo[0] = L.bo[m * 16 + lc16];
            } else {
                {   // projections: thread = (d-half, q | k, projection): 32 of the 64 terms each
                    const int col = cc.t & 63, which = (cc.t >> 6) & 1, hd = cc.t >> 7;
                    const float* xx = which ? xk : xq;
                    float u = 0.f, nn = 0.f;
#pragma unroll 8
                    for (int d = 0; d < 32; ++d) {
                        const float xv_ = xx[hd * 32 + d];
                        u += xv_ * oml[(hd * 32 + d) * PD_MF + col];
                        nn += xv_ * xv_;
                    }
                    upart[(hd * 2 + which) * 64 + col] = u;
                    if (col == 0) npart[hd * 2 + which] = nn;
                }
                PD_BARRIER();                                             // 2b
                const float cs = rsqrtf(sqrtf((float)PD_DH)), half_ln_f = 0.5f * logf((float)PD_F);
                float dn = 0.f;
                if (cc.t < PD_F) {
                    const int col = cc.t & (PD_MF - 1);
                    const float sgn = cc.t < PD_MF ? 1.f : -1.f;
                    const float uq = upart[col] + upart[128 + col], uk = upart[64 + col] + upart[192 + col];
                    const float nq = npart[0] + npart[2], nk = npart[1] + npart[3];
                    const float pq = __expf(sgn * cs * uq - (0.5f * cs * cs * nq + half_ln_f));
                    const float pk = __expf(sgn * cs * uk - (0.5f * cs * cs * nk + half_ln_f));
                    fq[cc.t] = pq;
                    fk[cc.t] = pk;
                    const float z = zold + pk;
                    L.z[sh * PD_F + cc.t] = z;
                    dn = pq * z;
                }
                dn = pd_wave_sum(dn);
                if (cc.lane == 0) dpart[cc.hw] = dn;
                PD_BARRIER();                                             // 2c
                {
                    const f32x4 vd = {xv[d4], xv[d4 + 1], xv[d4 + 2], xv[d4 + 3]};
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int f = fg + 16 * i;
                        const f32x4 sv = st[i] + fk[f] * vd;
                        *(gf32x4*)(Sb + f * PD_DH + d4) = sv;
                        acc += fq[f] * sv;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = rows4_sum(acc[i]);   // the wave's 4 rows per pass sit in lanes l, l^16, l^32, l^48
                    if (cc.lane < 16) *(f32x4*)(num + cc.hw * PD_DH + d4) = acc;
                }
                PD_BARRIER();                                             // 2d
                if (cc.t < PD_DH) {
                    float o = (num[cc.t] + num[PD_DH + cc.t]) + (num[2 * PD_DH + cc.t] + num[3 * PD_DH + cc.t]);
                    o = o / (dpart[0] + dpart[1] + a.eps);                // (waves 2, 3 of the half hold no features)
                    pd_publish_pair(gs + OFF_E3, jm * (PD_D / 2) + ((hm * PD_DH + cc.t) >> 1), ep + 2u, o, c.t, cc);
                }
            }
            // ---- P3
            PD_SYNC_OR_LEAVE();                                       // 3a
            const u64 tg3 = a.diag ? PD_NOW() : 0;
            pd_gemv<1, 4>(wo, bo, xa, PD_XS, part, cc);
            PD_SCHED_FENCE();
            if (a.diag) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PD_DIAG(l, 7, 2, PD_NOW() - tg3); }
            PD_BARRIER();                                             // 3b
            if (cc.t < 64) {
                const int s = cc.t >> 4, col = cc.t & 15, gc = m * 16 + col;
                pd_publish_pair(gs + OFF_E4, s * (PD_D / 2) + (gc >> 1), ep + 3u, pd_part_sum<1>(part, 0, s, col) + (float)(G2 ? xraw : xin)[s * PD_XS + gc], col, cc);
            }
            PD_DIAG(l, 3, 2, PD_NOW());
            PD_SCHED_FENCE();
            const gbf16_t* w2p = L.w2 + (size_t)m * (PD_HW * 1 * 16 * 512);                // next operand set: P5, in three slices
            // (nothing is requested between the E4 publish and barrier 4a: the pollers' E4 polls queued behind a slice issued here — r06 diagnostics:
            // first E4 poll pass 1.3 us against 0.4 us for E3, same payload)
            PD_SYNC_OR_LEAVE();                                       // 4a
            pd_load_w_part<1, 16, 0, 8>(w2, w2p, cc, nt);
            b2[0] = L.b2[m * 16 + lc16];
            PD_BARRIER();                                             // 4b
            pd_load_w_part<1, 16, 8, 16>(w2, w2p, cc, nt);
            // ---- P5
            PD_SYNC_OR_LEAVE();                                       // 5a
            const u64 tg5 = a.diag ? PD_NOW() : 0;
            pd_gemv<1, 16>(w2, b2, fh, PD_FS, part, cc);
            PD_SCHED_FENCE();
            if (a.diag) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PD_DIAG(l, 7, 3, PD_NOW() - tg5); }
            PD_BARRIER();                                             // 5b
            if (cc.t < 64) {
                const int s = cc.t >> 4, col = cc.t & 15, gc = m * 16 + col;
                pd_publish_pair(gs + OFF_E1, s * (PD_D / 2) + (gc >> 1), ep + 5u, pd_part_sum<1>(part, 0, s, col) + (float)(G2 ? hraw : x1)[s * PD_XS + gc], col, cc);
            }
            PD_DIAG(l, 5, 2, PD_NOW());
            PD_SCHED_FENCE();
            if (!G2 && !last) {                                       // next operand set: the next layer's P3 fragments now, its state slice after 1a / 1b
                pd_load_w<1, 4>(wo, LY[l + 1].wo + (size_t)m * (PD_HW * 1 * 4 * 512), cc, nt);
                bo[0] = LY[l + 1].bo[m * 16 + lc16];
            }
        }
        if (!has_logits) return;
        PD_SYNC_OR_LEAVE();                                           // Fa
        PD_BARRIER();                                                 // Fb
    }
}
}  // namespace

extern "C" int64_t emo_performer_decode_step_workspace_bytes(void) { return (int64_t)PD_WS_WORDS * 8; }

// The launch is PD_NG * PD_GM = 256 workgroups that spin-wait on each other: all of them must be resident at once, one per CU.  Checked once per
// process: the 96-KB dynamic LDS attribute could be set, the device has at least 256 CUs (partition modes / CU masks have fewer), and the
// occupancy query grants the kernel a workgroup per CU.  (Another process holding CUs cannot be seen from here: that case is the 50-ms give-up
// code in the workspace, emo_performer_decode_step's documented failure mode.)
static int pd_supported() {
    static int cached = -1;
    if (cached >= 0) return cached;
    const size_t lds = 96 * 1024;
    int dev = 0, cus = 0, per_cu = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= PD_NG * PD_GM;
    ok = ok && hipFuncSetAttribute((const void*)pd_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)pd_step_kernel<false>, PD_NT, lds) == hipSuccess && per_cu >= 1;
    ok = ok && hipFuncSetAttribute((const void*)pd_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)pd_step_kernel<true>, PD_NT, lds) == hipSuccess && per_cu >= 1;
    (void)hipGetLastError();
    cached = ok ? 1 : 0;
    return cached;
}
extern "C" int emo_performer_decode_step_supported(void) { return pd_supported(); }

static int pd_launch(const void* layer_table, int64_t n_layers, const int64_t* tok, const int64_t* seg, const float* E, const float* Sg, const float* pe,
                     float emb_scale, int64_t pos0, const int64_t* pos_ids, const void* wout_packed, const float* bout, int64_t n_token, float* logits,
                     int64_t n_streams, int64_t d_model, int64_t n_head, int64_t n_feat, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes, float eps,
                     float ln_eps, int64_t* diag, int samp_mode, float temperature, float top_p, const float* u_steps, int64_t* step, int64_t* seq,
                     int64_t ld_seq, int64_t col0, int64_t* tok_out, const float* logits_in, int64_t n_real, emo_stream_t stream, bool g2 = false,
                     const float* ln0 = nullptr, int64_t kv_tmax = 0) {
    EMO_CHECK(layer_table && E && pe && wout_packed && bout && logits && sync_ws, "emo_performer_decode_step: null pointer");
    if (g2) EMO_CHECK(ln0 && kv_tmax >= 1 && kv_tmax <= 2048, "emo_gpt2_decode_step: needs layer 0's ln_1 parameters and a KV cache of <= 2048 rows per (stream, head)");
    EMO_CHECK(d_model == PD_D && n_head == PD_H && n_feat == PD_F && d_ff == PD_FF,
              "emo_performer_decode_step: built for d_model 512 / 8 heads / 128 features / d_ff 2048 (got %lld / %lld / %lld / %lld)", (long long)d_model,
              (long long)n_head, (long long)n_feat, (long long)d_ff);
    EMO_CHECK(n_layers >= 1 && n_layers <= PD_MAX_LAYERS, "emo_performer_decode_step: 1 <= n_layers <= %d", PD_MAX_LAYERS);
    EMO_CHECK(n_streams >= PD_GS && n_streams <= PD_GS * PD_NG && n_streams % PD_GS == 0, "emo_performer_decode_step: n_streams must be a multiple of 4, <= 32");
    EMO_CHECK(n_token >= 1 && n_token <= 16 * PD_GM, "emo_performer_decode_step: n_token <= 512");
    EMO_CHECK(!(seg && !Sg), "emo_performer_decode_step: seg ids without a segment table");
    EMO_CHECK(sync_ws_bytes >= (int64_t)PD_WS_WORDS * 8 && ((uintptr_t)sync_ws & 15) == 0, "emo_performer_decode_step: workspace too small / unaligned");
    if (samp_mode) {
        EMO_CHECK(u_steps && step && tok_out && logits_in && n_real >= 1 && n_real <= n_streams && temperature > 0.f && n_token <= 1024,
                  "emo_performer_decode_step_sampled: bad sampling arguments");
    } else {
        EMO_CHECK(tok, "emo_performer_decode_step: null token pointer");
    }
    PdArgs a;
    a.layers = (const PdLayer*)layer_table; a.n_layers = (int)n_layers;
    a.tok = tok; a.seg = seg; a.E = E; a.Sg = Sg; a.pe = pe; a.emb_scale = emb_scale; a.pos0 = pos0; a.pos_ids = pos_ids;
    a.wout = (const bf16_t*)wout_packed; a.bout = bout; a.n_token = (int)n_token; a.logits = logits; a.n_streams = (int)n_streams;
    a.sync = (u64*)sync_ws; a.eps = eps; a.ln_eps = ln_eps; a.diag = (u64*)diag;
    a.samp_mode = samp_mode; a.temp = temperature; a.top_p = top_p; a.u_steps = u_steps; a.step = step; a.seq = seq; a.ld_seq = ld_seq; a.col0 = col0;
    a.tok_out = tok_out; a.logits_in = logits_in; a.n_real = (int)n_real;
    a.ln0 = ln0; a.kv_tmax = kv_tmax;
    { const char* e = getenv("EMO_PD_NT"); a.flags = e ? (atoi(e) & 3) : 0; }
    static_assert(LDS_TOTAL <= 96 * 1024, "LDS carve");
    const size_t lds = 96 * 1024;                                         // > half of the CU's LDS: one workgroup per CU
    EMO_CHECK(pd_supported(), "emo_performer_decode_step: this device / partition cannot hold the launch's %d workgroups at once (needs >= %d CUs with 96 KB "
              "of LDS each): use the chain of launches", PD_NG * PD_GM, PD_NG * PD_GM);
    if (g2) hipLaunchKernelGGL(pd_step_kernel<true>, dim3(PD_NG * PD_GM), dim3(PD_NT), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pd_step_kernel<false>, dim3(PD_NG * PD_GM), dim3(PD_NT), lds, (hipStream_t)stream, a);
    EMO_LAUNCH_CHECK();
    return EMO_OK;
}

extern "C" int emo_performer_decode_step(const void* layer_table, int64_t n_layers, const int64_t* tok, const int64_t* seg, const float* E, const float* Sg,
                                         const float* pe, float emb_scale, int64_t pos0, const int64_t* pos_ids, const void* wout_packed,
                                         const float* bout, int64_t n_token, float* logits, int64_t n_streams, int64_t d_model, int64_t n_head,
                                         int64_t n_feat, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes, float eps, float ln_eps,
                                         int64_t* diag, emo_stream_t stream) {
    return pd_launch(layer_table, n_layers, tok, seg, E, Sg, pe, emb_scale, pos0, pos_ids, wout_packed, bout, n_token, logits, n_streams, d_model, n_head,
                     n_feat, d_ff, sync_ws, sync_ws_bytes, eps, ln_eps, diag, 0, 1.f, 1.f, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, stream);
}

extern "C" int emo_performer_decode_step_sampled(const void* layer_table, int64_t n_layers, const int64_t* seg, const float* E, const float* Sg,
                                                 const float* pe, float emb_scale, int64_t pos0, const void* wout_packed, const float* bout,
                                                 int64_t n_token, float* logits, int64_t n_streams, int64_t n_real, int64_t d_model, int64_t n_head,
                                                 int64_t n_feat, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes, float eps, float ln_eps,
                                                 float temperature, float top_p, const float* u_steps, int64_t* step, int64_t* seq, int64_t ld_seq,
                                                 int64_t col0, int64_t* tok_out, emo_stream_t stream) {
    return pd_launch(layer_table, n_layers, nullptr, seg, E, Sg, pe, emb_scale, pos0, nullptr, wout_packed, bout, n_token, logits, n_streams, d_model, n_head,
                     n_feat, d_ff, sync_ws, sync_ws_bytes, eps, ln_eps, nullptr, 1, temperature, top_p, u_steps, step, seq, ld_seq, col0, tok_out, logits,
                     n_real, stream);
}

// ---------------------------------------------------------------------------------------------------------------- GPT-2 form (r06)
// The same launch for the GPT-2 backbone of BASELINE configs[3] (reference: stage2_accompaniment/model/music_gpt2.py -> HF GPT2Block, pre-LN,
// gelu_new, no ln_f; the loop of stage2_accompaniment/inference.py:250-277): pd_step_kernel<true>.  Differences to the Performer layer, all inside the
// same five edges and the same barrier sequence:
//   * the gathered rows are the RAW residual stream; the pollers normalise them out of place (E1 -> ln_1 of the next block, E4 -> ln_2), half B adds
//     the raw rows as residuals; layer 0's ln_1 parameters come from `ln0`, and the logits take the last block's output as it is;
//   * P2 is softmax attention of (head, stream) over the stream's head-major KV cache [n, H, kv_tmax, 64] (bf16): the member appends the token's key /
//     value row at index pos (= keys already cached; pos = pos0 + pos_ids[stream], or the sampler's counter) and reads the rows before it — the
//     table's S / z slots hold the K / V cache of the layer, the omega slot is unused;
//   * the table's g1 / be1 are ln_2 of the block, g2 / be2 are ln_1 of the NEXT block (any valid pointer for the last one); FFN activation gelu_new.
extern "C" int emo_gpt2_decode_step_supported(void) { return pd_supported(); }

extern "C" int emo_gpt2_decode_step(const void* layer_table, int64_t n_layers, const int64_t* tok, const int64_t* seg, const float* E, const float* Sg,
                                    const float* pe, float emb_scale, int64_t pos0, const int64_t* pos_ids, const float* ln0, int64_t kv_tmax,
                                    const void* wout_packed, const float* bout, int64_t n_token, float* logits, int64_t n_streams, int64_t d_model,
                                    int64_t n_head, int64_t d_ff, void* sync_ws, int64_t sync_ws_bytes, float ln_eps, int64_t* diag, emo_stream_t stream) {
    return pd_launch(layer_table, n_layers, tok, seg, E, Sg, pe, emb_scale, pos0, pos_ids, wout_packed, bout, n_token, logits, n_streams, d_model, n_head,
                     PD_F, d_ff, sync_ws, sync_ws_bytes, 0.f, ln_eps, diag, 0, 1.f, 1.f, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, stream, true, ln0,
                     kv_tmax);
}

extern "C" int emo_gpt2_decode_step_sampled(const void* layer_table, int64_t n_layers, const int64_t* seg, const float* E, const float* Sg, const float* pe,
                                            float emb_scale, int64_t pos0, const float* ln0, int64_t kv_tmax, const void* wout_packed, const float* bout,
                                            int64_t n_token, float* logits, int64_t n_streams, int64_t n_real, int64_t d_model, int64_t n_head, int64_t d_ff,
                                            void* sync_ws, int64_t sync_ws_bytes, float ln_eps, float temperature, float top_p, const float* u_steps,
                                            int64_t* step, int64_t* seq, int64_t ld_seq, int64_t col0, int64_t* tok_out, emo_stream_t stream) {
    return pd_launch(layer_table, n_layers, nullptr, seg, E, Sg, pe, emb_scale, pos0, nullptr, wout_packed, bout, n_token, logits, n_streams, d_model, n_head,
                     PD_F, d_ff, sync_ws, sync_ws_bytes, 0.f, ln_eps, nullptr, 1, temperature, top_p, u_steps, step, seq, ld_seq, col0, tok_out, logits, n_real,
                     stream, true, ln0, kv_tmax);
}
