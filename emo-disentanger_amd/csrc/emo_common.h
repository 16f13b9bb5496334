// Common device/host utilities for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/emo_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define EMO_WAVE 64

// ---------------------------------------------------------------- error handling (thread-local message)
void emo_set_error(const char* fmt, ...);
#define EMO_CHECK(cond, ...)                       \
    do {                                           \
        if (!(cond)) {                             \
            emo_set_error(__VA_ARGS__);            \
            return EMO_ERR_INVALID;                \
        }                                          \
    } while (0)
#define EMO_LAUNCH_CHECK()                                                   \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            emo_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,     \
                          hipGetErrorString(e__));                           \
            return EMO_ERR_LAUNCH;                                           \
        }                                                                    \
    } while (0)

// ---------------------------------------------------------------- scalar conversion helpers
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---------------------------------------------------------------- counter-based dropout RNG
// One keyed 32-bit hash per FOUR elements (16 random bits each, see drop_mult).  keep iff bits >= thr16,
// thr16 = round(p * 65536).  Forward and backward regenerate the same mask from
// (seed, offset, linear element index) — nothing is stored.
// The 64-bit key (two words derived from the 64-bit seed and the per-site offset) enters the hash at two
// points — XORed into the counter before the first round and again between the two
// multiply rounds — so two (seed, offset) streams are NOT index-shifted copies of one sequence (an
// additive 32-bit key in front of a fixed hash would make them exactly that).
__host__ __device__ __forceinline__ uint32_t emo_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct DropCtx {
    uint32_t key;    // mixed (seed, offset), word 0
    uint32_t key2;   // word 1
    uint32_t thr16;  // 0 => dropout disabled
    float scale;     // 1/(1-p)
};
__host__ __device__ __forceinline__ uint32_t emo_drop_hash(const DropCtx& d, uint32_t pair) {
    // (r04: the counter enters un-multiplied — the two multiply-xorshift rounds below are a full-avalanche bijection on their own, and the third
    // quarter-rate 32-bit multiply was a sixth of the hash; keep rate, lag-1 / lag-4 / row-stride correlations, cross-stream correlations and
    // avalanche re-checked on 2^24 samples: all at the sampling-noise level, as before)
    uint32_t x = pair ^ d.key;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= d.key2; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ DropCtx make_drop(float p, uint64_t seed, uint64_t offset) {
    DropCtx d;
    d.thr16 = p > 0.f ? (uint32_t)(p * 65536.f + 0.5f) : 0u;
    d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32), o0 = (uint32_t)offset, o1 = (uint32_t)(offset >> 32);
    uint32_t k = emo_hash32(s0 ^ 0x9E3779B9u);
    k = emo_hash32(k ^ s1);
    k = emo_hash32(k ^ o0 * 0x85EBCA6Bu ^ o1);
    uint32_t k2 = emo_hash32(s1 ^ 0xC2B2AE35u);
    k2 = emo_hash32(k2 + o0);
    k2 = emo_hash32(k2 ^ s0 * 0x27D4EB2Fu ^ o1 * 0x165667B1u);
    d.key = k;
    d.key2 = k2;
    return d;
}
// One keyed hash serves FOUR consecutive elements (16 random bits each): elements 4q, 4q+1 take the two halves of h = hash(q),
// elements 4q+2, 4q+3 the halves of xorshift32(h).  The hash's three 32-bit multiplies are quarter-rate VALU ops; r02 s_memtime
// stamps put the relu + dropout epilogue of a GEMM column tile at 41 % of a wave's cycles with one hash per element pair.
__host__ __device__ __forceinline__ uint32_t emo_xs32(uint32_t x) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    return x;
}
// multiplier (0 or scale) for linear element index idx
__host__ __device__ __forceinline__ float drop_mult(const DropCtx& d, uint64_t idx) {
    if (d.thr16 == 0u) return 1.f;
    const uint32_t quad = (uint32_t)(idx >> 2) ^ (uint32_t)(idx >> 34) * 0x9E3779B1u;
    uint32_t h = emo_drop_hash(d, quad);
    if (idx & 2) h = emo_xs32(h);
    const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
    return bits >= d.thr16 ? d.scale : 0.f;
}

// 4 consecutive elements: one hash when the start index is a multiple of 4; bit-identical to drop_mult()
__device__ __forceinline__ void drop_mult4(const DropCtx& d, uint64_t idx0, float (&m)[4]) {
    if (d.thr16 == 0u) { m[0] = m[1] = m[2] = m[3] = 1.f; return; }
    if (idx0 & 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = drop_mult(d, idx0 + i);
        return;
    }
    const uint32_t quad = (uint32_t)(idx0 >> 2) ^ (uint32_t)(idx0 >> 34) * 0x9E3779B1u;
    const uint32_t h = emo_drop_hash(d, quad), h2 = emo_xs32(h);
    m[0] = (h & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
    m[1] = (h >> 16) >= d.thr16 ? d.scale : 0.f;
    m[2] = (h2 & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
    m[3] = (h2 >> 16) >= d.thr16 ? d.scale : 0.f;
}

// The same for an aligned quad (idx0 % 4 == 0, dropout on), also returning the four keep decisions as bits pos .. pos + 3 OR-ed into `word`
// (the attention forward's keep words): the compares are shared between the multiplier and the bit.
// (quad = idx0 >> 2 of an index below 2^34, where the high-word term of the general form vanishes: callers that know their tensor is that small
// keep the quad index in 32-bit arithmetic — the 64-bit shifts and the extra quarter-rate multiply were a sixth of the hash's cost)
__device__ __forceinline__ void drop_mult4_bits_q(const DropCtx& d, uint32_t quad, float (&m)[4], uint32_t& word, int pos) {
    const uint32_t h = emo_drop_hash(d, quad), h2 = emo_xs32(h);
    const bool c0 = (h & 0xFFFFu) >= d.thr16, c1 = (h >> 16) >= d.thr16, c2 = (h2 & 0xFFFFu) >= d.thr16, c3 = (h2 >> 16) >= d.thr16;
    m[0] = c0 ? d.scale : 0.f; m[1] = c1 ? d.scale : 0.f; m[2] = c2 ? d.scale : 0.f; m[3] = c3 ? d.scale : 0.f;
    word |= (c0 ? 1u << pos : 0u) | (c1 ? 2u << pos : 0u) | (c2 ? 4u << pos : 0u) | (c3 ? 8u << pos : 0u);
}
__device__ __forceinline__ void drop_mult4_bits(const DropCtx& d, uint64_t idx0, float (&m)[4], uint32_t& word, int pos) {
    const uint32_t quad = (uint32_t)(idx0 >> 2) ^ (uint32_t)(idx0 >> 34) * 0x9E3779B1u;
    const uint32_t h = emo_drop_hash(d, quad), h2 = emo_xs32(h);
    const bool c0 = (h & 0xFFFFu) >= d.thr16, c1 = (h >> 16) >= d.thr16, c2 = (h2 & 0xFFFFu) >= d.thr16, c3 = (h2 >> 16) >= d.thr16;
    m[0] = c0 ? d.scale : 0.f; m[1] = c1 ? d.scale : 0.f; m[2] = c2 ? d.scale : 0.f; m[3] = c3 ? d.scale : 0.f;
    word |= (c0 ? 1u << pos : 0u) | (c1 ? 2u << pos : 0u) | (c2 ? 4u << pos : 0u) | (c3 ? 8u << pos : 0u);
}

// Key-stationary layout (attention dK/dV passes): a lane owns ONE key column j and 4 query rows t0 .. t0+3, so its four elements are T apart
// and drop_mult4 does not apply — but the 4 lanes of a quad own keys 4q .. 4q+3 of the SAME rows, i.e. the four 16-bit fields of one hash
// per row.  Lane c of the quad hashes row t0 + c and the words travel by DPP quad broadcasts: 1 hash per lane per 4 rows instead of 4 (the
// per-element hashes made the GPT-2 dK/dV kernel 2.8x the dQ kernel).  idx_own = linear index of (row t0 + (lane & 3), key j & ~3); it must
// be a multiple of 4 (T % 4 == 0 and j & 3 == lane & 3).  Bit-identical to drop_mult().
template <int R> __device__ __forceinline__ uint32_t emo_quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, R * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ void drop_mult_col4(const DropCtx& d, uint64_t idx_own, int lane, float (&m)[4]) {
    const uint32_t quad = (uint32_t)(idx_own >> 2) ^ (uint32_t)(idx_own >> 34) * 0x9E3779B1u;
    const uint32_t h = emo_drop_hash(d, quad), h2 = emo_xs32(h);
    const bool hi_word = lane & 2, hi_half = lane & 1;
    // (all eight broadcasts are executed by every lane BEFORE the per-lane selects: a `cond ? bcast(a) : bcast(b)` would put the DPP moves
    // under divergent EXEC masks, where a disabled source lane reads as 0)
    const uint32_t a0 = emo_quad_bcast<0>(h), a1 = emo_quad_bcast<1>(h), a2 = emo_quad_bcast<2>(h), a3 = emo_quad_bcast<3>(h);
    const uint32_t b0 = emo_quad_bcast<0>(h2), b1 = emo_quad_bcast<1>(h2), b2 = emo_quad_bcast<2>(h2), b3 = emo_quad_bcast<3>(h2);
    const uint32_t w0 = hi_word ? b0 : a0, w1 = hi_word ? b1 : a1, w2 = hi_word ? b2 : a2, w3 = hi_word ? b3 : a3;
    const uint32_t sh = hi_half ? 16u : 0u;
    m[0] = ((w0 >> sh) & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
    m[1] = ((w1 >> sh) & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
    m[2] = ((w2 >> sh) & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
    m[3] = ((w3 >> sh) & 0xFFFFu) >= d.thr16 ? d.scale : 0.f;
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Reductions over the FOUR 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane: two VALU lane swaps
// (v_permlane32_swap(u, u) = {[lo, lo], [hi, hi]}, v_permlane16_swap(u, u) = {[r0, r0, r2, r2], [r1, r1, r3, r3]}) instead of two
// ds_bpermute round trips through the LDS crossbar (r03: the shuffles of the FAVOR+ feature offsets serialised a 700-cycle chain per tile).
__device__ __forceinline__ float rows4_sum(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __builtin_bit_cast(float, (uint32_t)a[0]) + __builtin_bit_cast(float, (uint32_t)a[1]);
    const uint32_t w = __builtin_bit_cast(uint32_t, y);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    return __builtin_bit_cast(float, (uint32_t)b[0]) + __builtin_bit_cast(float, (uint32_t)b[1]);
}
__device__ __forceinline__ float rows4_max(float x) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = fmaxf(__builtin_bit_cast(float, (uint32_t)a[0]), __builtin_bit_cast(float, (uint32_t)a[1]));
    const uint32_t w = __builtin_bit_cast(uint32_t, y);
    const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)b[0]), __builtin_bit_cast(float, (uint32_t)b[1]));
}

__device__ __forceinline__ float gelu_new_f(float x) {
    const float c = 0.7978845608028654f;  // sqrt(2/pi)
    return 0.5f * x * (1.f + tanhf(c * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float dgelu_new_f(float x) {
    const float c = 0.7978845608028654f;
    float u = c * (x + 0.044715f * x * x * x);
    float t = tanhf(u);
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * c * (1.f + 3.f * 0.044715f * x * x);
}

// gelu_new / its derivative through tanh(u) = 1 - 2 / (1 + e^(2u)) with the hardware exp2 / rcp (saturates correctly at both ends): 8 / 14 VALU
// instructions per value where the libm tanhf of gelu_new_f / dgelu_new_f costs several dozen — inside a GEMM epilogue that was 3 x the tile's
// MFMA time (GPT-2 MLP: 236 us for the 69-GFLOP dgrad).  bf16 outputs only: the difference to tanhf (~1e-6) is far below the output rounding.
__device__ __forceinline__ float gelu_new_fast(float x) {
    const float x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(x * fmaf(x2, 2.f * 0.7978845608028654f * 0.044715f * 1.4426950408889634f, 2.f * 0.7978845608028654f * 1.4426950408889634f));
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    return fmaf(x, -r, x);                                        // x (1 - r) = 0.5 x (1 + tanh u)
}
__device__ __forceinline__ float dgelu_new_fast(float x) {
    const float x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(x * fmaf(x2, 2.f * 0.7978845608028654f * 0.044715f * 1.4426950408889634f, 2.f * 0.7978845608028654f * 1.4426950408889634f));
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float up = fmaf(x2, 3.f * 0.044715f * 0.7978845608028654f, 0.7978845608028654f);    // du/dx
    return (1.f - r) * fmaf(x * r * up, 2.f, 1.f);               // 0.5 (1 + t) + 0.5 x (1 - t^2) u',  1 - t^2 = 4 r (1 - r)
}

// bf16 outputs use the fast forms in EVERY kernel (a sequence's result must not depend on which GEMM kernel its batch size selects);
// fp32 outputs (parity mode) keep the libm tanhf.
// exact GELU (F.gelu: x Phi(x)) and its derivative Phi(x) + x phi(x) — the Performer stack's activation='gelu' (upstream TransformerEncoderLayer);
// libm erff / expf: this activation only runs on the generic epilogues
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float dgelu_erf_f(float x) { return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * expf(-0.5f * x * x); }
template <typename OutT> __device__ __forceinline__ float gelu_new_o(float x) { if constexpr (sizeof(OutT) == 2) return gelu_new_fast(x); else return gelu_new_f(x); }
template <typename OutT> __device__ __forceinline__ float dgelu_new_o(float x) { if constexpr (sizeof(OutT) == 2) return dgelu_new_fast(x); else return dgelu_new_f(x); }

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
