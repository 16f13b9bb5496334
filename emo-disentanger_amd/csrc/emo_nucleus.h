// Nucleus (top-p) draw of ONE stream by 512 threads — shared by nucleus_kernel (emo_elementwise.hip) and the one-launch decode step
// (emo_decode_persist.hip), which must pick the same token from the same logits bit for bit.  Reference: stage2_accompaniment/inference.py:71-100.
// probs = softmax(l/temp) (fp32, as NumPy on fp32 logits); rank sort (descending, ties by ascending index) of <= 1024 entries in LDS; inclusive
// cumsum in np.cumsum's sequential fp32 order; last_index = SECOND position whose cumsum exceeds top_p (the reference keeps the crossing token —
// SURVEY F12); where the reference would raise IndexError (single crossing) all sorted tokens are kept.  Draw: cdf over the renormalised (f64)
// candidates, searchsorted(u, right).
#pragma once
#include "emo_common.h"

constexpr int EMO_NUCLEUS_LDS = (3 * (1024 + 8)) * 4 + (1024 + 8) * 8 * 2 + 1024 * 4 + 64;      // bytes of LDS scratch (16-B aligned base)
constexpr int EMO_NUCLEUS_BARRIERS = 9;                                                          // sync() calls of emo_nucleus_draw (every path)

// tid in [0, 512); `sync` = a barrier over exactly the calling threads (+ any others that call it the same number of times).  Returns the picked
// token id in thread 0 (other threads: undefined).  u = the uniform draw of this stream.
template <typename Sync>
__device__ __forceinline__ int64_t emo_nucleus_draw(const float* __restrict__ l, int64_t V, float temp, float top_p, float u, char* lds, int tid, Sync sync) {
    float* sp = (float*)lds;
    float* sq = sp + 1024 + 8;
    float* cumf = sq + 1024 + 8;
    double* cumd = (double*)(cumf + 1024 + 8);
    unsigned long long* skey = (unsigned long long*)(cumd + 1024 + 8);
    int* si = (int*)(skey + 1024 + 8);
    float* red = (float*)(si + 1024);
    int* cnt = (int*)(red + 8);
    const int Vp = ((int)V + 7) & ~7;
    const int lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int c = tid; c < V; c += 512) mx = fmaxf(mx, l[c] / temp);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    sync();
    mx = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
    sync();
    float s = 0.f;
    for (int c = tid; c < V; c += 512) {
        const float e = expf(l[c] / temp - mx);
        sp[c] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    sync();
    const float tot = red[0] + red[1] + red[2] + red[3] + red[4] + red[5] + red[6] + red[7];
    // rank sort on 64-bit keys {probability bits, ~index}: key_j > key_c  <=>  p_j > p_c, or p_j == p_c and j < c (probabilities are >= 0, so their
    // bit patterns order like their values) - the stable descending order of the reference's argsort in ONE compare + ONE add-with-carry per pair
    // (r04: the float version spent ~7 VALU instructions per pair, 336 x 336 pairs on 6 waves = half of the kernel's 20 us).  Pad keys are 0.
    for (int c = tid; c < V; c += 512) {
        const float pc = sp[c] / tot;
        sq[c] = pc;
        skey[c] = ((unsigned long long)__builtin_bit_cast(unsigned, pc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)c);
    }
    for (int c = (int)V + tid; c < Vp + 8; c += 512) { sq[c] = -1.f; skey[c] = 0ull; }
    sync();
    for (int c = tid; c < Vp; c += 512) {
        if (c < V) {
            const float pc = sq[c];
            const unsigned long long kc = skey[c];
            int rank = 0;
            for (int j0 = 0; j0 < Vp; j0 += 2) {
                const u32x4 k2 = *(const u32x4*)(skey + j0);
                const unsigned long long ka = ((unsigned long long)k2[1] << 32) | k2[0], kb = ((unsigned long long)k2[3] << 32) | k2[2];
                rank += (int)(ka > kc) + (int)(kb > kc);
            }
            sp[rank] = pc;
            si[rank] = c;
        } else {
            sp[c] = 0.f;                                                   // sorted tail pad: adds nothing to either prefix
        }
    }
    sync();
    // Both prefixes stay SEQUENTIAL (np.cumsum order: the top-p crossing and the draw are rounding-sensitive), but a chunk of 32 sorted values is
    // fetched into registers before its dependent chain of adds runs (r04: the loops used to pay an LDS round trip per 4-8 values, and the f64 one
    // was ~9 of the kernel's 20 us).
    if (tid == 0) {                        // np.cumsum order, fp32
        float cum = 0.f;
        for (int i0 = 0; i0 < Vp; i0 += 32) {
            f32x4 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (i0 + 4 * j < Vp) ? *(const f32x4*)(sp + i0 + 4 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 ca;
                ca[0] = cum += a[j][0]; ca[1] = cum += a[j][1]; ca[2] = cum += a[j][2]; ca[3] = cum += a[j][3];
                if (i0 + 4 * j < Vp) *(f32x4*)(cumf + i0 + 4 * j) = ca;
            }
        }
    } else if (tid == 64) {                // sequential f64 prefix of the same sorted probabilities (candidate renormalisation + draw)
        double run = 0.0;
        for (int i0 = 0; i0 < Vp; i0 += 32) {
            f32x4 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (i0 + 4 * j < Vp) ? *(const f32x4*)(sp + i0 + 4 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double c0, c1, c2, c3;
                c0 = run += (double)a[j][0]; c1 = run += (double)a[j][1]; c2 = run += (double)a[j][2]; c3 = run += (double)a[j][3];
                if (i0 + 4 * j < Vp) { cumd[i0 + 4 * j] = c0; cumd[i0 + 4 * j + 1] = c1; cumd[i0 + 4 * j + 2] = c2; cumd[i0 + 4 * j + 3] = c3; }
            }
        }
    }
    sync();
    // first crossing i1 = #{i < V : cum_i <= top_p}; cum is non-decreasing, so the second crossing is i1 + 1
    int c1 = 0;
    for (int i = tid; i < V; i += 512) c1 += (cumf[i] <= top_p) ? 1 : 0;
    c1 = (int)wave_sum((float)c1);
    if (lane == 0) cnt[wave] = c1;
    sync();
    const int i1 = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5] + cnt[6] + cnt[7];
    int last;
    if (i1 >= V) last = V < 3 ? (int)V : 3;       // no crossing
    else if (i1 + 1 >= V) last = (int)V;          // single crossing (reference: IndexError)
    else last = i1 + 1;
    const double target = (double)u * cumd[last - 1];
    sync();
    int c2 = 0;
    for (int i = tid; i < last; i += 512) c2 += (cumd[i] <= target) ? 1 : 0;
    c2 = (int)wave_sum((float)c2);
    if (lane == 0) cnt[wave] = c2;
    sync();
    int64_t tok = 0;
    if (tid == 0) {
        int pick = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5] + cnt[6] + cnt[7];
        if (pick >= last) pick = last - 1;
        tok = (int64_t)si[pick];
    }
    return tok;
}
