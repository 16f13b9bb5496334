#!/usr/bin/env python3
"""Statistics of the counter-based dropout stream (csrc/emo_common.h: emo_drop_hash; r04 dropped the multiply in front of the first round —
ADVICE r04): keep rate, serial correlation inside a stream (lag 1, lag 4 = one hash word, lag = a row stride), correlation between the streams of
two SITES of one step (adjacent per-site offsets, the engine's base + 8 (l + 1) + {1, 2, 3}) and of two consecutive steps (base + 4096), and
avalanche of the 32-bit hash over single-bit flips of the counter.  The keep decisions are read out of the library itself (emo_dropout_apply on
ones: what every fused kernel regenerates); the avalanche test restates the un-keyed two-round core emo_hash32 in NumPy.  Everything should sit at the sampling-noise level
1 / sqrt(n).   usage: python tools/drop_hash_stats.py [log2_samples=22]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
U32 = np.uint32
M32 = np.uint64(0xFFFFFFFF)


def hash32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def keep_bits_from_library(n, p, seed, offset):
    import torch
    from emo_disentanger_amd import ops
    return (ops.dropout_apply(torch.ones(n, device='cuda'), p, seed, offset) != 0).cpu().numpy()


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n, p, seed = 1 << lg, 0.1, 12345
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:   # noqa: BLE001
        has_gpu = False
    if not has_gpu:
        print('no GPU: the statistics below need the library (the stream is defined by emo_drop_hash in libemo_hip.so); nothing measured')
        return
    noise = 1.0 / np.sqrt(n)
    base = 4096
    sites = {'L0.attn_out': base + 8 + 1, 'L0.ffn_hidden': base + 8 + 2, 'L0.ffn_out': base + 8 + 3, 'L1.attn_out': base + 16 + 1, 'next step L0.attn_out': 2 * base + 8 + 1}
    keep = {k: keep_bits_from_library(n, p, seed, off).astype(np.float64) for k, off in sites.items()}
    print('samples per stream 2^%d, sampling noise 1/sqrt(n) = %.2e' % (lg, noise))
    for k, v in keep.items():
        print('keep rate %-24s %.5f (target %.5f, thr16 = round(p 65536): %.5f)' % (k, v.mean(), 1 - p, 1 - round(p * 65536) / 65536))
    a = keep['L0.attn_out'] - keep['L0.attn_out'].mean()
    for lag in (1, 2, 4, 8, 512, 2048):
        print('serial correlation lag %-5d %+.2e' % (lag, float((a[:-lag] * a[lag:]).mean() / a.var())))
    names = list(keep)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            x, y = keep[names[i]] - keep[names[i]].mean(), keep[names[j]] - keep[names[j]].mean()
            print('cross-stream correlation %-24s x %-24s %+.2e' % (names[i], names[j], float((x * y).mean() / np.sqrt(x.var() * y.var()))))
    worst = max(abs(float(((keep[names[i]] - keep[names[i]].mean()) * (keep[names[j]] - keep[names[j]].mean())).mean() / keep[names[i]].var()))
                for i in range(len(names)) for j in range(i + 1, len(names)))
    print('largest |cross-stream correlation| %.2e = %.1f x the sampling noise' % (worst, worst / noise))
    # avalanche of the two multiply-xorshift rounds (the un-keyed core): flipping one counter bit flips each output bit with probability 1/2
    rng = np.random.default_rng(0)
    x = rng.integers(0, 1 << 32, size=1 << 18, dtype=np.uint64)
    h0 = hash32(x)
    dev = 0.0
    for b in range(32):
        d = h0 ^ hash32(x ^ np.uint64(1 << b))
        bits = ((d[:, None] >> np.arange(32, dtype=np.uint64)[None, :]) & np.uint64(1)).mean(0)
        dev = max(dev, float(np.abs(bits - 0.5).max()))
    print('avalanche: largest |P(output bit flips | one input bit flips) - 1/2| = %.4f (noise %.4f)' % (dev, 0.5 / np.sqrt(len(x))))


if __name__ == '__main__':
    main()
