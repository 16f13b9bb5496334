#!/usr/bin/env python3
"""Persistent 256 x 256 GEMM (csrc/emo_gemm_p256.hip, EMO_GEMM_P256=1) against the kernels it replaces (EMO_GEMM_P256=0: 256 x 256 tile per
block for K >= 1024, A-stationary for K = 512) on the training step's shapes: same-process A/B, alternating order, HIP-event timing of ITER
back-to-back launches, and the largest element difference between the two (same arithmetic up to the summation order inside a dot product)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from emo_disentanger_amd import ops  # noqa: E402

ITER = int(os.environ.get('ITER', 20))
M = int(os.environ.get('M', 131072))
bf = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s, sc=0.5: (torch.randn(*s, device='cuda', generator=g) * sc).to(bf)
SHAPES = [('FFN2 fwd  K=2048 bias+drop+res', 512, 2048, dict(bias=True, p_drop=0.1, seed=3, offset=11, res=True)),
          ('FFN1 dgrad K=2048 res', 512, 2048, dict(res=True)),
          ('QKV dgrad K=1536 res', 512, 1536, dict(res=True)),
          ('plainH K=2048', 512, 2048, {}),
          ('plain K=2048 A stride 0 (from L2)', 512, 2048, dict(l2=True)),
          ('FFN2 fwd  K=2048 b+d+r A stride 0', 512, 2048, dict(bias=True, p_drop=0.1, seed=3, offset=11, res=True, l2=True)),
          ('QKV fwd   K=512 N=1536 bias', 1536, 512, dict(bias=True)),
          ('out-proj  K=512 N=512 bias+drop+res', 512, 512, dict(bias=True, p_drop=0.1, seed=3, offset=12, res=True)),
          ('plain K=512 N=2048', 2048, 512, {})]
if os.environ.get('ONLY'):
    SHAPES = [s for s in SHAPES if os.environ['ONLY'] in s[0]]


VARIANT = os.environ.get('VARIANT', 'p256')                    # p256: the 256 x 256 persistent kernel; q512: the 128 x 512 ("full N") tile


def run(mode, a, w, o, kw):
    os.environ['EMO_GEMM_P256'] = mode if VARIANT == 'p256' else '0'
    os.environ['EMO_GEMM_Q512'] = mode if VARIANT == 'q512' else '0'
    return ops.gemm(a, w, out=o, **kw)


for name, n, k, spec in SHAPES:
    a, w = rnd(M, k), rnd(n, k, sc=0.05)
    if spec.get('l2'):
        a = a[:1].expand(M, k)
    kw = {}
    if spec.get('bias'):
        kw['bias'] = torch.randn(n, device='cuda', generator=g)
    if spec.get('res'):
        kw['residual'] = rnd(M, n)
    for key in ('p_drop', 'seed', 'offset'):
        if key in spec:
            kw[key] = spec[key]
    o1, o0 = torch.empty(M, n, device='cuda', dtype=bf), torch.empty(M, n, device='cuda', dtype=bf)
    run('1', a, w, o1, kw)
    k1 = ops.lib.emo_last_gemm_kernel() if hasattr(ops.lib, 'emo_last_gemm_kernel') else -1
    run('0', a, w, o0, kw)
    k0 = ops.lib.emo_last_gemm_kernel() if hasattr(ops.lib, 'emo_last_gemm_kernel') else -1
    torch.cuda.synchronize()
    ref = a[:512].double() @ w.double().t()
    diff = float((o1.float() - o0.float()).abs().max())
    scale = float(o0.float().abs().max())
    res = {}
    for rep in range(3):
        for mode in (('1', '0') if rep % 2 == 0 else ('0', '1')):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(mode, a, w, o1 if mode == '1' else o0, kw)
            e0.record()
            for _ in range(ITER):
                run(mode, a, w, o1 if mode == '1' else o0, kw)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / ITER * 1e3)
    fl = 2.0 * M * n * k
    t1, t0 = min(res['1']), min(res['0'])
    print('%-38s new  (kernel %d) %7.1f us = %6.1f TFLOP/s = %.3f of 2.5 PF | before (kernel %d) %7.1f us = %6.1f | max |diff| %.3g of %.3g %s' %
          (name, k1, t1, fl / t1 / 1e6, fl / t1 / 1e6 / 2500, k0, t0, fl / t0 / 1e6, diff, scale, '' if diff <= 0.02 * scale else '<-- MISMATCH'), flush=True)
os.environ.pop('EMO_GEMM_P256', None)
os.environ.pop('EMO_GEMM_Q512', None)
