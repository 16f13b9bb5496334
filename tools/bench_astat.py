#!/usr/bin/env python3
"""K = 512 GEMMs of one Performer layer at the bench shape (131072 tokens): A-stationary kernel vs the 128^2 tiled kernel, operands rotated
over several buffers (cold-ish), HIP events.  usage: python tools/bench_astat.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emo_disentanger_amd import ops  # noqa: E402

M, K = 131072, 512
dev = 'cuda'
shapes = [('QKV fwd (bias)', 1536, dict(bias=True)), ('out fwd (bias+drop+res)', 512, dict(bias=True, p_drop=0.1, residual=True)),
          ('FFN1 fwd (bias+relu+drop)', 2048, dict(bias=True, act=ops.ACT_RELU, p_drop=0.1)), ('FFN2 dgrad (mask)', 2048, dict(mask=True)), ('FFN2 dgrad (bitmask)', 2048, dict(bitmask=True)), ('FFN1 fwd (+mask_out)', 2048, dict(bias=True, act=ops.ACT_RELU, p_drop=0.1, mask_out=True)),
          ('out dgrad (plain)', 512, dict())]
NB = 3
As = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NB)]
for name, N, e in shapes:
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev) if e.get('bias') else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if e.get('residual') else None
    mask = (torch.rand(M, N, device=dev) > 0.5).to(torch.bfloat16) if e.get('mask') else None
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    kw = dict(bias=bias, act=e.get('act', ops.ACT_NONE), p_drop=e.get('p_drop', 0.0), seed=1, offset=2, residual=res)
    if mask is not None:
        kw.update(mul_aux=mask, mul_mode=ops.MUL_NONZERO, mul_scale=1.1)
    if e.get('bitmask'):
        kw.update(mul_aux=(torch.rand(M, N // 8, device=dev) * 255).to(torch.uint8), mul_mode=ops.MUL_BITMASK, mul_scale=1.1)
    if e.get('mask_out'):
        kw.update(mask_out=torch.empty(M, N // 8, device=dev, dtype=torch.uint8))
    line = '%-28s N=%4d ' % (name, N)
    for mode in (('astat',) if (e.get('bitmask') or e.get('mask_out')) else ('astat', 'tiled')):
        if mode == 'tiled':
            os.environ['EMO_GEMM_NO_ASTAT'] = '1'
        else:
            os.environ.pop('EMO_GEMM_NO_ASTAT', None)
        for i in range(3):
            ops.gemm(As[i % NB], W, out=outs[i % NB], **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        it = 12
        for i in range(it):
            ops.gemm(As[i % NB], W, out=outs[i % NB], **kw)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / it
        line += ' %s %.1f us = %.0f TFLOP/s |' % (mode, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
    print(line, flush=True)
