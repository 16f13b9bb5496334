#!/usr/bin/env python3
"""K = 512 products at small token counts (M = 8192: the reference YAML's batch size 4): A-stationary kernel with the column split
(EMO_ASTAT_SPLIT=s forces s column blocks) against the 128 x 128 tiling (EMO_GEMM_NO_ASTAT=1).  HIP-event timing of ITER back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from emo_disentanger_amd import ops  # noqa: E402

ITER = int(os.environ.get('ITER', 50))
M = int(os.environ.get('M', 8192))
bf = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s, sc=0.5: (torch.randn(*s, device='cuda', generator=g) * sc).to(bf)


def timed(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(ITER):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / ITER * 1e3)
    return best


for name, N, spec in (('QKV fwd bias', 1536, dict(bias=True)), ('out-proj fwd bias+drop+res', 512, dict(bias=True, p_drop=0.1, seed=3, offset=1, res=True)),
                      ('FFN1 fwd relu+drop+mask', 2048, dict(bias=True, act=ops.ACT_RELU, p_drop=0.1, seed=3, offset=2, mask=True)),
                      ('FFN2 dgrad bits', 2048, dict(bits=True)), ('out-proj dgrad', 512, {})):
    a, w = rnd(M, 512), rnd(N, 512, sc=0.05)
    o = torch.empty(M, N, device='cuda', dtype=bf)
    kw = {}
    if spec.get('bias'):
        kw['bias'] = torch.randn(N, device='cuda', generator=g)
    if spec.get('res'):
        kw['residual'] = rnd(M, N)
    for key in ('p_drop', 'seed', 'offset', 'act'):
        if key in spec:
            kw[key] = spec[key]
    mask = torch.zeros(M, N // 8, device='cuda', dtype=torch.uint8)
    row = []
    for split in os.environ.get('SPLITS', '0,1,2,4,8,16').split(','):
        k2 = dict(kw)
        os.environ.pop('EMO_GEMM_NO_ASTAT', None)
        os.environ.pop('EMO_ASTAT_SPLIT', None)
        if split == '0':
            os.environ['EMO_GEMM_NO_ASTAT'] = '1'
            if spec.get('bits'):
                k2.update(mul_aux=rnd(M, N), mul_mode=ops.MUL_NONZERO, mul_scale=1.1)
        else:
            if (N // 64) % int(split):
                continue
            os.environ['EMO_ASTAT_SPLIT'] = split
            if spec.get('mask'):
                k2['mask_out'] = mask
            if spec.get('bits'):
                k2.update(mul_aux=mask, mul_mode=ops.MUL_BITMASK, mul_scale=1.1)
        t = timed(lambda: ops.gemm(a, w, out=o, **k2))
        row.append('%s: %.1f us (kernel %d)' % ('tiled' if split == '0' else 'split %s' % split, t, ops.lib.emo_gemm_last_kernel()))
    print('%-28s N=%4d  %s' % (name, N, '  '.join(row)), flush=True)
