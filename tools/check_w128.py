#!/usr/bin/env python3
"""256 x 256 tile kernel (EMO_GEMM_W128=1) vs the default path and a float64 reference: shapes x epilogues (bias, dropout, residual),
run-to-run bitwise stability, same dropout mask as the default path."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emo_disentanger_amd import ops
torch.manual_seed(0)
bad = 0
for (M, N, K) in ((256, 256, 256), (512, 256, 320), (1024, 512, 2048), (2048, 768, 1024), (256, 1024, 4096), (4096, 512, 1536)):
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    for name, kw in (('plain', {}), ('bias', dict(bias=bias)), ('bias+drop+res', dict(bias=bias, p_drop=0.1, seed=5, offset=7, residual=res)), ('res', dict(residual=res))):
        os.environ['EMO_GEMM_W128'] = '1'
        y1 = ops.gemm(A, W, **kw)
        y2 = ops.gemm(A, W, **kw)
        os.environ['EMO_GEMM_W128'] = '0'
        y0 = ops.gemm(A, W, **kw)
        ref = A.double() @ W.double().t()
        if 'bias' in kw:
            ref = ref + bias.double()
        if 'p_drop' not in kw:
            if 'residual' in kw:
                ref = ref + res.double()
            e1 = ((y1.double() - ref).abs().max() / ref.abs().max()).item()
        else:
            e1 = 0.0
        d01 = (y1.float() - y0.float()).abs().max().item()
        same = torch.equal(y1, y2)
        ok = e1 < 6e-3 and same and d01 < 0.07
        print('M=%d N=%d K=%d %-14s rel err %.2e  max|w128 - default| %.3g  stable %s %s' % (M, N, K, name, e1, d01, same, '' if ok else '<-- BAD'))
        bad += not ok
print('FAIL' if bad else 'OK')
