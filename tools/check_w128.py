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
os.environ.pop('EMO_GEMM_W128', None)
# wgrad layout: dW[M,N] (+)= dY[K,M]^T X[K,N], fp32, with / without the bias gradient and accumulation
for (Kt, M, N) in ((8192, 2048, 512), (16384, 512, 2048), (8192, 1536, 512), (131072, 2048, 512)):
    dY = torch.randn(Kt, M, device='cuda').to(torch.bfloat16)
    X = torch.randn(Kt, N, device='cuda').to(torch.bfloat16)
    for acc in (False, True):
        outs = []
        for mode in ('1', '0'):
            os.environ['EMO_GEMM_W128_TN'] = mode
            dW = torch.full((M, N), 0.5, device='cuda')
            db = torch.full((M,), 0.25, device='cuda')
            ops.gemm(dY, X, a_trans=True, b_trans=True, out=dW, accumulate=acc, a_rowsum=db)
            dW2 = torch.full((M, N), 0.5, device='cuda')
            ops.gemm(dY, X, a_trans=True, b_trans=True, out=dW2, accumulate=acc)
            outs.append((dW, db, dW2))
        os.environ.pop('EMO_GEMM_W128_TN')
        ref = dY[:, :256].double().t() @ X.double() + (0.5 if acc else 0.0)
        e_ref = ((outs[0][0][:256].double() - ref).abs().max() / ref.abs().max()).item()
        d_w = ((outs[0][0] - outs[1][0]).abs().max() / outs[1][0].abs().max()).item()
        d_b = ((outs[0][1] - outs[1][1]).abs().max() / outs[1][1].abs().max()).item()
        same = torch.equal(outs[0][0], outs[0][2])
        ok = e_ref < 2e-5 and d_w < 2e-5 and d_b < 1e-4 and same
        print('TN K=%d M=%d N=%d acc=%d  vs fp64 %.2e  vs 128^2 kernel %.2e  bias grad %.2e  with/without rowsum equal %s %s' % (Kt, M, N, acc, e_ref, d_w, d_b, same, '' if ok else '<-- BAD'))
        bad += not ok
print('FAIL' if bad else 'OK')
