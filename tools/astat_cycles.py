#!/usr/bin/env python3
"""-DEMO_DIAG build only: where an A-stationary wave spends its cycles (s_memtime): mid-stage wait + barrier, epilogue, whole column sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
os.environ['EMO_GEMM_ABLATE'] = '8'
M, K = 131072, 512
for name, N, kw in (('QKV', 1536, {}), ('FFN1', 2048, dict(act=ops.ACT_RELU, p_drop=0.1, seed=1, offset=2)), ('FFN1+mask', 2048, dict(act=ops.ACT_RELU, p_drop=0.1, seed=1, offset=2, mask=True)), ('FFN1 plain', 2048, {}), ('out dgrad', 512, {}), ('out fwd drop+res', 512, dict(p_drop=0.1, seed=1, offset=3, res=True))):
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda')
    o = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    diag = torch.zeros(8, device='cuda', dtype=torch.int64)
    if kw.pop('res', False):
        kw['residual'] = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    if kw.pop('mask', False):
        kw['mask_out'] = torch.empty(M, N // 8, device='cuda', dtype=torch.uint8)
    ops.gemm(a, w, out=o, bias=b, **kw)
    diag.zero_()
    ops.gemm(a, w, out=o, bias=b, rln=(None, diag.view(torch.float32), None, None), **kw)
    torch.cuda.synchronize()
    d = diag.tolist()
    waves = d[3]
    print('%-10s N=%4d waves %d  loop %.0f cyc/wave  wait+barrier %.1f %%  epilogue %.1f %%  per stage: loop %.0f wait %.0f, per tile epilogue %.0f | before the loop (panel load) %.0f cyc' %
          (name, N, waves, d[2] / waves, 100 * d[0] / d[2], 100 * d[1] / d[2], d[2] / waves / (N / 16), d[0] / waves / (N / 16), d[1] / waves / (N / 64), d[4] / waves))
