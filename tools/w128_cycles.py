#!/usr/bin/env python3
"""-DEMO_DIAG build only (make EXTRA=-DEMO_DIAG -B): where a wave of the 256 x 256 tile kernel spends its cycles (s_memtime): prologue, K loop,
the mid-tile sync (lgkmcnt + counted vmcnt + s_barrier) inside it, epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
os.environ['EMO_GEMM_ABLATE'] = '8'
M = 131072
for name, N, K, kw in (('FFN2 fwd', 512, 2048, dict(p_drop=0.1, seed=1, offset=2, res=True)), ('QKV dgrad', 512, 1536, dict(res=True)), ('plain K=4096', 512, 4096, {})):
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda').to(torch.bfloat16) if kw.pop('res', False) else None
    o = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    diag = torch.zeros(8, device='cuda', dtype=torch.int64)
    for _ in range(3):
        ops.gemm(a, w, out=o, bias=b, residual=r, **kw)
    diag.zero_()
    ops.gemm(a, w, out=o, bias=b, residual=r, rln=(None, diag.view(torch.float32), None, None), **kw)
    torch.cuda.synchronize()
    d = diag.tolist()
    n = d[4]
    tot = d[0] + d[1] + d[3]
    print('%-14s waves %d  per wave: prologue %.0f  loop %.0f (sync %.0f = %.1f %% of loop; %.0f cyc per K-tile, of it sync %.0f)  epilogue %.0f  => loop share %.1f %%' %
          (name, n, d[0] / n, d[1] / n, d[2] / n, 100 * d[2] / d[1], d[1] / n / (K / 64), d[2] / n / (K / 64), d[3] / n, 100 * d[1] / tot))
