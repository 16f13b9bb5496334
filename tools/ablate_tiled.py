#!/usr/bin/env python3
"""-DEMO_DIAG build only: epilogue ablations of the 128^2 tiled kernel on the step's K >= 1536 GEMMs (16 no stores, 32 no residual loads, 64 no epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
M = 131072
bf = torch.bfloat16
f = [torch.randn(M, 2048, device='cuda').to(bf) for _ in range(2)]
W2 = (torch.randn(512, 2048, device='cuda') * 0.05).to(bf)
W1 = (torch.randn(2048, 512, device='cuda') * 0.05).to(bf)
res = torch.randn(M, 512, device='cuda').to(bf)
b = torch.randn(512, device='cuda')
outs = [torch.empty(M, 512, device='cuda', dtype=bf) for _ in range(2)]
def t(fn):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): fn(i)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for ab in ('0', '16', '32', '64'):
    os.environ['EMO_GEMM_ABLATE'] = ab
    a = t(lambda i: ops.gemm(f[i % 2], W2, out=outs[i % 2], bias=b, p_drop=0.1, seed=1, offset=2, residual=res))
    c = t(lambda i: ops.gemm(f[i % 2], W1, b_trans=True, out=outs[i % 2], residual=res))
    print('ablate %-3s FFN2 fwd NT K=2048 %.1f us   FFN1 dgrad NN K=2048 %.1f us' % (ab, a, c))
