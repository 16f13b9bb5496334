#!/usr/bin/env python3
"""HBM-bound kernels at the bench shape (M = 131072 tokens, D = 512, bf16): us per launch and effective TB/s."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops

def t(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

M, D = int(os.environ.get('M', 131072)), 512
bf = torch.bfloat16
x = torch.randn(M, D, device='cuda').to(bf); dy = torch.randn(M, D, device='cuda').to(bf); dres = torch.randn(M, D, device='cuda').to(bf)
g, b = torch.ones(D, device='cuda'), torch.zeros(D, device='cuda')
dg, db, dc = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
y, mean, rstd = ops.layernorm_fwd(x, g, b)
res = {}
us = t(lambda: ops.layernorm_fwd(x, g, b)); res['ln_fwd'] = (round(us, 1), round(2 * M * D * 2 / us / 1e6, 2))
us = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db)); res['ln_bwd_plain'] = (round(us, 1), round(3 * M * D * 2 / us / 1e6, 2))
us = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres, want_drop=True, p_drop=0.1, seed=1, offset=2, dcol=dc))
res['ln_bwd_res_drop_dcol'] = (round(us, 1), round(5 * M * D * 2 / us / 1e6, 2))
us = t(lambda: ops.colsum(dy, out=dc, accumulate=True)); res['colsum_512'] = (round(us, 1), round(M * D * 2 / us / 1e6, 2))
print(json.dumps(res))
