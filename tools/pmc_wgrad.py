#!/usr/bin/env python3
"""Driver for rocprofv3 --pmc passes over the dominant kernel's launch mix (wgrad dW = dY^T X, per-layer shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
M, d, f = 131072, 512, 2048
bf = torch.bfloat16
for rep in range(3):
    for n, k in ((3 * d, d), (d, d), (f, d), (d, f)):
        x = torch.randn(M, k, device='cuda').to(bf)
        dy = torch.randn(M, n, device='cuda').to(bf)
        dw = torch.zeros(n, k, device='cuda')
        ops.gemm(dy, x, a_trans=True, b_trans=True, out=dw, accumulate=True)
        torch.cuda.synchronize()
        del x, dy, dw
