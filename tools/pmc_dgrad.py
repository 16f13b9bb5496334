#!/usr/bin/env python3
"""Driver for rocprofv3 --pmc passes over the dgrad launch mix (dX = dY W, NN layout; per-layer shapes QKV / out / FFN1 / FFN2 at M = 131072)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
M, d, f = 131072, 512, 2048
bf = torch.bfloat16
for rep in range(3):
    for n, k in ((3 * d, d), (d, d), (f, d), (d, f)):      # W [n, k]: dX [M, k] = dY [M, n] @ W
        dy = torch.randn(M, n, device='cuda').to(bf)
        w = torch.randn(n, k, device='cuda').to(bf)
        dx = torch.empty(M, k, device='cuda', dtype=bf)
        ops.gemm(dy, w, b_trans=True, out=dx)
        torch.cuda.synchronize()
        del dy, w, dx
