#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc runs: a few launches of the three GEMM layouts at the perf-config shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
M, d, f = 131072, 512, 2048
bf = torch.bfloat16
rnd = lambda *s: torch.randn(*s, device='cuda').to(bf)
a, w, o = rnd(M, d), rnd(f, d), torch.empty(M, f, device='cuda', dtype=bf)
dy, dw = rnd(M, f), torch.zeros(f, d, device='cuda')
for _ in range(3):
    ops.gemm(a, w, out=o)                                            # fwd NT  (M x 2048 x 512)
    ops.gemm(dy, w, b_trans=True, out=a)                             # dgrad NN (M x 512 x 2048)
    ops.gemm(dy, a, a_trans=True, b_trans=True, out=dw, accumulate=True)   # wgrad TN
torch.cuda.synchronize()
