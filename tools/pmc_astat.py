#!/usr/bin/env python3
"""rocprofv3 --pmc driver: a few launches of the A-stationary K = 512 GEMM (QKV forward shape) and of the tiled kernel on the same operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
M, K, N = 131072, 512, 1536
bf = torch.bfloat16
a = torch.randn(M, K, device='cuda').to(bf)
w = (torch.randn(N, K, device='cuda') * 0.05).to(bf)
b = torch.randn(N, device='cuda')
o = torch.empty(M, N, device='cuda', dtype=bf)
for _ in range(4):
    ops.gemm(a, w, out=o, bias=b)
os.environ['EMO_GEMM_NO_ASTAT'] = '1'
for _ in range(4):
    ops.gemm(a, w, out=o, bias=b)
torch.cuda.synchronize()
