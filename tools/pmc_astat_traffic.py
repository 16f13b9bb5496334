#!/usr/bin/env python3
"""rocprofv3 --pmc driver for the HBM traffic of the A-stationary GEMM (FFN1 forward shape, the largest K = 512 product of the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
M, K, N = 131072, 512, 2048
bf = torch.bfloat16
As = [torch.randn(M, K, device='cuda').to(bf) for _ in range(3)]
w = (torch.randn(N, K, device='cuda') * 0.05).to(bf)
b = torch.randn(N, device='cuda')
os_ = [torch.empty(M, N, device='cuda', dtype=bf) for _ in range(3)]
for i in range(6):
    ops.gemm(As[i % 3], w, out=os_[i % 3], bias=b, act=ops.ACT_RELU, p_drop=0.1, seed=1, offset=2)
torch.cuda.synchronize()
