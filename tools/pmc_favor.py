#!/usr/bin/env python3
"""Driver for rocprofv3 --pmc passes over the FAVOR+ attention kernels at the bench shape (B=64, T=2048, H=8, dh=64, F=128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
B, T, H, dh, F = int(os.environ.get('B', 64)), 2048, 8, 64, 128
bf = torch.bfloat16
HD = H * dh
qkv = (torch.randn(B * T, 3 * HD, device='cuda') * 0.8).to(bf)
om = torch.randn(dh, F // 2, device='cuda')
dout = torch.randn(B * T, HD, device='cuda').to(bf)
for _ in range(2):
    out, den = ops.favor_attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], om, B, T, H)
    ops.favor_attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], om, out, dout, den, B, T, H)
torch.cuda.synchronize()
