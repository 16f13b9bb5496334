#!/usr/bin/env python3
"""FAVOR+ attention micro-benchmark at the bench shape: ms per forward / backward call (warm, 20 iterations, HIP events)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops

def t(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

res = {}
for B in [int(x) for x in os.environ.get('BS', '64,4').split(',')]:
    T, H, dh, F = 2048, 8, 64, 128
    HD = H * dh
    qkv = (torch.randn(B * T, 3 * HD, device='cuda') * 0.8).to(torch.bfloat16)
    om = torch.randn(dh, F // 2, device='cuda')
    dout = torch.randn(B * T, HD, device='cuda').to(torch.bfloat16)
    q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
    out, den = ops.favor_attn_fwd(q, k, v, om, B, T, H)
    res['B%d_fwd_ms' % B] = round(t(lambda: ops.favor_attn_fwd(q, k, v, om, B, T, H)), 4)
    res['B%d_bwd_ms' % B] = round(t(lambda: ops.favor_attn_bwd(q, k, v, om, out, dout, den, B, T, H)), 4)
print(json.dumps(res))
