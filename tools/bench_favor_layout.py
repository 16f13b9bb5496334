#!/usr/bin/env python3
"""Does the q/k/v row layout bound the FAVOR+ kernels?  Same 512 scans of 2048 tokens, (a) the model's layout: 64 sequences x 8 heads in one
fused [B*T, 3*512] buffer (each scan reads 128-B pieces at a 3 KB stride), (b) 512 single-head sequences in a [B*T, 3*64] buffer (384-B rows)."""
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
for B, H in ((64, 8), (512, 1)):
    T, dh, F = 2048, 64, 128
    HD = H * dh
    qkv = (torch.randn(B * T, 3 * HD, device='cuda') * 0.8).to(torch.bfloat16)
    om = torch.randn(dh, F // 2, device='cuda')
    dout = torch.randn(B * T, HD, device='cuda').to(torch.bfloat16)
    q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
    out, den = ops.favor_attn_fwd(q, k, v, om, B, T, H)
    res['B%d_H%d_fwd_ms' % (B, H)] = round(t(lambda: ops.favor_attn_fwd(q, k, v, om, B, T, H)), 4)
    res['B%d_H%d_bwd_ms' % (B, H)] = round(t(lambda: ops.favor_attn_bwd(q, k, v, om, out, dout, den, B, T, H)), 4)
print(json.dumps(res))
