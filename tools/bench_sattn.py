#!/usr/bin/env python3
"""GPT-2 causal softmax attention micro-benchmark at the gpt2 bench shape (B=16, T=2048, H=8, dh=64, bf16): ms and TFLOP/s of the forward
and backward calls with attention-probability dropout off / on (HIP events, warm)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops

def t(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

B, T, H, dh = int(os.environ.get('B', 16)), int(os.environ.get('T', 2048)), 8, 64
HD = H * dh
qkv = (torch.randn(B * T, 3 * HD, device='cuda') * 0.8).to(torch.bfloat16)
dout = torch.randn(B * T, HD, device='cuda').to(torch.bfloat16)
q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
res = {}
for p in (0.0, 0.1):
    out, lse, keep = ops.softmax_attn_fwd(q, k, v, B, T, H, p_drop=p, seed=5, offset=64, want_keep=True)     # (the training path: keep bits for the backward)
    f = t(lambda: ops.softmax_attn_fwd(q, k, v, B, T, H, p_drop=p, seed=5, offset=64, want_keep=True))
    b = t(lambda: ops.softmax_attn_bwd(q, k, v, out, dout, lse, B, T, H, p_drop=p, seed=5, offset=64, keep=keep))
    fl = 0.5 * 2 * 2.0 * B * H * T * T * dh
    res['p%.1f' % p] = {'fwd_ms': round(f, 4), 'fwd_tflops': round(fl / f / 1e9, 1), 'bwd_ms': round(b, 4), 'bwd_tflops': round(3.5 * fl / b / 1e9, 1)}
print(json.dumps(res))
