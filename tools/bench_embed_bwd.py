#!/usr/bin/env python3
"""Embedding backward at the bench shape (64 x 2048 tokens, d 512, V 327 + 2 segment rows): time vs EMO_EMBED_RPB (token rows per block)
and vs the token distribution (uniform ids / one hot id)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
B, T, D, V = 64, 2048, 512, 327
dout = torch.randn(B, T, D, device='cuda').to(torch.bfloat16)
seg = torch.randint(0, 2, (B, T), device='cuda')
def t(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for dist in ('uniform', 'onehot'):
    tok = torch.randint(0, V - 1, (B, T), device='cuda') if dist == 'uniform' else torch.full((B, T), 5, device='cuda')
    for rpb in (1024, 2048, 4096, 16384, 65536):
        os.environ['EMO_EMBED_RPB'] = str(rpb)
        dE, dS = torch.zeros(V, D, device='cuda'), torch.zeros(2, D, device='cuda')
        us = t(lambda: ops.embed_bwd(tok, seg, dout, dE, dS, 22.6, 0.1, 1, 2))
        us_noseg = t(lambda: ops.embed_bwd(tok, None, dout, dE, None, 22.6, 0.1, 1, 2))
        print(dist, 'rpb', rpb, 'us', round(us, 1), 'without segment rows', round(us_noseg, 1), flush=True)
