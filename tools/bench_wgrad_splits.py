#!/usr/bin/env python3
"""Sweep the split-K count of the wgrad (TN, fp32 accumulate) GEMMs: M tokens in {8192, 32768, 131072}.  Prints ms per shape and split count."""
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

d, f = 512, 2048
bf = torch.bfloat16
def rnd(*s): return torch.randn(*s, device='cuda').to(bf)
for M in [int(x) for x in os.environ.get('MS', '8192,32768,131072').split(',')]:
    for name, (n, k) in {'qkv': (3*d, d), 'out': (d, d), 'ffn1': (f, d), 'ffn2': (d, f)}.items():
        a, dy = rnd(M, k), rnd(M, n)
        dw = torch.zeros(n, k, device='cuda')
        row = {}
        for sp in ['auto', 1, 2, 4, 8, 16, 24, 32, 64]:
            if sp == 'auto': os.environ.pop('EMO_GEMM_SPLITS', None)
            else: os.environ['EMO_GEMM_SPLITS'] = str(sp)
            row[sp] = round(t(lambda: ops.gemm(dy, a, a_trans=True, b_trans=True, out=dw, accumulate=True)) * 1e3, 1)
        print(M, name, 'tiles', ((n + 127) // 128) * ((k + 127) // 128), 'us:', json.dumps(row), flush=True)
