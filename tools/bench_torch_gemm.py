#!/usr/bin/env python3
"""Yardstick only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the hot-path shapes,
to tell how far the hand-written kernels in csrc/emo_gemm.hip are from a tuned library.  Prints TFLOP/s per shape."""
import os, json
import torch

def t(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

M, d, f = int(os.environ.get('M', 131072)), 512, 2048
bf = torch.bfloat16
def rnd(*s): return torch.randn(*s, device='cuda').to(bf)
res = {}
for name, (m, n, k) in {'qkv': (M, 3*d, d), 'out': (M, d, d), 'ffn1': (M, f, d), 'ffn2': (M, d, f)}.items():
    a, w, dy = rnd(m, k), rnd(n, k), rnd(m, n)
    o = torch.empty(m, n, device='cuda', dtype=bf); dx = torch.empty(m, k, device='cuda', dtype=bf); dw = torch.empty(n, k, device='cuda', dtype=bf)
    ms = t(lambda: torch.matmul(a, w.t(), out=o)); res['fwd_' + name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
    ms = t(lambda: torch.matmul(dy, w, out=dx)); res['dgrad_' + name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
    ms = t(lambda: torch.matmul(dy.t(), a, out=dw)); res['wgrad_' + name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
print(json.dumps(res))
