#!/usr/bin/env python3
"""GEMM micro-benchmark on the perf-config shapes (fwd NT, dgrad NN, wgrad TN).  EMO_GEMM_VARIANT=1|2 selects the kernel."""
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

M, d, f = int(os.environ.get('M', 131072)), 512, 2048
res = {}
bf = torch.bfloat16
def rnd(*s): return torch.randn(*s, device='cuda').to(bf)
for name, (m, n, k) in {'qkv': (M, 3*d, d), 'out': (M, d, d), 'ffn1': (M, f, d), 'ffn2': (M, d, f)}.items():
    a, w, o = rnd(m, k), rnd(n, k), torch.empty(m, n, device='cuda', dtype=bf)
    ms = t(lambda: ops.gemm(a, w, out=o)); res['fwd_' + name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
    bias = torch.randn(n, device='cuda'); r = rnd(m, n)
    ms = t(lambda: ops.gemm(a, w, out=o, bias=bias, act=ops.ACT_RELU, p_drop=0.1, seed=1, offset=2, residual=r)); res['fwd_epi_' + name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
    # dgrad: dX[m,k] = dY[m,n] @ W[n,k]  (b_trans=True)
    dy, dx = rnd(m, n), torch.empty(m, k, device='cuda', dtype=bf)
    ms = t(lambda: ops.gemm(dy, w, b_trans=True, out=dx)); res['dgrad_' + name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
    # wgrad: dW[n,k] = dY^T X
    dw = torch.zeros(n, k, device='cuda')
    ms = t(lambda: ops.gemm(dy, a, a_trans=True, b_trans=True, out=dw, accumulate=True)); res['wgrad_' + name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
    cs = torch.zeros(n, device='cuda')
    ms = t(lambda: ops.colsum(dy, out=cs, accumulate=True)); res['colsum_' + name] = (round(ms, 4), round(m*n*2/ms/1e6, 1))
    del a, w, o, r, dy, dx
print(json.dumps(res))
