#!/usr/bin/env python3
"""256^2 four-phase GEMM (v6) vs the 128^2 kernels: correctness against torch, run-to-run bitwise stability (race screen) and time, on the
training shapes it is dispatched for.  Run once with EMO_GEMM_G6=0 and once without (the mode is read per process)."""
import json
import os
import sys
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


bf = torch.bfloat16
def rnd(*s): return torch.randn(*s, device='cuda').to(bf)
T = int(os.environ.get('TOKENS', 131072))
d, f = 512, 2048
res = {}
# ---- wgrad (TN): dW[n, k] = dY^T X
for name, (n, k) in {'w_qkv': (3 * d, d), 'w_out': (d, d), 'w_ffn1': (f, d), 'w_ffn2': (d, f)}.items():
    x, dy = rnd(T, k), rnd(T, n)
    dw = torch.zeros(n, k, device='cuda')
    bias = torch.zeros(n, device='cuda')
    def run(): 
        bias.zero_()
        return ops.gemm(dy, x, a_trans=True, b_trans=True, out=dw, accumulate=False, a_rowsum=bias)
    run()
    ref = dy[:16384].float().t() @ x[:16384].float()
    small = torch.zeros(n, k, device='cuda'); sb = torch.zeros(n, device='cuda')
    ops.gemm(dy[:16384], x[:16384], a_trans=True, b_trans=True, out=small, a_rowsum=sb)
    err = float((small - ref).abs().max() / ref.abs().max())
    berr = float((sb - dy[:16384].float().sum(0)).abs().max() / dy[:16384].float().sum(0).abs().max())
    outs = []
    for _ in range(4):
        run(); outs.append(dw.clone())
    stable = all(torch.equal(outs[0], o) for o in outs[1:])
    ms = t(run)
    res[name] = dict(ms=round(ms, 4), tflops=round(2.0 * T * n * k / ms / 1e9, 1), rel_err=err, bias_rel_err=berr, stable=stable)
    print(name, json.dumps(res[name]), flush=True)
# ---- long-K single-pass: FFN2 forward (NT, K = 2048) and FFN1 dgrad (NN, K = 2048), qkv dgrad (NN, K = 1536)
for name, (n, k, b_trans) in {'f_ffn2': (d, f, False), 'd_ffn1': (d, f, True), 'd_qkv': (d, 3 * d, True)}.items():
    a = rnd(T, k)
    w = rnd(k, n) if b_trans else rnd(n, k)
    out = torch.empty(T, n, device='cuda', dtype=bf)
    def run(): return ops.gemm(a, w, b_trans=b_trans, out=out)
    run()
    ref = a[:4096].float() @ (w.float() if b_trans else w.float().t())
    err = float((out[:4096].float() - ref).abs().max() / ref.abs().max())
    outs = []
    for _ in range(4):
        run(); outs.append(out.clone())
    stable = all(torch.equal(outs[0], o) for o in outs[1:])
    ms = t(run)
    res[name] = dict(ms=round(ms, 4), tflops=round(2.0 * T * n * k / ms / 1e9, 1), rel_err=err, stable=stable)
    print(name, json.dumps(res[name]), flush=True)
print(json.dumps({'g6': os.environ.get('EMO_GEMM_G6', 'auto'), 'splits': os.environ.get('EMO_GEMM_SPLITS', 'auto'), 'res': res}))
