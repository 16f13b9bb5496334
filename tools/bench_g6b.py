#!/usr/bin/env python3
"""v6 GEMM diagnostics: an L2/MALL-resident square problem (4096^3 NT) and streaming shapes, with EMO_GEMM_ABLATE variants."""
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
bf = torch.bfloat16
def rnd(*s): return torch.randn(*s, device='cuda').to(bf)
out = {}
for name, (M, N, K, at, bt, odt) in {'sq4096_NT': (4096, 4096, 4096, False, False, bf), 'sq8192_NT': (8192, 8192, 8192, False, False, bf),
                                     'sq4096_TN': (4096, 4096, 4096, True, True, bf),
                                     'ffn2_fwd': (131072, 512, 2048, False, False, bf), 'w_ffn1': (2048, 512, 131072, True, True, torch.float32)}.items():
    a = rnd(K, M) if at else rnd(M, K)
    b = rnd(K, N) if bt else rnd(N, K)
    c = torch.empty(M, N, device='cuda', dtype=odt)
    ms = t(lambda: ops.gemm(a, b, a_trans=at, b_trans=bt, out=c))
    out[name] = round(2.0 * M * N * K / ms / 1e9, 1)
print(json.dumps({'g6': os.environ.get('EMO_GEMM_G6', 'auto'), 'ablate': os.environ.get('EMO_GEMM_ABLATE', '0'), 'tflops': out}), flush=True)
