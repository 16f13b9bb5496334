#!/usr/bin/env python3
"""Launch one decode-step GEMM shape (M=32 streams) 200 times inside a hipGraph (so launches are back to back as in generation); run under
rocprofv3 --kernel-trace --stats to read its device time.  SHAPE=qkv|out|ffn1|ffn2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
d, f = 512, 2048
n, k = {'qkv': (3 * d, d), 'out': (d, d), 'ffn1': (f, d), 'ffn2': (d, f)}[os.environ.get('SHAPE', 'qkv')]
if 'NK' in os.environ: n, k = [int(x) for x in os.environ['NK'].split(',')]
MM = int(os.environ.get('MM', 32))
bf = torch.bfloat16
a = torch.randn(MM, k, device='cuda').to(bf)
ws = [torch.randn(n, k, device='cuda').to(bf) for _ in range(12)]      # 12 layers' weights: not L2-resident between uses, as in a decode step
b = torch.randn(n, device='cuda')
o = torch.empty(MM, n, device='cuda', dtype=bf)
for w in ws: ops.gemm(a, w, bias=b, out=o)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(4):
            for w in ws: ops.gemm(a, w, bias=b, out=o)
torch.cuda.current_stream().wait_stream(s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(10): g.replay()
e1.record(); e1.synchronize()
print(os.environ.get('SHAPE', 'qkv'), MM, n, k, 'us per launch (graph replay, wall):', round(e0.elapsed_time(e1) / 480 * 1e3, 2))
