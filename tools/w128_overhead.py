#!/usr/bin/env python3
"""Fixed cost per output tile of the 256 x 256 tile kernel: sustained time of M=131072, N=512 NT products at K = 1024 / 2048 / 4096 (t = a + b K)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emo_disentanger_amd import ops
M, N = 131072, 512
res = {}
for K in (1024, 2048, 4096):
    A = [torch.randn(M, K, device='cuda').to(torch.bfloat16) for _ in range(2)]
    W = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    O = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(2)]
    for i in range(3):
        ops.gemm(A[i % 2], W, out=O[i % 2])
    torch.cuda.synchronize()
    t_end = time.time() + 2.0
    ts = []
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            ops.gemm(A[i % 2], W, out=O[i % 2])
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    t = sum(ts[len(ts) // 2:]) / len(ts[len(ts) // 2:])
    res[K] = t
    print('K=%d  %.1f us  %.0f TFLOP/s' % (K, t, 2.0 * M * N * K / t / 1e6))
    del A
b = (res[4096] - res[1024]) / 3072
a = res[2048] - b * 2048
print('per-launch fixed cost %.1f us (4 tiles per CU -> %.1f us per tile), marginal rate %.0f TFLOP/s' % (a, a / 4, 2.0 * M * N / b / 1e6))
