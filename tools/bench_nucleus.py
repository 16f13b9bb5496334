#!/usr/bin/env python3
"""emo_sample_nucleus micro-benchmark: us per launch for flat vs peaked distributions (32 rows x V=327)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
V = 327
for name, scale in (('flat', 0.05), ('mid', 2.0), ('peaked', 12.0)):
    lg = (torch.randn(32, V, device='cuda') * scale).contiguous()
    u = torch.rand(32, device='cuda')
    for _ in range(5): ops.sample_nucleus(lg, 1.1, 0.9, u)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.sample_nucleus(lg, 1.1, 0.9, u)
    e1.record(); e1.synchronize()
    print(name, round(e0.elapsed_time(e1) / 200 * 1e3, 2), 'us')
