#!/usr/bin/env python3
"""Workload for rocprofv3: the fp32 parity mode's training step (bench.py: fp32_parity) at B = BS sequences x T = 2048, 3 + 3 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from emo_disentanger_amd import train as tr
from emo_disentanger_amd.optim import FusedAdam
B = int(os.environ.get('BS', 16))
r = bench.fp32_parity_bench(tr, FusedAdam, B, 2048, steps=3, warm=2)
print(r['ms_per_step'], r['value'], r['roofline'])
