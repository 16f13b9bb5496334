#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over the training step: the bench's model and batch, 1 warm-up + 2 measured steps of the product loop
(tools/pmc_classes.py turns the per-dispatch counters into profiles/r03_pmc_step.json)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from emo_disentanger_amd import train as tr
from emo_disentanger_amd.data import synthetic_batch
from emo_disentanger_amd.model.music_performer import MusicPerformer
from emo_disentanger_amd.optim import FusedAdam
C = bench.CFG
B = int(os.environ.get('BS', 64))
torch.manual_seed(0)
m = MusicPerformer(C['n_token'], C['n_layer'], C['n_head'], C['d_model'], C['d_ff'], C['d_model'], favor_feature_dims=C['n_feat'],
                   use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda().train()
opt = FusedAdam(m, lr=1e-4, max_grad_norm=0.5)
bs = [synthetic_batch(C['n_token'], B, C['seq'], seed=1234 + 100 * i, device='cuda') for i in range(2)]
cfg = tr.TrainConfig(redraw_prob=1.0, log_interval=10 ** 9, ckpt_dir=tempfile.mkdtemp(), verbose=False)
import time
n = int(os.environ.get('STEPS', 3))
if os.environ.get('TIME'):                                   # TIME=1: 3 untimed steps first, then report ms per step (same-box A/B of kernel variants)
    tr.train_model(1, m, [bs[i % 2] for i in range(3)], opt, None, C['n_token'] - 1, cfg=cfg)
torch.cuda.synchronize(); t0 = time.perf_counter()
tr.train_model(1, m, [bs[i % 2] for i in range(n)], opt, None, C['n_token'] - 1, cfg=cfg)
torch.cuda.synchronize()
if os.environ.get('TIME'):
    dt = (time.perf_counter() - t0) / n
    print('B=%d: %.3f ms/step = %.1f k tokens/s' % (B, dt * 1e3, B * C['seq'] / dt / 1e3))
