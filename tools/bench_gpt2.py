#!/usr/bin/env python3
"""Training-step timing of the GPT-2 backbone (secondary path; BASELINE benches the Performer)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
from emo_disentanger_amd.optim import FusedAdam
from emo_disentanger_amd.data import synthetic_batch
B, T = int(os.environ.get('B', 16)), int(os.environ.get('T', 2048))
m = MusicGPT2(327, 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, dropout=float(os.environ.get('DROPOUT', 0.1)), compute_dtype='bf16').cuda().train()
opt = FusedAdam(m, lr=1e-5, max_grad_norm=0.5)
b = synthetic_batch(327, B, T, device='cuda')
def step():
    opt.zero_grad()
    l = m.compute_loss(m(b['dec_input'], seg_inp=b['track_mask']), b['dec_target'])['total_loss']
    l.backward(); opt.step(); return l
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(json.dumps({'model': 'gpt2', 'B': B, 'T': T, 'ms_per_step': round(dt * 1e3, 2), 'tokens_per_s': round(B * T / dt, 1), 'loss': float(l)}))
