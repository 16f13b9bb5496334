#!/usr/bin/env python3
"""Per-shape GEMM timing INSIDE the training step (HIP events around every launch of 2 real steps at the bench configuration): the numbers
GEMM variants have to be judged by (cold-operand microbenchmarks mislead: DESIGN.md §4.1)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from emo_disentanger_amd import ops
from emo_disentanger_amd.data import synthetic_batch
from emo_disentanger_amd.model.music_performer import MusicPerformer
from emo_disentanger_amd.optim import FusedAdam
C = bench.CFG
B, T = int(os.environ.get('B', 64)), C['seq']
torch.manual_seed(0)
model = MusicPerformer(C['n_token'], C['n_layer'], C['n_head'], C['d_model'], C['d_ff'], C['d_model'], favor_feature_dims=C['n_feat'], use_segment_emb=True,
                       n_segment_types=2, dropout=0.1, compute_dtype='bf16', redraw='every_forward').cuda().train()
opt = FusedAdam(model, lr=1e-4, max_grad_norm=0.5, world_size=1)
b = synthetic_batch(C['n_token'], B, T, seed=1234, device='cuda')
def step():
    opt.zero_grad()
    logits = model(b['dec_input'], seg_inp=b['track_mask'])
    model.compute_loss(logits, b['dec_target'])['total_loss'].backward()
    opt.step()
for _ in range(3): step()
ops.GEMM_TIMING = []
for _ in range(2): step()
torch.cuda.synchronize()
rec, ops.GEMM_TIMING = ops.GEMM_TIMING, None
agg = {}
for kind, e0, e1, fl, by_, shape in rec:
    d = agg.setdefault((kind.split('/')[0], shape), [0.0, 0, fl])
    d[0] += e0.elapsed_time(e1); d[1] += 1
print('%-4s %-22s %6s %9s %9s %9s' % ('lay', '(M, N, K)', 'n/step', 'us', 'TFLOP/s', 'ms/step'))
for (kind, shape), (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%-4s %-22s %6d %9.1f %9.1f %9.2f' % (kind, shape, n // 2, ms / n * 1e3, fl / (ms / n) / 1e9, ms / 2))
