#!/usr/bin/env python3
"""Training-step timing of the stage-1 lead-sheet LM (BASELINE configs[4] shape: emopia_finetune.yaml = d512 L12 H8 d_ff 2048, tgt_len 512,
batch 4; vocabulary ~200), bf16, dropout 0.1, FusedAdam.  Secondary path: the stage-2 Performer is the benched one."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd.model.plain_transformer import PlainTransformer
from emo_disentanger_amd.optim import FusedAdam
B, T, V = int(os.environ.get('B', 4)), int(os.environ.get('T', 512)), 200
torch.manual_seed(0)
m = PlainTransformer(512, V, 12, 8, 512, 2048, 0, T, dec_dropout=0.1, pre_lnorm=True, compute_dtype='bf16').cuda().train()
opt = FusedAdam(m, lr=1e-5, max_grad_norm=0.5)
g = torch.Generator().manual_seed(1)
x = torch.randint(0, V - 1, (T, B), generator=g).cuda(); tgt = torch.randint(0, V - 1, (T, B), generator=g).cuda()
def step():
    opt.zero_grad()
    l = m.compute_loss(m(x, tuple())[0], tgt)['total_loss']
    l.backward(); opt.step(); return l
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(json.dumps({'model': 'stage1 txl', 'B': B, 'T': T, 'ms_per_step': round(dt * 1e3, 2), 'tokens_per_s': round(B * T / dt, 1), 'loss': float(l)}))
