#!/usr/bin/env python3
"""Is the reference-batch (B = 4) training step bound by the host?  One forward + backward + optimizer step of the product model: host time to
QUEUE the step (python + ctypes + launch calls, no synchronisation) against the time the GPU takes to drain it, and the number of launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from emo_disentanger_amd.data import synthetic_batch  # noqa: E402
from emo_disentanger_amd.model.music_performer import MusicPerformer  # noqa: E402
from emo_disentanger_amd.optim import FusedAdam  # noqa: E402

B = int(os.environ.get('B', 4))
if os.environ.get('MODEL', 'performer') == 'stage1':            # BASELINE configs[4] shape (tools/bench_stage1.py)
    from emo_disentanger_amd.model.plain_transformer import PlainTransformer
    T, V = 512, 200
    m = PlainTransformer(512, V, 12, 8, 512, 2048, 0, T, dec_dropout=0.1, pre_lnorm=True, compute_dtype='bf16').cuda().train()
    opt = FusedAdam(m, lr=1e-5, max_grad_norm=0.5)
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randint(0, V - 1, (T, B), generator=g).cuda(), torch.randint(0, V - 1, (T, B), generator=g).cuda()

    def step():
        opt.zero_grad()
        l = m.compute_loss(m(x, tuple())[0], tgt)['total_loss']
        l.backward()
        opt.step()
        return l
else:
    T = 2048
    m = MusicPerformer(327, 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, favor_feature_dims=128, compute_dtype='bf16').cuda().train()
    opt = FusedAdam(m, lr=1e-5, max_grad_norm=0.5)
    b = synthetic_batch(327, B, T, device='cuda')

    def step():
        opt.zero_grad()
        l = m.compute_loss(m(b['dec_input'], seg_inp=b['track_mask']), b['dec_target'])['total_loss']
        l.backward()
        opt.step()
        return l


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 20
q, tot = [], []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    q.append((t1 - t0) / N * 1e3)
    tot.append((t2 - t0) / N * 1e3)
print(os.environ.get('MODEL', 'performer') + ' B=%d: host queues a step in %.2f ms; step (queue + drain) %.2f ms; GPU still busy %.2f ms after the last launch call of %d steps' %
      (B, min(q), min(tot), (tot[q.index(min(q))] - min(q)) * N, N))
