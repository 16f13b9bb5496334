#!/usr/bin/env python3
"""cProfile of the host side of the reference-batch (B = 4) training step, which is bound by it (tools/b4_cpu_probe.py): top functions by own time."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from emo_disentanger_amd.data import synthetic_batch  # noqa: E402
from emo_disentanger_amd.model.music_performer import MusicPerformer  # noqa: E402
from emo_disentanger_amd.optim import FusedAdam  # noqa: E402

B = int(os.environ.get('B', 4))
STAGE1 = os.environ.get('MODEL', 'performer') == 'stage1'
if STAGE1:
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
# the backward runs on the autograd engine's thread, which a profiler enabled here does not see: wrap the Function's backward
from emo_disentanger_amd import engine  # noqa: E402

prb = cProfile.Profile()
if STAGE1:
    from emo_disentanger_amd.model import plain_transformer as _pt
    _cls = _pt.TXLStackFn
else:
    _cls = engine.DecoderStackFn
_orig = _cls.backward


def _prof_bwd(ctx, dout):
    prb.enable()
    try:
        return _orig(ctx, dout)
    finally:
        prb.disable()


_cls.backward = staticmethod(_prof_bwd)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
print('================ backward of the decoder stack (autograd thread)')
pstats.Stats(prb).sort_stats('tottime').print_stats(30)
