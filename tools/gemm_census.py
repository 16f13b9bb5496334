#!/usr/bin/env python3
"""Which emo_gemm calls does one training step make, and which kernel family serves each?  (B from the environment, default the reference YAML's 4.)
Prints one line per distinct (layout, M, N, K, dtypes, epilogue) with its count per step and the family code of emo_gemm_last_kernel()."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from emo_disentanger_amd import ops  # noqa: E402
from emo_disentanger_amd.data import synthetic_batch  # noqa: E402
from emo_disentanger_amd.model.music_performer import MusicPerformer  # noqa: E402
from emo_disentanger_amd.optim import FusedAdam  # noqa: E402

B, T = int(os.environ.get('B', 4)), 2048
MODEL = os.environ.get('MODEL', 'performer')                      # performer | gpt2 | stage1
if MODEL == 'stage1':
    from emo_disentanger_amd.model.plain_transformer import PlainTransformer
    T, V = 512, 200
    m = PlainTransformer(512, V, 12, 8, 512, 2048, 0, T, dec_dropout=0.1, pre_lnorm=True, compute_dtype='bf16').cuda().train()
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randint(0, V - 1, (T, B), generator=g).cuda(), torch.randint(0, V - 1, (T, B), generator=g).cuda()
    fwd = lambda: m.compute_loss(m(x, tuple())[0], tgt)['total_loss']
else:
    if MODEL == 'gpt2':
        from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
        m = MusicGPT2(327, 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda().train()
    else:
        m = MusicPerformer(327, 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, favor_feature_dims=128, compute_dtype='bf16').cuda().train()
    b = synthetic_batch(327, B, T, device='cuda')
    fwd = lambda: m.compute_loss(m(b['dec_input'], seg_inp=b['track_mask']), b['dec_target'])['total_loss']
opt = FusedAdam(m, lr=1e-5, max_grad_norm=0.5)


def step():
    opt.zero_grad()
    l = fwd()
    l.backward()
    opt.step()


step()
torch.cuda.synchronize()
seen = collections.Counter()
real = ops.gemm


def gemm(A, Bm, **kw):
    res = real(A, Bm, **kw)
    out = res[0] if isinstance(res, tuple) else res               # (lna=: the product comes with LN(A), mean, rstd)
    a_t, b_t = kw.get('a_trans', False), kw.get('b_trans', False)
    K, M = (A.shape if a_t else A.shape[::-1])
    N = Bm.shape[1] if b_t else Bm.shape[0]
    epi = '+'.join(k for k in ('lna', 'bias', 'residual', 'mul_aux', 'mask_out', 'aux_out', 'a_rowsum', 'b_rowsum') if kw.get(k) is not None)
    if kw.get('p_drop', 0.0) > 0:
        epi += '+drop'
    if kw.get('act', 0):
        epi += '+act%d' % kw['act']
    if kw.get('accumulate'):
        epi += '+acc'
    seen[(('T' if a_t else 'N') + ('N' if b_t else 'T'), M, N, K, str(A.dtype)[6:], str(out.dtype)[6:], epi, ops.lib.emo_gemm_last_kernel())] += 1
    return res


ops.gemm = gemm                                                  # (engine / plain_transformer call ops.gemm through the module attribute)
step()
torch.cuda.synchronize()
for k, n in sorted(seen.items(), key=lambda t: (-t[1], t[0])):
    print('%3d x  %s  M=%-7d N=%-5d K=%-7d %s -> %s  [%s]  kernel family %d%s' % (n, k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7] & 15, ' + split-K' if k[7] & 48 else ''))
