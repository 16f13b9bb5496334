import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from emo_disentanger_amd.model.music_performer import MusicPerformer
from emo_disentanger_amd import inference as inf
C = bench.CFG
torch.manual_seed(0)
m = MusicPerformer(C['n_token'], C['n_layer'], C['n_head'], C['d_model'], C['d_ff'], C['d_model'], favor_feature_dims=C['n_feat'],
                   use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda().eval()
g = torch.Generator().manual_seed(7)
ptok = torch.randint(0, 326, (32, 64), generator=g).cuda(); pseg = torch.ones(32, 64, dtype=torch.long).cuda()
inf.generate_streams(m, ptok, pseg, 8, seed=1); torch.cuda.synchronize()
os.environ['EMO_GEN_TIMING'] = '1'
for n_new in (256, 1024):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inf.generate_streams(m, ptok, pseg, n_new, seed=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('n_new', n_new, 'total %.1f ms = %.3f ms/step, %.0f tok/s' % (dt * 1e3, dt * 1e3 / n_new, 32 * n_new / dt))
