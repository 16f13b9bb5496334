"""Workload for rocprofv3: the GPT-2 backbone's 32-stream generation with the KV cache in HBM (bench.py: gen_gpt2), 64-token prompt + 512 new tokens."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
from emo_disentanger_amd import inference as inf
C = bench.CFG
torch.manual_seed(0)
m = MusicGPT2(C['n_token'], 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda().eval()
g = torch.Generator().manual_seed(7)
ptok = torch.randint(0, 326, (32, 64), generator=g).cuda(); pseg = torch.ones(32, 64, dtype=torch.long).cuda()
inf.generate_streams(m, ptok, pseg, 8, seed=1); torch.cuda.synchronize()
for n_new in (int(os.environ.get('N_NEW', 512)),):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inf.generate_streams(m, ptok, pseg, n_new, seed=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('n_new', n_new, 'total %.1f ms = %.3f ms/step, %.0f tok/s' % (dt * 1e3, dt * 1e3 / n_new, 32 * n_new / dt))
