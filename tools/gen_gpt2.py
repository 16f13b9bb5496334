#!/usr/bin/env python3
"""AR generation throughput of the GPT-2 backbone (32 streams, 64-token prompt, NNEW new tokens, KV cache + hipGraph)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
from emo_disentanger_amd import inference as inf
torch.manual_seed(0)
m = MusicGPT2(327, 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda().eval()
n, T0, n_new = 32, 64, int(os.environ.get('NNEW', 256))
g = torch.Generator().manual_seed(7)
ptok = torch.randint(0, 326, (n, T0), generator=g).cuda(); pseg = torch.ones(n, T0, dtype=torch.long).cuda()
inf.generate_streams(m, ptok, pseg, 8, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = inf.generate_streams(m, ptok, pseg, n_new, seed=2)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({'model': 'gpt2', 'streams': n, 'new_tokens': n_new, 'tokens_per_s': round(n * n_new / dt, 1), 'ms_per_token_step': round(1e3 * dt / n_new, 3)}))
