"""GPT-2 one-launch decode step (emo_gpt2_decode_step) timed at fixed context lengths: 32 streams, prefill CTX tokens, then K teacher-forced steps
between two events.  ms per step vs context = the latency floor of the 60 phases + the KV-cache stream (12 layers x 2 x ctx x 1 KB per stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
from emo_disentanger_amd import inference as inf
C = bench.CFG
torch.manual_seed(0)
m = MusicGPT2(C['n_token'], 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda().eval()
g = torch.Generator().manual_seed(7)
K = 40
for ctx in [int(x) for x in os.environ.get('CTXS', '16,64,256,512,1024,1536,2000').split(',')]:
    eng = inf.make_engine(m, 32)
    ptok = torch.randint(0, 326, (32, ctx), generator=g).cuda(); pseg = torch.ones(32, ctx, dtype=torch.long).cuda()
    eng.prefill(ptok, pseg)
    tok = torch.randint(0, 326, (32,), generator=g).cuda(); seg = torch.ones(32, dtype=torch.long).cuda()
    for _ in range(3):
        eng.step(tok, seg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        eng.step(tok, seg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    kv = 32 * 12 * 2 * (ctx + 3 + K / 2) * 1024
    print('ctx %5d: %.3f ms/step  (KV stream %.0f MB -> %.2f TB/s if it were all of the step)%s' % (ctx, ms, kv / 1e6, kv / ms / 1e9, '' if eng.persist is not None else '  [launch chain]'))
    if eng.persist is not None:
        eng.check_persistent()
