import sys, torch
sys.path.insert(0, '/root/repo')
from emo_disentanger_amd import inference as inf, ops
from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
from oracle.weights import make_state_dict
V, L, H, d, dff, n, T0, Tn = 80, 1, 4, 128, 256, 3, 37, 12
sd = make_state_dict('gpt2', V, L, H, d, dff, seed=8, scale=3.0)
m = MusicGPT2(V, L, H, d, dff, d, use_segment_emb=True, n_segment_types=2, compute_dtype='fp32')
m.load_state_dict(sd); m = m.cuda().eval()
gen = torch.Generator().manual_seed(0)
tok = torch.randint(0, V - 1, (n, T0 + Tn), generator=gen).cuda()
seg = torch.randint(0, 2, (n, T0 + Tn), generator=gen).cuda()
with torch.no_grad():
    full = m(tok, seg_inp=seg)
    pre = m(tok[:, :T0].contiguous(), seg_inp=seg[:, :T0].contiguous())
    pre_nc = m(tok[:, :T0], seg_inp=seg[:, :T0])
print('full vs prefix-forward (contig):', float((full[:, :T0] - pre).abs().max()))
print('full vs prefix-forward (non-contig):', float((full[:, :T0] - pre_nc).abs().max()))
eng = inf.make_engine(m, n)
lg = eng.prefill(tok[:, :T0].contiguous(), seg[:, :T0].contiguous())
print('engine prefill vs pre:', float((lg - pre[:, -1]).abs().max()))
x1 = eng._embed(tok[:, :T0].contiguous(), seg[:, :T0].contiguous(), 0)
ps = m._store
x2 = ops.embed_fwd(tok, seg, ps.f32('token_emb.emb_lookup.weight'), ps.f32('segemb.emb_lookup.weight'), m.pe.pe, torch.float32, float(m.token_emb.emb_scale)).view(n, T0+Tn, d)[:, :T0]
print('embed diff', float((x1.view(n, T0, d) - x2).abs().max()))
