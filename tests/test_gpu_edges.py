"""GPU edge cases of the path: maximum sequence length of the Performer YAMLs (T=3072), REMI vocabulary (V=370),
single-token sequences, all-pad targets, and the sliding-window fallback of generate_conditional."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _perf(V, L, H, d, dff, nf, dtype, seed=1, scale=2.0):
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from oracle.weights import make_state_dict
    sd = make_state_dict('performer', V, L, H, d, dff, favor_feature_dims=nf, seed=seed, scale=scale)
    m = MusicPerformer(V, L, H, d, dff, d, dropout=0.0, favor_feature_dims=nf, use_segment_emb=True, n_segment_types=2, compute_dtype=dtype,
                       redraw='fixed')
    m.load_state_dict(sd)
    return m.cuda(), sd


def test_max_len_3072_remi_vocab_matches_oracle():
    from oracle import model_ref
    from oracle.weights import synthetic_batch
    V, L, H, d, dff, nf, T = 370, 1, 2, 128, 256, 128, 3072           # d_head 64, 128 features: the production kernel instance
    m, sd = _perf(V, L, H, d, dff, nf, 'fp32')
    b = synthetic_batch(V, 1, T, seed=9, realistic_targets=True)
    rloss, rlogits, rgrads = model_ref.loss_and_grads('performer', sd, b, V, L, H, d)
    m.train()
    logits = m(b['dec_input'].cuda(), seg_inp=b['track_mask'].cuda())
    loss = m.compute_loss(logits, b['dec_target'].cuda())['total_loss']
    loss.backward()
    assert abs(float(loss) - float(rloss)) <= 1e-4
    np.testing.assert_allclose(logits.detach().cpu().numpy()[0, -64:], rlogits.numpy()[0, -64:], rtol=5e-4, atol=5e-4)
    gmax = max(float(g.abs().max()) for g in rgrads.values())
    for k, p in m.named_parameters():
        assert float((p.grad.cpu() - rgrads[k]).abs().max()) <= 3e-3 * gmax, k
    # bf16 speed mode at the same length
    m16, _ = _perf(V, L, H, d, dff, nf, 'bf16')
    l16 = m16.compute_loss(m16(b['dec_input'].cuda(), seg_inp=b['track_mask'].cuda()), b['dec_target'].cuda())['total_loss']
    assert abs(float(l16) - float(rloss)) <= 3e-2


@pytest.mark.parametrize('kind', ['performer', 'gpt2'])
def test_single_token_sequences_and_all_pad_targets(kind):
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from oracle import model_ref
    from oracle.weights import make_state_dict
    V, L, H, d, dff, nf = 50, 2, 4, 64, 128, 32
    sd = make_state_dict(kind, V, L, H, d, dff, favor_feature_dims=nf, seed=3, scale=3.0)
    if kind == 'performer':
        m, _ = _perf(V, L, H, d, dff, nf, 'fp32', seed=3, scale=3.0)
    else:
        m = MusicGPT2(V, L, H, d, dff, d, dropout=0.0, use_segment_emb=True, n_segment_types=2, compute_dtype='fp32')
        m.load_state_dict(sd)
        m = m.cuda()
    x = torch.tensor([[7], [3], [11]])
    seg = torch.tensor([[1], [0], [1]])
    ref = model_ref.forward(kind, sd, x, seg, L, H, d)
    out = m(x.cuda(), seg_inp=seg.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.numpy(), rtol=3e-4, atol=3e-4)            # T = 1
    tgt = torch.full((3, 1), V - 1)
    loss = m.compute_loss(out, tgt.cuda())['total_loss']
    assert torch.isnan(loss)                                                                          # mean over an empty set, like F.cross_entropy


def test_generate_conditional_sliding_window_fallback(monkeypatch):
    """Once len(generated) >= max_dec_inp_len the reference slides a window whose positions restart at 0: the cache is
    invalid there and the engine must fall back to the full-window forward; cached and uncached loops must agree."""
    import json
    import os
    from emo_disentanger_amd import inference as inf
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from oracle.weights import make_state_dict
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'generate.json')))
    e2i = {e: i for i, e in enumerate(g['events'])}
    i2e = {i: e for e, i in e2i.items()}
    mm = g['model']
    sd = make_state_dict('gpt2', mm['V'], mm['L'], mm['H'], mm['d'], mm['dff'], seed=mm['seed'], scale=mm['scale'])
    m = MusicGPT2(mm['V'], mm['L'], mm['H'], mm['d'], mm['dff'], mm['d'], use_segment_emb=True, n_segment_types=2, compute_dtype='fp32')
    m.load_state_dict(sd)
    m = m.cuda().eval()
    monkeypatch.setattr(inf, 'max_dec_inp_len', 24)
    outs = []
    for use_cache in (True, False):
        outs.append(inf.generate_conditional(m, e2i, i2e, [list(b) for b in g['lead']], list(g['primer']), max_events=60, skip_check=True,
                                             temp=1.2, top_p=0.97, model_type='gpt2', use_cache=use_cache, sampler=lambda p: int(np.argmax(p))))
    assert outs[0] == outs[1] and len(outs[0]) > 24
