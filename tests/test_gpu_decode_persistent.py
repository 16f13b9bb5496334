"""GPU: the one-launch Performer decode step (emo_performer_decode_step, csrc/emo_decode_persist.hip) against (a) the chain of launches it
replaces (same bf16 arithmetic up to the LayerNorm fold and the reduction order: logits within 2 % of the logit range, recurrent state within
1e-3 relative) and (b) the fp32 parity-mode engine, which is itself tied to the oracle's recurrent form (tests/test_gpu_generate.py) — bf16
tolerance 5 % of the logit range.  Reference: the token loop of stage2_accompaniment/inference.py:250-277."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(L, dtype, seed=0, scale=2.5):
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from oracle.weights import make_state_dict
    V, H, d, dff, nf = 327, 8, 512, 2048, 128
    sd = make_state_dict('performer', V, L, H, d, dff, favor_feature_dims=nf, seed=seed, scale=scale)
    m = MusicPerformer(V, L, H, d, dff, d, favor_feature_dims=nf, use_segment_emb=True, n_segment_types=2, compute_dtype=dtype, redraw='fixed')
    m.load_state_dict(sd)
    return m.cuda().eval()


def _run(model, ptok, pseg, toks, segs, persistent, monkeypatch):
    from emo_disentanger_amd import inference as inf
    monkeypatch.setenv('EMO_DECODE_PERSISTENT', '1' if persistent else '0')
    eng = inf.make_engine(model, ptok.shape[0], redraw=False)
    assert (eng.persist is not None) == (persistent and model.compute_dtype == torch.bfloat16)
    out = [eng.prefill(ptok, pseg).float().clone()]
    for t in range(toks.shape[1]):
        out.append(eng.step(toks[:, t], segs[:, t]).float().clone())
    if eng.persist is not None:
        eng.check_persistent()
    return torch.stack(out, 1), [s.clone() for s in eng.S], [z.clone() for z in eng.z]


@pytest.mark.parametrize('n,L', [(4, 1), (8, 3), (32, 12), (1, 2), (5, 2)])       # (1, 5: padded to the kernel's groups of 4 streams)
def test_one_launch_step_matches_launch_chain_and_fp32(n, L, monkeypatch):
    g = torch.Generator().manual_seed(5 + n)
    V, T0, K = 327, 24, 6
    ptok = torch.randint(0, V - 1, (n, T0), generator=g).cuda()
    pseg = torch.randint(0, 2, (n, T0), generator=g).cuda()
    toks = torch.randint(0, V - 1, (n, K), generator=g).cuda()
    segs = torch.randint(0, 2, (n, K), generator=g).cuda()
    mb = _model(L, 'bf16')
    one, S1, z1 = _run(mb, ptok, pseg, toks, segs, True, monkeypatch)
    chain, S0, z0 = _run(mb, ptok, pseg, toks, segs, False, monkeypatch)
    ref, _, _ = _run(_model(L, 'fp32'), ptok, pseg, toks, segs, False, monkeypatch)
    rng = float(ref.max() - ref.min())
    assert torch.equal(one[:, 0], chain[:, 0])                       # the prefill is shared
    e_chain = float((one - chain).abs().max()) / rng
    e_ref = float((one - ref).abs().max()) / rng
    e_chain_ref = float((chain - ref).abs().max()) / rng
    print('[one-launch decode] n=%d L=%d: vs launch chain %.4f, vs fp32 %.4f (launch chain vs fp32 %.4f) of the logit range' % (n, L, e_chain, e_ref, e_chain_ref))
    assert e_chain <= 0.02 and e_ref <= 0.05
    for a, b in zip(S1 + z1, S0 + z0):
        assert float((a - b).norm() / b.norm().clamp_min(1e-12)) <= 5e-3     # (the inputs differ by the bf16 rounding of the folded / explicit LayerNorm)
    # graph replay of the step = eager launches, bit for bit (the launch counter in the workspace advances on the device)
    from emo_disentanger_amd import inference as inf
    monkeypatch.setenv('EMO_DECODE_PERSISTENT', '1')
    a = inf.generate_streams(mb, ptok, pseg, 12, greedy=True, use_graph=False)
    b = inf.generate_streams(mb, ptok, pseg, 12, greedy=True, use_graph=True)
    c = inf.generate_streams(mb, ptok, pseg, 12, greedy=False, seed=3, use_graph=True)
    d = inf.generate_streams(mb, ptok, pseg, 12, greedy=False, seed=3, use_graph=False)
    assert torch.equal(a, b) and torch.equal(c, d)
    # the draw inside the launch picks the same tokens as the sampler kernel in front of it (same device code, same logits)
    monkeypatch.setenv('EMO_PD_SAMPLER', '0')
    e = inf.generate_streams(mb, ptok, pseg, 12, greedy=False, seed=3, use_graph=True)
    assert torch.equal(c, e)


@pytest.mark.parametrize('n', [4, 32])
def test_one_launch_step_matches_oracle_recurrent_form(n, monkeypatch):
    """The launch `gen` times, directly against the ORACLE: teacher-force 64 tokens on 4 / 32 streams at d512 / L12 / H8 / F128 and compare
    every step's logits of emo_performer_decode_step with oracle.model_ref.performer_forward(form='recurrent') — the token recurrence
    S += phi(k) v^T, z += phi(k), out = phi(q) S / (phi(q) z + eps) that the cached loop of stage2_accompaniment/inference.py:250-277 runs —
    on the same tokens.  bf16 bound: 5 % of the logit range (the bound of the bf16 training-parity tests); greedy ids must agree wherever the
    oracle's top-2 margin exceeds twice that bound's measured counterpart."""
    from oracle import model_ref
    from oracle.weights import make_state_dict
    from emo_disentanger_amd import inference as inf
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    V, L, H, d, dff, nf, T0, K = 327, 12, 8, 512, 2048, 128, 24, 64
    sd = make_state_dict('performer', V, L, H, d, dff, favor_feature_dims=nf, seed=3, scale=2.5)
    m = MusicPerformer(V, L, H, d, dff, d, favor_feature_dims=nf, use_segment_emb=True, n_segment_types=2, compute_dtype='bf16', redraw='fixed')
    m.load_state_dict(sd)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(50 + n)
    tok = torch.randint(0, V - 1, (n, T0 + K), generator=g)
    seg = torch.randint(0, 2, (n, T0 + K), generator=g)
    with torch.no_grad():
        ref = model_ref.performer_forward(sd, tok, seg, L, H, d, form='recurrent')[:, T0 - 1:]          # [n, K + 1, V]: positions T0-1 .. T0+K-1
    monkeypatch.setenv('EMO_DECODE_PERSISTENT', '1')
    eng = inf.make_engine(m, n, redraw=False)
    assert eng.persist is not None                                   # the one-launch step, not the launch chain
    tc, sc = tok.cuda(), seg.cuda()
    out = [eng.prefill(tc[:, :T0], sc[:, :T0]).float().clone()]
    for t in range(K):
        out.append(eng.step(tc[:, T0 + t], sc[:, T0 + t]).float().clone())
    eng.check_persistent()
    got = torch.stack(out, 1).cpu()
    rng = float(ref.max() - ref.min())
    err = (got - ref).abs()
    e_max, e_steps = float(err.max()), err.amax(dim=(0, 2))
    top2 = ref.topk(2, -1).values
    margin = top2[..., 0] - top2[..., 1]
    safe = margin > 2 * e_max
    print('[one-launch decode vs oracle recurrent] n=%d: max |dlogit| %.4f = %.4f of the logit range (step 1: %.4f, step %d: %.4f); greedy ids compared at %.0f %% of positions'
          % (n, e_max, e_max / rng, float(e_steps[1]), K, float(e_steps[-1]), 100 * float(safe.float().mean())))
    assert e_max <= 0.05 * rng
    assert float(e_steps[-8:].max()) <= 3 * float(e_steps[1:9].max()) + 0.01 * rng        # the recurrent state does not drift over the 64 steps
    assert float(safe.float().mean()) > 0.05
    assert torch.equal(got.argmax(-1)[safe], ref.argmax(-1)[safe])


def test_one_launch_step_refuses_what_it_was_not_built_for():
    from emo_disentanger_amd import ops
    from emo_disentanger_amd._lib import EmoError
    z = torch.zeros(16, device='cuda')
    zi = torch.zeros(16, dtype=torch.int64, device='cuda')
    ws = torch.zeros(ops.lib.emo_performer_decode_step_workspace_bytes() // 8, dtype=torch.int64, device='cuda')
    lg = torch.zeros(4, 327, device='cuda')
    with pytest.raises(EmoError, match='built for d_model 512'):
        ops.performer_decode_step(zi, 1, zi, None, z, None, z, 1.0, 0, None, z, z, 327, lg, 4, 256, 8, 128, 2048, ws)
    lg6 = torch.zeros(6, 327, device='cuda')
    with pytest.raises(EmoError, match='multiple of 4'):
        ops.performer_decode_step(zi, 1, zi, None, z, None, z, 1.0, 0, None, z, z, 327, lg6, 6, 512, 8, 128, 2048, ws)


# ------------------------------------------------------------------------------------------------ GPT-2 form (emo_gpt2_decode_step, r06)
def _gpt2(L, dtype, seed=0, scale=2.5):
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from oracle.weights import make_state_dict
    V, H, d, dff = 327, 8, 512, 2048
    sd = make_state_dict('gpt2', V, L, H, d, dff, seed=seed, scale=scale)
    m = MusicGPT2(V, L, H, d, dff, d, use_segment_emb=True, n_segment_types=2, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


def _run_gpt2(model, ptok, pseg, toks, segs, persistent, monkeypatch):
    from emo_disentanger_amd import inference as inf
    monkeypatch.setenv('EMO_DECODE_PERSISTENT', '1' if persistent else '0')
    eng = inf.make_engine(model, ptok.shape[0])
    assert (eng.persist is not None) == (persistent and model.compute_dtype == torch.bfloat16)
    out = [eng.prefill(ptok, pseg).float().clone()]
    for t in range(toks.shape[1]):
        out.append(eng.step(toks[:, t], segs[:, t]).float().clone())
    if eng.persist is not None:
        eng.check_persistent()
    T = ptok.shape[1] + toks.shape[1]
    return torch.stack(out, 1), [k[:, :, :T].clone() for k in eng.kc], [v[:, :, :T].clone() for v in eng.vc]


@pytest.mark.parametrize('n,L,T0', [(4, 1, 24), (8, 3, 2), (32, 12, 300), (1, 2, 33), (5, 2, 257)])      # (1, 5: padded to groups of 4 streams)
def test_gpt2_one_launch_step_matches_launch_chain_and_fp32(n, L, T0, monkeypatch):
    """emo_gpt2_decode_step against the chain of launches it replaces (skinny GEMMs with folded LayerNorms + sattn_decode: the same bf16 arithmetic up
    to the fold and the reduction order) and the fp32 parity-mode engine; the appended key / value rows must equal the chain's up to the bf16 rounding
    of their inputs.  Context lengths cover one row, a partial 256-row sweep, and several sweeps.  Reference loop: stage2_accompaniment/inference.py:250-277."""
    g = torch.Generator().manual_seed(15 + n)
    V, K = 327, 6
    ptok = torch.randint(0, V - 1, (n, T0), generator=g).cuda()
    pseg = torch.randint(0, 2, (n, T0), generator=g).cuda()
    toks = torch.randint(0, V - 1, (n, K), generator=g).cuda()
    segs = torch.randint(0, 2, (n, K), generator=g).cuda()
    mb, sd = _gpt2(L, 'bf16')
    one, K1, V1 = _run_gpt2(mb, ptok, pseg, toks, segs, True, monkeypatch)
    chain, K0, V0 = _run_gpt2(mb, ptok, pseg, toks, segs, False, monkeypatch)
    ref, _, _ = _run_gpt2(_gpt2(L, 'fp32')[0], ptok, pseg, toks, segs, False, monkeypatch)
    rng = float(ref.max() - ref.min())
    assert torch.equal(one[:, 0], chain[:, 0])                       # the prefill is shared
    e_chain = float((one - chain).abs().max()) / rng
    e_ref = float((one - ref).abs().max()) / rng
    e_chain_ref = float((chain - ref).abs().max()) / rng
    print('[one-launch GPT-2 decode] n=%d L=%d T0=%d: vs launch chain %.4f, vs fp32 %.4f (launch chain vs fp32 %.4f) of the logit range' % (n, L, T0, e_chain, e_ref, e_chain_ref))
    assert e_chain <= 0.02 and e_ref <= 0.05
    for a, b in zip(K1 + V1, K0 + V0):
        assert a.shape == b.shape and float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)) <= 1e-2
    from emo_disentanger_amd import inference as inf
    monkeypatch.setenv('EMO_DECODE_PERSISTENT', '1')
    a = inf.generate_streams(mb, ptok, pseg, 12, greedy=True, use_graph=False)
    b = inf.generate_streams(mb, ptok, pseg, 12, greedy=True, use_graph=True)
    c = inf.generate_streams(mb, ptok, pseg, 12, greedy=False, seed=3, use_graph=True)
    d = inf.generate_streams(mb, ptok, pseg, 12, greedy=False, seed=3, use_graph=False)
    assert torch.equal(a, b) and torch.equal(c, d)
    # the draw inside the launch picks the same tokens as the sampler kernel in front of the same launch (same device code, same logits)
    monkeypatch.setenv('EMO_PD_SAMPLER', '0')
    e = inf.generate_streams(mb, ptok, pseg, 12, greedy=False, seed=3, use_graph=True)
    assert torch.equal(c, e)


@pytest.mark.parametrize('n', [4, 32])
def test_gpt2_one_launch_step_matches_oracle_full_forward(n, monkeypatch):
    """The launch `gen_gpt2` times, directly against the ORACLE: teacher-force 64 tokens at d512 / L12 / H8 and compare every step's logits with
    oracle.model_ref.gpt2_forward over the whole prefix (what the reference's loop recomputes per token).  bf16 bound: 5 % of the logit range."""
    from oracle import model_ref
    from emo_disentanger_amd import inference as inf
    V, L, H, d, T0, K = 327, 12, 8, 512, 24, 64
    m, sd = _gpt2(L, 'bf16', seed=3)
    g = torch.Generator().manual_seed(70 + n)
    tok = torch.randint(0, V - 1, (n, T0 + K), generator=g)
    seg = torch.randint(0, 2, (n, T0 + K), generator=g)
    with torch.no_grad():
        ref = model_ref.gpt2_forward(sd, tok, seg, L, H, d)[:, T0 - 1:]
    monkeypatch.setenv('EMO_DECODE_PERSISTENT', '1')
    eng = inf.make_engine(m, n)
    assert eng.persist is not None
    tc, sc = tok.cuda(), seg.cuda()
    out = [eng.prefill(tc[:, :T0], sc[:, :T0]).float().clone()]
    for t in range(K):
        out.append(eng.step(tc[:, T0 + t], sc[:, T0 + t]).float().clone())
    eng.check_persistent()
    got = torch.stack(out, 1).cpu()
    rng = float(ref.max() - ref.min())
    err = (got - ref).abs()
    e_max, e_steps = float(err.max()), err.amax(dim=(0, 2))
    top2 = ref.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2 * e_max
    print('[one-launch GPT-2 decode vs oracle] n=%d: max |dlogit| %.4f = %.4f of the logit range (step 1: %.4f, step %d: %.4f); greedy ids compared at %.0f %% of positions'
          % (n, e_max, e_max / rng, float(e_steps[1]), K, float(e_steps[-1]), 100 * float(safe.float().mean())))
    assert e_max <= 0.05 * rng
    assert float(safe.float().mean()) > 0.05
    assert torch.equal(got.argmax(-1)[safe], ref.argmax(-1)[safe])


def test_gpt2_one_launch_step_fills_the_cache_to_its_last_row(monkeypatch):
    """Context up to the 2048-row window (the reference's max_dec_inp_len): the last step appends row 2047 and attends over all 2048 rows (4 full
    256-row sweeps); one more token does not fit and must be refused, not written past the cache."""
    from emo_disentanger_amd import inference as inf
    from emo_disentanger_amd._lib import EmoError
    g = torch.Generator().manual_seed(31)
    V, n, L, T0, K = 327, 4, 2, 2042, 6
    ptok = torch.randint(0, V - 1, (n, T0), generator=g).cuda()
    pseg = torch.randint(0, 2, (n, T0), generator=g).cuda()
    toks = torch.randint(0, V - 1, (n, K), generator=g).cuda()
    segs = torch.randint(0, 2, (n, K), generator=g).cuda()
    mb, _ = _gpt2(L, 'bf16')
    one, K1, V1 = _run_gpt2(mb, ptok, pseg, toks, segs, True, monkeypatch)
    chain, K0, V0 = _run_gpt2(mb, ptok, pseg, toks, segs, False, monkeypatch)
    rng = float(chain.max() - chain.min())
    assert float((one - chain).abs().max()) <= 0.02 * rng
    for a, b in zip(K1 + V1, K0 + V0):
        assert a.shape[2] == 2048 and float((a.float() - b.float()).norm() / b.float().norm()) <= 1e-2
    monkeypatch.setenv('EMO_DECODE_PERSISTENT', '1')
    eng = inf.make_engine(mb, n)
    eng.prefill(ptok, pseg)
    for t in range(K):
        eng.step(toks[:, t], segs[:, t])
    with pytest.raises(EmoError, match='past the positional-encoding table / the KV cache'):
        eng.step(toks[:, 0], segs[:, 0])


def test_gpt2_one_launch_step_refuses_what_it_was_not_built_for():
    from emo_disentanger_amd import ops
    from emo_disentanger_amd._lib import EmoError
    z = torch.zeros(1024, device='cuda')
    zi = torch.zeros(16, dtype=torch.int64, device='cuda')
    ws = torch.zeros(ops.lib.emo_performer_decode_step_workspace_bytes() // 8, dtype=torch.int64, device='cuda')
    lg = torch.zeros(4, 327, device='cuda')
    with pytest.raises(EmoError, match='built for d_model 512'):
        ops.gpt2_decode_step(zi, 1, zi, None, z, None, z, 1.0, 0, None, z[:512], 2048, z, z, 327, lg, 4, 256, 8, 2048, ws)
    with pytest.raises(EmoError, match='KV cache of <= 2048 rows'):
        ops.gpt2_decode_step(zi, 1, zi, None, z, None, z, 1.0, 0, None, z, 4096, z, z, 327, lg, 4, 512, 8, 2048, ws)
    lg6 = torch.zeros(6, 327, device='cuda')
    with pytest.raises(EmoError, match='multiple of 4'):
        ops.gpt2_decode_step(zi, 1, zi, None, z, None, z, 1.0, 0, None, z, 2048, z, z, 327, lg6, 6, 512, 8, 2048, ws)
