"""Parity WITH DROPOUT ON (the mode bench.py times).  The product regenerates every keep decision from a counter-based
hash; the tests read the multipliers out (tests/dropmask.py, and — for the attention probabilities — out of the attention kernels themselves
through an identity V) and run the oracle with the same draw (oracle.model_ref `masks=`), so loss, logits and every parameter gradient are
compared element for element instead of "finite and seeded".

Reference semantics: the three self.dropout sites of upstream TransformerEncoderLayer.forward reached from
stage2_accompaniment/model/fast_transformer_decoder.py:45-51, emb_dropout (music_performer.py:61-62), and for GPT-2 attn_dropout on the
probabilities (HF GPT2Attention._attn via music_gpt2.py:42-51,86), resid_dropout and the MLP dropout.
Tolerances: fp32 parity mode — loss 1e-4 (north_star), logits 3e-4, gradients 2e-3 of the largest gradient; bf16 — as the dropout-off tests."""
import math

import numpy as np
import pytest
import torch

from dropmask import export_masks, site_multipliers, slice_batch

pytestmark = pytest.mark.gpu
P = 0.1


def _ops():
    from emo_disentanger_amd import ops
    return ops


def _r(*shape, seed=0, dt=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt)


# ------------------------------------------------------------------------------------------------------ attention-probability dropout
def attention_keep_from_kernel(B, T, H, dh, p, seed, offset, dt):
    """Keep matrix [B, H, T, T] (bool, causal part) as the attention FORWARD kernel applies it: q = k = 0 makes every causal probability of
    row i exactly 1 / (i + 1); V = the identity on key block c (rows 64 c .. 64 c + dh - 1, zero elsewhere) makes out[i, :] the dropped
    probabilities of those keys.  out * (i + 1) is 0 (dropped) or 1 / (1 - p) (kept)."""
    ops = _ops()
    HD = H * dh
    keep = torch.zeros(B, H, T, T, dtype=torch.bool)
    rows = (torch.arange(T) + 1).double().view(1, T, 1, 1)
    for c in range((T + dh - 1) // dh):
        qkv = torch.zeros(B, T, 3, H, dh)
        n = min(dh, T - c * dh)
        qkv[:, c * dh:c * dh + n, 2] = torch.eye(dh)[:n].view(1, n, 1, dh)
        g = qkv.view(B * T, 3 * HD).to(dt).cuda()
        out, _ = ops.softmax_attn_fwd(g[:, :HD], g[:, HD:2 * HD], g[:, 2 * HD:], B, T, H, p_drop=p, seed=seed, offset=offset)
        o = out.double().cpu().view(B, T, H, dh) * rows * (1.0 - p)                  # -> 0 or 1 (bf16: within 1 %)
        assert bool(((o.abs() < 0.02) | ((o - 1).abs() < 0.02)).all())
        keep[:, :, :, c * dh:c * dh + n] = (o[..., :n] > 0.5).permute(0, 2, 1, 3)
    return keep


def _attn_ref(qkv, dout, B, T, H, dh, mult):
    """fp64 HF GPT2Attention._attn with the GIVEN dropout multipliers [B, H, T, T] + autograd backward."""
    HD = H * dh
    q, k, v = [qkv[:, i * HD:(i + 1) * HD].double().view(B, T, H, dh).permute(0, 2, 1, 3).clone().requires_grad_(True) for i in range(3)]
    w = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    w = w.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float('-inf')).softmax(-1)
    if mult is not None:
        w = w * mult.double()
    ref = (w @ v).permute(0, 2, 1, 3).reshape(B * T, HD)
    ref.backward(dout.double())
    back = lambda g: g.permute(0, 2, 1, 3).reshape(B * T, HD)
    return ref.detach(), back(q.grad), back(k.grad), back(v.grad)


@pytest.mark.parametrize('kernels', ['generic', '32x32'])
@pytest.mark.parametrize('B,T,H,dh,dt', [(2, 64, 2, 64, torch.float32), (2, 128, 2, 64, torch.bfloat16), (1, 256, 3, 64, torch.bfloat16),
                                         (1, 48, 4, 32, torch.float32), (2, 150, 2, 64, torch.bfloat16)])
def test_attention_probability_dropout_vs_fp64_with_kernel_mask(kernels, B, T, H, dh, dt, monkeypatch):
    """out, dQ, dK, dV of BOTH kernel sets (generic 64 x 64 tiles; 32 x 32 x 16 tiles where the shape qualifies: bf16, d_head 64, T % 128 == 0)
    under dropout against an fp64 reference that uses the mask READ OUT OF THE FORWARD KERNEL, which must also be the mask emo_dropout_apply
    exports for the site (that is what the model-level tests feed the oracle)."""
    ops = _ops()
    monkeypatch.setenv('EMO_SATTN32', '1' if kernels == '32x32' else '0')
    if kernels == '32x32' and not (dt == torch.bfloat16 and dh == 64 and T % 128 == 0):
        pytest.skip('shape runs on the generic kernels only')
    p, seed, off = 0.2, 5, 4105
    HD = H * dh
    keep = attention_keep_from_kernel(B, T, H, dh, p, seed, off, dt)
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    exported = site_multipliers((B, H, T, T), p, seed, off)
    assert torch.equal(keep & causal, (exported > 0) & causal)
    frac = float(keep[..., causal].float().mean())
    assert abs(frac - (1 - p)) < 0.02
    mult = keep.double() / (1.0 - p)
    qkv, dout = _r(B * T, 3 * HD, seed=51, dt=dt), _r(B * T, HD, seed=52, dt=dt)
    ref, rq, rk, rv = _attn_ref(qkv, dout, B, T, H, dh, mult)
    qc = qkv.cuda()
    out, lse = ops.softmax_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], B, T, H, p_drop=p, seed=seed, offset=off)
    dq, dk, dv = ops.softmax_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], out, dout.cuda(), lse, B, T, H, p_drop=p, seed=seed, offset=off)
    tol = 2e-5 if dt == torch.float32 else 3e-2
    gs = max(float(rq.abs().max()), float(rk.abs().max()), float(rv.abs().max()))
    for name, got, want, s, mul in (('out', out, ref, float(ref.abs().max()), 1.0), ('dq', dq, rq, gs, 3.0), ('dk', dk, rk, gs, 3.0), ('dv', dv, rv, gs, 3.0)):
        err = float((got.double().cpu() - want).abs().max())
        assert err <= mul * tol * s, (name, err, s)


@pytest.mark.parametrize('B,T,H', [(2, 256, 2), (1, 640, 3)])
def test_attention_32x32_backward_vs_fp64_without_dropout(B, T, H, monkeypatch):
    """The 32 x 32 x 16 backward (dK/dV pass) against fp64 at p = 0 (r03 compared only its forward with fp64)."""
    ops = _ops()
    monkeypatch.setenv('EMO_SATTN32', '1')
    dt, dh = torch.bfloat16, 64
    HD = H * dh
    qkv, dout = _r(B * T, 3 * HD, seed=61, dt=dt), _r(B * T, HD, seed=62, dt=dt)
    ref, rq, rk, rv = _attn_ref(qkv, dout, B, T, H, dh, None)
    qc = qkv.cuda()
    out, lse = ops.softmax_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], B, T, H)
    dq, dk, dv = ops.softmax_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], out, dout.cuda(), lse, B, T, H)
    gs = max(float(rq.abs().max()), float(rk.abs().max()), float(rv.abs().max()))
    for name, got, want, s, mul in (('out', out, ref, float(ref.abs().max()), 1.0), ('dq', dq, rq, gs, 3.0), ('dk', dk, rk, gs, 3.0), ('dv', dv, rv, gs, 3.0)):
        err = float((got.double().cpu() - want).abs().max())
        assert err <= mul * 3e-2 * s, (name, err, s)


# ------------------------------------------------------------------------------------------------------ model level
PERF_CASES = [dict(V=60, L=2, H=4, d=64, dff=128, nf=32, B=2, T=70, seed=3, scale=3.0),
              dict(V=327, L=2, H=8, d=256, dff=256, nf=128, B=1, T=96, seed=4, scale=2.0),
              dict(V=327, L=1, H=2, d=128, dff=256, nf=128, B=2, T=130, seed=5, scale=2.0)]
GPT2_CASES = [dict(V=40, L=2, H=4, d=64, dff=128, B=2, T=16, seed=1, scale=2.0),
              dict(V=327, L=2, H=4, d=64, dff=256, B=2, T=128, seed=2, scale=2.0),
              dict(V=327, L=1, H=8, d=512, dff=2048, B=1, T=256, seed=6, scale=1.5)]


def _build(kind, c, dtype, p):
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from oracle.weights import make_state_dict
    sd = make_state_dict(kind, c['V'], c['L'], c['H'], c['d'], c['dff'], favor_feature_dims=c.get('nf'), seed=c['seed'], scale=c['scale'])
    kw = dict(dropout=p, use_segment_emb=True, n_segment_types=2, compute_dtype=dtype)
    m = (MusicPerformer(c['V'], c['L'], c['H'], c['d'], c['dff'], c['d'], favor_feature_dims=c['nf'], redraw='fixed', **kw) if kind == 'performer'
         else MusicGPT2(c['V'], c['L'], c['H'], c['d'], c['dff'], c['d'], **kw))
    m.load_state_dict(sd)
    return m.cuda().train(), sd


def _step_and_compare(kind, c, dtype, b, n_fwd=1):
    """n_fwd-th training forward after set_dropout_seed: HIP path with dropout P vs the oracle with the exported masks of THAT forward."""
    from oracle import model_ref
    m, sd = _build(kind, c, dtype, P)
    seed = 991
    m.set_dropout_seed(seed)
    x, seg, tgt = b['dec_input'].cuda(), b['track_mask'].cuda(), b['dec_target'].cuda()
    for _ in range(n_fwd - 1):                                     # earlier forwards only advance the per-forward dropout base
        with torch.no_grad():
            m(x, seg_inp=seg)
    logits = m(x, seg_inp=seg)
    loss = m.compute_loss(logits, tgt)['total_loss']
    loss.backward()
    B, T = x.shape
    masks = export_masks(kind, P, seed, 4096 * n_fwd, B, T, c['d'], c['dff'], c['H'], c['L'])
    kw = dict(form='quadratic') if kind == 'performer' else {}
    rloss, rlogits, rgrads = model_ref.loss_and_grads(kind, sd, b, c['V'], c['L'], c['H'], c['d'], p_drop=P, training=True, masks=masks, **kw)
    # the draw matters: the same oracle WITHOUT dropout is far away
    nloss, nlogits, _ = model_ref.loss_and_grads(kind, sd, b, c['V'], c['L'], c['H'], c['d'], **kw)
    assert float((nlogits - rlogits).abs().max()) > 0.05
    return m, float(loss.detach()), logits.detach().cpu(), float(rloss), rlogits, rgrads


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('ci', range(len(PERF_CASES)))
def test_performer_dropout_on_matches_oracle_with_exported_masks(ci, dtype):
    from oracle.weights import synthetic_batch
    c = PERF_CASES[ci]
    b = synthetic_batch(c['V'], c['B'], c['T'], seed=77, realistic_targets=True)
    b['dec_target'][:, -3:] = 5
    m, loss, logits, rloss, rlogits, rgrads = _step_and_compare('performer', c, dtype, b, n_fwd=1 + ci)
    lt, gt = (1e-4, 2e-3) if dtype == 'fp32' else (3e-2, 6e-2)
    assert abs(loss - rloss) <= lt
    if dtype == 'fp32':
        np.testing.assert_allclose(logits.numpy(), rlogits.numpy(), rtol=3e-4, atol=3e-4)
        top2 = rlogits.topk(2, -1).values
        safe = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert (logits.argmax(-1)[safe] == rlogits.argmax(-1)[safe]).all()
    gmax = max(float(g.abs().max()) for g in rgrads.values())
    for k, p in m.named_parameters():
        err = float((p.grad.cpu() - rgrads[k]).abs().max())
        assert err <= gt * gmax, (k, err, gmax)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('ci', range(len(GPT2_CASES)))
def test_gpt2_dropout_on_matches_oracle_with_exported_masks(ci, dtype):
    """All four GPT-2 dropout sites, incl. the attention probabilities (case 2 = d512 / d_head 64 / T % 128 == 0: the 32 x 32 kernels in bf16)."""
    from oracle.weights import synthetic_batch
    c = GPT2_CASES[ci]
    b = synthetic_batch(c['V'], c['B'], c['T'], seed=78, realistic_targets=c['T'] >= 64)      # (T = 16: one segment run would pad every target)
    m, loss, logits, rloss, rlogits, rgrads = _step_and_compare('gpt2', c, dtype, b, n_fwd=2)
    lt, gt = (1e-4, 2e-3) if dtype == 'fp32' else (3e-2, 6e-2)
    assert abs(loss - rloss) <= lt
    if dtype == 'fp32':
        np.testing.assert_allclose(logits.numpy(), rlogits.numpy(), rtol=3e-4, atol=3e-4)
    gmax = max(float(g.abs().max()) for g in rgrads.values())
    for k, p in m.named_parameters():
        err = float((p.grad.cpu() - rgrads[k]).abs().max())
        assert err <= gt * gmax, (k, err, gmax)


BENCH_SHAPE = dict(V=327, L=12, H=8, d=512, dff=2048, nf=128, seed=0, scale=1.0)


def test_performer_dropout_on_at_benchmark_shape_fp32():
    """BASELINE configs[1]'s model (d512 / L12 / H8 / F128) at T = 2048, B = 1, dropout 0.1 ON, fp32 parity mode vs the oracle with the exported
    masks: loss 1e-4, logits 5e-4, every gradient element within 2e-3 of the largest gradient."""
    from oracle.weights import synthetic_batch
    c = dict(BENCH_SHAPE, B=1, T=2048)
    b = synthetic_batch(c['V'], 1, 2048, seed=1234)
    m, loss, logits, rloss, rlogits, rgrads = _step_and_compare('performer', c, 'fp32', b)
    gmax = max(float(g.abs().max()) for g in rgrads.values())
    gerr = max(float((p.grad.cpu() - rgrads[k]).abs().max()) for k, p in m.named_parameters()) / gmax
    lerr = float((logits - rlogits).abs().max())
    print('[dropout-on bench-shape parity] fp32: |dloss| %.3g  max|dlogit| %.3g  max|dgrad|/max|g| %.3g' % (abs(loss - rloss), lerr, gerr))
    assert abs(loss - rloss) <= 1e-4 and lerr <= 5e-4 and gerr <= 2e-3


@pytest.mark.parametrize('kind,B', [('performer', 64), ('gpt2', 16)])
def test_timed_kernel_instances_dropout_on_vs_oracle_bf16(kind, B):
    """The kernel INSTANCES bench.py times (bf16, B x T = 64 x 2048 tokens: A-stationary GEMMs with the fused ReLU + dropout + 1-bit mask and the
    bit-mask dgrad, 256 x 256 tile with dropout + residual, LayerNorm backward re-masking, FAVOR+ slice kernels; GPT-2 at its bench batch of
    16: the 32 x 32 attention kernels with probability dropout) run only at >= 32768 tokens, which the CPU oracle cannot do in a test.  With every
    target outside sequence `pick` set to the pad id, the full batch's loss and parameter gradients are those of that one sequence under ITS rows
    of the batch's dropout masks: the oracle runs that sequence alone (T = 2048) with the exported mask rows.  bf16 bounds = the dropout-off
    bounds of test_performer_at_benchmark_shape_matches_oracle (measured + 50 %): a mask that differs between a forward and a backward kernel, or
    from the exported one, on even 1 % of the elements is far outside them (checked below by perturbing the oracle's masks)."""
    from oracle import model_ref
    from oracle.weights import synthetic_batch
    c = dict(BENCH_SHAPE, scale=2.5 if kind == 'performer' else 1.0)
    T, pick, pad, seed = 2048, 5, c['V'] - 1, 4242
    m, sd = _build(kind, c, 'bf16', P)
    m.set_dropout_seed(seed)
    b = synthetic_batch(c['V'], B, T, seed=4321)
    tgt = torch.full_like(b['dec_target'], pad)
    tgt[pick] = b['dec_target'][pick]
    logits = m(b['dec_input'].cuda(), seg_inp=b['track_mask'].cuda())
    loss = m.compute_loss(logits, tgt.cuda())['total_loss']
    loss.backward()
    D, dff, H, L = c['d'], c['dff'], c['H'], c['L']
    # masks of sequence `pick` only: export the rows [pick T, (pick + 1) T) of every site (the flat element index is row-major over the batch)
    ops = _ops()

    def rows(width, off, per_row=1):
        full = ops.dropout_apply(torch.ones(B * T * per_row, width, device='cuda'), P, seed, off)
        return full[pick * T * per_row:(pick + 1) * T * per_row].cpu()
    masks = {'emb': rows(D, 4096).view(1, T, D)}
    for l in range(L):
        off = 4096 + 8 * (l + 1)
        if kind == 'performer':
            masks['L%d.attn_out' % l] = rows(D, off + 1).view(1, T, D)
            masks['L%d.ffn_hidden' % l] = rows(dff, off + 2).view(1, T, dff)
            masks['L%d.ffn_out' % l] = rows(D, off + 3).view(1, T, D)
        else:
            masks['L%d.attn_prob' % l] = rows(T, off + 1, per_row=H).view(1, H, T, T)
            masks['L%d.attn_out' % l] = rows(D, off + 2).view(1, T, D)
            masks['L%d.mlp_out' % l] = rows(D, off + 3).view(1, T, D)
    b1 = {k: (v[pick:pick + 1] if torch.is_tensor(v) else v) for k, v in b.items()}
    rloss, rlogits, rgrads = model_ref.loss_and_grads(kind, sd, b1, c['V'], L, H, D, p_drop=P, training=True, masks=masks)
    lg = logits.detach()[pick].cpu()
    loss_err, logit_err = abs(float(loss) - float(rloss)), float((lg - rlogits[0]).abs().max())
    gmax = max(float(g.abs().max()) for g in rgrads.values())
    gerr = max(float((p.grad.cpu() - rgrads[k]).abs().max()) for k, p in m.named_parameters()) / gmax
    gl2 = max(float((p.grad.cpu() - rgrads[k]).norm() / rgrads[k].norm().clamp_min(1e-12)) for k, p in m.named_parameters())
    # sensitivity: the same oracle with 1 % of ONE layer's FFN / MLP-output mask entries flipped
    flip = dict(masks)
    site = 'L%d.%s' % (L // 2, 'ffn_hidden' if kind == 'performer' else 'mlp_out')
    g = torch.Generator().manual_seed(1)
    sel = torch.rand(masks[site].shape, generator=g) < 0.01
    flip[site] = torch.where(sel, (1.0 / (1.0 - P)) - masks[site], masks[site])
    floss, flogits, fgrads = model_ref.loss_and_grads(kind, sd, b1, c['V'], L, H, D, p_drop=P, training=True, masks=flip)
    fl2 = max(float((fgrads[k] - rgrads[k]).norm() / rgrads[k].norm().clamp_min(1e-12)) for k in rgrads)
    print('[dropout-on timed instances] %s bf16 B=%d: |dloss| %.3g  max|dlogit| %.3g  max|dgrad|/max|g| %.3g  worst rel. L2 %.3g   (oracle with 1 %% of one '
          'mask flipped: rel. L2 %.3g, max|dlogit| %.3g)' % (kind, B, loss_err, logit_err, gerr, gl2, fl2, float((flogits - rlogits).abs().max())))
    assert loss_err <= 1e-3 and logit_err <= (5e-2 if kind == 'performer' else 0.3) and gerr <= 0.11 and gl2 <= 0.16, (loss_err, logit_err, gerr, gl2)
