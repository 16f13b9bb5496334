"""GPU: the stage-1 (Transformer-XL) training path with dropout ON — the mode it trains and is timed in — against the oracle under the SAME
draw.  Chain of evidence: the imported reference's F.dropout run == oracle/txl_ref.py with the replayed masks (CPU, tests/test_oracle_txl.py,
fixtures txl_dropout_*.npz); here: the HIP path == that oracle with the multipliers the kernels themselves use (exported per site through
emo_dropout_apply; the attention-probability site is read out of relattn_fwd itself).  Covers the reference's eight sites incl. dropatt followed
by the renormalisation p / (sum p + 1e-8) (/root/reference/stage1_compose/model/optimus_txl_decoder.py:361-363) and the o_net output dropout
(:375), with the cross-layer backward fusions of model/plain_transformer.py (EMO_S1_FUSE) both on and off."""
import json
import os

import numpy as np
import pytest
import torch

import dropmask

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
DROP_CASES = sorted(json.load(open(os.path.join(G, 'txl_dropout_manifest.json'))).items())


def _ops():
    from emo_disentanger_amd import ops
    return ops


def relattn_keep_from_kernel(B, T, H, dh, p, seed, offset, dt):
    """Multipliers [B, H, T, T] (0 or 1 / (1 - p) on the causal part, 0 above the diagonal) as relattn_fwd applies them: q = k = 0 and zero
    biases make every causal probability of row i exactly 1 / (i + 1); after dropout and the renormalisation the weight of key j is
    keep_ij / (number of kept keys of row i); V = the identity on key block c makes out[i, :] those weights: > 0 <=> kept."""
    ops = _ops()
    HD = H * dh
    keep = torch.zeros(B, H, T, T, dtype=torch.bool)
    R = torch.zeros(T, HD, dtype=dt, device='cuda')
    zb = torch.zeros(H, dh, device='cuda')
    rows = (torch.arange(T) + 1).double().view(1, T, 1, 1)
    for c in range((T + dh - 1) // dh):
        qkv = torch.zeros(B, T, 3, H, dh)
        n = min(dh, T - c * dh)
        qkv[:, c * dh:c * dh + n, 2] = torch.eye(dh)[:n].view(1, n, 1, dh)
        g = qkv.view(B * T, 3 * HD).to(dt).cuda()
        out, _, _ = ops.relpos_attn_fwd(g[:, :HD], g[:, HD:2 * HD], g[:, 2 * HD:], R, zb, zb, B, T, H, p_drop=p, seed=seed, offset=offset)
        o = out.double().cpu().view(B, T, H, dh) * rows                              # kept: (i + 1) / n_kept >= 1; dropped: 0
        assert bool(((o.abs() < 1e-3) | (o > 0.98)).all())
        keep[:, :, :, c * dh:c * dh + n] = (o[..., :n] > 0.5).permute(0, 2, 1, 3)
    return keep.float() / (1.0 - p)


def _relattn_ref_masked(q, k, v, R, u, vb, mult, renorm=True):
    """fp64 RelPartialLearnableMultiHeadAttn core (:340-366) with explicit distances and the GIVEN dropout multipliers [B, H, T, T]."""
    B, T, H, dh = q.shape
    AC = torch.einsum('bihd,bjhd->bhij', q + u, k)
    idx = (torch.arange(T)[:, None] - torch.arange(T)[None, :]).clamp(min=0)
    BD = torch.einsum('bihd,ijhd->bhij', q + vb, R[idx])
    sc = (AC + BD) / dh ** 0.5
    sc = sc.masked_fill(torch.triu(torch.ones(T, T), 1).bool(), -float('inf'))
    p = torch.softmax(sc, -1)
    if mult is not None:
        p = p * mult.double()                                                        # dropatt (:361)
    if renorm:
        p = p / (p.sum(-1, keepdim=True) + 1e-8)                                     # renormalisation (:363)
    return torch.einsum('bhij,bjhd->bihd', p, v)


@pytest.mark.parametrize('B,T,H,dh,dt', [(2, 64, 2, 64, torch.float32), (2, 128, 2, 64, torch.bfloat16), (1, 150, 3, 32, torch.float32),
                                         (1, 256, 2, 64, torch.bfloat16), (2, 33, 2, 16, torch.float32), (1, 512, 1, 64, torch.bfloat16)])
def test_relattn_kernels_with_dropout_and_renormalisation_vs_fp64(B, T, H, dh, dt):
    """out, dq (content + relative), dk, dv, dR, d r_w_bias, d r_r_bias of the four relattn kernels at p = 0.2 against fp64 autograd of the
    reference formula under the mask read out of the forward kernel, which must equal the mask emo_dropout_apply exports for the site."""
    ops = _ops()
    p, seed, off = 0.2, 7, 4105
    HD = H * dh
    mult = relattn_keep_from_kernel(B, T, H, dh, p, seed, off, dt)
    exported = dropmask.site_multipliers((B, H, T, T), p, seed, off)
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    assert torch.equal((mult != 0)[..., causal], (exported != 0)[..., causal])       # same element -> hash mapping as the exported site
    rate = float((mult != 0)[..., causal].float().mean())
    assert abs(rate - (1 - p)) < 0.03
    g = torch.Generator().manual_seed(1)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dt)
    qkv, R = rn(B * T, 3 * HD, sc=0.7), rn(T, HD, sc=0.7)
    u, vb = rn(H, dh, sc=0.3).float(), rn(H, dh, sc=0.3).float()
    leaf = [t.double().requires_grad_(True) for t in (qkv, R, u, vb)]
    q, k, v = [leaf[0][:, i * HD:(i + 1) * HD].view(B, T, H, dh) for i in range(3)]
    ref = _relattn_ref_masked(q, k, v, leaf[1].view(T, H, dh), leaf[2], leaf[3], mult)
    dout = rn(B, T, H, dh)
    ref.backward(dout.double())
    qc = qkv.cuda()
    out, lse, zden = ops.relpos_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], R.cuda(), u.cuda(), vb.cuda(), B, T, H, p_drop=p, seed=seed, offset=off)
    dqkv, dR, du, dvb = ops.relpos_attn_bwd(qc, R.cuda(), u.cuda(), vb.cuda(), out, dout.view(B * T, HD).cuda(), lse, zden, B, T, H, p_drop=p, seed=seed, offset=off)
    tol = 2e-5 if dt == torch.float32 else 3e-2

    def close(got, want, scale, mult_=1.0, what=''):
        err = float((got.detach().double().cpu() - want.detach().double()).abs().max())
        assert err <= mult_ * tol * max(scale, 1e-6), '%s: max err %.3e vs scale %.3e' % (what, err, scale)
    close(out.view(B, T, H, dh), ref, float(ref.abs().max()), 3, 'out')
    gs = float(leaf[0].grad.abs().max())
    close(dqkv[:, 2 * HD:], leaf[0].grad[:, 2 * HD:], gs, 4, 'dv')
    close(dqkv[:, HD:2 * HD], leaf[0].grad[:, HD:2 * HD], gs, 4, 'dk')
    close(dqkv[:, :HD], leaf[0].grad[:, :HD], gs, 4, 'dq')
    close(dR, leaf[1].grad, float(leaf[1].grad.abs().max()), 6, 'dR')
    close(du, leaf[2].grad, float(leaf[2].grad.abs().max()), 8, 'd r_w_bias')
    close(dvb, leaf[3].grad, float(leaf[3].grad.abs().max()), 8, 'd r_r_bias')
    # the renormalisation is not a no-op here and the test sees it: plain inverted dropout (no renormalisation) is far outside the tolerance
    with torch.no_grad():
        plain = _relattn_ref_masked(q, k, v, leaf[1].view(T, H, dh), leaf[2], leaf[3], mult, renorm=False)
    assert float((plain - ref).abs().max()) > 0.1 * float(ref.abs().max()) > 3 * tol * float(ref.abs().max())


def _step(c, dtype, x, tgt, p, seed, fuse, monkeypatch):
    """One training forward + backward of the product's PlainTransformer with dropout p; returns the model, loss, logits (CPU)."""
    from emo_disentanger_amd.model import plain_transformer as pt
    from oracle.txl_ref import make_state_dict_txl
    monkeypatch.setattr(pt, '_FUSE_BELOW', bool(fuse))
    sd = make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    m = pt.PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], 0, c['T'], dec_dropout=p, pre_lnorm=True, compute_dtype=dtype)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.set_dropout_seed(seed)
    logits, _ = m(x.cuda(), tuple())
    loss = m.compute_loss(logits, tgt.cuda())['total_loss']
    loss.backward()
    return m, sd, float(loss.detach()), logits.detach().float().cpu()


_ORACLE = {}


def _oracle(c, sd, x, tgt, p, seed, dtype):
    """Oracle loss / logits / gradients under the product's masks (cached per shape: the masks depend on (seed, site, index) only, not on the
    compute dtype or the backward schedule)."""
    from oracle import txl_ref
    key = (tuple(sorted((k, v) for k, v in c.items() if not isinstance(v, (list, dict)))), tuple(x.shape), p, seed)
    if key in _ORACLE:
        return _ORACLE[key]
    B, T, D, H, L, dh = x.shape[1], x.shape[0], c['d'], c['H'], c['L'], c['d'] // c['H']
    dt = torch.float32 if dtype == 'fp32' else torch.bfloat16
    base = 4096                                                                      # first forward after set_dropout_seed
    exported = lambda l: dropmask.site_multipliers((B, H, T, T), p, seed, base + 8 * (l + 1) + 1)
    masks = dropmask.export_txl_masks(p, seed, base, B, T, D, c['dff'], H, L, exported)
    # the attention site once more from the kernel itself (layer 0): the exported mask IS what relattn_fwd applies at this shape
    k0 = relattn_keep_from_kernel(B, T, H, dh, p, seed, base + 8 + 1, dt)
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    assert torch.equal((k0 != 0)[..., causal], (exported(0) != 0)[..., causal])
    _ORACLE[key] = txl_ref.loss_and_grads(sd, x, tgt, L, H, masks=masks)
    return _ORACLE[key]


def _compare(m, loss, logits, rloss, rlogits, rgrads, dtype, gt32=2e-3):
    # gradient bounds: fp32 2e-3 of the largest gradient element; bf16 1e-1 — measured on MI355X with dropout OFF the bf16 path is already at
    # 0.033 / 0.067 on the two fixture shapes (word_emb: every layer's bf16 backward rounding ends in it, times emb_scale) and with dropout ON at
    # 0.047 / 0.073 (tools/diag_s1_dropout.py prints both): the masks add nothing measurable, which is what this test is about
    lt, gt = (1e-4, gt32) if dtype == 'fp32' else (3e-2, 1e-1)
    assert abs(loss - float(rloss)) <= lt, (loss, float(rloss))
    scale = float(rlogits.abs().max())
    lerr = float((logits - rlogits).abs().max())
    assert lerr <= (5e-4 if dtype == 'fp32' else 6e-2) * max(scale, 1.0), lerr
    gmax = max(float(g.abs().max()) for g in rgrads.values())
    worst = ('', 0.0)
    for k, prm in m.named_parameters():
        err = float((prm.grad.cpu() - rgrads[k]).abs().max())
        worst = max(worst, (k, err / gmax), key=lambda t: t[1])
        assert err <= gt * gmax, (k, err, gmax)
    return abs(loss - float(rloss)), lerr, worst


@pytest.mark.parametrize('fuse', [1, 0])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('name,c', DROP_CASES)
def test_stage1_dropout_on_matches_oracle_with_exported_masks(name, c, dtype, fuse, monkeypatch):
    """Fixture shapes (the ones the masked oracle is pinned on by the imported reference): loss 1e-4, logits 5e-4, every gradient element
    within 2e-3 of the largest gradient in fp32; the bf16 bounds of the stage-2 dropout tests in bf16."""
    g = np.load(os.path.join(G, name + '.npz'))
    x, tgt = torch.from_numpy(g['x']), torch.from_numpy(g['tgt'])
    p, seed = c['p'], 31
    m, sd, loss, logits = _step(c, dtype, x, tgt, p, seed, fuse, monkeypatch)
    rloss, rlogits, rgrads = _oracle(c, sd, x, tgt, p, seed, dtype)
    _compare(m, loss, logits, rloss, rlogits, rgrads, dtype)
    # the masks matter: the dropout-off oracle is far outside the bound
    from oracle import txl_ref
    off, _ = txl_ref.forward(sd, x, c['L'], c['H'])
    assert float((off - rlogits).abs().max()) > 1e-2 * float(rlogits.abs().max())


@pytest.mark.parametrize('fuse', [1, 0])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_stage1_dropout_on_at_bench_shape_matches_oracle(dtype, fuse, monkeypatch):
    """The shape bench.py's stage-1 leg times (BASELINE configs[4]: d512 / L12 / H8 / d_ff 2048, tgt_len 512, batch 4, dropout 0.1)."""
    c = dict(V=200, L=12, H=8, d=512, dff=2048, T=512, seed=5, scale=2.0)
    B, p, seed = 4, 0.1, 77
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], B), dtype=np.int64))
    tgt = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], B), dtype=np.int64))
    tgt[-37:, 1] = c['V'] - 1
    m, sd, loss, logits = _step(c, dtype, x, tgt, p, seed, fuse, monkeypatch)
    rloss, rlogits, rgrads = _oracle(c, sd, x, tgt, p, seed, dtype)
    # fp32 gradient bound 4e-3 here: with dropout OFF this shape measures 2.5e-3 (loss 2e-6, logits 4e-6: the forward is exact; the backward
    # differs where a ReLU pre-activation of the 12 x 4 M hidden units lies within rounding of 0 and the two sides disagree on its sign), with
    # dropout ON 2.2e-3 (tools/diag_s1_dropout.py)
    dl, lerr, worst = _compare(m, loss, logits, rloss, rlogits, rgrads, dtype, gt32=4e-3)
    print('[stage-1 dropout-on bench-shape parity] %s fuse=%d: |dloss| %.3g  max|dlogit| %.3g  worst grad %s %.3g of max|g|' % (dtype, fuse, dl, lerr, worst[0], worst[1]))


def test_stage1_fused_and_unfused_backward_agree_bf16(monkeypatch):
    """EMO_S1_FUSE moves the output-dropout re-mask and the CoreNet.3 bias gradient of layer l-1 into the LayerNorm backward of layer l: same
    masks, same sums — the two schedules must give the same gradients up to summation order."""
    c = dict(V=200, L=4, H=8, d=512, dff=2048, T=256, seed=6, scale=2.0)
    rng = np.random.default_rng(10)
    x = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], 4), dtype=np.int64))
    tgt = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], 4), dtype=np.int64))
    m1, _, l1, _ = _step(c, 'bf16', x, tgt, 0.1, 3, 1, monkeypatch)
    m0, _, l0, _ = _step(c, 'bf16', x, tgt, 0.1, 3, 0, monkeypatch)
    assert abs(l1 - l0) < 1e-5                                                       # (same forward; the loss reduction's summation order is free)
    gmax = max(float(p.grad.abs().max()) for p in m0.parameters())
    for (k, a), (_, b) in zip(m1.named_parameters(), m0.named_parameters()):
        assert float((a.grad - b.grad).abs().max()) <= 2e-3 * gmax, k
