"""Performer oracle: upstream pytorch-fast-transformers is absent => parity UNPINNED upstream.
The restatement is checked by independent identities."""
import math

import numpy as np
import torch

from oracle import model_ref
from oracle.weights import make_state_dict, orthogonal_omega, synthetic_batch


def _qkv(N=2, L=37, H=3, dh=16, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(N, L, H, dh, generator=g) * scale for _ in range(3)]


def test_unpinned_three_forms_agree():
    q, k, v = _qkv()
    om = orthogonal_omega(16, 32, np.random.default_rng(3))
    a = model_ref.causal_linear_attention(q, k, v, om, form='prefix')
    b = model_ref.causal_linear_attention(q, k, v, om, form='quadratic')
    c = model_ref.causal_linear_attention(q, k, v, om, form='recurrent')
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
    assert torch.allclose(a, c, rtol=1e-4, atol=1e-5)


def test_unpinned_c_kernel_backward_matches_autograd_quadratic():
    q, k, v = [t.double().requires_grad_(True) for t in _qkv(N=1, L=19, H=2, dh=8, seed=1)]
    om = orthogonal_omega(8, 16, np.random.default_rng(4)).double()
    out_q = model_ref.causal_linear_attention(q, k, v, om, form='quadratic')
    w = torch.randn_like(out_q)
    gq = torch.autograd.grad((out_q * w).sum(), (q, k, v))
    q32, k32, v32 = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
    out_p = model_ref.causal_linear_attention(q32, k32, v32, om.float(), form='prefix')
    gp = torch.autograd.grad((out_p * w.float()).sum(), (q32, k32, v32))
    for a, b in zip(gq, gp):
        assert torch.allclose(a.float(), b, rtol=2e-3, atol=2e-4)


def test_unpinned_favor_approximates_softmax_kernel():
    dh, F = 16, 4096
    rng = np.random.default_rng(0)
    om = orthogonal_omega(dh, F, rng)
    q = torch.randn(64, dh) * 0.7
    k = torch.randn(64, dh) * 0.7
    est = model_ref.favor_features(q, om) @ model_ref.favor_features(k, om).T
    true = torch.exp(q @ k.T / math.sqrt(dh))
    assert ((est - true).abs() / true).mean() < 0.1


def test_unpinned_prefix_causality_and_model_shapes():
    V, L, H, d, dff = 50, 2, 4, 64, 128
    sd = make_state_dict('performer', V, L, H, d, dff, favor_feature_dims=32, seed=3, scale=3.0)
    b = synthetic_batch(V, 2, 48, seed=1)
    full = model_ref.forward('performer', sd, b['dec_input'], b['track_mask'], L, H, d)
    pre = model_ref.forward('performer', sd, b['dec_input'][:, :20], b['track_mask'][:, :20], L, H, d)
    assert full.shape == (2, 48, V)
    assert torch.allclose(full[:, :20], pre, rtol=1e-4, atol=1e-5)
    loss, _, grads = model_ref.loss_and_grads('performer', sd, b, V, L, H, d)
    assert torch.isfinite(loss) and all(torch.isfinite(g).all() for g in grads.values())
    # parameter registration order = Appendix D (optimizer-state compatibility)
    keys = [k for k in sd if 'decoder_layers.0.' in k]
    assert keys[0].endswith('feature_map.omega') and keys[1].endswith('query_projection.weight') and keys[-1].endswith('norm2.bias')


def test_omega_generator_statistics():
    om = orthogonal_omega(64, 128, np.random.default_rng(1)).double()
    gram = om.T @ om                                   # columns orthogonal (one 64x64 block)
    off = gram - torch.diag(torch.diag(gram))
    assert off.abs().max() < 1e-4
    # column norms are chi(64)-distributed row norms of a Gaussian block: mean ~ sqrt(63.5)
    assert abs(torch.diag(gram).sqrt().mean().item() - math.sqrt(63.5)) < 1.0
