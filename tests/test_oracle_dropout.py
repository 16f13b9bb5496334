"""CPU: the oracle's explicit-mask dropout (oracle.model_ref `masks=`, used by the dropout-ON GPU parity tests) IS the reference's F.dropout —
replaying torch's own Bernoulli draws as masks reproduces the F.dropout run bit for bit, forward and backward, for both backbones (the
GPT-2 / prologue / loss arithmetic around the sites is pinned by the golden vectors of the imported reference, tests/test_oracle_golden.py)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import model_ref
from oracle.weights import make_state_dict, synthetic_batch


def _replay_masks(kind, p, seed, B, T, D, dff, H, L):
    """The multipliers F.dropout draws, in the order the oracle forward calls it (same shapes -> same consumption of the CPU generator)."""
    torch.manual_seed(seed)
    draw = lambda *shape: F.dropout(torch.ones(*shape), p, True)
    m = {'emb': draw(B, T, D)}
    for l in range(L):
        if kind == 'performer':
            m['L%d.attn_out' % l], m['L%d.ffn_hidden' % l], m['L%d.ffn_out' % l] = draw(B, T, D), draw(B, T, dff), draw(B, T, D)
        else:
            m['L%d.attn_prob' % l], m['L%d.attn_out' % l], m['L%d.mlp_out' % l] = draw(B, H, T, T), draw(B, T, D), draw(B, T, D)
    return m


@pytest.mark.parametrize('kind', ['performer', 'gpt2'])
def test_explicit_masks_reproduce_F_dropout(kind):
    V, L, H, D, dff, B, T, p = 50, 2, 4, 64, 128, 2, 24, 0.1
    sd = make_state_dict(kind, V, L, H, D, dff, favor_feature_dims=32, seed=3, scale=2.0)
    b = synthetic_batch(V, B, T, seed=5)
    kw = dict(form='quadratic') if kind == 'performer' else {}
    torch.manual_seed(17)
    loss0, logits0, grads0 = model_ref.loss_and_grads(kind, sd, b, V, L, H, D, p_drop=p, training=True, **kw)
    masks = _replay_masks(kind, p, 17, B, T, D, dff, H, L)
    assert set(torch.unique(masks['emb']).tolist()) == {0.0, float(torch.tensor(1.0) / (1.0 - p))}
    loss1, logits1, grads1 = model_ref.loss_and_grads(kind, sd, b, V, L, H, D, p_drop=p, training=True, masks=masks, **kw)
    assert torch.equal(logits0, logits1) and torch.equal(loss0, loss1)
    for k in grads0:
        assert torch.allclose(grads0[k], grads1[k], rtol=0, atol=1e-6 * float(grads0[k].abs().max() + 1e-12)), k
    # eval mode / p = 0 ignore the masks
    loss2, logits2, _ = model_ref.loss_and_grads(kind, sd, b, V, L, H, D, p_drop=p, training=False, masks=masks, **kw)
    loss3, logits3, _ = model_ref.loss_and_grads(kind, sd, b, V, L, H, D, **kw)
    assert torch.equal(logits2, logits3) and not torch.equal(logits2, logits1)
