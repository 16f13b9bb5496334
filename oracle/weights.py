"""ORACLE (test infrastructure). Deterministic, seed-driven weights and inputs.

Fixtures store only I/O; weights are regenerated from NumPy seeds so that the
golden files stay small.  Init rule follows the reference:
  stage2_accompaniment/model/transformer_helpers.py:24-40 (``weights_init``):
    Linear/Embedding ~ N(0, 0.01), bias 0, LayerNorm weight ~ N(1, 0.01);
  HF ``Conv1D`` (GPT-2) is not matched by ``weights_init`` and keeps HF's
  N(0, 0.02) init.
``scale`` > 1 multiplies the matrix std-devs to obtain "trained-like" weights
whose logits are not flat (needed for meaningful argmax parity).
State-dict key names/order follow SURVEY.md Appendix D.
"""
from collections import OrderedDict
import math

import numpy as np
import torch


def positional_encoding(d_embed, max_pos=12000):
    """transformer_helpers.py:43-55 — same torch fp32 expression (not float64)."""
    pe = torch.zeros(max_pos, d_embed)
    position = torch.arange(0, max_pos, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_embed, 2).float() * (-math.log(10000.0) / d_embed))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).transpose(0, 1).contiguous()  # [max_pos, 1, d]


def orthogonal_omega(d_head, n_dims, rng):
    """fast_transformers.feature_maps.fourier_features.orthogonal_random_matrix_
    (upstream, restated): per block of d_head columns, G ~ N(0,1), Q = qr(G),
    columns rescaled by the row norms of G.  Returns [d_head, n_dims//2] fp32."""
    cols = n_dims // 2
    w = np.zeros((d_head, cols), dtype=np.float64)
    start = 0
    while start < cols:
        end = min(start + d_head, cols)
        block = rng.standard_normal((d_head, d_head))
        norms = np.sqrt((block * block).sum(1))
        q, _ = np.linalg.qr(block)
        w[:, start:end] = q[:, : end - start] * norms[None, : end - start]
        start += d_head
    return torch.from_numpy(w.astype(np.float32))


def _normal(rng, shape, std, mean=0.0):
    return torch.from_numpy((rng.standard_normal(shape) * std + mean).astype(np.float32))


def make_state_dict(kind, n_token, n_layer, n_head, d_model, d_ff, d_embed=None,
                    n_segment_types=2, favor_feature_dims=None, seed=0, scale=1.0,
                    max_pos=12000, with_omega=True):
    """kind in {'performer','gpt2'}.  Key order == reference registration order."""
    d_embed = d_embed or d_model
    rng = np.random.default_rng(seed)
    lin, conv = 0.01 * scale, 0.02 * scale
    sd = OrderedDict()
    sd['token_emb.emb_lookup.weight'] = _normal(rng, (n_token, d_embed), lin)
    if d_embed != d_model:
        sd['token_emb.emb_proj.weight'] = _normal(rng, (d_model, d_embed), lin)
    sd['pe.pe'] = positional_encoding(d_embed, max_pos)
    sd['dec_out_proj.weight'] = _normal(rng, (n_token, d_model), lin)
    sd['dec_out_proj.bias'] = _normal(rng, (n_token,), 0.01 * (scale > 1.0))
    bstd = 0.01 * (scale > 1.0)  # reference init has zero biases; trained-like adds some
    if kind == 'performer':
        d_head = d_model // n_head
        fdim = favor_feature_dims or 2 * d_head
        for l in range(n_layer):
            p = 'transformer_decoder.decoder_layers.%d.' % l
            if with_omega:
                sd[p + 'attention.inner_attention.feature_map.omega'] = orthogonal_omega(
                    d_head, fdim, np.random.default_rng(100 + l + 1000 * seed))
            for nm in ('query', 'key', 'value', 'out'):
                sd[p + 'attention.%s_projection.weight' % nm] = _normal(rng, (d_model, d_model), lin)
                sd[p + 'attention.%s_projection.bias' % nm] = _normal(rng, (d_model,), bstd)
            sd[p + 'linear1.weight'] = _normal(rng, (d_ff, d_model), lin)
            sd[p + 'linear1.bias'] = _normal(rng, (d_ff,), bstd)
            sd[p + 'linear2.weight'] = _normal(rng, (d_model, d_ff), lin)
            sd[p + 'linear2.bias'] = _normal(rng, (d_model,), bstd)
            for nm in ('norm1', 'norm2'):
                sd[p + nm + '.weight'] = _normal(rng, (d_model,), 0.01, 1.0)
                sd[p + nm + '.bias'] = _normal(rng, (d_model,), bstd)
    elif kind == 'gpt2':
        for i in range(n_layer):
            p = 'transformer_decoder.%d.' % i
            sd[p + 'ln_1.weight'] = _normal(rng, (d_model,), 0.01, 1.0)
            sd[p + 'ln_1.bias'] = _normal(rng, (d_model,), bstd)
            sd[p + 'attn.c_attn.weight'] = _normal(rng, (d_model, 3 * d_model), conv)
            sd[p + 'attn.c_attn.bias'] = _normal(rng, (3 * d_model,), bstd)
            sd[p + 'attn.c_proj.weight'] = _normal(rng, (d_model, d_model), conv)
            sd[p + 'attn.c_proj.bias'] = _normal(rng, (d_model,), bstd)
            sd[p + 'ln_2.weight'] = _normal(rng, (d_model,), 0.01, 1.0)
            sd[p + 'ln_2.bias'] = _normal(rng, (d_model,), bstd)
            sd[p + 'mlp.c_fc.weight'] = _normal(rng, (d_model, d_ff), conv)
            sd[p + 'mlp.c_fc.bias'] = _normal(rng, (d_ff,), bstd)
            sd[p + 'mlp.c_proj.weight'] = _normal(rng, (d_ff, d_model), conv)
            sd[p + 'mlp.c_proj.bias'] = _normal(rng, (d_model,), bstd)
    else:
        raise NotImplementedError(kind)
    if n_segment_types:
        sd['segemb.emb_lookup.weight'] = _normal(rng, (n_segment_types, d_embed), lin)
        if d_embed != d_model:
            sd['segemb.emb_proj.weight'] = _normal(rng, (d_model, d_embed), lin)
    return sd


def synthetic_batch(n_token, B, T, seed=1234, realistic_targets=False):
    """SURVEY §8(d) synthetic EMOPIA-shaped batch; dict keys follow
    stage2_accompaniment/dataloader.py:221-231.  pad id = n_token-1, EOS = n_token-2."""
    rng = np.random.default_rng(seed)
    pad, eos = n_token - 1, n_token - 2
    inp = rng.integers(0, n_token - 1, size=(B, T), dtype=np.int64)
    seg = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        t, cur = 0, 0
        while t < T:
            run = int(rng.integers(8, 65))
            seg[b, t:t + run] = cur
            cur ^= 1
            t += run
    tgt = np.empty_like(inp)
    tgt[:, :-1] = inp[:, 1:]
    tgt[:, -1] = eos
    if realistic_targets:
        tgt[seg == 0] = pad
    z = np.zeros((B, T), dtype=np.int64)
    return {
        'id': torch.arange(B), 'dec_input': torch.from_numpy(inp), 'dec_target': torch.from_numpy(tgt),
        'track_mask': torch.from_numpy(seg), 'chord_idx': torch.from_numpy(z.copy()),
        'melody_idx': torch.from_numpy(z.copy()), 'length': torch.full((B,), T),
    }
