"""Parameter holders of the embedding prologue: the state-dict CONTRACT of
/root/reference/stage2_accompaniment/model/transformer_helpers.py (:24-40 weights_init,
:43-63 PositionalEncoding, :66-87 TokenEmbedding).  Two pieces here necessarily follow the reference expression for expression and are
not claimed as original work: the `pe` buffer of PositionalEncoding (it is part of the checkpoint — `pe.pe` — and must be bit-identical:
tests/test_host_logic.py compares it with rows dumped from the imported reference) and the class-name dispatch of weights_init (the init
rule IS the behaviour).  The arithmetic of the prologue runs in the fused HIP kernel ``emo_embed_fwd`` (gather + segment gather +
*sqrt(d) + PE + dropout in one pass); nothing in this file computes on the hot path."""
import math

import torch
from torch import nn


def weights_init(m):
    """Reference init rule (transformer_helpers.py:24-40): Linear/Embedding ~ N(0, 0.01), bias 0,
    LayerNorm weight ~ N(1, 0.01).  HF-style Conv1D modules are NOT matched and keep N(0, 0.02)."""
    classname = m.__class__.__name__
    if classname.find('Linear') != -1:
        if getattr(m, 'weight', None) is not None:
            nn.init.normal_(m.weight, 0.0, 0.01)
        if getattr(m, 'bias', None) is not None:
            nn.init.constant_(m.bias, 0.0)
    elif classname.find('Embedding') != -1:
        if hasattr(m, 'weight'):
            nn.init.normal_(m.weight, 0.0, 0.01)
    elif classname.find('LayerNorm') != -1:
        if hasattr(m, 'weight'):
            nn.init.normal_(m.weight, 1.0, 0.01)
        if getattr(m, 'bias', None) is not None:
            nn.init.constant_(m.bias, 0.0)


class PositionalEncoding(nn.Module):
    def __init__(self, d_embed, max_pos=12000):
        super().__init__()
        self.d_embed, self.max_pos = d_embed, max_pos
        pe = torch.zeros(max_pos, d_embed)
        position = torch.arange(0, max_pos, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_embed, 2).float() * (-math.log(10000.0) / d_embed))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1).contiguous())   # [max_pos, 1, d] (in state_dict as pe.pe)

    def forward(self, seq_len, bsz=None):
        pos_encoding = self.pe[:seq_len, :]
        if bsz is not None:
            pos_encoding = pos_encoding.expand(seq_len, bsz, -1)
        return pos_encoding


class TokenEmbedding(nn.Module):
    """emb_lookup [n_token, d_embed] (+ emb_proj: Linear(d_embed, d_proj, bias=False) when the widths differ), output scaled by
    sqrt(d_proj) — transformer_helpers.py:66-87.  The engine never runs this module's forward inside a model: because the projection
    is linear, `emb_proj(emb_lookup(ids))` equals a gather from the PROJECTED table `emb_lookup.weight @ emb_proj.weight^T`, which is
    one small GEMM per forward (engine.embedding_table) in front of the same fused gather kernel."""

    def __init__(self, n_token, d_embed, d_proj, emb_scale=0.5, pad_idx=None):
        super().__init__()
        self.n_token, self.d_embed, self.d_proj = n_token, d_embed, d_proj
        self.emb_scale = d_proj ** emb_scale
        self.emb_lookup = nn.Embedding(n_token, d_embed, padding_idx=pad_idx)
        self.emb_proj = nn.Linear(d_embed, d_proj, bias=False) if d_proj != d_embed else None

    def forward(self, inp_tokens):
        from emo_disentanger_amd import ops
        table = self.emb_lookup.weight.detach()
        if self.emb_proj is not None:
            table = ops.gemm(table.float().contiguous(), self.emb_proj.weight.detach().float().contiguous())
        zero_pe = torch.zeros(inp_tokens.shape[1], self.d_proj, device=inp_tokens.device)
        return ops.embed_fwd(inp_tokens, None, table, None, zero_pe, torch.float32, float(self.emb_scale))
