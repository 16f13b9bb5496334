"""MusicGPT2 — drop-in for /root/reference/stage2_accompaniment/model/music_gpt2.py:18-103: twelve
HF-4.28 ``GPT2Block``s (pre-LN, Conv1D weights [in,out], gelu_new, NO final ln_f) behind the same
prologue / logits / loss as MusicPerformer; executed by libemo_hip.so.  Checkpoints written by the
reference carry HF's persistent buffers ``attn.bias`` [1,1,4096,4096] / ``attn.masked_bias``; they
are accepted and dropped on load (the causal mask is implicit in the fused attention kernel)."""
import torch
from torch import nn

from .transformer_helpers import weights_init
from ._base import MusicLMBase


class Conv1D(nn.Module):
    """HF Conv1D parameter holder: weight [nx, nf] (in, out), y = x @ W + b; init N(0, 0.02), bias 0."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf))
        self.bias = nn.Parameter(torch.zeros(nf))
        nn.init.normal_(self.weight, std=0.02)


class _GPT2Attention(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        self.c_attn = Conv1D(3 * d_model, d_model)
        self.c_proj = Conv1D(d_model, d_model)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in (prefix + 'bias', prefix + 'masked_bias'):     # HF<=4.28 persistent causal-mask buffers
            state_dict.pop(k, None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class _GPT2MLP(nn.Module):
    def __init__(self, d_model, d_ff):
        super().__init__()
        self.c_fc = Conv1D(d_ff, d_model)
        self.c_proj = Conv1D(d_model, d_ff)


class GPT2Block(nn.Module):
    def __init__(self, d_model, d_ff):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d_model, eps=1e-5)
        self.attn = _GPT2Attention(d_model)
        self.ln_2 = nn.LayerNorm(d_model, eps=1e-5)
        self.mlp = _GPT2MLP(d_model, d_ff)


class MusicGPT2(MusicLMBase):
    kind = 'gpt2'

    def __init__(self, n_token, n_layer, n_head, d_model, d_ff, d_embed,
                 activation='relu', dropout=0.1, use_pe=True,
                 use_segment_emb=False, n_segment_types=None,
                 use_chord_mhot_emb=False, compute_dtype=None):
        super().__init__()
        self._init_common(n_token, n_layer, n_head, d_model, d_ff, d_embed, activation, dropout, use_pe, use_segment_emb,
                          n_segment_types, use_chord_mhot_emb, compute_dtype)
        self.transformer_decoder = nn.ModuleList([GPT2Block(d_model, d_ff) for _ in range(n_layer)])
        self._init_tail(use_segment_emb, n_segment_types)
        self.apply(weights_init)
        print('[info] model init completed')

    def _layer_prefix(self, l):
        return 'transformer_decoder.%d.' % l
