"""Shared implementation of the two stage-2 language models (the reference duplicates it in
music_performer.py / music_gpt2.py).  Object contract = SURVEY.md §8(b)."""
import os

import torch
from torch import nn

from emo_disentanger_amd import engine
from emo_disentanger_amd._lib import EmoError

from .transformer_helpers import PositionalEncoding, TokenEmbedding


class MusicLMBase(nn.Module):
    kind = None   # 'performer' | 'gpt2'

    def _init_common(self, n_token, n_layer, n_head, d_model, d_ff, d_embed, activation, dropout, use_pe, use_segment_emb,
                     n_segment_types, use_chord_mhot_emb, compute_dtype):
        self.n_token, self.n_layer, self.n_head = n_token, n_layer, n_head
        self.d_model, self.d_ff, self.d_embed = d_model, d_ff, d_embed
        self.dropout, self.activation, self.use_pe = dropout, activation, use_pe
        self.use_chord_mhot_emb = bool(use_chord_mhot_emb)
        self.token_emb = TokenEmbedding(n_token, d_embed, d_model)
        self.pe = PositionalEncoding(d_embed)
        self.dec_out_proj = nn.Linear(d_model, n_token)
        self._compute_dtype = engine._dt(compute_dtype or os.environ.get('EMO_COMPUTE_DTYPE', 'bf16'))
        self._store = None
        self._fwd_counter = 0
        self._seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF

    def _init_tail(self, use_segment_emb, n_segment_types):
        self.emb_dropout = nn.Dropout(self.dropout)
        self.use_segment_emb = use_segment_emb
        if use_segment_emb:
            self.segemb = TokenEmbedding(n_segment_types, self.d_embed, self.d_model)
            self.n_segment_types = n_segment_types
        else:
            self.segemb = None
        if self.use_chord_mhot_emb:                      # music_performer.py:42-44 (registered after segemb: same parameter order)
            self.chord_emb = nn.Linear(12, self.d_model)

    # ------------------------------------------------------------------ engine plumbing
    @property
    def compute_dtype(self):
        return self._compute_dtype

    def set_compute_dtype(self, name):
        """'bf16' (speed mode) or 'fp32' (parity mode: exact-f32 MFMA everywhere)."""
        self._compute_dtype = engine._dt(name)
        self._store = None
        return self

    def set_dropout_seed(self, seed):
        self._seed, self._fwd_counter = int(seed), 0

    def _fused_groups(self):
        return []

    def _ensure_store(self):
        if self._store is None or not self._store.intact() or self._store.compute_dtype != self._compute_dtype:
            self._store = engine.ParamStore(self, self._compute_dtype, self._fused_groups())
        self._store.sync_mirror()
        return self._store

    def _next_dropout_base(self):
        self._fwd_counter += 1
        return self._seed, self._fwd_counter * 4096

    def _zero_pe(self, T, D):
        return torch.zeros(T, D, device=self._store.device)

    def _layer_prefix(self, l):
        raise NotImplementedError

    # ------------------------------------------------------------------ reference API
    def forward(self, x, seg_inp=None, chord_inp=None, keep_last_only=False, attn_kwargs=None):
        """music_performer.py:50-70 / music_gpt2.py:70-92.  x, seg_inp: int64 [B,T] on the GPU; chord_inp: [B,T,12] multi-hot pitch
        classes, used only by a model built with use_chord_mhot_emb (x_emb += chord_emb(chord_inp), :56-57).
        Returns fp32 logits [B,T,V] (or [B,V] with keep_last_only)."""
        if not x.is_cuda:
            raise EmoError('inputs must be GPU tensors (the HIP path has no CPU fallback)')
        self._ensure_store()
        self._attn_kwargs = dict(attn_kwargs) if attn_kwargs else {}
        anchor = self.token_emb.emb_lookup.weight
        need_bwd = torch.is_grad_enabled() and anchor.requires_grad
        if seg_inp is not None and not self.use_segment_emb:
            seg_inp = None
        if chord_inp is not None and not self.use_chord_mhot_emb:
            chord_inp = None
        h = engine.DecoderStackFn.apply(self, x.long(), None if seg_inp is None else seg_inp.long(), anchor, need_bwd, chord_inp)
        if keep_last_only:
            h = h[:, -1, :]
        return engine.LogitsFn.apply(self, h)

    def compute_loss(self, dec_logits, dec_tgt, reduction='mean'):
        """music_performer.py:72-81."""
        if reduction != 'mean':
            raise NotImplementedError("only reduction='mean' (the reference's only use) is built")
        recons_loss = engine.XentFn.apply(dec_logits, dec_tgt.long(), self.n_token - 1).float()
        return {'recons_loss': recons_loss, 'total_loss': recons_loss}
