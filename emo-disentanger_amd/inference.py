"""Autoregressive generation — the sampling loop of
/root/reference/stage2_accompaniment/inference.py (temperature :71-83, nucleus :86-100,
generate_conditional :231-327) on top of a recurrent-state / KV-cache decode engine.

The reference re-runs the full model over the whole (<=2048-token) prefix for every sampled token
(SURVEY F7).  Here the Performer keeps its FAVOR+ scan state (S [F x dh], z [F] per layer/head) and
GPT-2 keeps a KV cache in HBM, so a step costs one token of work; results equal full recompute
(tests/test_gpu_generate.py).  Rejected samples (Beat going backwards, PAD, early EOS) re-sample from
the SAME logits without touching the state.  Once the window slides (len >= 2048) absolute positions
restart at 0 every step, which invalidates any cache: the engine then falls back to the reference's
full-window forward.  Host-side sampling helpers keep the reference's NumPy semantics (global RNG,
F12 nucleus indexing); `sample_on_device` is the batched on-GPU path (emo_sample_nucleus).
"""
import os
import time

import numpy as np
import torch

from . import engine, ops
from ._lib import EmoError

max_dec_inp_len = 2048


# ------------------------------------------------------------------------------------------------ sampling
from .sampling import beat_position, nucleus, temperature  # noqa: E402,F401  (host path with the reference's NumPy semantics)


def sample_on_device(logits, temp, top_p, u=None, greedy=False):
    """logits fp32 [n, V] on the GPU -> int64 [n] (no host round trip)."""
    if greedy:
        return ops.argmax(logits.contiguous())
    if u is None:
        u = torch.rand(logits.shape[0], device=logits.device)
    return ops.sample_nucleus(logits.contiguous(), temp, top_p, u)


# ------------------------------------------------------------------------------------------------ decode engines
class _EngineBase:
    def __init__(self, model, n_streams, max_len=max_dec_inp_len):
        self.model, self.n, self.max_len = model, n_streams, max_len
        self.ps = model._ensure_store()
        self.dev, self.dt = self.ps.device, self.ps.compute_dtype
        self.pos = 0
        self.pos_dev = torch.zeros(n_streams, device=self.dev, dtype=torch.int64)   # device-side positions (hipGraph replay)
        self._tables = None
        self.dev_pos0, self.pos_auto = 0, True   # position = dev_pos0 + pos_dev[stream]; pos_auto: the engine advances pos_dev itself

    def _embed(self, tok, seg, pos0, dev_pos=False):
        m = self.model
        if self._tables is None:                                  # snapshot, like the engine's omegas / folded weights
            self._tables = (engine.embedding_table(self.ps, 'token_emb.'), engine.embedding_table(self.ps, 'segemb.') if m.use_segment_emb else None)
        E, S = self._tables
        S = S if seg is not None else None
        pe = m.pe.pe if m.use_pe else m._zero_pe(self.max_len, m.d_model)
        return ops.embed_fwd(tok, seg if S is not None else None, E, S, pe, self.dt, float(m.token_emb.emb_scale),
                             pos0=self.dev_pos0 if dev_pos else pos0, pos_ids=self.pos_dev if dev_pos else None).view(-1, m.d_model)

    def _logits(self, h, out=None):
        return ops.gemm(h, self.ps.w('dec_out_proj.weight'), bias=self.ps.f32('dec_out_proj.bias'), out=out, out_dtype=torch.float32)

    @torch.no_grad()
    def append(self, tok, seg):
        """tok, seg: int64 [n, k]; consumes k tokens per stream, returns logits [n, V] after the last one."""
        if self.pos == 0:
            return self.prefill(tok, seg)
        out = None
        for i in range(tok.shape[1]):
            out = self.step(tok[:, i], seg[:, i])
        return out


class PerformerDecodeEngine(_EngineBase):
    """FAVOR+ recurrent state per layer: S [n,H,F,dh], z [n,H,F] fp32 (6.4 MB / stream at the perf config).
    omega is fixed for the lifetime of the engine (the state is only meaningful under one feature map)."""

    def __init__(self, model, n_streams, redraw=True, persistent=True):
        """persistent=False keeps the chain of launches (callers that run several engines side by side on different streams, and the fall-back of
        the reference loop when a one-launch step gave up)."""
        super().__init__(model, n_streams)
        if redraw and model.redraw != 'fixed':
            model.draw_feature_maps()
        self.omegas = [lyr.attention.inner_attention.feature_map.omega.clone() for lyr in model.transformer_decoder.decoder_layers]
        self.act = ops.ACT_GELU if getattr(model, 'activation', 'relu') == 'gelu' else ops.ACT_RELU
        self.S, self.z = [None] * model.n_layer, [None] * model.n_layer
        # bf16 one-token steps: norm1 / norm2 are folded into the GEMMs around them (2 launches fewer per layer, see emo_hip.h: ln_c1 / rln_*)
        self.fold = None
        if self.dt == torch.bfloat16 and n_streams <= 32 and os.environ.get('EMO_DECODE_LN_FOLD', '1') != '0':
            self._prepare_folds()
        # the whole token step as ONE persistent launch (emo_performer_decode_step) for the benchmark architecture: bf16, d_model 512, 8 heads,
        # 128 features, d_ff 2048, up to 32 streams (the kernel's groups own 4 streams each: other counts — the reference's own one-piece-at-a-time
        # loop is n = 1 — are padded with idle streams whose state is zero and whose logits nobody reads).  EMO_DECODE_PERSISTENT=0 keeps the chain
        # of launches (tests compare the two).
        self.persist = None
        ff = model.transformer_decoder.decoder_layers[0].linear1.weight.shape[0]
        nf = 2 * self.omegas[0].shape[1]
        self.n_pad = (n_streams + 3) // 4 * 4
        if (self.dt == torch.bfloat16 and 1 <= n_streams <= 32 and model.d_model == 512 and model.n_head == 8 and nf == 128
                and ff == 2048 and model.n_layer <= 15 and model.n_token <= 512 and persistent and self.act == ops.ACT_RELU
                and os.environ.get('EMO_DECODE_PERSISTENT', '1') != '0'
                and ops.lib.emo_performer_decode_step_supported() == 1):
            # (emo_performer_decode_step_supported: the launch's 256 workgroups spin-wait on each other and must all be resident — >= 256 CUs,
            # 96 KB LDS each, one per CU by the occupancy query; partitions / CU masks with fewer keep the chain of launches)
            self._prepare_persist()

    # ------------------------------------------------------------------------------------------ one-launch step
    @staticmethod
    def _pack_fragments(W, tile_idx, kpw):
        """bf16 nn.Linear weight [N, K] -> [members][4 waves][tiles per member][kpw][64 lanes x 8]: the MFMA B fragment (16 output columns x 32 k)
        of column tile t and k step ks holds, in lane l, W[16 t + l % 16][32 ks + 8 (l // 16) .. + 8]; wave w of the compute half that owns the
        product holds the k steps [w kpw, (w + 1) kpw) of all of the member's tiles (emo_hip.h: emo_performer_decode_step)."""
        N, K = W.shape
        assert N % 16 == 0 and K == 32 * 4 * kpw
        frags = W.reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(N // 16, K // 32, 512)      # [tile][k step][lane * 8 + j]
        sel = frags[tile_idx]                                                                                  # [members, tiles per member, k steps, 512]
        members, tpm = tile_idx.shape
        return sel.reshape(members, tpm, 4, kpw, 512).permute(0, 2, 1, 3, 4).contiguous()

    def _prepare_persist(self):
        m, ps, dev = self.model, self.ps, self.dev
        D, L = m.d_model, m.n_layer
        mem = torch.arange(32, device=dev)
        t_qkv = torch.stack([mem, 32 + mem, 64 + mem], 1)            # member (h, j) = 4 h + j: rows 64 h + 16 j .. of q, of k (+ 512), of v (+ 1024)
        t_one = mem.view(32, 1)                                      # member m: output columns 16 m ..
        t_ffn = (4 * mem).view(32, 1) + torch.arange(4, device=dev).view(1, 4)
        pk = self._pack_fragments
        self.persist = {'w': [], 'table': None}
        for l in range(L):
            pfx = m._layer_prefix(l)
            q = pfx + 'attention.query_projection.'
            self.persist['w'].append(dict(
                wqkv=pk(ps.w(q + 'weight', 3 * D), t_qkv, 4), bqkv=ps.f32(q + 'bias', 3 * D),
                wo=pk(ps.w(pfx + 'attention.out_projection.weight'), t_one, 4), bo=ps.f32(pfx + 'attention.out_projection.bias'),
                g1=ps.f32(pfx + 'norm1.weight'), be1=ps.f32(pfx + 'norm1.bias'),
                w1=pk(ps.w(pfx + 'linear1.weight'), t_ffn, 4), b1=ps.f32(pfx + 'linear1.bias'),
                w2=pk(ps.w(pfx + 'linear2.weight'), t_one, 16), b2=ps.f32(pfx + 'linear2.bias'),
                g2=ps.f32(pfx + 'norm2.weight'), be2=ps.f32(pfx + 'norm2.bias')))
        V = m.n_token
        Vp = (V + 15) // 16 * 16
        wout = torch.zeros(Vp, D, device=dev, dtype=torch.bfloat16)
        wout[:V] = ps.w('dec_out_proj.weight')
        self.persist['wout'] = pk(wout, torch.arange(Vp // 16, device=dev).view(-1, 1), 4)
        self.persist['bout'] = ps.f32('dec_out_proj.bias')
        self.persist['sync'] = torch.zeros(ops.lib.emo_performer_decode_step_workspace_bytes() // 8, device=dev, dtype=torch.int64)   # zeroed ONCE
        self.persist['logits'] = torch.zeros(self.n_pad, V, device=dev, dtype=torch.float32)
        if self.n_pad != self.n:                                     # padded inputs of the idle streams: token 0, segment 0, position 0
            self.persist['tok'] = torch.zeros(self.n_pad, dtype=torch.int64, device=dev)
            self.persist['seg'] = torch.zeros(self.n_pad, dtype=torch.int64, device=dev)
            self.persist['pos'] = torch.zeros(self.n_pad, dtype=torch.int64, device=dev)

    def _persist_table(self):
        """[L][16] device pointers (emo_hip.h); built once the recurrent state exists (prefill)."""
        pp = self.persist
        if pp['table'] is None:
            rows = []
            for l, w in enumerate(pp['w']):
                assert self.S[l].is_contiguous() and self.z[l].is_contiguous() and self.omegas[l].is_contiguous()
                rows.append([w['wqkv'].data_ptr(), w['bqkv'].data_ptr(), w['wo'].data_ptr(), w['bo'].data_ptr(), w['g1'].data_ptr(), w['be1'].data_ptr(),
                             w['w1'].data_ptr(), w['b1'].data_ptr(), w['w2'].data_ptr(), w['b2'].data_ptr(), w['g2'].data_ptr(), w['be2'].data_ptr(),
                             self.omegas[l].data_ptr(), self.S[l].data_ptr(), self.z[l].data_ptr(), 0])
            pp['table'] = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        return pp['table']

    def _step_persistent(self, tok, seg, dev_pos, logits_out):
        """Returns `logits_out` when given, else the engine's STATIC logits buffer (or a view of its first n rows): the next step overwrites it —
        callers that keep logits across steps clone them (the chain of launches returns a fresh tensor; hipGraph capture needs the static one)."""
        m, pp = self.model, self.persist
        if self._tables is None:
            self._tables = (engine.embedding_table(self.ps, 'token_emb.'), engine.embedding_table(self.ps, 'segemb.') if m.use_segment_emb else None)
        E, Sg = self._tables
        seg = seg if (Sg is not None and seg is not None) else None
        pe = m.pe.pe if m.use_pe else m._zero_pe(self.max_len, m.d_model)
        if not dev_pos and self.pos >= pe.shape[0]:                  # (ops.embed_fwd makes the same check on the launch-chain path)
            raise EmoError('decode position %d is past the positional-encoding table (%d rows)' % (self.pos, pe.shape[0]))
        nf = 2 * self.omegas[0].shape[1]
        pos_ids = self.pos_dev if dev_pos else None
        padded = self.n_pad != self.n
        if padded:
            pp['tok'][:self.n].copy_(tok)
            tok = pp['tok']
            if seg is not None:
                pp['seg'][:self.n].copy_(seg)
                seg = pp['seg']
            if pos_ids is not None:
                pp['pos'][:self.n].copy_(pos_ids)
                pos_ids = pp['pos']
        out = logits_out if (logits_out is not None and not padded) else pp['logits']
        ops.performer_decode_step(self._persist_table(), m.n_layer, tok, seg, E, Sg if seg is not None else None, pe, float(m.token_emb.emb_scale),
                                  self.dev_pos0 if dev_pos else self.pos, pos_ids, pp['wout'], pp['bout'], m.n_token, out,
                                  self.n_pad, m.d_model, m.n_head, nf, 2048, pp['sync'], diag=pp.get('diag'))
        if padded:
            if logits_out is not None:
                logits_out.copy_(out[:self.n])
                return logits_out
            return out[:self.n]
        return out

    def step_sampled(self, seg_padded, temp, top_p, U, step_ctr, seq, col0, tok_out, pos0):
        """One token step with the nucleus draw INSIDE the launch (emo_performer_decode_step_sampled): draws from the logits the previous step (or
        the prefill: see load_logits) left in the engine's buffer, writes token / sequence / step counter like emo_sample_nucleus_step, then runs
        the step on the drawn tokens.  seg_padded: int64 [n_pad] (or None)."""
        m, pp = self.model, self.persist
        if self._tables is None:
            self._tables = (engine.embedding_table(self.ps, 'token_emb.'), engine.embedding_table(self.ps, 'segemb.') if m.use_segment_emb else None)
        E, Sg = self._tables
        seg = seg_padded if Sg is not None else None
        pe = m.pe.pe if m.use_pe else m._zero_pe(self.max_len, m.d_model)
        ops.performer_decode_step_sampled(self._persist_table(), m.n_layer, seg, E, Sg if seg is not None else None, pe, float(m.token_emb.emb_scale), pos0,
                                          pp['wout'], pp['bout'], m.n_token, pp['logits'], self.n_pad, self.n, m.d_model, m.n_head,
                                          2 * self.omegas[0].shape[1], 2048, pp['sync'], temp, top_p, U, step_ctr, seq, col0, tok_out)
        return pp['logits'][:self.n]

    def load_logits(self, logits):
        """Put externally produced logits (the prefill's) where step_sampled draws from."""
        self.persist['logits'][:self.n].copy_(logits)

    def check_persistent(self):
        """Raises if a one-launch step gave up (synchronises; call it where the caller reads results anyway)."""
        if self.persist is not None:
            code = int(self.persist['sync'][-8].item())
            if code != 0:
                raise EmoError('emo_performer_decode_step gave up (code 0x%x): a workgroup of the persistent launch did not get a compute unit next to the '
                               'others within 50 ms (is another process using the GPU?); set EMO_DECODE_PERSISTENT=0 for the chain of launches' % code)

    def _prepare_folds(self):
        """gamma-scaled weights, c1[n] = sum_k gamma_k W[n,k] (of the ROUNDED bf16 product, the one the MFMA sees) and bias + W.beta for every
        GEMM whose input is a LayerNorm output: linear1 (norm1), the next layer's q/k/v projection and the final logits (norm2).  Built once
        per engine from the fp32 masters: the engine is a snapshot of the weights, like its omegas."""
        m, ps = self.model, self.ps
        D = m.d_model

        def fold(wname, bname, rows, gamma, beta):
            W = ps.f32(wname, rows)
            Wg = (W * gamma[None, :]).to(torch.bfloat16).contiguous()
            return Wg, Wg.float().sum(1).contiguous(), (ps.f32(bname, rows) + (W * beta[None, :]).sum(1)).contiguous()

        L = m.n_layer
        pf = [m._layer_prefix(l) for l in range(L)]
        self.fold = {'ffn1': [fold(pf[l] + 'linear1.weight', pf[l] + 'linear1.bias', None, ps.f32(pf[l] + 'norm1.weight'), ps.f32(pf[l] + 'norm1.bias'))
                              for l in range(L)],
                     'qkv': [None] + [fold(pf[l] + 'attention.query_projection.weight', pf[l] + 'attention.query_projection.bias', 3 * D,
                                           ps.f32(pf[l - 1] + 'norm2.weight'), ps.f32(pf[l - 1] + 'norm2.bias')) for l in range(1, L)],
                     'out': fold('dec_out_proj.weight', 'dec_out_proj.bias', None, ps.f32(pf[L - 1] + 'norm2.weight'), ps.f32(pf[L - 1] + 'norm2.bias'))}
        self.stats1 = torch.empty(self.n, 2, device=self.dev, dtype=torch.float32)
        self.stats2 = torch.empty(self.n, 2, device=self.dev, dtype=torch.float32)

    def _step_folded(self, x, logits_out):
        """One token per stream, LayerNorms folded: per layer q/k/v GEMM, state update, out-projection, linear1, linear2 (5 launches)."""
        m, ps, fd = self.model, self.ps, self.fold
        D, H = m.d_model, m.n_head
        res = None                                   # (raw tensor, stats, gamma, beta) whose LayerNorm is the current hidden state
        for l in range(m.n_layer):
            pfx = m._layer_prefix(l)
            q = pfx + 'attention.query_projection.'
            if res is None:
                qkv = ops.gemm(x, ps.w(q + 'weight', 3 * D), bias=ps.f32(q + 'bias', 3 * D))
            else:
                Wg, c1, bb = fd['qkv'][l]
                qkv = ops.gemm(res[0], Wg, bias=bb, ln_c1=c1, ln_stats_out=self.stats2)
            attn = ops.favor_decode_step(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], self.omegas[l], self.S[l], self.z[l], H)
            ow, ob = ps.w(pfx + 'attention.out_projection.weight'), ps.f32(pfx + 'attention.out_projection.bias')
            x1 = ops.gemm(attn, ow, bias=ob, residual=x) if res is None else ops.gemm(attn, ow, bias=ob, rln=res)
            Wg, c1, bb = fd['ffn1'][l]
            f = ops.gemm(x1, Wg, bias=bb, act=self.act, ln_c1=c1, ln_stats_out=self.stats1)
            x2 = ops.gemm(f, ps.w(pfx + 'linear2.weight'), bias=ps.f32(pfx + 'linear2.bias'),
                          rln=(x1, self.stats1, ps.f32(pfx + 'norm1.weight'), ps.f32(pfx + 'norm1.bias')))
            res = (x2, self.stats2, ps.f32(pfx + 'norm2.weight'), ps.f32(pfx + 'norm2.bias'))
        Wg, c1, bb = fd['out']
        return ops.gemm(res[0], Wg, bias=bb, ln_c1=c1, out=logits_out, out_dtype=torch.float32)

    @torch.no_grad()
    def prefill(self, tok, seg):
        m, ps = self.model, self.ps
        B, T = tok.shape
        D, H = m.d_model, m.n_head
        x = self._embed(tok, seg, 0)
        for l in range(m.n_layer):
            pfx = m._layer_prefix(l)
            q = pfx + 'attention.query_projection.'
            qkv = ops.gemm(x, ps.w(q + 'weight', 3 * D), bias=ps.f32(q + 'bias', 3 * D))
            attn, _, self.S[l], self.z[l] = ops.favor_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], self.omegas[l], B, T, H, want_state=True)
            x = self._tail(pfx, x, attn)
        if self.persist is not None:
            self.persist['table'] = None                         # new state tensors: the pointer table is rebuilt at the next step
            if self.n_pad != B:                                  # state of the padded (idle) streams: zero; S[l] / z[l] stay the views of the real ones
                for l in range(m.n_layer):
                    Sp = torch.zeros((self.n_pad,) + tuple(self.S[l].shape[1:]), device=self.dev, dtype=self.S[l].dtype)
                    zp = torch.zeros((self.n_pad,) + tuple(self.z[l].shape[1:]), device=self.dev, dtype=self.z[l].dtype)
                    Sp[:B].copy_(self.S[l])
                    zp[:B].copy_(self.z[l])
                    self.S[l], self.z[l] = Sp[:B], zp[:B]
        self.pos = T
        self.pos_dev.fill_(T)
        return self._logits(x.view(B, T, D)[:, -1].contiguous())

    def _tail(self, pfx, x, attn):
        ps = self.ps
        x1 = ops.gemm(attn, ps.w(pfx + 'attention.out_projection.weight'), bias=ps.f32(pfx + 'attention.out_projection.bias'), residual=x)
        h1, _, _ = ops.layernorm_fwd(x1, ps.f32(pfx + 'norm1.weight'), ps.f32(pfx + 'norm1.bias'))
        f = ops.gemm(h1, ps.w(pfx + 'linear1.weight'), bias=ps.f32(pfx + 'linear1.bias'), act=self.act)
        x2 = ops.gemm(f, ps.w(pfx + 'linear2.weight'), bias=ps.f32(pfx + 'linear2.bias'), residual=h1)
        out, _, _ = ops.layernorm_fwd(x2, ps.f32(pfx + 'norm2.weight'), ps.f32(pfx + 'norm2.bias'))
        return out

    @torch.no_grad()
    def step(self, tok, seg, dev_pos=False, logits_out=None):
        """dev_pos=True: positions come from the device array `pos_dev` (and are advanced on the device), so the whole step is
        capturable in a hipGraph and replayable.  logits_out: write the logits into this static buffer (no copy kernel)."""
        m, ps = self.model, self.ps
        D, H = m.d_model, m.n_head
        if self.persist is not None:
            out = self._step_persistent(tok.reshape(-1), None if seg is None else seg.reshape(-1), dev_pos, logits_out)
            if dev_pos:
                if self.pos_auto:
                    self.pos_dev.add_(1)
            else:
                self.pos += 1
            return out
        x = self._embed(tok.view(-1, 1), None if seg is None else seg.view(-1, 1), self.pos, dev_pos)
        if self.fold is not None:
            out = self._step_folded(x, logits_out)
            if dev_pos:
                if self.pos_auto:
                    self.pos_dev.add_(1)
            else:
                self.pos += 1
            return out
        for l in range(m.n_layer):
            pfx = m._layer_prefix(l)
            q = pfx + 'attention.query_projection.'
            qkv = ops.gemm(x, ps.w(q + 'weight', 3 * D), bias=ps.f32(q + 'bias', 3 * D))
            attn = ops.favor_decode_step(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], self.omegas[l], self.S[l], self.z[l], H)
            x = self._tail(pfx, x, attn)
        if dev_pos:
            if self.pos_auto:
                self.pos_dev.add_(1)
        else:
            self.pos += 1
        return self._logits(x, logits_out)


class GPT2DecodeEngine(_EngineBase):
    """KV cache in HBM: per layer k,v [n, max_len, D] in the compute dtype (1.6 GB for 32 streams x 2048 x 12 layers, bf16)."""

    def __init__(self, model, n_streams, max_len=max_dec_inp_len, persistent=True):
        super().__init__(model, n_streams, max_len)
        D = model.d_model
        self.n_pad = (n_streams + 3) // 4 * 4
        # head-major cache [n, H, max_len, dh] (r06; HF's own past_key_values layout): the decode attention gives one workgroup to a (stream, head),
        # whose keys are then one contiguous run instead of 128-byte pieces 1 KB apart.  EMO_KV_HEAD_MAJOR=0: [n, max_len, D] (r05, same-box A/B)
        self.head_major = os.environ.get('EMO_KV_HEAD_MAJOR', '1') != '0'
        # the whole token step as ONE persistent launch (emo_gpt2_decode_step; r06), under the conditions of the Performer engine's: bf16, d_model 512,
        # 8 heads, d_ff 2048, <= 32 streams (padded to a multiple of 4 with idle streams), head-major cache of <= 2048 rows.  EMO_DECODE_PERSISTENT=0 /
        # EMO_GPT2_PERSISTENT=0 keep the chain of launches (tests compare the two).
        ff = self.ps.f32(model._layer_prefix(0) + 'mlp.c_fc.bias').numel()
        want_persist = (self.dt == torch.bfloat16 and 1 <= n_streams <= 32 and D == 512 and model.n_head == 8 and ff == 2048 and model.n_layer <= 15
                        and model.n_token <= 512 and max_len <= 2048 and self.head_major and persistent
                        and os.environ.get('EMO_DECODE_PERSISTENT', '1') != '0' and os.environ.get('EMO_GPT2_PERSISTENT', '1') != '0'
                        and ops.lib.emo_gpt2_decode_step_supported() == 1)
        rows = self.n_pad if want_persist else n_streams
        shp = (rows, model.n_head, max_len, D // model.n_head) if self.head_major else (rows, max_len, D)
        self.kc_all = [torch.zeros(*shp, device=self.dev, dtype=self.dt) for _ in range(model.n_layer)]
        self.vc_all = [torch.zeros(*shp, device=self.dev, dtype=self.dt) for _ in range(model.n_layer)]
        self.kc = [t[:n_streams] for t in self.kc_all]             # (rows n .. n_pad: the idle padding streams of the one-launch step)
        self.vc = [t[:n_streams] for t in self.vc_all]
        self.persist = None
        self.lens = torch.zeros(n_streams, device=self.dev, dtype=torch.int64)
        # bf16 one-token steps: the Conv1D weights ([in, out]) are transposed ONCE to the k-contiguous layout of the skinny decode GEMM,
        # ln_1 / ln_2 are folded into c_attn / c_fc (emo_hip.h: ln_c1) and the new k / v rows are appended by the attention kernel:
        # 5 launches per layer instead of 9 general-shape ones.
        self.fold = None
        if self.dt == torch.bfloat16 and n_streams <= 32 and os.environ.get('EMO_DECODE_LN_FOLD', '1') != '0':
            self._prepare_folds()
        if want_persist:
            self._prepare_persist()

    # ------------------------------------------------------------------------------------------ one-launch step
    def _prepare_persist(self):
        m, ps, dev = self.model, self.ps, self.dev
        L = m.n_layer
        mem = torch.arange(32, device=dev)
        t_qkv = torch.stack([mem, 32 + mem, 64 + mem], 1)            # member (h, j) = 4 h + j: columns 64 h + 16 j .. of q, of k (+ 512), of v (+ 1024)
        t_one = mem.view(32, 1)
        t_ffn = (4 * mem).view(32, 1) + torch.arange(4, device=dev).view(1, 4)
        pk = PerformerDecodeEngine._pack_fragments

        def lin(name):                                               # Conv1D [in, out] -> nn.Linear layout [out, in]
            return ps.w(name).t().contiguous()

        pf = [m._layer_prefix(l) for l in range(L)]
        self.persist = {'w': [], 'table': None}
        for l in range(L):
            nx = pf[l + 1] if l + 1 < L else pf[0]                   # (the last block's slot is loaded and never applied: this GPT-2 has no ln_f)
            self.persist['w'].append(dict(
                wqkv=pk(lin(pf[l] + 'attn.c_attn.weight'), t_qkv, 4), bqkv=ps.f32(pf[l] + 'attn.c_attn.bias'),
                wo=pk(lin(pf[l] + 'attn.c_proj.weight'), t_one, 4), bo=ps.f32(pf[l] + 'attn.c_proj.bias'),
                g1=ps.f32(pf[l] + 'ln_2.weight'), be1=ps.f32(pf[l] + 'ln_2.bias'),
                w1=pk(lin(pf[l] + 'mlp.c_fc.weight'), t_ffn, 4), b1=ps.f32(pf[l] + 'mlp.c_fc.bias'),
                w2=pk(lin(pf[l] + 'mlp.c_proj.weight'), t_one, 16), b2=ps.f32(pf[l] + 'mlp.c_proj.bias'),
                g2=ps.f32(nx + 'ln_1.weight'), be2=ps.f32(nx + 'ln_1.bias')))
        self.persist['ln0'] = torch.cat([ps.f32(pf[0] + 'ln_1.weight'), ps.f32(pf[0] + 'ln_1.bias')]).contiguous()
        V = m.n_token
        Vp = (V + 15) // 16 * 16
        wout = torch.zeros(Vp, m.d_model, device=dev, dtype=torch.bfloat16)
        wout[:V] = ps.w('dec_out_proj.weight')
        self.persist['wout'] = pk(wout, torch.arange(Vp // 16, device=dev).view(-1, 1), 4)
        self.persist['bout'] = ps.f32('dec_out_proj.bias')
        self.persist['sync'] = torch.zeros(ops.lib.emo_performer_decode_step_workspace_bytes() // 8, device=dev, dtype=torch.int64)   # zeroed ONCE
        self.persist['logits'] = torch.zeros(self.n_pad, V, device=dev, dtype=torch.float32)
        if self.n_pad != self.n:                                     # padded inputs of the idle streams: token 0, segment 0, position 0
            self.persist['tok'] = torch.zeros(self.n_pad, dtype=torch.int64, device=dev)
            self.persist['seg'] = torch.zeros(self.n_pad, dtype=torch.int64, device=dev)
            self.persist['pos'] = torch.zeros(self.n_pad, dtype=torch.int64, device=dev)

    def _persist_table(self):
        """[L][16] device pointers (emo_hip.h: emo_gpt2_decode_step); the caches live as long as the engine."""
        pp = self.persist
        if pp['table'] is None:
            rows = []
            for l, w in enumerate(pp['w']):
                rows.append([w['wqkv'].data_ptr(), w['bqkv'].data_ptr(), w['wo'].data_ptr(), w['bo'].data_ptr(), w['g1'].data_ptr(), w['be1'].data_ptr(),
                             w['w1'].data_ptr(), w['b1'].data_ptr(), w['w2'].data_ptr(), w['b2'].data_ptr(), w['g2'].data_ptr(), w['be2'].data_ptr(),
                             0, self.kc_all[l].data_ptr(), self.vc_all[l].data_ptr(), 0])
            pp['table'] = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        return pp['table']

    def _persist_inputs(self):
        m = self.model
        if self._tables is None:
            self._tables = (engine.embedding_table(self.ps, 'token_emb.'), engine.embedding_table(self.ps, 'segemb.') if m.use_segment_emb else None)
        E, Sg = self._tables
        pe = m.pe.pe if m.use_pe else m._zero_pe(self.max_len, m.d_model)
        return E, Sg, pe

    def _step_persistent(self, tok, seg, dev_pos, logits_out):
        """Returns `logits_out` when given, else the engine's STATIC logits buffer (or a view of its first n rows), like the Performer engine's."""
        m, pp = self.model, self.persist
        E, Sg, pe = self._persist_inputs()
        seg = seg if (Sg is not None and seg is not None) else None
        if not dev_pos and self.pos >= min(pe.shape[0], self.max_len):
            raise EmoError('decode position %d is past the positional-encoding table / the KV cache (%d rows)' % (self.pos, min(pe.shape[0], self.max_len)))
        pos_ids = self.pos_dev if dev_pos else None
        padded = self.n_pad != self.n
        if padded:
            pp['tok'][:self.n].copy_(tok)
            tok = pp['tok']
            if seg is not None:
                pp['seg'][:self.n].copy_(seg)
                seg = pp['seg']
            if pos_ids is not None:
                pp['pos'][:self.n].copy_(pos_ids)
                pos_ids = pp['pos']
        out = logits_out if (logits_out is not None and not padded) else pp['logits']
        ops.gpt2_decode_step(self._persist_table(), m.n_layer, tok, seg, E, Sg if seg is not None else None, pe, float(m.token_emb.emb_scale),
                             self.dev_pos0 if dev_pos else self.pos, pos_ids, pp['ln0'], self.max_len, pp['wout'], pp['bout'], m.n_token, out,
                             self.n_pad, m.d_model, m.n_head, 2048, pp['sync'], diag=pp.get('diag'))
        if padded:
            if logits_out is not None:
                logits_out.copy_(out[:self.n])
                return logits_out
            return out[:self.n]
        return out

    def step_sampled(self, seg_padded, temp, top_p, U, step_ctr, seq, col0, tok_out, pos0):
        """One token step with the nucleus draw INSIDE the launch (emo_gpt2_decode_step_sampled); see PerformerDecodeEngine.step_sampled."""
        m, pp = self.model, self.persist
        E, Sg, pe = self._persist_inputs()
        seg = seg_padded if Sg is not None else None
        ops.gpt2_decode_step_sampled(self._persist_table(), m.n_layer, seg, E, Sg if seg is not None else None, pe, float(m.token_emb.emb_scale), pos0,
                                     pp['ln0'], self.max_len, pp['wout'], pp['bout'], m.n_token, pp['logits'], self.n_pad, self.n, m.d_model, m.n_head,
                                     2048, pp['sync'], temp, top_p, U, step_ctr, seq, col0, tok_out)
        return pp['logits'][:self.n]

    def load_logits(self, logits):
        """Put externally produced logits (the prefill's) where step_sampled draws from."""
        self.persist['logits'][:self.n].copy_(logits)

    def check_persistent(self):
        """Raises if a one-launch step gave up (synchronises; call it where the caller reads results anyway)."""
        if self.persist is not None:
            code = int(self.persist['sync'][-8].item())
            if code != 0:
                raise EmoError('emo_gpt2_decode_step gave up (code 0x%x): a workgroup of the persistent launch did not get a compute unit next to the '
                               'others within 50 ms (is another process using the GPU?); set EMO_DECODE_PERSISTENT=0 for the chain of launches' % code)

    def _prepare_folds(self):
        m, ps = self.model, self.ps

        def tr(wname, bname, gamma=None, beta=None):
            Wt = ps.f32(wname).t().contiguous()                    # [out, in]
            if gamma is None:
                return Wt.to(torch.bfloat16).contiguous(), None, ps.f32(bname)
            Wg = (Wt * gamma[None, :]).to(torch.bfloat16).contiguous()
            return Wg, Wg.float().sum(1).contiguous(), (ps.f32(bname) + (Wt * beta[None, :]).sum(1)).contiguous()

        self.fold = []
        for l in range(m.n_layer):
            pfx = m._layer_prefix(l)
            self.fold.append({'attn': tr(pfx + 'attn.c_attn.weight', pfx + 'attn.c_attn.bias', ps.f32(pfx + 'ln_1.weight'), ps.f32(pfx + 'ln_1.bias')),
                              'proj': tr(pfx + 'attn.c_proj.weight', pfx + 'attn.c_proj.bias'),
                              'fc': tr(pfx + 'mlp.c_fc.weight', pfx + 'mlp.c_fc.bias', ps.f32(pfx + 'ln_2.weight'), ps.f32(pfx + 'ln_2.bias')),
                              'mlp': tr(pfx + 'mlp.c_proj.weight', pfx + 'mlp.c_proj.bias')})

    def _step_folded(self, x, lens, lens_off, logits_out):
        m = self.model
        D, H = m.d_model, m.n_head
        for l in range(m.n_layer):
            fd = self.fold[l]
            Wg, c1, bb = fd['attn']
            qkv = ops.gemm(x, Wg, bias=bb, ln_c1=c1)
            a = ops.softmax_attn_decode(qkv[:, :D], self.kc[l], self.vc[l], lens, H, lens_off=lens_off, k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:])
            Wt, _, bb = fd['proj']
            h = ops.gemm(a, Wt, bias=bb, residual=x)
            Wg, c1, bb = fd['fc']
            f = ops.gemm(h, Wg, bias=bb, act=ops.ACT_GELU_NEW, ln_c1=c1)
            Wt, _, bb = fd['mlp']
            x = ops.gemm(f, Wt, bias=bb, residual=h)
        return self._logits(x, logits_out)

    def _block_tail(self, pfx, x, a):
        ps = self.ps
        h = ops.gemm(a, ps.w(pfx + 'attn.c_proj.weight'), b_trans=True, bias=ps.f32(pfx + 'attn.c_proj.bias'), residual=x)
        n2, _, _ = ops.layernorm_fwd(h, ps.f32(pfx + 'ln_2.weight'), ps.f32(pfx + 'ln_2.bias'))
        f = ops.gemm(n2, ps.w(pfx + 'mlp.c_fc.weight'), b_trans=True, bias=ps.f32(pfx + 'mlp.c_fc.bias'), act=ops.ACT_GELU_NEW)
        return ops.gemm(f, ps.w(pfx + 'mlp.c_proj.weight'), b_trans=True, bias=ps.f32(pfx + 'mlp.c_proj.bias'), residual=h)

    @torch.no_grad()
    def prefill(self, tok, seg):
        m, ps = self.model, self.ps
        B, T = tok.shape
        D, H = m.d_model, m.n_head
        x = self._embed(tok, seg, 0)
        for l in range(m.n_layer):
            pfx = m._layer_prefix(l)
            n1, _, _ = ops.layernorm_fwd(x, ps.f32(pfx + 'ln_1.weight'), ps.f32(pfx + 'ln_1.bias'))
            qkv = ops.gemm(n1, ps.w(pfx + 'attn.c_attn.weight'), b_trans=True, bias=ps.f32(pfx + 'attn.c_attn.bias'))
            if self.head_major:
                self.kc[l][:, :, :T].copy_(qkv[:, D:2 * D].view(B, T, H, D // H).permute(0, 2, 1, 3))
                self.vc[l][:, :, :T].copy_(qkv[:, 2 * D:].view(B, T, H, D // H).permute(0, 2, 1, 3))
            else:
                self.kc[l][:, :T].copy_(qkv[:, D:2 * D].view(B, T, D))
                self.vc[l][:, :T].copy_(qkv[:, 2 * D:].view(B, T, D))
            a, _ = ops.softmax_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, T, H)
            x = self._block_tail(pfx, x, a)
        self.pos = T
        self.pos_dev.fill_(T)
        self.lens.fill_(T)
        return self._logits(x.view(B, T, D)[:, -1].contiguous())

    @torch.no_grad()
    def step(self, tok, seg, dev_pos=False, logits_out=None):
        m, ps = self.model, self.ps
        D, H = m.d_model, m.n_head
        if self.persist is not None:
            out = self._step_persistent(tok.reshape(-1), None if seg is None else seg.reshape(-1), dev_pos, logits_out)
            self.lens.add_(1)
            if dev_pos:
                if self.pos_auto:
                    self.pos_dev.add_(1)
            else:
                self.pos += 1
            return out
        x = self._embed(tok.view(-1, 1), None if seg is None else seg.view(-1, 1), self.pos, dev_pos)
        if self.fold is not None:
            if dev_pos and not self.pos_auto:            # positions AND key counts come from the sampler's step counter
                return self._step_folded(x, self.pos_dev, self.dev_pos0 + 1, logits_out)
            self.lens.add_(1)
            out = self._step_folded(x, self.lens, 0, logits_out)
            if dev_pos:
                self.pos_dev.add_(1)
            else:
                self.pos += 1
            return out
        self.lens.add_(1)
        ext = dev_pos and not self.pos_auto                       # positions AND key counts come from the sampler's step counter (as in the folded path)
        for l in range(m.n_layer):
            pfx = m._layer_prefix(l)
            n1, _, _ = ops.layernorm_fwd(x, ps.f32(pfx + 'ln_1.weight'), ps.f32(pfx + 'ln_1.bias'))
            qkv = ops.gemm(n1, ps.w(pfx + 'attn.c_attn.weight'), b_trans=True, bias=ps.f32(pfx + 'attn.c_attn.bias'))
            # (the general-shape path — fp32 parity mode, more than 32 streams — lets the attention kernel append the rows as the folded path does)
            a = ops.softmax_attn_decode(qkv[:, :D], self.kc[l], self.vc[l], self.pos_dev if ext else self.lens, H, lens_off=self.dev_pos0 + 1 if ext else 0,
                                        k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:])
            x = self._block_tail(pfx, x, a)
        if dev_pos:
            self.pos_dev.add_(1)
        else:
            self.pos += 1
        return self._logits(x, logits_out)


def make_engine(model, n_streams, **kw):
    return PerformerDecodeEngine(model, n_streams, **kw) if model.kind == 'performer' else GPT2DecodeEngine(model, n_streams, **kw)


# ------------------------------------------------------------------------------------------------ reference loop
def generate_conditional(model, event2idx, idx2event, lead_sheet_events, primer,
                         max_events=10000, skip_check=False, max_bars=None,
                         temp=1.2, top_p=0.9, inadmissibles=None,
                         model_type="performer", use_cache=True, sampler=None, verbose=False):
    """Arguments and return value of the reference's generate_conditional (inference.py:231-327): the accompaniment of every
    lead-sheet bar is sampled token by token until the model emits Track_LeadSheet, then the next bar's lead sheet is injected.
    The grammar (Beat positions never go back, PAD / premature EOS rejected, 256 consecutive rejections = stuck) lives in
    `_Stream.offer`; this function is the one-stream driver of it.  `sampler(probs)` defaults to nucleus(probs, top_p) on NumPy's
    global RNG, like the reference.  use_cache=False (or a context at the 2048-token window) runs the reference's full-window
    forward per sampled token."""
    note = print if verbose else (lambda *a, **k: None)
    draw = sampler if sampler is not None else (lambda probs: nucleus(probs, top_p))
    dev = next(model.parameters()).device
    s = _Stream(event2idx, lead_sheet_events, primer, max_bars)
    primed = len(s.generated)
    eng = make_engine(model, 1) if use_cache else None
    cached = None
    t0 = time.time()
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            while not s.done:
                if eng is not None and len(s.generated) < max_dec_inp_len:
                    if s.consumed < len(s.generated):            # new tokens since the last step (a sampled word, or an injected bar)
                        cached = eng.append(torch.tensor([s.generated[s.consumed:]], dtype=torch.long, device=dev),
                                            torch.tensor([s.seg[s.consumed:]], dtype=torch.long, device=dev))
                        s.consumed = len(s.generated)
                    logits = cached                              # a rejected sample is re-drawn from the same logits
                else:
                    kw = {'attn_kwargs': {'omit_feature_map_draw': len(s.generated) > primed}} if model_type == 'performer' else {}
                    logits = model(torch.tensor([s.generated[-max_dec_inp_len:]], dtype=torch.long, device=dev),
                                   seg_inp=torch.tensor([s.seg[-max_dec_inp_len:]], dtype=torch.long, device=dev), keep_last_only=True, **kw)
                logits_np = logits[0].cpu().numpy().copy()
                if eng is not None and getattr(eng, 'persist', None) is not None:
                    try:
                        eng.check_persistent()                   # (the copy above already synchronised)
                    except EmoError as e:
                        # the one-launch step gave up (its workgroups were not all resident within 50 ms): logits and recurrent state of this
                        # engine are void.  Re-run the piece so far through the chain of launches and continue there.
                        note('[gen] %s -> falling back to the chain of launches' % e)
                        eng = make_engine(model, 1, redraw=False, persistent=False) if model.kind == 'performer' else make_engine(model, 1, persistent=False)
                        s.consumed, cached = 0, None
                        continue
                probs = temperature(logits_np, temp, inadmissibles=inadmissibles)
                bars = s.generated_bars
                if not s.offer(int(draw(probs)), event2idx, idx2event, skip_check, max_events):
                    note('[gen] sample rejected (%d in a row)' % s.failed_cnt)
                elif s.generated_bars != bars:
                    note('[gen] bar %d / %d done, %d events' % (s.generated_bars, s.target_bars, len(s.generated)))
    finally:
        model.train(was_training)
    note('[gen] %d events in %.2f s%s' % (len(s.generated), time.time() - t0, ' (stuck: 256 rejected samples)' if s.stuck else ''))
    return s.result()


# ------------------------------------------------------------------------------------------------ batched reference loop (SURVEY f-4)
class _Stream:
    """Per-stream state of generate_conditional's loop (inference.py:233-250)."""

    def __init__(self, event2idx, lead_sheet_events, primer, max_bars):
        self.lead = lead_sheet_events
        self.generated = list(primer) + [event2idx['Track_LeadSheet']] + list(lead_sheet_events[0]) + [event2idx['Track_Full']]
        self.seg = [0] * len(self.generated)
        self.seg[-1] = 1
        self.target_bars = len(lead_sheet_events) if max_bars is None else min(max_bars, len(lead_sheet_events))
        self.generated_bars, self.cur_pos, self.failed_cnt = 0, 0, 0
        self.consumed = 0            # tokens already folded into the engine state
        self.done, self.stuck = self.target_bars <= 0, False

    def offer(self, word, event2idx, idx2event, skip_check, max_events):
        """One sampled word through the grammar of inference.py:276-318.  Returns False when the sample is rejected (the caller
        re-samples from the SAME distribution, like the reference's `continue`), True when the stream advanced or ended."""
        ev = idx2event[word]
        if not skip_check and 'Beat' in ev:
            pos = beat_position(ev)
            if not pos >= self.cur_pos:
                self.failed_cnt += 1
                if self.failed_cnt >= 256:          # reference: returns `generated` as is (no [:-1])
                    self.done = self.stuck = True
                    return True
                return False
            self.cur_pos, self.failed_cnt = pos, 0
        if ev == 'Track_LeadSheet':
            self.generated.append(word)
            self.seg.append(0)
            self.generated_bars += 1
            if self.generated_bars < self.target_bars:
                nxt = self.lead[self.generated_bars]
                self.generated.extend(nxt)
                self.seg.extend([0] * len(nxt))
                self.generated.append(event2idx['Track_Full'])
                self.seg.append(1)
                self.cur_pos = 0
            else:
                self.done = True
            return True
        if ev == 'PAD_None' or (ev == 'EOS_None' and self.generated_bars < self.target_bars - 1):
            return False
        if ev == 'EOS_None' and self.generated_bars == self.target_bars - 1:
            self.generated.append(word)
            self.done = True
            return True
        self.generated.append(word)
        self.seg.append(1)
        if len(self.generated) > max_events:
            self.done = True
        return True

    def result(self):
        return self.generated if self.stuck else self.generated[:-1]


def generate_conditional_batch(model, event2idx, idx2event, lead_sheets, primers, max_events=10000, skip_check=False, max_bars=None,
                               temp=1.2, top_p=0.9, inadmissibles=None, samplers=None, seeds=None):
    """n independent generate_conditional() runs (one lead sheet + primer each) in lock-step on ONE decode engine: per-stream bar
    counter, Beat position, rejection counter and RNG; every engine step feeds each unfinished stream its next pending token (a
    sampled word, or the next token of an injected lead-sheet bar), so streams of different lengths stay aligned in position.
    `samplers[i](probs)` defaults to nucleus(probs, top_p, rng=RandomState(seeds[i])).  Stream i returns exactly what
    generate_conditional(..., sampler=samplers[i]) returns for it alone (tests/test_gpu_generate.py), as long as it stays inside
    the 2048-token window (max_dec_inp_len); a stream that reaches the window is finished by the single-stream windowed path."""
    n = len(lead_sheets)
    assert n == len(primers) and n > 0
    if samplers is None:
        rss = [np.random.RandomState((seeds[i] if seeds is not None else i)) for i in range(n)]
        samplers = [(lambda probs, rs=rs: nucleus(probs, top_p, rng=rs)) for rs in rss]
    dev = next(model.parameters()).device
    st = [_Stream(event2idx, lead_sheets[i], primers[i], max_bars) for i in range(n)]
    pad = event2idx.get('PAD_None', 0)
    was_training = model.training
    model.eval()
    overflow = []
    try:
        with torch.no_grad():
            eng = make_engine(model, n)
            L0 = min(len(s.generated) for s in st)
            tok = torch.tensor([s.generated[:L0] for s in st], dtype=torch.long, device=dev)
            seg = torch.tensor([s.seg[:L0] for s in st], dtype=torch.long, device=dev)
            logits = eng.prefill(tok, seg)
            for s in st:
                s.consumed = L0
            while True:
                logits_np = None
                for i, s in enumerate(st):
                    if s.done:
                        continue
                    if len(s.generated) >= max_dec_inp_len:      # window slides: positions restart, the recurrent state is void
                        s.done = True
                        overflow.append(i)
                        continue
                    if s.consumed < len(s.generated):
                        continue
                    if logits_np is None:
                        logits_np = logits.cpu().numpy()
                        if getattr(eng, 'persist', None) is not None:
                            eng.check_persistent()
                    while True:                                  # a rejected sample re-derives probs from the same logits (reference: `continue`)
                        probs = temperature(logits_np[i].copy(), temp, inadmissibles=inadmissibles)
                        if s.offer(int(samplers[i](probs)), event2idx, idx2event, skip_check, max_events):
                            break
                if all(s.done for s in st):
                    break
                nxt_tok = [s.generated[s.consumed] if s.consumed < len(s.generated) else pad for s in st]
                nxt_seg = [s.seg[s.consumed] if s.consumed < len(s.seg) else 1 for s in st]
                for s in st:
                    if s.consumed < len(s.generated):
                        s.consumed += 1
                logits = eng.step(torch.tensor(nxt_tok, dtype=torch.long, device=dev), torch.tensor(nxt_seg, dtype=torch.long, device=dev))
    finally:
        model.train(was_training)
    out = [s.result() for s in st]
    for i in overflow:       # rare: hand the stream to the reference-shaped single-stream loop (full-window forward per token)
        out[i] = _resume_windowed(model, event2idx, idx2event, st[i], max_events, skip_check, temp, inadmissibles, samplers[i])
    return out


def _resume_windowed(model, event2idx, idx2event, s, max_events, skip_check, temp, inadmissibles, sampler):
    dev = next(model.parameters()).device
    s.done = False
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            while not s.done:
                dec_input = torch.tensor([s.generated[-max_dec_inp_len:]], dtype=torch.long, device=dev)
                dec_seg = torch.tensor([s.seg[-max_dec_inp_len:]], dtype=torch.long, device=dev)
                kw = {'attn_kwargs': {'omit_feature_map_draw': True}} if model.kind == 'performer' else {}
                logits_np = model(dec_input, seg_inp=dec_seg, keep_last_only=True, **kw)[0].cpu().numpy().copy()
                while True:
                    probs = temperature(logits_np.copy(), temp, inadmissibles=inadmissibles)
                    if s.offer(int(sampler(probs)), event2idx, idx2event, skip_check, max_events):
                        break
    finally:
        model.train(was_training)
    return s.result()


class _Chain:
    """One lock-step group of streams: engine + device-side loop state + (optionally) the captured step graph on its own HIP stream."""

    def __init__(self, model, ptok, pseg, n_new, U, temp, top_p, greedy, seg_value, redraw, persistent=True):
        self.n, self.T0 = ptok.shape
        n, T0, dev = self.n, self.T0, ptok.device
        self.eng = make_engine(model, n, redraw=redraw, persistent=persistent) if model.kind == 'performer' else make_engine(model, n, persistent=persistent)
        eng = self.eng
        self.out = torch.empty(n, T0 + n_new, dtype=torch.long, device=dev)
        self.out[:, :T0] = ptok
        out = self.out
        seg_col = torch.full((n,), seg_value, dtype=torch.long, device=dev)
        logits_buf = eng.prefill(ptok, pseg).clone()
        if not greedy:
            # all loop state on the device inside OUR kernels: the sampler reads u[step[r], r], writes the token into out[r, T0 + step[r]] and
            # advances step[r]; the embedding takes position (T0 - 1) + step[r]; the logits GEMM writes straight into logits_buf
            # (5 fewer launches per token than the torch index_select / scatter_ / add_ / copy_ version below).
            step_ctr = torch.zeros(n, dtype=torch.long, device=dev)
            eng.pos_dev, eng.dev_pos0, eng.pos_auto = step_ctr, T0 - 1, False
            nxt_buf = torch.empty(n, dtype=torch.long, device=dev)

            if getattr(eng, 'persist', None) is not None and os.environ.get('EMO_PD_SAMPLER', '1') != '0':
                # one launch per token: the draw runs inside the persistent step (the same device code as emo_sample_nucleus_step)
                eng.load_logits(logits_buf)
                seg_p = torch.zeros(eng.n_pad, dtype=torch.long, device=dev)
                seg_p[:n] = seg_col
                Uc = U.contiguous()

                def one_step():
                    eng.step_sampled(seg_p, temp, top_p, Uc, step_ctr, out, T0, nxt_buf, T0 - 1)
            else:
                def one_step():
                    ops.sample_nucleus_step(logits_buf, temp, top_p, U, step_ctr, seq=out, col0=T0, out=nxt_buf)
                    eng.step(nxt_buf, seg_col, dev_pos=True, logits_out=logits_buf)
        else:
            step_idx = torch.zeros(1, dtype=torch.long, device=dev)

            def one_step():
                u = U.index_select(0, step_idx).view(n)
                nxt = sample_on_device(logits_buf, temp, top_p, u, greedy)
                out.scatter_(1, (step_idx + T0).expand(n, 1), nxt.view(n, 1))
                logits_buf.copy_(eng.step(nxt, seg_col, dev_pos=True))
                step_idx.add_(1)
        self.one_step = one_step
        self.graph, self.stream = None, None

    def capture(self, steps=1):
        """One graph of `steps` consecutive token steps (all loop state lives on the device, so a replay continues wherever the streams are):
        a replay costs the device a fixed ~10 us of idle time whatever it holds, so several token steps per replay amortise it (r04:
        EMO_GEN_GRAPH_STEPS, default 16; the single-step graph serves the remainder)."""
        dev = self.out.device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            with torch.cuda.graph(g, stream=self.stream):
                for _ in range(steps):
                    self.one_step()
        return g


@torch.no_grad()
def generate_streams(model, prompt_tok, prompt_seg, n_new, temp=1.1, top_p=0.9, greedy=False, seed=0, seg_value=1, use_graph=True, chains=None):
    """BASELINE configs[3]: n parallel streams in lock-step (grammar checks off => fixed token count).  Everything stays
    on the GPU: recurrent/KV state, positions, sampling, token buffer.  One decode step is ~64 small DEPENDENT launches, so the
    step (sample -> append -> embed -> 12 layers -> logits) is captured ONCE in a hipGraph and replayed per token.  `chains` > 1 splits the
    streams into independent groups whose graphs replay on separate HIP streams (same tokens as one group: every stream keeps its own
    uniform draws); measured r01: SLOWER (0.52 / 0.89 / 0.60 / 1.11 ms per step for 1 / 2 / 4 / 8 chains) — the replays are issued by one
    host thread and do not overlap — so the default is one chain.
    Returns int64 [n, T0 + n_new]."""
    n, T0 = prompt_tok.shape
    dev = prompt_tok.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    U = torch.rand(max(n_new, 1), n, device=dev, generator=gen)
    if chains is None:
        chains = int(os.environ.get('EMO_GEN_CHAINS', 1))
    if not use_graph or n_new <= 2 or chains < 1 or n % chains != 0 or n // chains < 1:
        chains = 1
    m = n // chains
    cs = []
    for c in range(chains):
        rows = slice(c * m, (c + 1) * m)
        cs.append(_Chain(model, prompt_tok[rows].contiguous(), prompt_seg[rows].contiguous(), n_new, U[:, rows].contiguous(), temp, top_p, greedy,
                         seg_value, redraw=(c == 0), persistent=(chains == 1)))      # (two persistent launches on two streams can each be
                                                                                      # partially resident and starve each other: one chain only)
    if n_new > 0:
        for ch in cs:
            ch.one_step()                        # eager first step (also warms every kernel / attribute cache)
        if not use_graph:
            for _ in range(n_new - 1):
                cs[0].one_step()
        elif n_new > 1:
            torch.cuda.synchronize()
            left = n_new - 1
            k = max(1, int(os.environ.get('EMO_GEN_GRAPH_STEPS', 16)))
            for ch in cs:
                ch.graph = ch.capture(1)
                ch.graph_k = ch.capture(k) if (k > 1 and left >= 2 * k) else None
            main = torch.cuda.current_stream()
            t_host = time.perf_counter()
            while left > 0:
                many = cs[0].graph_k is not None and left >= k
                for ch in cs:
                    with torch.cuda.stream(ch.stream):
                        (ch.graph_k if many else ch.graph).replay()
                left -= k if many else 1
            if os.environ.get('EMO_GEN_TIMING'):            # diagnostics: host time to ENQUEUE the replays vs time until the GPU is done
                t_enq = time.perf_counter() - t_host
                torch.cuda.synchronize()
                print('[gen timing] enqueue %.3f ms/step, total %.3f ms/step' % (1e3 * t_enq / (n_new - 1), 1e3 * (time.perf_counter() - t_host) / (n_new - 1)))
            for ch in cs:
                main.wait_stream(ch.stream)
    for ch in cs:
        if getattr(ch.eng, 'persist', None) is not None:
            ch.eng.check_persistent()
    return cs[0].out if chains == 1 else torch.cat([ch.out for ch in cs], 0)


# ------------------------------------------------------------------------------------------------ command line (reference inference.py:330-485)
def read_lead_sheet(path, event2idx):
    """A stage-1 output file: one event per line, optionally a Key_* line first, bars opened by Bar_None.  -> (key event, [[ids of bar 0], ...])"""
    events = open(path).read().splitlines()
    key = events[0] if events and 'Key' in events[0] else 'Key_C'
    starts = [i for i, e in enumerate(events) if e == 'Bar_None'] + [len(events)]
    return key, [[event2idx[e] for e in events[a:b]] for a, b in zip(starts[:-1], starts[1:])]


def emotions_of(file_name):
    for tag, cands in (('Positive', ['Q1', 'Q4']), ('Negative', ['Q2', 'Q3']), ('Q1', ['Q1']), ('Q2', ['Q2']), ('Q3', ['Q3']), ('Q4', ['Q4']), ('None', ['None'])):
        if tag in file_name:
            return cands
    raise ValueError('wrong emotion label')


def main(argv=None):
    """Same flags as the reference's stage-2 inference.py (-m / -c / -r / -i / -o): every lead sheet found in the output directory gets its
    accompaniment, all jobs of a run in lock-step on ONE decode engine (--streams at a time).  The generated events are written as text
    (`<piece>_<emotion>_full.txt`, one event per line); turning them into MIDI is the reference's convert2midi.py (needs miditoolkit) and
    is run on those files when that module is importable."""
    import argparse
    import yaml
    from . import train as tr
    from .data import load_vocab
    ap = argparse.ArgumentParser(description='stage-2 accompaniment generation on MI355X')
    req = ap.add_argument_group('required arguments')
    req.add_argument('-m', '--model_type', choices=['performer', 'gpt2'], required=True)
    req.add_argument('-c', '--configuration', required=True)
    req.add_argument('-r', '--representation', choices=['remi', 'functional'], required=True)
    ap.add_argument('-i', '--inference_params', required=True, help='checkpoint (.pt state dict)')
    ap.add_argument('-o', '--output_dir', required=True, help='directory holding the stage-1 lead sheets; results are written next to them')
    ap.add_argument('--streams', type=int, default=32, help='lead sheets generated in lock-step on one decode engine')
    ap.add_argument('--max_bars', type=int, default=128)
    ap.add_argument('--dtype', default=None, choices=[None, 'bf16', 'fp32'])
    args = ap.parse_args(argv)
    conf = yaml.load(open(args.configuration), Loader=yaml.FullLoader)
    torch.cuda.set_device(conf['training']['gpuid'])
    event2idx, idx2event, pad = load_vocab(conf['data_loader']['vocab_path'].format(args.representation))
    model = tr.build_model(args.model_type, pad + 1, conf['model'], args.dtype).cuda()
    tr.load_pretrained(model, args.inference_params)
    model.eval()
    temp, top_p = (1.1, 0.99) if args.model_type == 'performer' else (1.2, 0.97)
    print('[info] temp = %s | top_p = %s' % (temp, top_p))
    pat = 'roman.txt' if args.representation == 'functional' else '.txt'
    jobs = []
    for f in sorted(os.listdir(args.output_dir)):
        if pat not in f or f.endswith('_full.txt'):
            continue
        key, bars = read_lead_sheet(os.path.join(args.output_dir, f), event2idx)
        for e in emotions_of(f):
            out = os.path.join(args.output_dir, '_'.join(f.split('_')[:2]) + '_' + e + '_full.txt')
            if os.path.exists(out):
                print('[info] %s exists, skipping ...' % out)
                continue
            primer = [event2idx['Emotion_%s' % e]] + ([event2idx[key]] if args.representation == 'functional' else []) + [event2idx['Tempo_110']]
            jobs.append((out, key, bars, primer))
    print('[# jobs]', len(jobs))
    for i in range(0, len(jobs), args.streams):
        group = jobs[i:i + args.streams]
        gen = generate_conditional_batch(model, event2idx, idx2event, [g[2] for g in group], [g[3] for g in group], max_bars=args.max_bars,
                                         temp=temp, top_p=top_p, seeds=list(range(i, i + len(group))))
        for (out, key, _, _), ids in zip(group, gen):
            with open(out, 'w') as fh:
                fh.write('\n'.join([key] + [idx2event[w] for w in ids]) + '\n')
            print('[info] wrote', out, len(ids), 'events')
    try:
        import convert2midi  # noqa: F401  (the reference's module, if the user put it on the path together with miditoolkit)
        print('[info] convert2midi is importable: run it on the *_full.txt files to obtain MIDI')
    except ImportError:
        print('[info] event files written; MIDI conversion (reference convert2midi.py, needs miditoolkit) is outside this package')


if __name__ == '__main__':
    main()
