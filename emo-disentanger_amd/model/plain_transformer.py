"""Stage-1 lead-sheet language model (Transformer-XL decoder), SURVEY §8 f-1 — inference path.

Mirrors /root/reference/stage1_compose/model/plain_transformer.py:14-93 (PlainTransformer: constructor, forward, generate, compute_loss) on
top of OptimusTXLDecoder with attn_type 0 (optimus_txl_decoder.py:299-391, 700-925): same constructor arguments, parameter names, registration
order (= optimizer state order) and state-dict keys, so reference checkpoints load unchanged.  Evaluation forward, training forward /
backward (all eight dropout sites of the reference) and token-by-token generation with memory run on the HIP kernels (relative-position
attention: emo_relpos_attn_fwd / _bwd / _decode); the backward of the attention is a first version (dense by-products + GEMMs).

The reference keeps `mems` = the hidden states of the last mem_len positions per layer and re-projects them to keys / values at every step
(:309-316); the key / value of a position never changes, so the engine caches K and V instead (`TXLMemory`, returned where the reference
returns its list of tensors and accepted back by generate()).  R = r_net(pos_emb) depends only on the distance and is computed once per
engine.  Layout: the reference is time-major ([T, B] tokens, [T, B, V] logits); the kernels work batch-major internally."""
import os

import torch
from torch import nn
import torch.nn.functional as F

from emo_disentanger_amd import engine, ops
from emo_disentanger_amd._lib import EmoError


def weights_init(m):
    """stage1_compose/model/transformer_helpers.py:24-66 (the branches this model reaches)."""
    name = m.__class__.__name__
    if name.find('Linear') != -1:
        nn.init.normal_(m.weight, 0.0, 0.01)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0.0)
    elif name.find('Embedding') != -1:
        if hasattr(m, 'weight'):
            nn.init.normal_(m.weight, 0.0, 0.01)
    elif name.find('LayerNorm') != -1:
        nn.init.normal_(m.weight, 1.0, 0.01)
        nn.init.constant_(m.bias, 0.0)
    elif name.find('TXLDecoder') != -1:
        nn.init.normal_(m.r_w_bias, 0.0, 0.01)
        nn.init.normal_(m.r_r_bias, 0.0, 0.01)


class WordEmbedding(nn.Module):
    def __init__(self, n_token, d_embed, d_proj, emb_scale=0.5, pad_idx=None):
        super().__init__()
        if d_proj != d_embed:
            raise NotImplementedError('d_word_embed != d_model (emb_proj) is not used by any stage-1 YAML and is not built')
        self.n_token, self.d_embed, self.d_proj = n_token, d_embed, d_proj
        self.emb_scale = d_proj ** emb_scale
        self.emb_lookup = nn.Embedding(n_token, d_embed, padding_idx=n_token - 1 if pad_idx is None else pad_idx)


class PositionalEmbedding(nn.Module):
    def __init__(self, demb):
        super().__init__()
        self.demb = demb
        self.register_buffer('inv_freq', 1 / (10000 ** (torch.arange(0.0, demb, 2.0) / demb)))

    def forward(self, pos_seq):
        s = torch.outer(pos_seq, self.inv_freq)
        return torch.cat([s.sin(), s.cos()], dim=-1)


class _RelAttn(nn.Module):                      # parameter container: RelPartialLearnableMultiHeadAttn
    def __init__(self, n_head, d_model, d_head):
        super().__init__()
        self.qkv_net = nn.Linear(d_model, 3 * n_head * d_head, bias=False)
        self.o_net = nn.Linear(n_head * d_head, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model)
        self.r_net = nn.Linear(d_model, n_head * d_head, bias=False)


class _PosFF(nn.Module):                        # parameter container: PositionwiseFF (CoreNet indices 0 and 3 hold the Linears)
    def __init__(self, d_model, d_inner, dropout):
        super().__init__()
        self.CoreNet = nn.Sequential(nn.Linear(d_model, d_inner), nn.ReLU(inplace=True), nn.Dropout(dropout), nn.Linear(d_inner, d_model), nn.Dropout(dropout))
        self.layer_norm = nn.LayerNorm(d_model)


class _DecoderLayer(nn.Module):
    def __init__(self, n_head, d_model, d_head, d_inner, dropout):
        super().__init__()
        self.dec_attn = _RelAttn(n_head, d_model, d_head)
        self.pos_ff = _PosFF(d_model, d_inner, dropout)


class OptimusTXLDecoder(nn.Module):             # the class name matters: weights_init matches 'TXLDecoder'
    def __init__(self, n_layer, n_head, d_model, d_head, d_inner, dropout, tgt_len, mem_len, pre_lnorm):
        super().__init__()
        self.n_layer, self.n_head, self.d_model, self.d_head = n_layer, n_head, d_model, d_head
        self.tgt_len, self.mem_len, self.ext_len, self.pre_lnorm = tgt_len, mem_len, 0, pre_lnorm
        self.r_w_bias = nn.Parameter(torch.zeros(n_head, d_head))
        self.r_r_bias = nn.Parameter(torch.zeros(n_head, d_head))
        self.layers = nn.ModuleList([_DecoderLayer(n_head, d_model, d_head, d_inner, dropout) for _ in range(n_layer)])
        self.pos_emb = PositionalEmbedding(d_model)


def _txl_layer_fwd(ps, p, x, r_dist, B, T, H, pd, seed, off, pre, save, mem=None):
    """RelPartialLearnableDecoderLayer (:526-557) = rel. attention (:301-391) + PositionwiseFF (:28-66), training or evaluation.
    mem [B, mlen, D] (segment recurrence, :312-321): keys / values come from LayerNorm + qkv_net of cat([mem, x]) and query i sees keys
    j <= mlen + i at distance mlen + i - j — exactly causal attention over the concatenated sequence restricted to its last T query rows,
    which is how it is run (the first mlen output rows are dropped; their queries cost compute but no extra kernel)."""
    D = x.shape[1]
    mlen = 0 if mem is None else mem.shape[1]
    K = mlen + T
    xq = x
    if mlen:
        x = torch.cat([mem, x.view(B, T, D)], 1).reshape(B * K, D)
    a, f = p + 'dec_attn.', p + 'pos_ff.'
    if not pre:
        raise NotImplementedError('post-LN (pre_lnorm=False) training is not built: every stage-1 YAML sets pre_lnorm: True')
    n, m1, r1 = ops.layernorm_fwd(x, ps.f32(a + 'layer_norm.weight'), ps.f32(a + 'layer_norm.bias'))
    qkv = ops.gemm(n, ps.w(a + 'qkv_net.weight'))
    vec, lse, zden = ops.relpos_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], r_dist, ps.f32('decoder.r_w_bias'), ps.f32('decoder.r_r_bias'),
                                         B, K, H, p_drop=pd, seed=seed, offset=off + 1)
    vq = vec.view(B, K, D)[:, mlen:].reshape(B * T, D) if mlen else vec
    h = ops.gemm(vq, ps.w(a + 'o_net.weight'), p_drop=pd, seed=seed, offset=off + 2, residual=xq)
    n2, m2, r2 = ops.layernorm_fwd(h, ps.f32(f + 'layer_norm.weight'), ps.f32(f + 'layer_norm.bias'))
    g = ops.gemm(n2, ps.w(f + 'CoreNet.0.weight'), bias=ps.f32(f + 'CoreNet.0.bias'), act=ops.ACT_RELU, p_drop=pd, seed=seed, offset=off + 3)
    o = ops.gemm(g, ps.w(f + 'CoreNet.3.weight'), bias=ps.f32(f + 'CoreNet.3.bias'), p_drop=pd, seed=seed, offset=off + 4, residual=h)
    if save is not None:
        save.update(x=x, m1=m1, r1=r1, n=n, qkv=qkv, vec=vec, vq=vq, mlen=mlen, lse=lse, zden=zden, h=h, m2=m2, r2=r2, n2=n2, g=g, r_dist=r_dist)
    return o


import os as _os
_FUSE_BELOW = _os.environ.get('EMO_S1_FUSE', '1') != '0'      # (A/B switch of the cross-layer LayerNorm-backward fusion)


def _txl_layer_bwd(ps, p, dout, pe_d, B, T, H, pd, seed, off, s, acc, dyd=None, bias3_done=False, below=None, dR_out=None, dq_rel_out=None):
    """dyd / bias3_done: dout with this layer's output dropout already re-applied and the CoreNet.3 bias gradient already accumulated — by the
    LayerNorm backward of the layer ABOVE, which produced dout (below = (offset of the output-dropout site, CoreNet.3 bias gradient) of the
    layer below: this layer's last LayerNorm backward does the same for it).  Saves three ~3-us launches per layer of a launch-bound step.
    Returns (dx, dx with the lower layer's dropout, bias gradient done)."""
    D = dout.shape[1]
    a, f = p + 'dec_attn.', p + 'pos_ff.'
    inv = 1.0 / (1.0 - pd) if pd > 0 else 1.0
    # weight gradients on the engine's side stream, as the Performer does below 32768 rows (r03 measured no gain, 7.98 against 8.02 ms/step: the
    # host then queued a step no faster than the GPU ran it; r05, with the host at 4.5 ms per step, the GPU bounds the step and the overlap counts)
    wg = lambda dy, xin, wname, bname=None: ops.gemm(dy, xin, a_trans=True, b_trans=True, out=ps.g(wname), accumulate=True,
                                                     a_rowsum=None if bname is None else ps.g(bname), stream=engine._side_fork(dy, xin))
    if dyd is None:
        dyd = ops.dropout_apply(dout, pd, seed, off + 4) if pd > 0 else dout
    wg(dyd, s['g'], f + 'CoreNet.3.weight', None if bias3_done else f + 'CoreNet.3.bias')
    dg = ops.gemm(dyd, ps.w(f + 'CoreNet.3.weight'), b_trans=True, mul_aux=s['g'], mul_mode=ops.MUL_NONZERO, mul_scale=inv)
    wg(dg, s['n2'], f + 'CoreNet.0.weight', f + 'CoreNet.0.bias')
    dn2 = ops.gemm(dg, ps.w(f + 'CoreNet.0.weight'), b_trans=True)
    dh, dad = ops.layernorm_bwd(dn2, s['h'], ps.f32(f + 'layer_norm.weight'), s['m2'], s['r2'], ps.g(f + 'layer_norm.weight'), ps.g(f + 'layer_norm.bias'),
                                dres=dout, want_drop=pd > 0, p_drop=pd, seed=seed, offset=off + 2)       # dad = dh re-masked with the o_net output dropout
    if dad is None:
        dad = dh
    wg(dad, s['vq'], a + 'o_net.weight')
    dvec = ops.gemm(dad, ps.w(a + 'o_net.weight'), b_trans=True)
    mlen = s['mlen']
    K = mlen + T
    if mlen:                                                    # memory rows: no output gradient (their outputs were dropped), no input gradient
        pad = lambda t: torch.cat([t.new_zeros(B, mlen, D), t.view(B, T, D)], 1).reshape(B * K, D)
        dvec, dh_res = pad(dvec), pad(dh)
    else:
        dh_res = dh
    # (the column sums behind d r_w_bias / d r_r_bias — parameters shared by all layers — accumulate into `acc`; TXLStackFn.backward adds them once)
    dqkv, dR, _, _ = ops.relpos_attn_bwd(s['qkv'], s['r_dist'], ps.f32('decoder.r_w_bias'), ps.f32('decoder.r_r_bias'), s['vec'], dvec, s['lse'],
                                         s['zden'], B, K, H, p_drop=pd, seed=seed, offset=off + 1, acc_dq=None, acc_rr=acc[3 * D:], dR_out=dR_out, dq_rel_out=dq_rel_out if not mlen else None)
    if dR_out is None:
        wg(dR.to(ps.compute_dtype), pe_d, a + 'r_net.weight')                 # R = r_net(dropout(pos_emb)): dW_r += dR^T pos_emb
    # (colsum(dq) = the first D entries of the column sums of dqkv, which the weight-gradient GEMM takes from its operand fragments: acc[:3D])
    ops.gemm(dqkv, s['n'], a_trans=True, b_trans=True, out=ps.g(a + 'qkv_net.weight'), accumulate=True, a_rowsum=acc[:3 * D], stream=engine._side_fork(dqkv, s['n']))
    dn = ops.gemm(dqkv, ps.w(a + 'qkv_net.weight'), b_trans=True)
    if below is not None and not mlen:                           # the layer below wants dx * its output dropout and colsum of that (its CoreNet.3 bias)
        dx, dxd = ops.layernorm_bwd(dn, s['x'], ps.f32(a + 'layer_norm.weight'), s['m1'], s['r1'], ps.g(a + 'layer_norm.weight'), ps.g(a + 'layer_norm.bias'),
                                    dres=dh_res, want_drop=pd > 0, p_drop=pd, seed=seed, offset=below[0], dcol=below[1])
        return dx, (dxd if dxd is not None else dx), True
    dx, _ = ops.layernorm_bwd(dn, s['x'], ps.f32(a + 'layer_norm.weight'), s['m1'], s['r1'], ps.g(a + 'layer_norm.weight'), ps.g(a + 'layer_norm.bias'),
                              dres=dh_res)
    return (dx.view(B, K, D)[:, mlen:].reshape(B * T, D) if mlen else dx), None, False


class TXLStackFn(torch.autograd.Function):
    """tokens [B, T] -> final hidden states [B, T, D] (OptimusTXLDecoder._forward :750-925 with attn_type 0, mem_len 0).  Dropout sites as in
    the reference: emb_dropout AND decoder.drop on the embedding, decoder.drop on pos_emb, dropatt on the probabilities, drop on the attention
    output, the two CoreNet dropouts, decoder.drop on the final hidden state.  Parameter gradients go straight into the flat grad buffer."""

    @staticmethod
    def forward(ctx, model, tok, anchor, need_bwd, mems=None):
        """mems: None or L+1 tensors [B, mlen, D] (compute dtype): the previous segments' layer inputs.  The layer inputs of THIS segment
        (the reference's `hids`, :783-841) are left in model._hids for the memory update."""
        ps = model._store
        B, T = tok.shape
        mlen = 0 if mems is None else mems[0].shape[1]
        D, H, L = model.dec_d_model, model.dec_n_head, model.dec_n_layer
        pd = model.dec_dropout if model.training else 0.0
        model._fwd_counter += 1
        seed, base = model._seed, model._fwd_counter * 4096
        x = ops.embed_fwd(tok, None, ps.f32('word_emb.emb_lookup.weight'), None, model._zero_pe(T), ps.compute_dtype, float(model.word_emb.emb_scale),
                          p_drop=pd, seed=seed, offset=base).view(B * T, D)
        if pd > 0:
            x = ops.dropout_apply(x, pd, seed, base + 1)
        pe = model.decoder.pos_emb(torch.arange(mlen + T, device=ps.device, dtype=torch.float32)).to(ps.compute_dtype).contiguous()   # row d = distance d
        pe_d = ops.dropout_apply(pe, pd, seed, base + 3) if pd > 0 else pe
        saves, hids = [], [x]
        r_all = ops.gemm(pe_d, ps.w('decoder.layers.0.dec_attn.r_net.weight', L * D))      # [n_dist, L*D]: layer l's R = columns l*D .. (l+1)*D
        for l in range(L):
            sv = {} if need_bwd else None
            r_dist = r_all[:, l * D:(l + 1) * D]
            x = _txl_layer_fwd(ps, 'decoder.layers.%d.' % l, x, r_dist, B, T, H, pd, seed, base + 8 * (l + 1), model.decoder.pre_lnorm, sv,
                               mem=None if not mlen else mems[l])
            saves.append(sv)
            hids.append(x)
        model._hids = hids if model.dec_mem_len > 0 else None
        if pd > 0:
            x = ops.dropout_apply(x, pd, seed, base + 2)
        ctx.model, ctx.saves, ctx.tok, ctx.pe_d = model, saves, tok, pe_d
        ctx.cfg = (B, T, D, H, L, pd, seed, base)
        return x.view(B, T, D)

    @staticmethod
    def backward(ctx, dout):
        model, ps = ctx.model, ctx.model._store
        B, T, D, H, L, pd, seed, base = ctx.cfg
        ps.ensure_grads()
        dx = dout.reshape(B * T, D)
        if dx.dtype != ps.compute_dtype or not dx.is_contiguous():
            dx = dx.to(ps.compute_dtype).contiguous()
        if pd > 0:
            dx = ops.dropout_apply(dx, pd, seed, base + 2)
        acc = torch.zeros(4 * D, device=dx.device, dtype=torch.float32)       # [colsum(dqkv) (3D), colsum(dq_relative) (D)] summed over the layers
        dyd, b3 = None, False
        n_dist = ctx.pe_d.shape[0]
        dR_all = torch.empty(n_dist, L * D, device=dx.device, dtype=torch.float32)       # every layer's dR (its kernel writes all n_dist rows of its block)
        dq_rel_all = torch.empty(L, B * T, D, device=dx.device, dtype=ps.compute_dtype)   # every layer's relative part of dq: ONE column sum below
        stacked = True
        for l in reversed(range(L)):
            stacked = stacked and not (ctx.saves[l]['mlen'])
            below = (base + 8 * l + 4, ps.g('decoder.layers.%d.pos_ff.CoreNet.3.bias' % (l - 1))) if (l > 0 and _FUSE_BELOW) else None
            dx, dyd, b3 = _txl_layer_bwd(ps, 'decoder.layers.%d.' % l, dx, ctx.pe_d, B, T, H, pd, seed, base + 8 * (l + 1), ctx.saves[l], acc,
                                         dyd=dyd, bias3_done=b3, below=below, dR_out=dR_all[:, l * D:(l + 1) * D], dq_rel_out=dq_rel_all[l])
            ctx.saves[l] = None
        if stacked:
            ops.colsum(dq_rel_all.view(L * B * T, D), out=acc[3 * D:], accumulate=True)
        engine.join_side_stream()                                             # acc[:3 D] and the weight gradients are complete
        # R_l = r_net_l(dropout(pos_emb)): dW_r[l] += dR_l^T pos_emb, all layers in one product (the weights are adjacent in the store)
        ops.gemm(dR_all.to(ps.compute_dtype), ctx.pe_d, a_trans=True, b_trans=True, out=ps.g('decoder.layers.0.dec_attn.r_net.weight', L * D), accumulate=True)
        ps.g('decoder.r_r_bias').add_(acc[3 * D:].view(H, D // H))             # d r_r_bias = colsum(dq_relative)
        ps.g('decoder.r_w_bias').add_((acc[:D] - acc[3 * D:]).view(H, D // H)) # d r_w_bias = colsum(dq) - colsum(dq_relative)
        if pd > 0:
            dx = ops.dropout_apply(dx, pd, seed, base + 1)
        gE = ps.g('word_emb.emb_lookup.weight')
        ops.embed_bwd(ctx.tok, None, dx, gE, None, float(model.word_emb.emb_scale), p_drop=pd, seed=seed, offset=base)
        gE[model.word_emb.emb_lookup.padding_idx].zero_()            # nn.Embedding(padding_idx): that row never receives a gradient
        return None, None, None, None, None


class TXLMemory:
    """What generate() hands back in place of the reference's list of `mems` tensors: per-layer K / V caches of one lock-step group of streams."""

    def __init__(self, model, n_streams, max_len):
        ps = model._ensure_store()
        self.n, self.max_len, self.len = n_streams, max_len, 0
        D, L = model.dec_d_model, model.dec_n_layer
        self.kc = [torch.zeros(n_streams, max_len, D, device=ps.device, dtype=ps.compute_dtype) for _ in range(L)]
        self.vc = [torch.zeros(n_streams, max_len, D, device=ps.device, dtype=ps.compute_dtype) for _ in range(L)]
        self.lens = torch.zeros(n_streams, device=ps.device, dtype=torch.int64)
        self.r_dist = model._r_by_distance(max_len)

    def __len__(self):                    # the reference's callers only test / print len(mems)
        return len(self.kc) + 1


class PlainTransformer(nn.Module):
    def __init__(self, d_word_embed, vocab_size, dec_n_layer, dec_n_head, dec_d_model, dec_d_ff, dec_mem_len, dec_tgt_len,
                 dec_dropout=0.1, dec_activation='relu', pad_index=None, pre_lnorm=False, compute_dtype=None, max_gen_len=4096):
        super().__init__()
        self.d_word_embed, self.vocab_size = d_word_embed, vocab_size
        self.dec_n_layer, self.dec_n_head, self.dec_d_model, self.dec_d_ff = dec_n_layer, dec_n_head, dec_d_model, dec_d_ff
        self.dec_dropout, self.dec_activation, self.dec_mem_len, self.dec_tgt_len = dec_dropout, dec_activation, dec_mem_len, dec_tgt_len
        self.word_emb = WordEmbedding(vocab_size, d_word_embed, dec_d_model)
        self.emb_dropout = nn.Dropout(dec_dropout)
        self.pad_index = vocab_size - 1 if pad_index is None else pad_index
        self.decoder = OptimusTXLDecoder(dec_n_layer, dec_n_head, dec_d_model, dec_d_model // dec_n_head, dec_d_ff, dec_dropout,
                                         tgt_len=dec_tgt_len, mem_len=dec_mem_len, pre_lnorm=pre_lnorm)
        self.dec_out_proj = nn.Linear(dec_d_model, vocab_size)
        self.apply(weights_init)
        self._compute_dtype = engine._dt(compute_dtype or os.environ.get('EMO_COMPUTE_DTYPE', 'bf16'))
        self._store, self._max_gen_len, self._zero_pe_buf = None, max_gen_len, None
        self._fwd_counter, self._seed = 0, int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ engine plumbing
    def _ensure_store(self):
        if self._store is None or not self._store.intact() or self._store.compute_dtype != self._compute_dtype:
            # (the r_net weights of all layers sit back to back: R of every layer is ONE [n_dist, L*D] product of the shared position embedding,
            # and their weight gradients one product of the concatenated dR — 44 launches per training step fewer, r04)
            rnets = ['decoder.layers.%d.dec_attn.r_net.weight' % l for l in range(self.dec_n_layer)]
            self._store = engine.ParamStore(self, self._compute_dtype, [rnets] if len(rnets) > 1 else [])
        self._store.sync_mirror()
        return self._store

    def set_compute_dtype(self, name):
        self._compute_dtype, self._store = engine._dt(name), None
        return self

    def set_dropout_seed(self, seed):
        self._seed, self._fwd_counter = int(seed), 0

    def _zero_pe(self, n):
        if self._zero_pe_buf is None or self._zero_pe_buf.shape[0] < n or self._zero_pe_buf.device != self._store.device:
            self._zero_pe_buf = torch.zeros(max(n, 64), self.dec_d_model, device=self._store.device)
        return self._zero_pe_buf

    def _r_by_distance(self, n_dist):
        """R[l][d] = r_net_l(pos_emb(d)) for d = 0 .. n_dist-1 (evaluation: no dropout on pos_emb), in the compute dtype.  Row d is the
        reference's r_head_k[klen-1-d] (optimus_txl_decoder.py:318, 791-796)."""
        ps = self._ensure_store()
        pe = self.decoder.pos_emb(torch.arange(n_dist, device=ps.device, dtype=torch.float32)).to(ps.compute_dtype).contiguous()
        return [ops.gemm(pe, ps.w('decoder.layers.%d.dec_attn.r_net.weight' % l)) for l in range(self.dec_n_layer)]     # (contiguous per layer: the decode kernels index rows)

    def _embed(self, tok_bm, pos0=0):
        ps = self._store
        B, T = tok_bm.shape
        return ops.embed_fwd(tok_bm, None, ps.f32('word_emb.emb_lookup.weight'), None, self._zero_pe(pos0 + T), ps.compute_dtype,
                             float(self.word_emb.emb_scale)).view(-1, self.dec_d_model)

    def _layer(self, l, x, attn_fn):
        """One RelPartialLearnableDecoderLayer (:526-557) in evaluation mode; attn_fn(qkv) -> attention vectors [M, D]."""
        ps, p = self._store, 'decoder.layers.%d.' % l
        a, f = p + 'dec_attn.', p + 'pos_ff.'
        pre = self.decoder.pre_lnorm
        n = ops.layernorm_fwd(x, ps.f32(a + 'layer_norm.weight'), ps.f32(a + 'layer_norm.bias'))[0] if pre else x
        vec = attn_fn(ops.gemm(n, ps.w(a + 'qkv_net.weight')))
        h = ops.gemm(vec, ps.w(a + 'o_net.weight'), residual=x)
        if not pre:
            h = ops.layernorm_fwd(h, ps.f32(a + 'layer_norm.weight'), ps.f32(a + 'layer_norm.bias'))[0]
        n2 = ops.layernorm_fwd(h, ps.f32(f + 'layer_norm.weight'), ps.f32(f + 'layer_norm.bias'))[0] if pre else h
        g = ops.gemm(n2, ps.w(f + 'CoreNet.0.weight'), bias=ps.f32(f + 'CoreNet.0.bias'), act=ops.ACT_RELU)
        o = ops.gemm(g, ps.w(f + 'CoreNet.3.weight'), bias=ps.f32(f + 'CoreNet.3.bias'), residual=h)
        return o if pre else ops.layernorm_fwd(o, ps.f32(f + 'layer_norm.weight'), ps.f32(f + 'layer_norm.bias'))[0]

    def _logits(self, h):
        ps = self._store
        return ops.gemm(h, ps.w('dec_out_proj.weight'), bias=ps.f32('dec_out_proj.bias'), out_dtype=torch.float32)

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def _prefill(self, dec_input, mem=None):
        """Full-segment pass: dec_input int64 [T, B] -> (hidden [B*T, D], B, T); fills `mem` (K / V of every position) when given."""
        if not dec_input.is_cuda:
            raise EmoError('inputs must be GPU tensors (the HIP path has no CPU fallback)')
        ps = self._ensure_store()
        tok = dec_input.t().contiguous().long()
        B, T = tok.shape
        D, H = self.dec_d_model, self.dec_n_head
        r_dist = mem.r_dist if mem is not None else self._r_by_distance(T)
        x = self._embed(tok)
        rw, rr = ps.f32('decoder.r_w_bias'), ps.f32('decoder.r_r_bias')
        for l in range(self.dec_n_layer):
            def attn(qkv, l=l):
                if mem is not None:
                    mem.kc[l][:, :T].copy_(qkv[:, D:2 * D].view(B, T, D))
                    mem.vc[l][:, :T].copy_(qkv[:, 2 * D:].view(B, T, D))
                return ops.relpos_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], r_dist[l], rw, rr, B, T, H)[0]
            x = self._layer(l, x, attn)
        if mem is not None:
            mem.len = T
            mem.lens.fill_(T)
        return x, B, T

    def forward(self, dec_input, dec_mems, dec_seg_len=None, return_avg_attn=False):
        """plain_transformer.py:62-80.  dec_input int64 [T, B]; returns (logits fp32 [T, B, V], new_mems).  mem_len = 0 (every training / validation
        YAML): new_mems is the empty list, as in the reference."""
        if return_avg_attn:
            raise NotImplementedError('return_avg_attn is an analysis path of the reference and is not built')
        anchor = self.word_emb.emb_lookup.weight
        if self.dec_mem_len > 0 or (dec_mems is not None and len(dec_mems) > 0):
            return self._forward_with_memory(dec_input, dec_mems, dec_seg_len, anchor)
        if self.training or (torch.is_grad_enabled() and anchor.requires_grad):
            if not dec_input.is_cuda:
                raise EmoError('inputs must be GPU tensors (the HIP path has no CPU fallback)')
            self._ensure_store()
            tok = dec_input.t().contiguous().long()
            h = TXLStackFn.apply(self, tok, anchor, torch.is_grad_enabled() and anchor.requires_grad)
            return engine.LogitsFn.apply(self, h).permute(1, 0, 2), []
        h, B, T = self._prefill(dec_input)
        logits = self._logits(h).view(B, T, self.vocab_size).permute(1, 0, 2)
        return logits, []

    def _forward_with_memory(self, dec_input, dec_mems, dec_seg_len, anchor):
        """Segment-level recurrence (optimus_txl_decoder.py:750-925 with mems, memory update :702-748).  dec_mems: () / [] for the first
        segment, then the list this method returned: L+1 tensors [mlen, B, D] (time-major like the reference's, compute dtype, detached)."""
        if not dec_input.is_cuda:
            raise EmoError('inputs must be GPU tensors (the HIP path has no CPU fallback)')
        if self.dec_mem_len <= 0:
            raise NotImplementedError('incoming mems with mem_len = 0 (attend to a memory that is never updated) is not built')
        ps = self._ensure_store()
        L, D = self.dec_n_layer, self.dec_d_model
        tok = dec_input.t().contiguous().long()
        B, T = tok.shape
        if isinstance(dec_mems, (tuple, list)) and len(dec_mems) == 1 and isinstance(dec_mems[0], (tuple, list)):
            dec_mems = dec_mems[0]                               # the reference accepts the list wrapped in a 1-tuple (:754-756)
        mems = None
        if dec_mems is not None and len(dec_mems) > 0 and dec_mems[0].numel() > 0:
            assert len(dec_mems) == L + 1, 'len(mems) must be n_layer + 1'
            mems = [m.detach().to(ps.compute_dtype).permute(1, 0, 2).contiguous() for m in dec_mems]      # [B, mlen, D]
        mlen = 0 if mems is None else mems[0].shape[1]
        h = TXLStackFn.apply(self, tok, anchor, torch.is_grad_enabled() and anchor.requires_grad, mems)
        logits = engine.LogitsFn.apply(self, h).permute(1, 0, 2)
        hids, self._hids = self._hids, None
        new_mems = []
        with torch.no_grad():
            for i in range(L + 1):
                hid = hids[i].detach().view(B, T, D)
                old = mems[i] if mems is not None else hid.new_zeros(B, 0, D)
                if dec_seg_len is None:                          # :719-724 (ext_len = 0): the last mem_len of the mlen + T cached steps
                    cat = torch.cat([old, hid], 1)
                    new_mems.append(cat[:, max(0, mlen + T - self.dec_mem_len):].permute(1, 0, 2).contiguous())
                else:                                            # :726-746: per sample, its first dec_seg_len[b] steps; left-padded with zeros
                    assert dec_seg_len.shape[0] == B
                    rows = []
                    for b_i in range(B):
                        c = torch.cat([old[b_i], hid[b_i, :int(dec_seg_len[b_i])]], 0)
                        rows.append(c[max(0, c.shape[0] - self.dec_mem_len):])
                    width = max(r.shape[0] for r in rows)
                    rows = [torch.cat([r.new_zeros(width - r.shape[0], D), r], 0) for r in rows]
                    new_mems.append(torch.stack(rows, 1))
        return logits, new_mems

    @torch.no_grad()
    def generate(self, dec_input, dec_mems):
        """plain_transformer.py:52-59: first call = the whole primer [L, B] with dec_mems = tuple(); later calls one token [[id]] with the memory
        returned by the previous call.  Returns (logits fp32 [V] of the last position of stream 0, memory)."""
        if self.training:
            raise NotImplementedError('generate() is an evaluation path: call .eval()')
        D, H = self.dec_d_model, self.dec_n_head
        if not isinstance(dec_mems, TXLMemory):
            mem = TXLMemory(self, dec_input.shape[1], self._max_gen_len)
            h, B, T = self._prefill(dec_input, mem)
            return self._logits(h.view(B, T, D)[:, -1].contiguous())[0], mem
        mem, ps = dec_mems, self._ensure_store()
        if dec_input.shape[0] != 1:
            # the reference loop re-submits the whole primer with the memory of the primer when the very first sample is rejected
            # (inference_utils.py:67-70 + `continue`); token i of a segment sees the memory and tokens <= i, which is what feeding them one at a
            # time does as long as the window does not slide inside the segment
            if self.dec_mem_len > 0 and mem.len + dec_input.shape[0] > self.dec_mem_len + 1:
                raise NotImplementedError('a multi-token segment that overflows mem_len while it is processed is not built')
            logits = None
            for i in range(dec_input.shape[0]):
                logits, mem = self.generate(dec_input[i:i + 1], mem)
            return logits, mem
        if mem.len >= mem.max_len:
            raise EmoError('generation longer than max_gen_len=%d: construct the model with a larger max_gen_len' % mem.max_len)
        tok = dec_input.t().contiguous().long()
        x = self._embed(tok)
        mem.lens.add_(1)
        mem.len += 1
        rw, rr = ps.f32('decoder.r_w_bias'), ps.f32('decoder.r_r_bias')
        for l in range(self.dec_n_layer):
            def attn(qkv, l=l):
                return ops.relpos_attn_decode(qkv[:, :D], mem.kc[l], mem.vc[l], mem.lens, H, mem.r_dist[l], rw, rr, mem_len=self.dec_mem_len,
                                              k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:])
            x = self._layer(l, x, attn)
        return self._logits(x)[0], mem

    def compute_loss(self, dec_logits, dec_tgt, reduction='mean'):
        """plain_transformer.py:82-93 (evaluation use: validation loss); the fused cross-entropy of the stage-2 path."""
        if reduction != 'mean':
            raise NotImplementedError("only reduction='mean' (the reference's only use) is built")
        V = dec_logits.size(-1)
        ce = engine.XentFn.apply(dec_logits.reshape(-1, V).contiguous(), dec_tgt.contiguous().view(-1).long(), self.pad_index).float()
        return {'ce_loss': ce, 'total_loss': ce}
