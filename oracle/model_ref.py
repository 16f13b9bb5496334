"""ORACLE (test infrastructure, NOT product code).

Functional PyTorch-fp32 restatement of the stage-2 causal LM forward pass —
same op sequence as the reference so that on CPU the
GPT-2 path is bit-identical to the imported reference (golden-pinned), and
autograd over these ops is the backward oracle.

Reference citations (relative to /root/reference):
  prologue      stage2_accompaniment/model/music_performer.py:50-62,
                model/transformer_helpers.py:81-87, :57-63
  performer     model/fast_transformer_decoder.py:54-74 + upstream
                pytorch-fast-transformers (AttentionLayer, CausalLinearAttention,
                Favor, TransformerEncoderLayer) — **parity unpinned** (absent dep)
  gpt2 block    model/music_gpt2.py:84-86 + HF transformers==4.28.0 GPT2Block
  logits        music_performer.py:65-68 ; loss music_performer.py:72-81
"""
import ctypes
import math
import os
import subprocess

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c_oracle(force=False):
    """Compile oracle/causal_product_ref.c -> oracle/_build/libcausal_ref.so (gcc -O2 -fopenmp)."""
    out_dir = os.path.join(_HERE, '_build')
    so = os.path.join(out_dir, 'libcausal_ref.so')
    src = os.path.join(_HERE, 'causal_product_ref.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-fopenmp', '-shared', '-fPIC', '-o', so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c_oracle())
    return _LIB


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class CausalDotProductC(torch.autograd.Function):
    """upstream fast_transformers/causal_product/__init__.py: autograd.Function
    around the native CPU kernel; here bound to oracle/causal_product_ref.c."""

    @staticmethod
    def forward(ctx, Q, K, V):
        Q, K, V = Q.contiguous().float(), K.contiguous().float(), V.contiguous().float()
        N, H, L, E = Q.shape
        M = V.shape[-1]
        out = torch.zeros(N, H, L, M)
        i64 = ctypes.c_int64
        _lib().causal_dot_product_ref(_p(Q), _p(K), _p(V), _p(out), i64(N), i64(H), i64(L), i64(E), i64(M))
        ctx.save_for_backward(Q, K, V)
        return out

    @staticmethod
    def backward(ctx, dO):
        Q, K, V = ctx.saved_tensors
        dO = dO.contiguous().float()
        N, H, L, E = Q.shape
        M = V.shape[-1]
        dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
        i64 = ctypes.c_int64
        _lib().causal_dot_backward_ref(_p(Q), _p(K), _p(V), _p(dO), _p(dQ), _p(dK), _p(dV),
                                       i64(N), i64(H), i64(L), i64(E), i64(M))
        return dQ, dK, dV


# ----------------------------------------------------------------------------- FAVOR+
def favor_features(x, omega):
    """upstream feature_maps/fourier_features.py Favor.forward (stabilize=False).
    x [..., d_head]; omega [d_head, F/2]; returns [..., F], all > 0."""
    d_head = x.shape[-1]
    n_dims = 2 * omega.shape[1]
    softmax_temp = 1.0 / math.sqrt(d_head)
    x = x * math.sqrt(softmax_temp)
    norm_x_squared = torch.einsum("...d,...d->...", x, x).unsqueeze(-1)
    u = x.unsqueeze(-2).matmul(omega).squeeze(-2)
    offset = norm_x_squared * 0.5 + 0.5 * math.log(n_dims)
    return torch.cat([torch.exp(u - offset), torch.exp(-u - offset)], dim=-1)


def causal_linear_attention(q, k, v, omega, eps=1e-6, form='prefix'):
    """upstream attention/causal_linear_attention.py CausalLinearAttention.forward.
    q,k,v [N,L,H,dh] -> [N,L,H,dh].  `form` selects one of three algebraically
    identical evaluations used to cross-check each other:
      'prefix'    cumsum normaliser + sequential causal product (C kernel) — upstream's
      'quadratic' O(L^2) masked form
      'recurrent' token-by-token state update (what a decode engine does)"""
    Q, K = favor_features(q, omega), favor_features(k, omega)
    if form == 'prefix':
        Z = 1 / (torch.einsum("nlhi,nlhi->nlh", Q, K.cumsum(1)) + eps)
        Vn = CausalDotProductC.apply(Q.permute(0, 2, 1, 3).contiguous(), K.permute(0, 2, 1, 3).contiguous(),
                                     v.permute(0, 2, 1, 3).contiguous()).permute(0, 2, 1, 3)
        return Vn * Z[:, :, :, None]
    if form == 'quadratic':
        L = q.shape[1]
        A = torch.einsum("nlhi,nshi->nhls", Q, K) * torch.tril(torch.ones(L, L, dtype=Q.dtype))
        num = torch.einsum("nhls,nshd->nlhd", A, v)
        den = A.sum(-1).permute(0, 2, 1) + eps
        return num / den[..., None]
    if form == 'recurrent':
        N, L, H, Fd = Q.shape
        S = torch.zeros(N, H, Fd, v.shape[-1], dtype=Q.dtype)
        z = torch.zeros(N, H, Fd, dtype=Q.dtype)
        outs = []
        for t in range(L):
            S = S + K[:, t, :, :, None] * v[:, t, :, None, :]
            z = z + K[:, t]
            num = torch.einsum("nhf,nhfd->nhd", Q[:, t], S)
            den = torch.einsum("nhf,nhf->nh", Q[:, t], z) + eps
            outs.append(num / den[..., None])
        return torch.stack(outs, 1)
    raise ValueError(form)


# ----------------------------------------------------------------------------- dropout with a GIVEN mask
def _drop(x, p_drop, training, masks=None, site=None):
    """F.dropout(x, p, training) is x * keep / (1 - p) with keep ~ Bernoulli(1 - p) (torch nn/functional.py dropout).  `masks` (dict site ->
    MULTIPLIER tensor keep / (1 - p), broadcastable to x) replaces the draw by a given one, so that a run of the HIP path with dropout ON can be
    compared element for element: tests export the multipliers the product's kernels used (tests/dropmask.py).  Sites: 'emb'; Performer layer l:
    'L<l>.attn_out', 'L<l>.ffn_hidden', 'L<l>.ffn_out' (upstream TransformerEncoderLayer.forward: the three self.dropout calls); GPT-2 block l:
    'L<l>.attn_prob' (HF GPT2Attention._attn attn_dropout), 'L<l>.attn_out' (resid_dropout after c_proj), 'L<l>.mlp_out' (GPT2MLP dropout)."""
    if masks is None:
        return F.dropout(x, p_drop, training)
    if not training or p_drop == 0.0:
        return x
    return x * masks[site].to(x.dtype).view(x.shape)


# ----------------------------------------------------------------------------- prologue / epilogue
def prologue(sd, x, seg_inp, d_model, p_drop=0.0, training=False, chord_inp=None, masks=None):
    """music_performer.py:51-62: (E[x] (*proj))*sqrt(d) + (S[seg])*sqrt(d) (+ chord_emb(chord_inp), :56-57) + PE[:T] -> dropout."""
    emb = F.embedding(x, sd['token_emb.emb_lookup.weight'])
    if 'token_emb.emb_proj.weight' in sd:
        emb = F.linear(emb, sd['token_emb.emb_proj.weight'])
    emb = emb * (d_model ** 0.5)
    if seg_inp is not None and 'segemb.emb_lookup.weight' in sd:
        s = F.embedding(seg_inp, sd['segemb.emb_lookup.weight'])
        if 'segemb.emb_proj.weight' in sd:
            s = F.linear(s, sd['segemb.emb_proj.weight'])
        emb = emb + s * (d_model ** 0.5)
    if chord_inp is not None and 'chord_emb.weight' in sd:
        emb = emb + F.linear(chord_inp, sd['chord_emb.weight'], sd['chord_emb.bias'])
    T = x.size(1)
    h = emb + sd['pe.pe'][:T].permute(1, 0, 2)
    return _drop(h, p_drop, training, masks, 'emb')


def logits_head(sd, h, keep_last_only=False):
    out = F.linear(h, sd['dec_out_proj.weight'], sd['dec_out_proj.bias'])
    return out[:, -1, :] if keep_last_only else out


def compute_loss(logits, tgt, n_token, reduction='mean'):
    """music_performer.py:72-81."""
    return F.cross_entropy(logits.view(-1, logits.size(-1)), tgt.contiguous().view(-1),
                           ignore_index=n_token - 1, reduction=reduction).float()


# ----------------------------------------------------------------------------- Performer
def performer_layer(sd, p, h, n_head, omega, p_drop=0.0, training=False, form='prefix', masks=None, site='', activation='relu'):
    """upstream transformers.py TransformerEncoderLayer.forward (post-LN; activation = F.relu if activation == "relu" else F.gelu) around
    attention_layer.py AttentionLayer.forward."""
    N, L, D = h.shape
    dh = D // n_head
    q = F.linear(h, sd[p + 'attention.query_projection.weight'], sd[p + 'attention.query_projection.bias']).view(N, L, n_head, dh)
    k = F.linear(h, sd[p + 'attention.key_projection.weight'], sd[p + 'attention.key_projection.bias']).view(N, L, n_head, dh)
    v = F.linear(h, sd[p + 'attention.value_projection.weight'], sd[p + 'attention.value_projection.bias']).view(N, L, n_head, dh)
    a = causal_linear_attention(q, k, v, omega, form=form).reshape(N, L, D)
    a = F.linear(a, sd[p + 'attention.out_projection.weight'], sd[p + 'attention.out_projection.bias'])
    x = h + _drop(a, p_drop, training, masks, site + 'attn_out')
    y = x = F.layer_norm(x, (D,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-5)
    act = F.relu if activation == 'relu' else F.gelu
    y = _drop(act(F.linear(y, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])), p_drop, training, masks, site + 'ffn_hidden')
    y = _drop(F.linear(y, sd[p + 'linear2.weight'], sd[p + 'linear2.bias']), p_drop, training, masks, site + 'ffn_out')
    return F.layer_norm(x + y, (D,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-5)


def performer_forward(sd, x, seg_inp, n_layer, n_head, d_model, omegas=None, keep_last_only=False,
                      p_drop=0.0, training=False, form='prefix', chord_inp=None, masks=None, activation='relu'):
    """MusicPerformer.forward (music_performer.py:50-70). `omegas`: list of [dh,F/2]
    (the reference redraws omega every forward — SURVEY F8 — so parity runs inject it)."""
    h = prologue(sd, x, seg_inp, d_model, p_drop, training, chord_inp, masks)
    for l in range(n_layer):
        p = 'transformer_decoder.decoder_layers.%d.' % l
        om = omegas[l] if omegas is not None else sd[p + 'attention.inner_attention.feature_map.omega']
        h = performer_layer(sd, p, h, n_head, om, p_drop, training, form, masks, 'L%d.' % l, activation)
    return logits_head(sd, h, keep_last_only)


# ----------------------------------------------------------------------------- GPT-2
def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def gpt2_block(sd, p, h, n_head, p_drop=0.0, training=False, masks=None, site=''):
    """HF 4.28 GPT2Block.forward (pre-LN) with GPT2Attention._attn eager math."""
    N, L, D = h.shape
    dh = D // n_head
    n = F.layer_norm(h, (D,), sd[p + 'ln_1.weight'], sd[p + 'ln_1.bias'], 1e-5)
    qkv = torch.addmm(sd[p + 'attn.c_attn.bias'], n.view(-1, D), sd[p + 'attn.c_attn.weight']).view(N, L, 3 * D)
    q, k, v = qkv.split(D, dim=2)
    q = q.view(N, L, n_head, dh).permute(0, 2, 1, 3)
    k = k.view(N, L, n_head, dh).permute(0, 2, 1, 3)
    v = v.view(N, L, n_head, dh).permute(0, 2, 1, 3)
    w = torch.matmul(q, k.transpose(-1, -2))
    w = w / torch.full([], dh ** 0.5, dtype=w.dtype)
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool))[None, None]
    w = torch.where(causal, w, torch.full([], torch.finfo(w.dtype).min, dtype=w.dtype))
    w = _drop(F.softmax(w, dim=-1), p_drop, training, masks, site + 'attn_prob')
    a = torch.matmul(w, v).permute(0, 2, 1, 3).contiguous().view(N, L, D)
    a = torch.addmm(sd[p + 'attn.c_proj.bias'], a.view(-1, D), sd[p + 'attn.c_proj.weight']).view(N, L, D)
    h = h + _drop(a, p_drop, training, masks, site + 'attn_out')
    m = F.layer_norm(h, (D,), sd[p + 'ln_2.weight'], sd[p + 'ln_2.bias'], 1e-5)
    f = torch.addmm(sd[p + 'mlp.c_fc.bias'], m.view(-1, D), sd[p + 'mlp.c_fc.weight'])
    f = gelu_new(f)
    f = torch.addmm(sd[p + 'mlp.c_proj.bias'], f, sd[p + 'mlp.c_proj.weight']).view(N, L, D)
    return h + _drop(f, p_drop, training, masks, site + 'mlp_out')


def gpt2_forward(sd, x, seg_inp, n_layer, n_head, d_model, keep_last_only=False, p_drop=0.0, training=False, chord_inp=None, masks=None):
    """MusicGPT2.forward (music_gpt2.py:70-92): no final ln_f."""
    h = prologue(sd, x, seg_inp, d_model, p_drop, training, chord_inp, masks)
    for i in range(n_layer):
        h = gpt2_block(sd, 'transformer_decoder.%d.' % i, h, n_head, p_drop, training, masks, 'L%d.' % i)
    return logits_head(sd, h, keep_last_only)


def forward(kind, sd, x, seg_inp, n_layer, n_head, d_model, **kw):
    if kind == 'performer':
        return performer_forward(sd, x, seg_inp, n_layer, n_head, d_model, **kw)
    kw.pop('omegas', None), kw.pop('form', None), kw.pop('activation', None)
    return gpt2_forward(sd, x, seg_inp, n_layer, n_head, d_model, **kw)


def loss_and_grads(kind, sd, batch, n_token, n_layer, n_head, d_model, **kw):
    """fwd + CE + autograd backward over every floating-point parameter (pe.pe, omega are buffers)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith('pe.pe') and 'omega' not in k}
    full = dict(sd)
    full.update(params)
    if torch.is_tensor(batch.get('chords_mhot')) and 'chord_emb.weight' in sd:
        kw = dict(kw, chord_inp=batch['chords_mhot'])
    logits = forward(kind, full, batch['dec_input'], batch['track_mask'], n_layer, n_head, d_model, **kw)
    loss = compute_loss(logits, batch['dec_target'], n_token)
    loss.backward()
    return loss.detach(), logits.detach(), {k: p.grad for k, p in params.items()}
