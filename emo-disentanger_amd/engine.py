"""Execution engine under MusicPerformer / MusicGPT2: flat parameter storage, per-layer forward /
backward schedules over the C-ABI ops, and the single autograd node that wraps the whole decoder
stack (one Python-level node => ~10 kernel launches per layer, no per-op autograd bookkeeping).

Storage layout (HBM): all trainable parameters live in ONE fp32 buffer (``flat32``) with a matching
``flat_grad`` (one RCCL all-reduce, one fused Adam launch) and, in bf16 mode, a bf16 mirror
(``flat16``) that the MFMA GEMMs read.  The three Performer projections q/k/v are adjacent in the
buffer so that one [3D, D] view feeds a fused QKV GEMM; ``nn.Parameter`` objects are views into the
flat buffers, so ``state_dict()`` keys / shapes / registration order stay the reference's.
Activations are [B*T, D] row-major in the compute dtype; LayerNorm statistics, softmax / FAVOR
normalisers, logits and all gradients of parameters are fp32.
"""
import math
import weakref

import torch

from . import _lib, ops
from ._lib import EmoError

ALIGN = 8  # elements; keeps every parameter 16-B aligned in the bf16 mirror


def _dt(name):
    if name in ('bf16', 'bfloat16', torch.bfloat16):
        return torch.bfloat16
    if name in ('fp32', 'f32', 'float32', torch.float32):
        return torch.float32
    raise ValueError('compute dtype must be bf16 or fp32, got %r' % (name,))


class ParamStore:
    """Flat fp32 master / grad / (bf16 mirror) buffers; parameters are re-pointed to views."""

    def __init__(self, module, compute_dtype, fused_groups):
        named = list(module.named_parameters())
        dev = named[0][1].device
        if dev.type != 'cuda':
            raise EmoError('emo-disentanger_amd models run on the GPU only (no CPU fallback): call .cuda() first')
        byname = dict(named)
        order, seen = [], set()
        for n, _ in named:
            if n in seen:
                continue
            grp = next((g for g in fused_groups if n == g[0]), None)
            for m in (grp if grp else [n]):
                order.append(m)
                seen.add(m)
        self.offsets, off = {}, 0
        fused = {g[0]: g for g in fused_groups}
        for n in order:
            self.offsets[n] = off
            sz = byname[n].numel()
            in_group = any(n in g for g in fused_groups)
            off += sz if in_group else (sz + ALIGN - 1) // ALIGN * ALIGN
        total = (off + ALIGN - 1) // ALIGN * ALIGN
        self.device, self.total, self.compute_dtype = dev, total, compute_dtype
        self.flat32 = torch.zeros(total, device=dev, dtype=torch.float32)
        # one extra ALIGN-sized tail: slot [total] carries the rank's non-pad token count through the DP all-reduce (dp.py)
        self.flat_grad_ext = torch.zeros(total + ALIGN, device=dev, dtype=torch.float32)
        self.flat_grad = self.flat_grad_ext[:total]
        self.flat16 = torch.zeros(total, device=dev, dtype=torch.bfloat16) if compute_dtype == torch.bfloat16 else None
        self.params = byname
        self.shapes = {n: tuple(p.shape) for n, p in named}
        with torch.no_grad():
            for n, p in named:
                o = self.offsets[n]
                view = self.flat32[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        self._fused = fused
        self._views = {}
        self._pad = {}
        self._mirror_version, self._mirror_epoch, self._wT = -1, 0, {}
        self._first, self._last = named[0][1], named[-1][1]
        self._plist = [p for _, p in named]

    # -- consistency ---------------------------------------------------------------------------
    def intact(self):
        """False after module._apply (.cuda()/.to()/.float()) replaced the parameter storage."""
        lo, hi = self.flat32.data_ptr(), self.flat32.data_ptr() + 4 * self.total
        return all(p.device == self.device and lo <= p.data_ptr() < hi for p in (self._first, self._last))

    def _write_signal(self):
        """Changes whenever torch wrote the fp32 master: through the flat buffer (broadcast, flat32.copy_) or through a parameter
        (optimizer.step, load_state_dict, init_ — parameters were attached with `p.data = view`, which keeps each parameter's OWN
        version counter, so the base's counter alone does not see those writes)."""
        return self.flat32._version + sum(p._version for p in self._plist)

    def sync_mirror(self):
        """Refresh the bf16 mirror if anything wrote the fp32 master through torch since the last refresh."""
        if self.flat16 is not None:
            sig = self._write_signal()
            if sig != self._mirror_version:
                ops.cast(self.flat32, self.flat16)
                self._mirror_version = sig
                self._mirror_epoch += 1

    def mark_mirror_fresh(self):
        """The caller's kernel wrote fp32 master and bf16 mirror together (fused Adam)."""
        self._mirror_version = self._write_signal()
        self._mirror_epoch += 1

    def wT(self, name, fused_rows=None):
        """bf16 TRANSPOSED copy [in, out] of the nn.Linear weight `name` ([out, in]; fused_rows: the [3D, D] q/k/v view), refreshed when the
        mirror changes: the dgrad dX = dY W then is the same k-contiguous NT product as a forward (K = 512: the A-stationary kernel; the
        long reductions: the 256 x 256 tile kernel).  All registered mirrors are refreshed by ONE batched transpose launch per mirror epoch
        (emo_transpose_batch) instead of one copy kernel each."""
        key = (name, fused_rows)
        ent = self._wT.get(key)
        if ent is None:
            shape = self.shapes[name] if fused_rows is None else (fused_rows, self.shapes[name][1])
            ent = self._wT[key] = [torch.empty(shape[::-1], device=self.device, dtype=torch.bfloat16), -1, shape]
            self._wT_desc = None                                  # table rebuilt with the new record
        if ent[1] != self._mirror_epoch:
            self._refresh_wT()
        return ent[0]

    def padded(self, name, n, fill=0.0, master=False):
        """Copy of parameter `name` with its first dimension padded to n entries of `fill` (compute dtype; master=True: fp32), refreshed when the
        mirror changes.  The output projection's vocabulary (327) padded to 512: engine.LogitsFn."""
        key = (name, n, master)
        ent = self._pad.get(key)
        if ent is None:
            shp = self.shapes[name]
            ent = self._pad[key] = [torch.full((n,) + tuple(shp[1:]), fill, device=self.device, dtype=torch.float32 if master else self.compute_dtype), -1]
        if ent[1] != self._mirror_epoch:
            src = self.f32(name) if master else self.w(name)
            ent[0][:src.shape[0]].copy_(src)
            ent[1] = self._mirror_epoch
        return ent[0]

    def _refresh_wT(self):
        if getattr(self, '_wT_desc', None) is None:
            rec, tile = [], 0
            for (name, fused_rows), (buf, _, shape) in self._wT.items():
                rows, cols = shape
                tpr = (cols + 63) // 64
                rec.append([self.w(name, fused_rows).data_ptr(), buf.data_ptr(), rows, cols, tile, tpr])
                tile += ((rows + 63) // 64) * tpr
            self._wT_desc = (torch.tensor(rec, dtype=torch.int64, device=self.device), len(rec), tile)
        desc, n, tiles = self._wT_desc
        ops.transpose_batch(desc, n, tiles)
        for ent in self._wT.values():
            ent[1] = self._mirror_epoch

    def invalidate_mirror(self):
        self._mirror_version = -1

    def ensure_grads(self):
        """Re-attach .grad views (zero_grad(set_to_none=True) drops them)."""
        missing = [n for n, p in self.params.items() if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * self.offsets[n]]
        if not missing:
            return
        if len(missing) == len(self.params):
            self.flat_grad.zero_()
        for n in missing:
            p, o = self.params[n], self.offsets[n]
            g = self.flat_grad[o:o + p.numel()].view(p.shape)
            if len(missing) != len(self.params):
                if p.grad is not None:
                    g.copy_(p.grad)
                else:
                    g.zero_()
            p.grad = g

    # -- views ------------------------------------------------------------------------------------
    def _view(self, buf, name, rows=None):
        # (views of the three flat buffers are cached: ~340 per training step, 2.7 us each to slice and reshape — a ninth of the host time that
        # bounds the step at the reference YAML's batch size 4, r05.  The buffers live as long as the store; the cache is keyed by their address.)
        key = (buf.data_ptr(), name, rows)
        v = self._views.get(key)
        if v is None:
            o, shp = self.offsets[name], self.shapes[name]
            if name in self._fused and rows is not None:
                n = sum(self.params[m].numel() for m in self._fused[name])
                v = buf[o:o + n].view(rows, -1) if len(shp) == 2 else buf[o:o + n]
            else:
                v = buf[o:o + self.params[name].numel()].view(shp)
            self._views[key] = v
        return v

    def w(self, name, fused_rows=None):
        """GEMM operand view (compute dtype)."""
        return self._view(self.flat16 if self.flat16 is not None else self.flat32, name, fused_rows)

    def f32(self, name, fused_rows=None):
        return self._view(self.flat32, name, fused_rows)

    def g(self, name, fused_rows=None):
        return self._view(self.flat_grad, name, fused_rows)


def embedding_table(ps, prefix):
    """fp32 table the fused gather reads for TokenEmbedding `prefix` ('token_emb.' / 'segemb.'): emb_lookup.weight, or its projection
    emb_lookup.weight @ emb_proj.weight^T [n, d_model] when d_embed != d_model (transformer_helpers.py:75-78,84-85)."""
    E = ps.f32(prefix + 'emb_lookup.weight')
    if prefix + 'emb_proj.weight' in ps.params:
        return ops.gemm(E, ps.f32(prefix + 'emb_proj.weight'))
    return E


def embedding_table_bwd(ps, prefix, d_table):
    """Back through the projection: dE += dT P, dP += dT^T E (d_table: gradient of the projected table, fp32 [n, d_model])."""
    E, P = ps.f32(prefix + 'emb_lookup.weight'), ps.f32(prefix + 'emb_proj.weight')
    ops.gemm(d_table, P, b_trans=True, out=ps.g(prefix + 'emb_lookup.weight'), accumulate=True)
    ops.gemm(d_table, E, a_trans=True, b_trans=True, out=ps.g(prefix + 'emb_proj.weight'), accumulate=True)


def check_ids(t, n, what, also=None):
    """Index inputs are range-checked like torch's embedding / cross_entropy do (the kernels index LDS tables and logits rows with
    them): one asynchronous device-side assertion, no host sync.  `also`: one extra admissible value (ignore_index).  EMO_CHECK_IDS=0
    removes the three tiny launches."""
    if t is None or not _CHECK_IDS:
        return
    ok = (t >= 0) & (t < n)
    if also is not None:
        ok = ok | (t == also)
    torch._assert_async(ok.all(), '%s out of range [0, %d)' % (what, n))


import os as _os0
_CHECK_IDS = _os0.environ.get('EMO_CHECK_IDS', '1') != '0'


# =================================================================================================== layer schedules
class LayerCtx:
    __slots__ = ('t',)

    def __init__(self):
        self.t = {}


def performer_layer_fwd(ps, pfx, x, omega, B, T, H, p, seed, off, save, act='relu', ln_in=None, defer_norm2=False):
    """Post-LN encoder layer with FAVOR+ causal attention.  x: [M, D].  act: 'relu' (every YAML of the reference) or 'gelu'
    (upstream's F.gelu, passed through at fast_transformer_decoder.py:50: exact erf form on the generic epilogue, pre-activation saved).
    LayerNorm inside the product that consumes it (emo_hip.h lna_*, r05): norm1 runs inside the linear1 product; with defer_norm2 the layer returns
    its un-normalised x2 and the NEXT layer passes ln_in = (gamma2, beta2, this layer's save) to run norm2 inside its QKV product (the statistics land
    in this layer's save for the backward)."""
    D = x.shape[1]
    q = pfx + 'attention.query_projection.'
    if ln_in is not None:                                          # x is the previous layer's raw x2
        qkv, x, m_prev, r_prev = ops.gemm(x, ps.w(q + 'weight', 3 * D), bias=ps.f32(q + 'bias', 3 * D), lna=(ln_in[0], ln_in[1], 1e-5))
        if ln_in[2] is not None:
            ln_in[2].t['m2'], ln_in[2].t['r2'] = m_prev, r_prev
    else:
        qkv = ops.gemm(x, ps.w(q + 'weight', 3 * D), bias=ps.f32(q + 'bias', 3 * D))
    # (training with a time-segmented scan — B*H < 256: the call's workspace stays with the layer, its K-state increments serve the backward)
    fws = None
    if save is not None:
        attn, den, fws = ops.favor_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], omega, B, T, H, keep_ws=True)
    else:
        attn, den = ops.favor_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], omega, B, T, H)
    x1 = ops.gemm(attn, ps.w(pfx + 'attention.out_projection.weight'), bias=ps.f32(pfx + 'attention.out_projection.bias'),
                  p_drop=p, seed=seed, offset=off + 1, residual=x)
    W1 = ps.w(pfx + 'linear1.weight')
    ln1_in = act == 'relu' and ops.gemm_lna_ok(x.shape[0], W1.shape[0], D, x1.dtype)
    if not ln1_in:
        h1, m1, r1 = ops.layernorm_fwd(x1, ps.f32(pfx + 'norm1.weight'), ps.f32(pfx + 'norm1.bias'))
    # 1-bit relu.dropout mask for the FFN2 dgrad (1/16 of the bytes of re-reading f), when the shape runs on the A-stationary kernel
    fmask = torch.empty(x.shape[0], W1.shape[0] // 8, device=x.device, dtype=torch.uint8) \
        if (act == 'relu' and save is not None and ops.gemm_bitmask_ok(x.shape[0], W1.shape[0], D, x1.dtype, x1.dtype)) else None
    z = None
    # the feed-forward block in ONE launch (emo_hip.h: emo_ffn_fwd, r06): LayerNorm1 in registers, the hidden chunk handed from the FFN1 accumulators
    # straight into the FFN2 product, residual from the registers; f / mask / h1 / statistics are written for the backward as before.  Bit-identical to
    # the two launches and, measured inside the step, exactly as fast (695 us against 383 + 245 us per layer; same-box 44.00 vs 43.94 ms per step over
    # four alternations: DESIGN §7) — opt-in with EMO_FFN_FUSED=1, the two-launch form stays the default.
    if (ln1_in and fmask is not None and save is not None and _os.environ.get('EMO_FFN_FUSED', '0') == '1'
            and ops.ffn_fwd_ok(x.shape[0], D, W1.shape[0], x1.dtype)):
        f, h1, m1, r1, fmask, x2 = ops.ffn_fwd(x1, ps.f32(pfx + 'norm1.weight'), ps.f32(pfx + 'norm1.bias'), W1, ps.f32(pfx + 'linear1.bias'),
                                               ps.w(pfx + 'linear2.weight'), ps.f32(pfx + 'linear2.bias'), p_drop=p, seed=seed, offset_f=off + 2, offset_y=off + 3)
        if defer_norm2:
            out, m2, r2 = x2, None, None
        else:
            out, m2, r2 = ops.layernorm_fwd(x2, ps.f32(pfx + 'norm2.weight'), ps.f32(pfx + 'norm2.bias'))
        save.t = dict(x=x, qkv=qkv, attn=attn, den=den, x1=x1, m1=m1, r1=r1, h1=h1, f=f, fmask=fmask, z=z, x2=x2, m2=m2, r2=r2, omega=omega, fws=fws)
        return out
    chf = int(_os.environ.get('EMO_FFN_CHUNK_F', 0))
    if ln1_in and chf and x.shape[0] > chf and x.shape[0] % chf == 0:      # PROBE: FFN1 -> FFN2 per row chunk (the hidden chunk stays in the Infinity Cache)
        M_ = x.shape[0]
        f = torch.empty(M_, W1.shape[0], device=x.device, dtype=x1.dtype)
        h1, x2 = torch.empty_like(x1), torch.empty_like(x1)
        m1, r1 = torch.empty(M_, device=x.device, dtype=torch.float32), torch.empty(M_, device=x.device, dtype=torch.float32)
        for r0 in range(0, M_, chf):
            sl = slice(r0, r0 + chf)
            ops.gemm(x1[sl], W1, bias=ps.f32(pfx + 'linear1.bias'), act=ops.ACT_RELU, p_drop=p, seed=seed, offset=off + 2, mask_out=None if fmask is None else fmask[sl],
                     lna=(ps.f32(pfx + 'norm1.weight'), ps.f32(pfx + 'norm1.bias'), 1e-5), lna_out=(h1[sl], m1[sl], r1[sl]), out=f[sl])
            ops.gemm(f[sl], ps.w(pfx + 'linear2.weight'), bias=ps.f32(pfx + 'linear2.bias'), p_drop=p, seed=seed, offset=off + 3, residual=h1[sl], out=x2[sl])
        if defer_norm2:
            out, m2, r2 = x2, None, None
        else:
            out, m2, r2 = ops.layernorm_fwd(x2, ps.f32(pfx + 'norm2.weight'), ps.f32(pfx + 'norm2.bias'))
        if save is not None:
            save.t = dict(x=x, qkv=qkv, attn=attn, den=den, x1=x1, m1=m1, r1=r1, h1=h1, f=f, fmask=fmask, z=z, x2=x2, m2=m2, r2=r2, omega=omega, fws=fws)
        return out
    if ln1_in:
        f, h1, m1, r1 = ops.gemm(x1, W1, bias=ps.f32(pfx + 'linear1.bias'), act=ops.ACT_RELU, p_drop=p, seed=seed, offset=off + 2, mask_out=fmask,
                                 lna=(ps.f32(pfx + 'norm1.weight'), ps.f32(pfx + 'norm1.bias'), 1e-5))
    elif act == 'relu':
        f = ops.gemm(h1, W1, bias=ps.f32(pfx + 'linear1.bias'), act=ops.ACT_RELU, p_drop=p, seed=seed, offset=off + 2, mask_out=fmask)
    else:
        z = torch.empty(x.shape[0], W1.shape[0], device=x.device, dtype=h1.dtype) if save is not None else None      # pre-activation for gelu'
        f = ops.gemm(h1, W1, bias=ps.f32(pfx + 'linear1.bias'), act=ops.ACT_GELU, aux_out=z, p_drop=p, seed=seed, offset=off + 2)
    x2 = ops.gemm(f, ps.w(pfx + 'linear2.weight'), bias=ps.f32(pfx + 'linear2.bias'), p_drop=p, seed=seed, offset=off + 3, residual=h1)
    if defer_norm2:
        out, m2, r2 = x2, None, None                              # norm2 runs inside the next layer's QKV product, which fills m2 / r2 of this save
    else:
        out, m2, r2 = ops.layernorm_fwd(x2, ps.f32(pfx + 'norm2.weight'), ps.f32(pfx + 'norm2.bias'))
    if save is not None:
        save.t = dict(x=x, qkv=qkv, attn=attn, den=den, x1=x1, m1=m1, r1=r1, h1=h1, f=f, fmask=fmask, z=z, x2=x2, m2=m2, r2=r2, omega=omega, fws=fws)
    return out


# Weight-gradient GEMMs (and the remaining bias column sums) only feed the optimizer, so they CAN run on a second HIP stream
# next to the dgrad / attention-backward chain of the main stream (EMO_WGRAD_STREAM=1).  Measured r01: worth 3.5 % while the
# wgrad kernel still paid for split-K atomics (79.9 -> 77.1 ms/step); with the workspace split-K the kernels no longer leave
# gaps to fill and sharing the CUs costs more than it hides (72.9 ms with, 70.8 ms without) -> off at the benchmark batch.  At small token
# counts (the reference YAML's batch size 4: 8192 tokens, GEMM grids of 64-256 blocks on 256 CUs) the chip is underfilled and the second stream
# wins (r03, same box: 9.08 -> 8.84 ms/step) -> EMO_WGRAD_STREAM unset = on below 32768 tokens; =1 always, =0 never.
import os as _os
_SIDE = {'stream': None, 'raw': None, 'device': None, 'on': _os.environ.get('EMO_WGRAD_STREAM', '') == '1', 'auto': _os.environ.get('EMO_WGRAD_STREAM', '') == '', 'used': False}


def _side_on(tensors):
    if _SIDE['on']:
        return True
    t = next((x for x in tensors if x is not None), None)
    return bool(_SIDE['auto'] and t is not None and t.is_cuda and t.shape[0] < 32768)


_AUX = {'stream': None}     # auxiliary stream of the omega redraw (MusicLM forward)


def _side_fork(*tensors):
    """Raw handle of the side stream, made to wait for everything queued on the main stream so far — or None when the weight gradients stay on
    the main stream.  The tensors are marked as used by the side stream so the caching allocator does not recycle them early.
    (r05: the launches take the handle explicitly.  The `with torch.cuda.stream(side)` form this replaces — current_stream(), wait_stream(),
    the context manager's set_stream twice — cost 40-70 us of host time per weight gradient, 2-3 ms of the 8 ms in which the host queues a
    training step at the reference YAML's batch size 4, a step the host bounds: tools/b4_cpu_probe.py, tools/b4_host_profile.py.)"""
    if not _side_on(tensors):
        return None
    dev = torch.cuda.current_device()
    if _SIDE['stream'] is None or _SIDE['device'] != dev:
        _SIDE['stream'], _SIDE['device'] = torch.cuda.Stream(device=dev), dev
        _SIDE['raw'] = _SIDE['stream'].cuda_stream
    _SIDE['used'] = True
    _lib.check(_lib.lib.emo_stream_wait(_SIDE['raw'], _lib.stream()))
    side = _SIDE['stream']
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    return _SIDE['raw']


def join_side_stream():
    if _SIDE['used'] and _SIDE['stream'] is not None:
        _lib.check(_lib.lib.emo_stream_wait(_lib.stream(), _SIDE['raw']))
        _SIDE['used'] = False


def _timed_wgrad(a, b, out, a_rowsum=None, b_rowsum=None, stream=None):
    """dW += a^T b.  a_rowsum / b_rowsum: the bias gradient (column sums of dY) taken inside the same GEMM from the operand fragments."""
    ops.gemm(a, b, a_trans=True, b_trans=True, out=out, accumulate=True, a_rowsum=a_rowsum, b_rowsum=b_rowsum, stream=stream)


def _wgrad(ps, wname, bname, dy, xin, fused_rows=None, bias_done=False):
    """dW[N,K] += dy[M,N]^T xin[M,K] ; db[N] += colsum(dy)   (nn.Linear layout).  bias_done: the column sums were already
    accumulated by the LayerNorm-backward kernel that produced dy."""
    _timed_wgrad(dy, xin, ps.g(wname, fused_rows), a_rowsum=None if bias_done else ps.g(bname, fused_rows), stream=_side_fork(dy, xin))


def performer_layer_bwd(ps, pfx, dout, B, T, H, p, seed, off, save):
    s = save.t
    D = dout.shape[1]
    inv = 1.0 / (1.0 - p) if p > 0 else 1.0
    g2, dyd = ops.layernorm_bwd(dout, s['x2'], ps.f32(pfx + 'norm2.weight'), s['m2'], s['r2'], ps.g(pfx + 'norm2.weight'), ps.g(pfx + 'norm2.bias'),
                                want_drop=p > 0, p_drop=p, seed=seed, offset=off + 3, dcol=ps.g(pfx + 'linear2.bias'))
    if dyd is None:
        dyd = g2
    bf = ps.flat16 is not None and D == 512 and dout.shape[0] % 128 == 0 and dout.shape[0] >= ops.ASTAT_MIN_ROWS
    chb = int(_os.environ.get('EMO_FFN_CHUNK_B', 0))
    chunked = bool(bf and chb and s['fmask'] is not None and s.get('z') is None and dout.shape[0] > chb and dout.shape[0] % chb == 0)
    if chunked:                                                     # PROBE: the FFN backward per row chunk (df stays in the Infinity Cache for its two readers)
        dh1 = torch.empty_like(dyd)
        for r0 in range(0, dout.shape[0], chb):
            sl = slice(r0, r0 + chb)
            _wgrad(ps, pfx + 'linear2.weight', pfx + 'linear2.bias', dyd[sl], s['f'][sl], bias_done=True)
            dfc = ops.gemm(dyd[sl], ps.wT(pfx + 'linear2.weight'), mul_aux=s['fmask'][sl], mul_mode=ops.MUL_BITMASK, mul_scale=inv)
            _wgrad(ps, pfx + 'linear1.weight', pfx + 'linear1.bias', dfc, s['h1'][sl])
            ops.gemm(dfc, ps.wT(pfx + 'linear1.weight'), residual=g2[sl], out=dh1[sl])
    else:
        _wgrad(ps, pfx + 'linear2.weight', pfx + 'linear2.bias', dyd, s['f'], bias_done=True)
    bf = ps.flat16 is not None and D == 512 and dout.shape[0] % 128 == 0 and dout.shape[0] >= ops.ASTAT_MIN_ROWS     # the A-stationary K = 512 class
    if chunked:
        pass
    elif s.get('z') is not None:                       # activation = 'gelu': df = (dyd W2) * gelu'(z), then the hidden dropout's multipliers again
        df = ops.gemm(dyd, ps.w(pfx + 'linear2.weight'), b_trans=True, mul_aux=s['z'], mul_mode=ops.MUL_DGELU)
        if p > 0:
            df = ops.dropout_apply(df, p, seed, off + 2)
    elif bf and s['fmask'] is not None:
        df = ops.gemm(dyd, ps.wT(pfx + 'linear2.weight'), mul_aux=s['fmask'], mul_mode=ops.MUL_BITMASK, mul_scale=inv)
    elif bf:     # K = 512 dgrads against the transposed mirror (NT, A-stationary kernel)
        df = ops.gemm(dyd, ps.wT(pfx + 'linear2.weight'), mul_aux=s['f'], mul_mode=ops.MUL_NONZERO, mul_scale=inv)
    else:
        df = ops.gemm(dyd, ps.w(pfx + 'linear2.weight'), b_trans=True, mul_aux=s['f'], mul_mode=ops.MUL_NONZERO, mul_scale=inv)
    if not chunked:
        _wgrad(ps, pfx + 'linear1.weight', pfx + 'linear1.bias', df, s['h1'])
    # the long-reduction dgrads (K = 2048 / 1536) as NT products against transposed mirrors too: 256 x 256 tile kernel (emo_gemm_w128.hip).
    # From 32768 tokens, where M / 256 * N / 256 >= 256 tiles holds for N = 512; below that (the reference batch size 4) the long-reduction
    # dgrads stay NN products on the 128 x 128 kernel (the K = 512 ones above already run on the column-split A-stationary kernel from 4096 tokens).
    nt_long = bf and dout.shape[0] >= 32768 and dout.shape[0] % 256 == 0 and _os.environ.get('EMO_DGRAD_NT', '1') != '0'
    if not chunked:
        dh1 = ops.gemm(df, ps.wT(pfx + 'linear1.weight'), residual=g2) if nt_long else ops.gemm(df, ps.w(pfx + 'linear1.weight'), b_trans=True, residual=g2)
    g1, da = ops.layernorm_bwd(dh1, s['x1'], ps.f32(pfx + 'norm1.weight'), s['m1'], s['r1'], ps.g(pfx + 'norm1.weight'), ps.g(pfx + 'norm1.bias'),
                               want_drop=p > 0, p_drop=p, seed=seed, offset=off + 1, dcol=ps.g(pfx + 'attention.out_projection.bias'))
    if da is None:
        da = g1
    _wgrad(ps, pfx + 'attention.out_projection.weight', pfx + 'attention.out_projection.bias', da, s['attn'], bias_done=True)
    qkv = s['qkv']
    # d(attention output) leaves the out-projection dgrad already divided by the FAVOR+ normaliser (emo_hip.h: hdiv, emo_favor_attn_bwd_dn; r06):
    # one rounding from the fp32 accumulators, and the two backward sweeps lose their normaliser stream, reciprocals and rescaled operand copies
    dn = bool(bf and D // H == 64 and s.get('fws') is None and _os.environ.get('EMO_FAVOR_DN', '1') != '0'
              and ops.favor_bwd_dn_ok(qkv.dtype, B, T, H, D // H, 2 * s['omega'].shape[1]))
    if dn:
        dattn = ops.gemm(da, ps.wT(pfx + 'attention.out_projection.weight'), hdiv=(s['den'], T))
    else:
        dattn = ops.gemm(da, ps.wT(pfx + 'attention.out_projection.weight')) if bf else ops.gemm(da, ps.w(pfx + 'attention.out_projection.weight'), b_trans=True)
    dq, dk, dv = ops.favor_attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], s['omega'], s['attn'], dattn, s['den'], B, T, H, ws_saved=s.get('fws'), dn=dn)
    dqkv = dq._base if dq._base is not None else torch.cat([dq, dk, dv], 1)
    q = pfx + 'attention.query_projection.'
    _wgrad(ps, q + 'weight', q + 'bias', dqkv, s['x'], fused_rows=3 * D)
    if nt_long:
        return ops.gemm(dqkv, ps.wT(q + 'weight', 3 * D), residual=g1)
    return ops.gemm(dqkv, ps.w(q + 'weight', 3 * D), b_trans=True, residual=g1)


def gpt2_block_fwd(ps, pfx, x, B, T, H, p, seed, off, save):
    """HF GPT2Block (pre-LN, Conv1D weights [in,out], gelu_new, no final ln_f) — SURVEY App. B/C."""
    D = x.shape[1]
    # at >= 32768 tokens the Conv1D products run as k-contiguous NT products against transposed mirrors ([out, in], ParamStore.wT): the K = 512
    # ones on the A-stationary kernel (the [in, out] layout only has the 128 x 128 tiled kernel), the K = 2048 one on the 256 x 256 tile
    nt = ps.flat16 is not None and D == 512 and x.shape[0] % 256 == 0 and x.shape[0] >= 32768 and _os.environ.get('EMO_DGRAD_NT', '1') != '0'

    def lin(inp, wname, **kw):
        if nt:
            return ops.gemm(inp, ps.wT(wname), **kw)
        return ops.gemm(inp, ps.w(wname), b_trans=True, **kw)

    # LayerNorm inside the product that consumes it (emo_hip.h lna_*, r05): the A-stationary kernel holds complete rows of its A panel in registers
    def ln_lin(inp, ln, wname, **kw):
        g, b = ps.f32(pfx + ln + '.weight'), ps.f32(pfx + ln + '.bias')
        if nt and ops.gemm_lna_ok(inp.shape[0], ps.shapes[wname][1], D, inp.dtype):
            return ops.gemm(inp, ps.wT(wname), lna=(g, b, 1e-5), **kw)          # -> (product, LN(inp), mean, rstd)
        n, m, r = ops.layernorm_fwd(inp, g, b)
        return lin(n, wname, **kw), n, m, r

    qkv, n1, m1, r1 = ln_lin(x, 'ln_1', pfx + 'attn.c_attn.weight', bias=ps.f32(pfx + 'attn.c_attn.bias'))
    # (training: the forward leaves the attention-dropout keep bits for the backward's dK/dV pass — one hash evaluation per score instead of two)
    a, lse, keep = ops.softmax_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, T, H, p_drop=p, seed=seed, offset=off + 1, want_keep=True) \
        if (save is not None and p > 0) else ops.softmax_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, T, H, p_drop=p, seed=seed, offset=off + 1) + (None,)
    h = lin(a, pfx + 'attn.c_proj.weight', bias=ps.f32(pfx + 'attn.c_proj.bias'), p_drop=p, seed=seed, offset=off + 2, residual=x)
    z = torch.empty(x.shape[0], ps.shapes[pfx + 'mlp.c_fc.weight'][1], device=x.device, dtype=x.dtype) if save is not None else None
    f, n2, m2, r2 = ln_lin(h, 'ln_2', pfx + 'mlp.c_fc.weight', bias=ps.f32(pfx + 'mlp.c_fc.bias'), act=ops.ACT_GELU_NEW, aux_out=z)
    out = lin(f, pfx + 'mlp.c_proj.weight', bias=ps.f32(pfx + 'mlp.c_proj.bias'), p_drop=p, seed=seed, offset=off + 3, residual=h)
    if save is not None:
        save.t = dict(x=x, m1=m1, r1=r1, n1=n1, qkv=qkv, a=a, lse=lse, keep=keep, h=h, m2=m2, r2=r2, n2=n2, z=z, f=f)
    return out


def _wgrad_conv1d(ps, wname, bname, xin, dy):
    """dW[K,N] += xin[M,K]^T dy[M,N] ; db[N] += colsum(dy)   (HF Conv1D layout)."""
    _timed_wgrad(xin, dy, ps.g(wname), b_rowsum=ps.g(bname), stream=_side_fork(dy, xin))


def gpt2_block_bwd(ps, pfx, dout, B, T, H, p, seed, off, save, doutd=None, below_off=None):
    """doutd: dout already multiplied with this block's MLP-output dropout mask (by the LayerNorm backward of the block above).  below_off: dropout
    offset whose mask the returned gradient is ALSO wanted with — the block below's MLP-output dropout (off_below + 3) or the embedding dropout —
    produced by this block's last LayerNorm backward instead of a separate pass (r05: 24 dropout_apply launches per step).  Returns (dx, dx masked)."""
    s = save.t
    D = dout.shape[1]
    dyd = doutd if doutd is not None else (ops.dropout_apply(dout, p, seed, off + 3) if p > 0 else dout)
    _wgrad_conv1d(ps, pfx + 'mlp.c_proj.weight', pfx + 'mlp.c_proj.bias', s['f'], dyd)
    dz = ops.gemm(dyd, ps.w(pfx + 'mlp.c_proj.weight'), mul_aux=s['z'], mul_mode=ops.MUL_DGELU_NEW)
    _wgrad_conv1d(ps, pfx + 'mlp.c_fc.weight', pfx + 'mlp.c_fc.bias', s['n2'], dz)
    dn2 = ops.gemm(dz, ps.w(pfx + 'mlp.c_fc.weight'))
    fuse = _os.environ.get('EMO_GPT2_FUSE_MASK', '1') != '0'       # (0: the separate dropout_apply passes, same-box A/B)
    dh, dad = ops.layernorm_bwd(dn2, s['h'], ps.f32(pfx + 'ln_2.weight'), s['m2'], s['r2'], ps.g(pfx + 'ln_2.weight'), ps.g(pfx + 'ln_2.bias'), dres=dout,
                                want_drop=p > 0 and fuse, p_drop=p, seed=seed, offset=off + 2)       # dad = dh with the attention-output dropout's mask
    if dad is None:
        dad = ops.dropout_apply(dh, p, seed, off + 2) if p > 0 else dh
    if not fuse:
        below_off = None
    _wgrad_conv1d(ps, pfx + 'attn.c_proj.weight', pfx + 'attn.c_proj.bias', s['a'], dad)
    da = ops.gemm(dad, ps.w(pfx + 'attn.c_proj.weight'))
    qkv = s['qkv']
    dq, dk, dv = ops.softmax_attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], s['a'], da, s['lse'], B, T, H, p_drop=p, seed=seed, offset=off + 1, keep=s['keep'])
    dqkv = dq._base
    _wgrad_conv1d(ps, pfx + 'attn.c_attn.weight', pfx + 'attn.c_attn.bias', s['n1'], dqkv)
    dn1 = ops.gemm(dqkv, ps.w(pfx + 'attn.c_attn.weight'))
    dx, dxd = ops.layernorm_bwd(dn1, s['x'], ps.f32(pfx + 'ln_1.weight'), s['m1'], s['r1'], ps.g(pfx + 'ln_1.weight'), ps.g(pfx + 'ln_1.bias'), dres=dh,
                                want_drop=p > 0 and below_off is not None, p_drop=p, seed=seed, offset=below_off if below_off is not None else 0)
    return dx, dxd


# =================================================================================================== autograd nodes
class DecoderStackFn(torch.autograd.Function):
    """tokens -> final hidden states [B,T,D].  Parameter gradients are accumulated straight into the
    flat grad buffer (the ``.grad`` views); the node returns None for them."""

    @staticmethod
    def forward(ctx, model, tok, seg, anchor, need_bwd, chord=None):
        ps = model._store
        B, T = tok.shape
        D, H, L = model.d_model, model.n_head, model.n_layer
        p = model.dropout if model.training else 0.0
        seed, base = model._next_dropout_base()
        use_seg = seg is not None and model.use_segment_emb
        seg = seg if use_seg else None
        E = embedding_table(ps, 'token_emb.')
        S = embedding_table(ps, 'segemb.') if use_seg else None
        check_ids(tok, E.shape[0], 'token ids')
        check_ids(seg, 0 if S is None else S.shape[0], 'segment ids')
        if model.use_pe and model.d_embed != D:
            raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 2: positional '
                               'encoding of width d_embed cannot be added to d_model-wide embeddings (use_pe=True needs d_embed == d_model, '
                               'as in the reference, music_performer.py:59-60)' % (D, model.d_embed))
        pe = model.pe.pe if model.use_pe else model._zero_pe(T, D)
        cp = None
        # The omega redraw (randn + one Gram-Schmidt launch of 64 dependent column steps: ~180 us on 12 workgroups, pure latency) depends on nothing
        # in this forward: it runs on an auxiliary stream beside the embedding / first LayerNorm / first QKV product, the main stream waits for it
        # in front of the first layer (r04: hidden completely at the benchmark's batch, about a third of it at the YAML's batch size 4).
        omegas, omega_ev = None, None
        if model.kind == 'performer':
            if tok.is_cuda and not torch.cuda.is_current_stream_capturing() and _os.environ.get('EMO_OMEGA_STREAM', '1') != '0':
                main = torch.cuda.current_stream()
                if _AUX['stream'] is None or _AUX['stream'].device != main.device:
                    _AUX['stream'] = torch.cuda.Stream(device=main.device)
                aux = _AUX['stream']
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    omegas = model._omegas()
                    omega_ev = aux.record_event()
            else:
                omegas = model._omegas()
        if chord is None:
            x = ops.embed_fwd(tok, seg, E, S, pe, ps.compute_dtype, float(model.token_emb.emb_scale), p_drop=p, seed=seed, offset=base).view(B * T, D)
        else:
            # x_emb += chord_emb(chord_inp) (music_performer.py:56-57) sits between the gather and the dropout the gather kernel fuses, so
            # this (never-used-by-the-call-sites) option runs unfused: gather without dropout, one K = 12 (padded to 32) GEMM, dropout.
            cp = torch.zeros(B * T, 32, device=tok.device, dtype=ps.compute_dtype)
            cp[:, :12] = chord.reshape(B * T, 12)
            wp = torch.zeros(D, 32, device=tok.device, dtype=ps.compute_dtype)
            wp[:, :12] = ps.w('chord_emb.weight')
            e = ops.embed_fwd(tok, seg, E, S, pe, torch.float32, float(model.token_emb.emb_scale)).view(B * T, D)
            x = (e + ops.gemm(cp, wp, bias=ps.f32('chord_emb.bias'), out_dtype=torch.float32)).to(ps.compute_dtype)
            if p > 0.0:
                x = ops.dropout_apply(x, p, seed, base)
        saves = []
        if omega_ev is not None:
            torch.cuda.current_stream().wait_event(omega_ev)
            for t in omegas:
                t.record_stream(torch.cuda.current_stream())
        # norm2 of layer l inside layer l + 1's QKV product (performer_layer_fwd): possible when that product runs on the A-stationary kernel
        chain = model.kind == 'performer' and ops.gemm_lna_ok(B * T, 3 * D, D, ps.compute_dtype) and _os.environ.get('EMO_LN2_IN_GEMM', '1') != '0'
        ln_in = None
        for l in range(L):
            sv = LayerCtx() if need_bwd else None
            if model.kind == 'performer':
                pfx = model._layer_prefix(l)
                defer = chain and l + 1 < L
                x = performer_layer_fwd(ps, pfx, x, omegas[l], B, T, H, p, seed, base + 8 * (l + 1), sv, act=model.activation, ln_in=ln_in, defer_norm2=defer)
                ln_in = (ps.f32(pfx + 'norm2.weight'), ps.f32(pfx + 'norm2.bias'), sv) if defer else None
            else:
                x = gpt2_block_fwd(ps, model._layer_prefix(l), x, B, T, H, p, seed, base + 8 * (l + 1), sv)
            saves.append(sv)
        ctx.model, ctx.saves, ctx.tok, ctx.seg, ctx.chord = model, saves, tok, seg, cp
        ctx.cfg = (B, T, D, H, L, p, seed, base)
        return x.view(B, T, D)

    @staticmethod
    def backward(ctx, dout):
        model, ps = ctx.model, ctx.model._store
        B, T, D, H, L, p, seed, base = ctx.cfg
        ps.ensure_grads()
        dx = dout.reshape(B * T, D)
        if dx.dtype != ps.compute_dtype or not dx.is_contiguous():
            dx = dx.to(ps.compute_dtype).contiguous()
        dxd = None                                                # GPT-2: dx with the dropout mask its consumer wants (the block below / the embedding)
        n_seg = model.n_segment_types if ctx.seg is not None else 0
        eg = _os.environ.get('EMO_EMBED_GEMM', '')                 # '0' never / '1' always (tests, A/B); default: from 32768 tokens
        emb_gemm = (ctx.chord is None and model.d_embed == D and ps.compute_dtype == torch.bfloat16 and (eg == '1' or (eg != '0' and dx.shape[0] >= 32768))
                    and model.n_token + n_seg <= 512 and D % 256 == 0 and dx.shape[0] % 64 == 0)
        for l in reversed(range(L)):
            if model.kind == 'performer':
                dx = performer_layer_bwd(ps, model._layer_prefix(l), dx, B, T, H, p, seed, base + 8 * (l + 1), ctx.saves[l])
            else:
                dx, dxd = gpt2_block_bwd(ps, model._layer_prefix(l), dx, B, T, H, p, seed, base + 8 * (l + 1), ctx.saves[l], doutd=dxd,
                                         below_off=(base + 8 * l + 3) if l > 0 else (base if emb_gemm else None))
            ctx.saves[l] = None
            hook = getattr(model, '_bwd_hook', None)                 # data parallel: dp.GradExchange starts the late layers' all-reduce here
            if hook is not None:
                hook(l)
        proj = model.d_embed != D
        if proj:                                                 # gradients of the PROJECTED tables, then back through emb_proj
            dE = torch.zeros(model.n_token, D, device=dx.device)
            dS = torch.zeros(model.n_segment_types, D, device=dx.device) if ctx.seg is not None else None
        else:
            dE = ps.g('token_emb.emb_lookup.weight')
            dS = ps.g('segemb.emb_lookup.weight') if ctx.seg is not None else None
        if emb_gemm:
            # The scatter-add of the embedding gradient as ONE weight-gradient product against a 0 / 1 indicator matrix [M, 512] (column = token
            # id, n_token + segment id): bf16 x 1.0 is exact and the sums are fp32 in a fixed order, without float atomics.  r05, benchmark batch:
            # emo_embed_bwd 375 us (its LDS float atomics and 128-B row loads) -> dropout pass + indicator + the 256 x 256 wgrad kernel,
            # same-box -0.32 ms per step.  Rounding: the dropped / rescaled gradient passes through bf16 once before the sum.
            dxm = dxd if dxd is not None else (ops.dropout_apply(dx.contiguous(), p, seed, base) if p > 0.0 else dx.contiguous())
            ind = torch.zeros(dx.shape[0], 512, device=dx.device, dtype=torch.bfloat16)
            ind.scatter_(1, ctx.tok.reshape(-1, 1), 1.0)
            if n_seg:
                ind.scatter_(1, ctx.seg.reshape(-1, 1) + model.n_token, 1.0)
            dEp = ops.gemm(ind, dxm, a_trans=True, b_trans=True, out_dtype=torch.float32)
            sc = float(model.token_emb.emb_scale)
            dE.add_(dEp[:model.n_token], alpha=sc)
            if n_seg:
                dS.add_(dEp[model.n_token:model.n_token + n_seg], alpha=sc)
        elif ctx.chord is None:
            ops.embed_bwd(ctx.tok, ctx.seg, dx, dE, dS, float(model.token_emb.emb_scale), p_drop=p, seed=seed, offset=base)
        else:
            dxm = ops.dropout_apply(dx.contiguous(), p, seed, base) if p > 0.0 else dx.contiguous()
            ops.embed_bwd(ctx.tok, ctx.seg, dxm, dE, dS, float(model.token_emb.emb_scale))
            ps.g('chord_emb.weight').add_(ops.gemm(dxm, ctx.chord, a_trans=True, b_trans=True, out_dtype=torch.float32)[:, :12])
            ops.colsum(dxm, out=ps.g('chord_emb.bias'), accumulate=True)
        if proj:
            embedding_table_bwd(ps, 'token_emb.', dE)
            if dS is not None:
                embedding_table_bwd(ps, 'segemb.', dS)
        join_side_stream()          # all weight gradients are complete before anything downstream (all-reduce, optimizer) runs
        return None, None, None, None, None, None


LOGIT_PAD_FILL = -1e30         # logit of a pad column: exp(pad - lse) == 0 exactly in fp32, never the row maximum


def logit_pad(M, V, dtype):
    """Padded vocabulary size of the output projection (0 = none).  From 32768 rows in the bf16 mode the three products around the logits
    (forward, d hidden, dW: 45 GFLOP each at the benchmark batch) ran on the generic kernels that take N, K or M = 327 — 314 + 109 + 388 us,
    r05 — where the tiled kernels need 90 / 77 / ~100: the projection is computed for 512 columns (zero weight rows, bias LOGIT_PAD_FILL), the
    loss kernels see a 512-column problem whose pad columns contribute exactly nothing, and the caller gets the [.., :V] view."""
    mode = _os.environ.get('EMO_LOGIT_PAD', '')                    # '0' never / '1' always (tests, A/B)
    if dtype != torch.bfloat16 or V % 128 == 0 or mode == '0' or (mode != '1' and M < 32768):
        return 0
    return (V + 511) // 512 * 512 if V <= 512 else (V + 127) // 128 * 128


class LogitsFn(torch.autograd.Function):
    """dec_out_proj: fp32 logits = h W^T + b (untied nn.Linear, music_performer.py:27,65)."""

    @staticmethod
    def forward(ctx, model, h):
        ps = model._store
        shp = h.shape
        h2 = h.reshape(-1, shp[-1])
        V = ps.shapes['dec_out_proj.weight'][0]
        Vp = logit_pad(h2.shape[0], V, ps.compute_dtype)
        ctx.model, ctx.h2, ctx.shp = model, h2, shp
        if Vp:
            buf = ops.gemm(h2, ps.padded('dec_out_proj.weight', Vp), bias=ps.padded('dec_out_proj.bias', Vp, LOGIT_PAD_FILL, master=True), out_dtype=torch.float32)
            _tag_padded(buf)
            return buf[:, :V].view(*shp[:-1], V)                  # strided view: XentFn / accuracy find the padded buffer behind it
        logits = ops.gemm(h2, ps.w('dec_out_proj.weight'), bias=ps.f32('dec_out_proj.bias'), out_dtype=torch.float32)
        return logits.view(*shp[:-1], -1)

    @staticmethod
    def backward(ctx, dlogits):
        ps = ctx.model._store
        ps.ensure_grads()
        M, V = ctx.h2.shape[0], dlogits.shape[-1]
        dl = dlogits.reshape(M, V)
        Vp = (V + 7) // 8 * 8
        base = dl._base
        if (base is not None and base.dim() == 2 and base.shape[0] == M and base.shape[1] >= Vp and base.shape[1] % 8 == 0 and base.is_contiguous()
                and base.dtype == torch.float32 and dl.data_ptr() == base.data_ptr() and dl.stride() == (base.shape[1], 1)):
            padded, Vp = base, base.shape[1]                      # produced by XentFn: already padded (to 8, or to logit_pad), pad columns are zero
        else:
            padded = torch.zeros(M, Vp, device=dl.device, dtype=torch.float32)
            padded[:, :V].copy_(dl)
        if ps.compute_dtype == torch.bfloat16:
            p16 = torch.empty(M, Vp, device=dl.device, dtype=torch.bfloat16)
            ops.cast(padded, p16)
            padded = p16
        if Vp % 128 == 0 and Vp != V and Vp == logit_pad(M, V, ps.compute_dtype):
            # the padded problem: d hidden against the zero-padded weight, dW / db of all Vp rows into a scratch whose first V rows are added
            dWp = torch.zeros(Vp, ctx.h2.shape[1], device=dl.device, dtype=torch.float32)
            dbp = torch.zeros(Vp, device=dl.device, dtype=torch.float32)
            _timed_wgrad(padded, ctx.h2, dWp, a_rowsum=dbp, stream=_side_fork(padded, ctx.h2))
            dh = ops.gemm(padded, ps.padded('dec_out_proj.weight', Vp), b_trans=True)
            join_side_stream()
            ps.g('dec_out_proj.weight').add_(dWp[:V])
            ps.g('dec_out_proj.bias').add_(dbp[:V])
            return None, dh.view(ctx.shp)
        g = padded[:, :V]
        _timed_wgrad(g, ctx.h2, ps.g('dec_out_proj.weight'), a_rowsum=ps.g('dec_out_proj.bias'), stream=_side_fork(padded, ctx.h2))
        dh = ops.gemm(g, ps.w('dec_out_proj.weight'), b_trans=True)
        join_side_stream()
        return None, dh.view(ctx.shp)


class XentFn(torch.autograd.Function):
    """F.cross_entropy(ignore_index, reduction='mean') on fp32 logits (music_performer.py:72-81)."""

    @staticmethod
    def forward(ctx, logits, tgt, ignore_index):
        V = logits.shape[-1]
        l2 = padded_logits(logits)                                # LogitsFn's padded buffer [M, Vp] behind the view, if there is one
        if l2 is None:
            l2 = logits.reshape(-1, V)
            if not l2.is_contiguous():
                l2 = l2.contiguous()
        t = tgt.reshape(-1)
        check_ids(t, V, 'cross-entropy targets', also=ignore_index)
        lse, acc = ops.xent_fwd(l2, t, ignore_index)
        ctx.save_for_backward(l2, t, lse, acc)
        ctx.ignore, ctx.shape = ignore_index, logits.shape
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, gout):
        l2, t, lse, acc = ctx.saved_tensors
        gscale = (gout.float() / acc[1]).reshape(1)
        V = ctx.shape[-1]
        dl = ops.xent_bwd(l2, t, lse, gscale, ctx.ignore, torch.float32)
        g = dl.view(ctx.shape) if dl.shape[1] == V else _as_view(dl, V, ctx.shape)
        return g, None, None


_PADDED = {}                   # data_ptr -> weakref of a buffer LogitsFn.forward produced (pad columns = LOGIT_PAD_FILL)


def _tag_padded(buf):
    key = buf.data_ptr()

    def drop(ref, key=key):
        if _PADDED.get(key) is ref:
            del _PADDED[key]
    _PADDED[key] = weakref.ref(buf, drop)


def padded_logits(logits):
    """The contiguous [M, Vp] buffer of LogitsFn's padded projection when `logits` is its [.., :V] view (pad columns = LOGIT_PAD_FILL), else None.
    Only buffers that LogitsFn.forward itself produced and that are still alive qualify (r05 advisor finding: geometry alone also matched a
    user's [.., :V] slice of any wider fp32 tensor, whose extra columns are real logits)."""
    base = logits._base
    V = logits.shape[-1]
    if base is None or base.dim() != 2 or not base.is_contiguous() or base.dtype != torch.float32 or base.shape[1] <= V or base.shape[1] % 128:
        return None
    ref = _PADDED.get(base.data_ptr())
    tagged = ref() if ref is not None else None
    if tagged is None or tagged.shape != base.shape:
        return None
    M = logits.numel() // V
    if base.shape[0] != M or logits.data_ptr() != base.data_ptr() or logits.reshape(-1, V).stride() != (base.shape[1], 1):
        return None
    return base


def _as_view(dl, V, shape):
    """[M, Vp] padded buffer -> [..., V] strided VIEW (no copy) so LogitsFn.backward can reuse the buffer."""
    M, Vp = dl.shape
    lead = shape[:-1]
    strides, s = [], Vp
    for d in reversed(lead):
        strides.append(s)
        s *= d
    return dl.as_strided(tuple(lead) + (V,), tuple(reversed(strides)) + (1,))
