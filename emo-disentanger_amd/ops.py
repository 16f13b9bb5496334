"""Thin tensor-level wrappers over the C-ABI (no autograd here; see model/ for the autograd glue).

Every function takes GPU tensors, validates layout, and launches on torch's current stream.
2-D operands may be row-strided views (``stride(1) == 1``), e.g. the q/k/v slices of a fused
projection."""
import ctypes

import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_GELU_NEW, ACT_NONE, ACT_RELU, BF16, F32, MUL_BITMASK, MUL_DGELU, MUL_DGELU_NEW, MUL_NONE, MUL_NONZERO, Epilogue, check,
                   dtype_code, lib, ptr, stream)

__all__ = ['gemm', 'colsum', 'embed_fwd', 'embed_bwd', 'layernorm_fwd', 'layernorm_bwd', 'dropout_apply', 'favor_attn_fwd',
           'favor_attn_bwd', 'favor_decode_step', 'performer_decode_step', 'performer_decode_step_sampled', 'favor_draw_omega', 'softmax_attn_fwd', 'softmax_attn_bwd', 'softmax_attn_decode', 'relpos_attn_fwd', 'relpos_attn_bwd', 'relpos_attn_decode', 'xent_fwd',
           'xent_bwd', 'argmax', 'sample_nucleus', 'sample_nucleus_step', 'accuracy_counts', 'sumsq', 'clip_coef', 'adam_step', 'cast', 'add_bias2',
           'ACT_NONE', 'ACT_RELU', 'ACT_GELU_NEW', 'ACT_GELU', 'MUL_NONE', 'MUL_NONZERO', 'MUL_DGELU_NEW', 'MUL_DGELU', 'MUL_BITMASK', 'gemm_bitmask_ok', 'gemm_lna_ok', 'bitmask_rows', 'favor_bwd_dn_ok', 'ffn_fwd', 'ffn_fwd_ok']


def _c(t):
    """contiguous int64 tensor (or None).  NEVER call .contiguous() inside a ptr(...) argument: the temporary dies before
    the launch and the caching allocator hands its block to the next temporary."""
    if t is None:
        return None
    assert t.dtype == torch.int64
    return t if t.is_contiguous() else t.contiguous()


def _rows(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'need a 2-D tensor with unit column stride'
    return t.stride(0)


_stream = stream   # (gemm() has a keyword argument of that name)
_ws_cache = {}
_ws_need = {}      # emo_gemm_workspace_bytes by problem (a pure function of the shape: one library call per shape, not per launch)
GEMM_TIMING = None   # bench.py sets this to a list: every GEMM launch is then bracketed by HIP events on its launch stream (in situ)
KERNEL_TIMING = None  # bench.py sets this to a dict kind -> list of (event0, event1, algorithmic flops, algorithmic bytes) for the attention kernels


class _timed:
    """with _timed(kind, flops, bytes): brackets the launches inside with HIP events on torch's current stream when bench.py asked for it."""

    def __init__(self, kind, flops, nbytes):
        self.kind, self.flops, self.nbytes = kind, flops, nbytes

    def __enter__(self):
        if KERNEL_TIMING is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if KERNEL_TIMING is not None:
            self.e1.record()
            KERNEL_TIMING.setdefault(self.kind, []).append((self.e0, self.e1, self.flops, self.nbytes))
        return False


def _workspace(kind, device, need, st=None):
    """Caller-owned kernel scratch (the library never allocates): one buffer per (kind, device, stream), grown on demand.  Consumers on one
    stream run in launch order, so consecutive launches can share it.  The size REPORTED to the kernel is the size it asked for, not the
    (possibly larger) cached buffer: the GEMM's split-K count depends on the workspace it is offered, and a result must not depend on which
    shapes the process happened to run before (r03: a trace test passed alone and failed after other tests by 1e-4 of a parameter sum)."""
    if need == 0:
        return None, 0
    key = (kind, device, st if st is not None else stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < need:
        if buf is not None and st is not None and st != stream():
            # (r05 advisor finding) the scratch of a side-stream launch is allocated from the CURRENT stream's pool: the buffer that is being
            # replaced may still be in use by an earlier launch on `st`, and without this mark the caching allocator would hand its block to the
            # next main-stream allocation as soon as the reference drops
            buf.record_stream(torch.cuda.ExternalStream(st))
        buf = torch.empty(need, device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf, need


def gemm(A, B, *, a_trans=False, b_trans=False, out=None, out_dtype=None, accumulate=False, bias=None, act=ACT_NONE,
         aux_out=None, mul_aux=None, mul_mode=MUL_NONE, mul_scale=1.0, p_drop=0.0, seed=0, offset=0, residual=None,
         ln_c1=None, ln_eps=1e-5, ln_stats_out=None, rln=None, a_rowsum=None, b_rowsum=None, mask_out=None, stream=None, lna=None, lna_out=None, hdiv=None):
    """C[M,N] = epilogue(op(A) @ op(B));  a_trans: A stored [K,M];  b_trans=False: B stored [N,K] (nn.Linear),
    b_trans=True: B stored [K,N] (HF Conv1D).  ln_c1 / ln_stats_out / rln: LayerNorm folded around a decode-step GEMM (include/emo_hip.h).
    stream: raw hipStream_t to launch on (default: torch's current stream).
    lna = (gamma, beta, eps): A is the RAW input of a LayerNorm whose output is the operand — the A-stationary kernel normalises its row panel in
    registers (gemm_lna_ok shapes only); returns (C, LN(A), mean, rstd) instead of C."""
    st = stream if stream is not None else _stream()
    K, M = (A.shape if a_trans else A.shape[::-1])
    Kb, N = (B.shape if b_trans else B.shape[::-1])
    assert K == Kb, 'inner dims differ: %s vs %s' % (tuple(A.shape), tuple(B.shape))
    assert A.dtype == B.dtype
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=out_dtype or A.dtype)
    ws, ws_bytes = None, 0
    plain = bias is None and residual is None and aux_out is None and mul_aux is None and act == ACT_NONE and p_drop == 0.0
    if (out.dtype == torch.float32 and plain) or (out.dtype == torch.bfloat16 and not accumulate and M > 32):
        # fp32 plain outputs (wgrad): split-K partial sums; bf16 outputs on a small tile grid with a long reduction: split-K + reduce-with-epilogue
        wkey = (M, N, K, A.dtype, out.dtype)
        need = _ws_need.get(wkey)
        if need is None:
            need = _ws_need[wkey] = lib.emo_gemm_workspace_bytes(M, N, K, dtype_code(A.dtype), dtype_code(out.dtype))
        ws, ws_bytes = _workspace('gemm', A.device, need, st)
    rx, rstats, rgamma, rbeta = rln if rln is not None else (None, None, None, None)     # residual = LayerNorm(rx) from exported statistics
    assert rx is None or (rx.dtype == out.dtype and _rows(rx) == _rows(out))
    ln_out = ln_mean = ln_rstd = None
    if lna is not None:
        assert gemm_lna_ok(M, N, K, A.dtype) and not a_trans and not b_trans and lna[0].dtype == torch.float32 and lna[1].dtype == torch.float32
        assert _rows(A) % 8 == 0 and _rows(B) % 8 == 0 and _rows(out) % 8 == 0 and (A.data_ptr() | B.data_ptr() | out.data_ptr()) % 16 == 0, \
            'lna=: the A-stationary kernel needs 16-B aligned operands and row strides that are multiples of 8 elements'
        ln_eps = lna[2]
        if lna_out is not None:                                    # caller's (rows, mean, rstd) buffers: row chunks of one tensor
            ln_out, ln_mean, ln_rstd = lna_out
            assert ln_out.shape == (M, K) and ln_out.is_contiguous() and ln_out.dtype == A.dtype and ln_mean.numel() == M and ln_rstd.numel() == M
        else:
            ln_out = torch.empty(M, K, device=A.device, dtype=A.dtype)
            ln_mean = torch.empty(M, device=A.device, dtype=torch.float32)
            ln_rstd = torch.empty(M, device=A.device, dtype=torch.float32)
    epi = Epilogue(ptr(bias), act, ptr(aux_out), ptr(mul_aux), mul_mode, mul_scale, p_drop, seed, offset, ptr(residual),
                   ptr(ln_c1), ln_eps, ptr(ln_stats_out), ptr(rx), ptr(rstats), ptr(rgamma), ptr(rbeta), ptr(a_rowsum), ptr(b_rowsum), ptr(mask_out), ptr(ws), ws_bytes,
                   ptr(lna[0]) if lna is not None else None, ptr(lna[1]) if lna is not None else None, ptr(ln_out), ptr(ln_mean), ptr(ln_rstd),
                   ptr(hdiv[0]) if hdiv is not None else None, int(hdiv[1]) if hdiv is not None else 0)
    if hdiv is not None:      # (den, T): C[m][n] /= den[m / T, n / 64, m % T]  (emo_hip.h: hdiv — the out-projection dgrad leaves dN = dout / den)
        assert hdiv[0].dtype == torch.float32 and hdiv[0].is_contiguous() and hdiv[0].numel() == M * (N // 64) and plain and mask_out is None and lna is None
    if not plain or mask_out is not None:
        for t in (aux_out, residual) + (() if mul_mode == MUL_BITMASK else (mul_aux,)):
            assert t is None or (t.dtype == out.dtype and _rows(t) == _rows(out))
        for t in (mask_out,) + ((mul_aux,) if mul_mode == MUL_BITMASK else ()):
            assert t is None or (t.dtype == torch.uint8 and t.is_contiguous() and t.shape == (M, N // 8) and gemm_bitmask_ok(M, N, K, A.dtype, out.dtype))
        assert bias is None or bias.dtype == torch.float32
    if GEMM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tstream = torch.cuda.ExternalStream(st) if stream is not None else torch.cuda.current_stream()     # the events go where the launch goes
        e0.record(tstream)
    check(lib.emo_gemm(ptr(A), int(a_trans), _rows(A), ptr(B), int(b_trans), _rows(B), ptr(out), _rows(out), M, N, K,
                       dtype_code(A.dtype), dtype_code(out.dtype), int(accumulate), ctypes.byref(epi), st))
    if GEMM_TIMING is not None:
        e1.record(tstream)
        kind = ('T' if a_trans else 'N') + ('N' if b_trans else 'T')
        fam = lib.emo_gemm_last_kernel() & 15          # the kernel family that RAN (emo_hip.h), not the one the shape suggests
        if fam == 4:
            kind += '/K>1024'
        elif fam == 2:       # one class per kernel INSTANCE (template argument = epilogue flags), as the rocprofv3 summary lists them
            tag = [t for t, on in (('relu', act == ACT_RELU), ('gelu', act == ACT_GELU_NEW), ('drop', p_drop > 0), ('res', residual is not None),
                                   ('bits', mul_mode == MUL_BITMASK), ('mul', mul_aux is not None and mul_mode != MUL_BITMASK), ('mask', mask_out is not None),
                                   ('hdiv', hdiv is not None)) if on]
            kind += '/K=512/' + ('+'.join(tag) if tag else 'plain')
        elif fam == 3:       # 256 x 256 tile wgrad (with / without the bias gradient)
            kind += '/rs' if (a_rowsum is not None or b_rowsum is not None) else '/plain'
        elif kind == 'TN':
            kind += '/128'
        elif fam == 1:
            kind += '/skinny'
        GEMM_TIMING.append((kind, e0, e1, 2.0 * M * N * K,
                            (M * K + N * K) * A.element_size() + M * N * out.element_size(), (M, N, K)))
    if lna is not None:
        return out, ln_out, ln_mean, ln_rstd
    return out


ASTAT_MIN_ROWS = int(os.environ.get('EMO_ASTAT_MIN_ROWS', 0)) or 128 * 32       # emo_gemm_astat.hip (AS_MIN_BLOCKS): 128-row panels; below one panel per CU the kernel splits the columns over the grid


def gemm_bitmask_ok(M, N, K, in_dtype, out_dtype):
    """Shape class in which emo_gemm writes / reads the 1-bit epilogue mask (mask_out / MUL_BITMASK): the A-stationary K = 512 kernel."""
    return in_dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and K == 512 and M % 128 == 0 and M >= ASTAT_MIN_ROWS and N % 64 == 0 and 64 <= N <= 2048


def gemm_lna_ok(M, N, K, in_dtype):
    """Shape class in which emo_gemm can take the LayerNorm of its A operand (lna=): the A-stationary K = 512 kernel (emo_hip.h: lna_*)."""
    # (mirrors every refusal of emo_gemm_astat_try that does not depend on the tensors: shape class, EMO_GEMM_NO_ASTAT, the safe-transpose
    # diagnostics mode; strides / alignment are asserted by gemm() itself, which owns the tensors — lna_* has no other kernel to fall back to)
    return (in_dtype == torch.bfloat16 and K == 512 and M % 128 == 0 and M >= ASTAT_MIN_ROWS and N % 64 == 0 and 64 <= N <= 2048
            and os.environ.get('EMO_GEMM_NO_ASTAT') is None and os.environ.get('EMO_LN_IN_GEMM', '1') != '0'
            and os.environ.get('EMO_GEMM_SAFE_TR', '0') in ('', '0'))


def bitmask_rows(mask, M, N):
    """Row-major view [M, N/8] (byte (m, n/8), bit j = column 8 (n/8) + j) of a mask in emo_gemm's tiled layout (emo_hip.h: mask_out) — tests / diagnostics."""
    t = mask.reshape(M // 32, N // 64, 4, 16, 2, 2)           # [row panel, column tile, g = column group, r = row % 16, i = row / 16, h = column / 32]
    return t.permute(0, 4, 3, 1, 5, 2).reshape(M, N // 8)


def ffn_fwd_ok(M, d_model, d_ff, dtype):
    """True when ffn_fwd serves the problem (emo_hip.h: emo_ffn_fwd — the feed-forward block in one launch)."""
    return dtype == torch.bfloat16 and bool(lib.emo_ffn_fwd_supported(dtype_code(dtype), M, d_model, d_ff))


def ffn_fwd(x1, gamma, beta, W1, b1, W2, b2, p_drop=0.0, seed=0, offset_f=0, offset_y=0, eps=1e-5):
    """(f, h1, mean, rstd, mask, x2) of the fused feed-forward block: h1 = LN(x1), f = drop(relu(h1 W1^T + b1)), x2 = h1 + drop(f W2^T + b2)."""
    M, D = x1.shape
    Hd = W1.shape[0]
    assert x1.is_contiguous() and W1.is_contiguous() and W2.is_contiguous() and W1.shape == (Hd, D) and W2.shape == (D, Hd)
    assert b1.dtype == torch.float32 and b2.dtype == torch.float32 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    dev, dt = x1.device, x1.dtype
    h1, x2 = torch.empty_like(x1), torch.empty_like(x1)
    f = torch.empty(M, Hd, device=dev, dtype=dt)
    mean, rstd = torch.empty(M, device=dev, dtype=torch.float32), torch.empty(M, device=dev, dtype=torch.float32)
    mask = torch.empty(M, Hd // 8, device=dev, dtype=torch.uint8)
    flops, nbytes = 2.0 * 2.0 * M * D * Hd, (3.0 * M * D + M * Hd) * x1.element_size() + M * Hd / 8
    with _timed('ffn_fused_fwd', flops, nbytes):
        check(lib.emo_ffn_fwd(ptr(x1), ptr(gamma), ptr(beta), eps, ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(h1), ptr(mean), ptr(rstd), ptr(f), ptr(mask),
                              ptr(x2), M, D, Hd, dtype_code(dt), p_drop, seed, offset_f, offset_y, stream()))
    return f, h1, mean, rstd, mask, x2


def colsum(X, out=None, accumulate=False):
    M, N = X.shape
    if out is None:
        out = torch.empty(N, device=X.device, dtype=torch.float32)
    check(lib.emo_colsum(ptr(X), dtype_code(X.dtype), M, N, _rows(X), ptr(out), int(accumulate), stream()))
    return out


def embed_fwd(tok, seg, E, S, pe, dtype, scale, pos0=0, p_drop=0.0, seed=0, offset=0, pos_ids=None):
    B, T = tok.shape
    D = E.shape[1]
    out = torch.empty(B, T, D, device=tok.device, dtype=dtype)
    assert pe.is_contiguous() and pe.shape[-1] == D and pe.numel() >= (pos0 + T) * D
    tok, seg, pos_ids = _c(tok), _c(seg), _c(pos_ids)       # keep the contiguous copies alive until the launch is queued
    check(lib.emo_embed_fwd(ptr(tok), ptr(seg), ptr(E), ptr(S), ptr(pe), ptr(out),
                            dtype_code(dtype), B, T, D, E.shape[0], 0 if S is None else S.shape[0], pos0, ptr(pos_ids), scale, p_drop, seed,
                            offset, stream()))
    return out


def embed_bwd(tok, seg, dout, dE, dS, scale, p_drop=0.0, seed=0, offset=0):
    B, T = tok.shape
    D = dE.shape[1]
    assert dout.is_contiguous()
    tok, seg = _c(tok), _c(seg)
    check(lib.emo_embed_bwd(ptr(tok), ptr(seg), ptr(dout), dtype_code(dout.dtype),
                            ptr(dE), ptr(dS), B, T, D, dE.shape[0], 0 if dS is None else dS.shape[0], scale, p_drop, seed, offset,
                            stream()))


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    M, D = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    check(lib.emo_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), dtype_code(x.dtype), M, D, eps, stream()))
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, want_drop=False, p_drop=0.0, seed=0, offset=0, dcol=None):
    M, D = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and (dres is None or dres.is_contiguous())
    dx = torch.empty_like(x)
    dxd = torch.empty_like(x) if want_drop else None
    ws, ws_bytes = _workspace('ln_bwd', x.device, lib.emo_layernorm_bwd_workspace_bytes(dtype_code(x.dtype), M, D))
    check(lib.emo_layernorm_bwd_ws(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dxd), ptr(dgamma), ptr(dbeta),
                                   ptr(dcol), dtype_code(x.dtype), M, D, p_drop, seed, offset, ptr(ws), ws_bytes, stream()))
    return dx, dxd


def dropout_apply(x, p_drop, seed, offset):
    assert x.is_contiguous()
    out = torch.empty_like(x)
    check(lib.emo_dropout_apply(ptr(x), ptr(out), dtype_code(x.dtype), x.numel(), p_drop, seed, offset, stream()))
    return out


def _favor_workspace(device, B, T, H, dh, n_feat):
    """Scratch for the segment-parallel scan (see include/emo_hip.h)."""
    return _workspace('favor', device, lib.emo_favor_attn_workspace_bytes(B, T, H, dh, n_feat))


def favor_attn_fwd(q, k, v, omega, B, T, H, eps=1e-6, want_state=False, keep_ws=False):
    """q,k,v: [B*T, H*dh] (row-strided views allowed). Returns out [B*T, H*dh], den [B,H,T] (, S, z).
    keep_ws: the call gets a PRIVATE workspace, returned as a third value (None when the scan is not segmented) for favor_attn_bwd(ws_saved=):
    the backward then skips recomputing the K-state increments (include/emo_hip.h, emo_favor_attn_bwd_kstate)."""
    M, HD = q.shape
    dh = HD // H
    n_feat = 2 * omega.shape[1]
    assert M == B * T and omega.shape[0] == dh and omega.dtype == torch.float32 and omega.is_contiguous()
    assert _rows(q) == _rows(k) == _rows(v)
    out = torch.empty(M, HD, device=q.device, dtype=q.dtype)
    den = torch.empty(B, H, T, device=q.device, dtype=torch.float32)
    S = torch.empty(B, H, n_feat, dh, device=q.device, dtype=torch.float32) if want_state else None
    z = torch.empty(B, H, n_feat, device=q.device, dtype=torch.float32) if want_state else None
    if keep_ws:
        ws_bytes = lib.emo_favor_attn_workspace_bytes(B, T, H, dh, n_feat)
        ws = torch.empty(ws_bytes, device=q.device, dtype=torch.uint8) if ws_bytes else None
    else:
        ws, ws_bytes = _favor_workspace(q.device, B, T, H, dh, n_feat)
    # algorithmic HBM bytes: read q, k, v + write out = 4 * H*dh * e per token
    with _timed('favor_fwd', 0.0, 4.0 * M * HD * q.element_size()):
        check(lib.emo_favor_attn_fwd(ptr(q), ptr(k), ptr(v), _rows(q), ptr(omega), ptr(out), HD, ptr(den), ptr(S), ptr(z), dtype_code(q.dtype),
                                     B, T, H, dh, n_feat, eps, ptr(ws), ws_bytes, stream()))
    if keep_ws:
        assert not want_state
        return out, den, ws
    return (out, den, S, z) if want_state else (out, den)


def favor_bwd_dn_ok(dtype, B, T, H, dh, n_feat):
    """True when favor_attn_bwd(dn=True) serves the problem (emo_hip.h: emo_favor_attn_bwd_dn — the single-segment slice kernels)."""
    return dtype == torch.bfloat16 and bool(lib.emo_favor_attn_bwd_dn_supported(dtype_code(dtype), B, T, H, dh, n_feat))


def favor_attn_bwd(q, k, v, omega, out, dout, den, B, T, H, dqkv=None, eps=1e-6, ws_saved=None, dn=False):
    """Returns (dq, dk, dv) views of one fused [B*T, 3*H*dh] buffer (ready for the fused-QKV dgrad/wgrad).
    dn=True: `dout` is already dN = dout / den (gemm(hdiv=(den, T)) produced it); den is not read."""
    M, HD = q.shape
    dh = HD // H
    n_feat = 2 * omega.shape[1]
    assert out.is_contiguous() and dout.is_contiguous()
    if dqkv is None:
        dqkv = torch.empty(M, 3 * HD, device=q.device, dtype=q.dtype)
    dq, dk, dv = dqkv[:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:]
    if dn:
        with _timed('favor_bwd', 0.0, 7.0 * M * HD * q.element_size()):
            check(lib.emo_favor_attn_bwd_dn(ptr(q), ptr(k), ptr(v), _rows(q), ptr(omega), ptr(out), ptr(dout), HD, ptr(dq), ptr(dk), ptr(dv), 3 * HD,
                                            dtype_code(q.dtype), B, T, H, dh, n_feat, eps, stream()))
        return dq, dk, dv
    if ws_saved is not None:                                      # the forward's private workspace: its K-state increments are still there
        ws, ws_bytes, kvalid = ws_saved, ws_saved.numel(), 1
        assert ws_bytes == lib.emo_favor_attn_workspace_bytes(B, T, H, dh, n_feat)
    else:
        (ws, ws_bytes), kvalid = _favor_workspace(q.device, B, T, H, dh, n_feat), 0
    # read q, k, v, dout (+ out) + write dq, dk, dv = 7 * H*dh * e per token
    with _timed('favor_bwd', 0.0, 7.0 * M * HD * q.element_size()):
        check(lib.emo_favor_attn_bwd_kstate(ptr(q), ptr(k), ptr(v), _rows(q), ptr(omega), ptr(out), ptr(dout), HD, ptr(den), ptr(dq), ptr(dk), ptr(dv),
                                            3 * HD, dtype_code(q.dtype), B, T, H, dh, n_feat, eps, ptr(ws), ws_bytes, kvalid, stream()))
    return dq, dk, dv


def favor_decode_step(q, k, v, omega, state_S, state_z, H, eps=1e-6):
    n, HD = q.shape
    dh = HD // H
    out = torch.empty(n, HD, device=q.device, dtype=q.dtype)
    check(lib.emo_favor_decode_step(ptr(q), ptr(k), ptr(v), _rows(q), ptr(omega), ptr(state_S), ptr(state_z), ptr(out), HD,
                                    dtype_code(q.dtype), n, H, dh, 2 * omega.shape[1], eps, stream()))
    return out


def performer_decode_step(layer_table, n_layers, tok, seg, E, Sg, pe, emb_scale, pos0, pos_ids, wout_packed, bout, n_token, logits, n_streams,
                          d_model, n_head, n_feat, d_ff, sync_ws, eps=1e-6, ln_eps=1e-5, diag=None):
    """One token step of every stream in ONE persistent launch (emo_hip.h: emo_performer_decode_step)."""
    tok, seg, pos_ids = _c(tok), _c(seg), _c(pos_ids)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape == (n_streams, n_token)
    check(lib.emo_performer_decode_step(ptr(layer_table), n_layers, ptr(tok), ptr(seg), ptr(E), ptr(Sg), ptr(pe), emb_scale, pos0, ptr(pos_ids),
                                        ptr(wout_packed), ptr(bout), n_token, ptr(logits), n_streams, d_model, n_head, n_feat, d_ff,
                                        ptr(sync_ws), sync_ws.numel() * sync_ws.element_size(), eps, ln_eps, ptr(diag), stream()))
    return logits


def performer_decode_step_sampled(layer_table, n_layers, seg, E, Sg, pe, emb_scale, pos0, wout_packed, bout, n_token, logits, n_streams, n_real,
                                  d_model, n_head, n_feat, d_ff, sync_ws, temperature, top_p, u_steps, step, seq, col0, tok_out, eps=1e-6, ln_eps=1e-5):
    """emo_performer_decode_step with the next token drawn inside the launch (emo_hip.h)."""
    seg = _c(seg)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape == (n_streams, n_token)
    assert u_steps.dtype == torch.float32 and u_steps.is_contiguous() and u_steps.shape[1] == n_real and step.dtype == torch.int64 and tok_out.dtype == torch.int64
    assert seq is None or (seq.dtype == torch.int64 and seq.stride(1) == 1)
    check(lib.emo_performer_decode_step_sampled(ptr(layer_table), n_layers, ptr(seg), ptr(E), ptr(Sg), ptr(pe), emb_scale, pos0, ptr(wout_packed), ptr(bout),
                                                n_token, ptr(logits), n_streams, n_real, d_model, n_head, n_feat, d_ff, ptr(sync_ws),
                                                sync_ws.numel() * sync_ws.element_size(), eps, ln_eps, temperature, top_p, ptr(u_steps), ptr(step), ptr(seq),
                                                0 if seq is None else seq.stride(0), col0, ptr(tok_out), stream()))
    return logits


def gpt2_decode_step(layer_table, n_layers, tok, seg, E, Sg, pe, emb_scale, pos0, pos_ids, ln0, kv_tmax, wout_packed, bout, n_token, logits, n_streams,
                     d_model, n_head, d_ff, sync_ws, ln_eps=1e-5, diag=None):
    """One GPT-2 token step of every stream in ONE persistent launch (emo_hip.h: emo_gpt2_decode_step)."""
    tok, seg, pos_ids = _c(tok), _c(seg), _c(pos_ids)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape == (n_streams, n_token)
    assert ln0.dtype == torch.float32 and ln0.is_contiguous() and ln0.numel() == 2 * d_model
    check(lib.emo_gpt2_decode_step(ptr(layer_table), n_layers, ptr(tok), ptr(seg), ptr(E), ptr(Sg), ptr(pe), emb_scale, pos0, ptr(pos_ids), ptr(ln0), kv_tmax,
                                   ptr(wout_packed), ptr(bout), n_token, ptr(logits), n_streams, d_model, n_head, d_ff,
                                   ptr(sync_ws), sync_ws.numel() * sync_ws.element_size(), ln_eps, ptr(diag), stream()))
    return logits


def gpt2_decode_step_sampled(layer_table, n_layers, seg, E, Sg, pe, emb_scale, pos0, ln0, kv_tmax, wout_packed, bout, n_token, logits, n_streams, n_real,
                             d_model, n_head, d_ff, sync_ws, temperature, top_p, u_steps, step, seq, col0, tok_out, ln_eps=1e-5):
    """emo_gpt2_decode_step with the next token drawn inside the launch (emo_hip.h)."""
    seg = _c(seg)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape == (n_streams, n_token)
    assert ln0.dtype == torch.float32 and ln0.is_contiguous() and ln0.numel() == 2 * d_model
    assert u_steps.dtype == torch.float32 and u_steps.is_contiguous() and u_steps.shape[1] == n_real and step.dtype == torch.int64 and tok_out.dtype == torch.int64
    assert seq is None or (seq.dtype == torch.int64 and seq.stride(1) == 1)
    check(lib.emo_gpt2_decode_step_sampled(ptr(layer_table), n_layers, ptr(seg), ptr(E), ptr(Sg), ptr(pe), emb_scale, pos0, ptr(ln0), kv_tmax, ptr(wout_packed),
                                           ptr(bout), n_token, ptr(logits), n_streams, n_real, d_model, n_head, d_ff, ptr(sync_ws),
                                           sync_ws.numel() * sync_ws.element_size(), ln_eps, temperature, top_p, ptr(u_steps), ptr(step), ptr(seq),
                                           0 if seq is None else seq.stride(0), col0, ptr(tok_out), stream()))
    return logits


def favor_draw_omega(gauss, omega):
    """gauss [L, nb, dh, dh] ~ N(0,1) -> omega [L, dh, n_feat/2] (orthogonal blocks scaled by row norms)."""
    L, nb, dh, _ = gauss.shape
    assert gauss.is_contiguous() and omega.is_contiguous() and omega.shape[:2] == (L, dh)
    check(lib.emo_favor_draw_omega(ptr(gauss), ptr(omega), L, dh, 2 * omega.shape[2], stream()))
    return omega


def softmax_attn_fwd(q, k, v, B, T, H, p_drop=0.0, seed=0, offset=0, want_keep=False):
    """want_keep: also return the dropout keep words for softmax_attn_bwd (None when the call has none: include/emo_hip.h)."""
    M, HD = q.shape
    dh = HD // H
    assert M == B * T and _rows(q) == _rows(k) == _rows(v)
    out = torch.empty(M, HD, device=q.device, dtype=q.dtype)
    lse = torch.empty(B, H, T, device=q.device, dtype=torch.float32)
    keep, kbytes = None, 0
    if want_keep and all(t.data_ptr() % 16 == 0 for t in (q, k, v)):
        kbytes = lib.emo_softmax_attn_keep_bytes(dtype_code(q.dtype), B, T, H, dh, p_drop)
        if kbytes:
            keep = torch.empty(kbytes // 4, device=q.device, dtype=torch.int32)
    # 2 matmuls (Q K^T, P V) of 2*T*T*dh FLOP per (b, h), half of the tiles skipped by the causal mask
    with _timed('sattn_fwd', 0.5 * 2 * 2.0 * B * H * T * T * dh, 4.0 * M * HD * q.element_size()):
        check(lib.emo_softmax_attn_fwd_keep(ptr(q), ptr(k), ptr(v), _rows(q), ptr(out), HD, ptr(lse), dtype_code(q.dtype), B, T, H, dh, p_drop,
                                            seed, offset, ptr(keep), kbytes, stream()))
    return (out, lse, keep) if want_keep else (out, lse)


def softmax_attn_bwd(q, k, v, out, dout, lse, B, T, H, p_drop=0.0, seed=0, offset=0, dqkv=None, keep=None):
    M, HD = q.shape
    dh = HD // H
    assert out.is_contiguous() and dout.is_contiguous()
    if dqkv is None:
        dqkv = torch.empty(M, 3 * HD, device=q.device, dtype=q.dtype)
    dq, dk, dv = dqkv[:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:]
    delta = torch.empty(B, H, T, device=q.device, dtype=torch.float32)          # dO.O per query row, handed from the dQ to the dK/dV pass
    # dQ pass: S, dP, dQ; dK/dV pass: S, dP, dV, dK = 7 matmuls, causal half
    with _timed('sattn_bwd', 0.5 * 7 * 2.0 * B * H * T * T * dh, 8.0 * M * HD * q.element_size()):
        check(lib.emo_softmax_attn_bwd_keep(ptr(q), ptr(k), ptr(v), _rows(q), ptr(out), ptr(dout), HD, ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), 3 * HD,
                                            dtype_code(q.dtype), B, T, H, dh, p_drop, seed, offset, ptr(keep), 0 if keep is None else keep.numel() * 4, stream()))
    return dq, dk, dv


def softmax_attn_decode(q, kcache, vcache, lens, H, lens_off=0, k_new=None, v_new=None):
    """One query row per stream against the KV caches; k_new / v_new: the new token's rows, appended in-kernel at position lens + lens_off - 1.
    Caches: [n, T_max, H * dh], or head-major [n, H, T_max, dh] (4-D tensors)."""
    n, HD = q.shape
    head_major = kcache.dim() == 4
    T_max = kcache.shape[2] if head_major else kcache.shape[1]
    assert kcache.is_contiguous() and vcache.is_contiguous() and lens.dtype == torch.int64 and kcache.shape == vcache.shape
    assert not head_major or (kcache.shape[1] == H and kcache.shape[3] == HD // H)
    assert (k_new is None) == (v_new is None) and (k_new is None or _rows(k_new) == _rows(v_new))
    out = torch.empty(n, HD, device=q.device, dtype=q.dtype)
    check(lib.emo_softmax_attn_decode_layout(ptr(q), _rows(q), ptr(kcache), ptr(vcache), T_max, ptr(lens), lens_off, ptr(k_new), ptr(v_new),
                                             0 if k_new is None else _rows(k_new), ptr(out), HD, dtype_code(q.dtype), n, H, HD // H, int(head_major), stream()))
    return out


def relpos_attn_fwd(q, k, v, r_dist, r_w_bias, r_r_bias, B, T, H, p_drop=0.0, seed=0, offset=0):
    """Transformer-XL relative-position causal attention (stage 1).  q,k,v [B*T, H*dh] views; r_dist [>=T, H*dh] indexed by distance."""
    M, HD = q.shape
    dh = HD // H
    assert M == B * T and _rows(q) == _rows(k) == _rows(v) and r_dist.shape[0] >= T and r_dist.dtype == q.dtype
    assert r_w_bias.dtype == torch.float32 and r_r_bias.dtype == torch.float32 and r_w_bias.is_contiguous() and r_r_bias.is_contiguous()
    out = torch.empty(M, HD, device=q.device, dtype=q.dtype)
    lse = torch.empty(B, H, T, device=q.device, dtype=torch.float32)
    zden = torch.empty(B, H, T, device=q.device, dtype=torch.float32)
    check(lib.emo_relpos_attn_fwd(ptr(q), ptr(k), ptr(v), _rows(q), ptr(r_dist), _rows(r_dist), r_dist.shape[0], ptr(r_w_bias), ptr(r_r_bias), ptr(out), HD,
                                  ptr(lse), ptr(zden), dtype_code(q.dtype), B, T, H, dh, p_drop, seed, offset, stream()))
    return out, lse, zden


def relpos_attn_bwd(qkv, r_dist, r_w_bias, r_r_bias, out, dout, lse, zden, B, T, H, p_drop=0.0, seed=0, offset=0, acc_dq=None, acc_rr=None, dR_out=None, dq_rel_out=None):
    """Backward of relpos_attn_fwd.  qkv [B*T, 3*H*dh] (the fused projection).  Returns dqkv [B*T, 3*H*dh], dR [T, H*dh] fp32 (gradient of
    r_dist rows 0..T-1), d r_w_bias [H, dh], d r_r_bias [H, dh] fp32.  Three kernels, each recomputing the probabilities of its tiles:
    query-tile pass (dq = content + relative part), key-tile pass (dk, dv), distance-window pass (dR) — include/emo_hip.h."""
    M, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    dt, dev = qkv.dtype, qkv.device
    assert M == B * T and out.is_contiguous() and dout.is_contiguous() and qkv.is_contiguous()
    dqkv = torch.empty(M, D3, device=dev, dtype=dt)
    # dq_rel_out: caller's [M, D] buffer for the relative part of dq (the training stack keeps all layers' in one tensor and takes ONE column sum)
    dq_rel = dq_rel_out if dq_rel_out is not None else torch.empty(M, D, device=dev, dtype=dt)
    assert dq_rel.shape == (M, D) and dq_rel.dtype == dt and dq_rel.is_contiguous()
    delta = torch.empty(B, H, T, device=dev, dtype=torch.float32)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    check(lib.emo_relpos_attn_bwd(ptr(q), ptr(k), ptr(v), D3, ptr(r_dist), _rows(r_dist), r_dist.shape[0], ptr(r_w_bias), ptr(r_r_bias), ptr(out),
                                  ptr(dout), D, ptr(lse), ptr(zden), ptr(dqkv), D3, ptr(dq_rel), D, ptr(delta), dtype_code(dt), B, T, H, dh, p_drop,
                                  seed, offset, stream()))
    if acc_rr is not None:                                      # training stack: column sums accumulated over the layers, no per-layer ATen ops —
        if acc_dq is not None:                                  # acc_dq += colsum(dq) (None: the caller takes it from the qkv_net weight-gradient
            colsum(dqkv[:, :D], out=acc_dq, accumulate=True)    # GEMM's a_rowsum, one launch fewer), acc_rr += colsum(dq_rel); the caller forms
        if dq_rel_out is None:
            colsum(dq_rel, out=acc_rr, accumulate=True)         # d r_r_bias = acc_rr, d r_w_bias = acc_dq - acc_rr once per backward
        d_rr = d_rw = None
    else:
        d_rr = colsum(dq_rel)                                   # sum over (b, i) of the relative part of dq = d r_r_bias
        d_rw = (colsum(dqkv[:, :D]) - d_rr).view(H, dh)         # ... of the content part = d r_w_bias
        d_rr = d_rr.view(H, dh)
    qu, qv = add_bias2(q, r_w_bias, r_r_bias)                   # q + r_w_bias, q + r_r_bias
    check(lib.emo_relpos_attn_bwd_kv(ptr(qu), ptr(qv), D, ptr(k), ptr(v), D3, ptr(r_dist), _rows(r_dist), r_dist.shape[0], ptr(dout), D, ptr(lse),
                                     ptr(zden), ptr(delta), ptr(dqkv[:, D:2 * D]), ptr(dqkv[:, 2 * D:]), D3, dtype_code(dt), B, T, H, dh, p_drop, seed,
                                     offset, stream()))
    if dR_out is not None:                                      # caller's fp32 [>= T, D] view (row-strided: one column block of a buffer shared by all layers)
        assert dR_out.dtype == torch.float32 and dR_out.shape[1] == D and dR_out.shape[0] >= T and dR_out.stride(1) == 1
        dR = dR_out
    else:
        dR = torch.empty(T, D, device=dev, dtype=torch.float32)
    ws, ws_bytes = _workspace('relpos_dr', dev, lib.emo_relpos_attn_bwd_r_workspace_bytes(B, T, H, dh))
    check(lib.emo_relpos_attn_bwd_r(ptr(qu), ptr(qv), D, ptr(k), ptr(v), D3, ptr(r_dist), _rows(r_dist), r_dist.shape[0], ptr(dout), D, ptr(lse),
                                    ptr(zden), ptr(delta), ptr(dR), _rows(dR), ptr(ws), ws_bytes, dtype_code(dt), B, T, H, dh, p_drop, seed, offset, stream()))
    return dqkv, dR, d_rw, d_rr


def relpos_attn_decode(q, kcache, vcache, lens, H, r_dist, r_w_bias, r_r_bias, mem_len=0, lens_off=0, k_new=None, v_new=None):
    n, HD = q.shape
    T_max = kcache.shape[1]
    assert kcache.is_contiguous() and vcache.is_contiguous() and lens.dtype == torch.int64 and r_dist.shape[0] >= T_max
    out = torch.empty(n, HD, device=q.device, dtype=q.dtype)
    check(lib.emo_relpos_attn_decode(ptr(q), _rows(q), ptr(kcache), ptr(vcache), T_max, ptr(lens), lens_off, mem_len, ptr(k_new), ptr(v_new),
                                     0 if k_new is None else _rows(k_new), ptr(r_dist), _rows(r_dist), r_dist.shape[0], ptr(r_w_bias), ptr(r_r_bias),
                                     ptr(out), HD, dtype_code(q.dtype), n, H, HD // H, stream()))
    return out


def xent_fwd(logits, tgt, ignore_index):
    M, V = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32
    lse = torch.empty(M, device=logits.device, dtype=torch.float32)
    acc = torch.zeros(2, device=logits.device, dtype=torch.float32)
    tgt = _c(tgt)
    check(lib.emo_xent_fwd(ptr(logits), ptr(tgt), M, V, ignore_index, ptr(lse), ptr(acc), stream()))
    return lse, acc


def xent_bwd(logits, tgt, lse, gscale, ignore_index, out_dtype, ld_out=None):
    M, V = logits.shape
    ld_out = ld_out or ((V + 7) // 8) * 8
    dl = torch.empty(M, ld_out, device=logits.device, dtype=out_dtype)
    tgt = _c(tgt)
    check(lib.emo_xent_bwd(ptr(logits), ptr(tgt), ptr(lse), ptr(gscale), ptr(dl), ld_out, dtype_code(out_dtype), M, V,
                           ignore_index, stream()))
    return dl


def argmax(logits):
    rows, V = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32
    out = torch.empty(rows, device=logits.device, dtype=torch.int64)
    check(lib.emo_argmax(ptr(logits), rows, V, ptr(out), stream()))
    return out


def sample_nucleus(logits, temperature, top_p, u):
    rows, V = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32 and u.dtype == torch.float32 and u.numel() == rows
    out = torch.empty(rows, device=logits.device, dtype=torch.int64)
    check(lib.emo_sample_nucleus(ptr(logits), rows, V, temperature, top_p, ptr(u), ptr(out), stream()))
    return out


def sample_nucleus_step(logits, temperature, top_p, u_steps, step, seq=None, col0=0, out=None):
    """One lock-step sampling step with device-side loop state (see include/emo_hip.h): uses u_steps[step[r], r], writes the id to out[r]
    and seq[r, col0 + step[r]], increments step[r].  Allocation-free when `out` is given (hipGraph capture)."""
    rows, V = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32 and u_steps.dtype == torch.float32 and u_steps.is_contiguous()
    assert u_steps.shape[-1] == rows and step.dtype == torch.int64 and step.numel() == rows
    assert seq is None or (seq.dtype == torch.int64 and seq.stride(1) == 1 and seq.shape[0] == rows)
    if out is None:
        out = torch.empty(rows, device=logits.device, dtype=torch.int64)
    check(lib.emo_sample_nucleus_step(ptr(logits), rows, V, temperature, top_p, ptr(u_steps), ptr(step), ptr(seq), 0 if seq is None else seq.stride(0),
                                      col0, ptr(out), stream()))
    return out


def accuracy_counts(logits, tgt, chord, melody, pad):
    M, V = logits.shape
    counts = torch.zeros(6, device=logits.device, dtype=torch.int64)
    assert logits.is_contiguous() and logits.dtype == torch.float32
    tgt, chord, melody = _c(tgt), _c(chord), _c(melody)
    check(lib.emo_accuracy_counts(ptr(logits), ptr(tgt), ptr(chord), ptr(melody), M, V, pad, ptr(counts), stream()))
    return counts


SUMSQ_FLOATS = 1026     # emo_hip.h: EMO_SUMSQ_FLOATS


def sumsq(x, acc):
    """acc[0] = sum(x*x) in a fixed summation order; acc: SUMSQ_FLOATS fp32 scratch, zeroed once by the caller."""
    assert acc.dtype == torch.float32 and acc.numel() >= SUMSQ_FLOATS
    check(lib.emo_sumsq(ptr(x), x.numel(), ptr(acc), stream()))


def clip_coef(sumsq_t, max_norm, pre, coef, denom=None):
    """coef = pre' * min(1, max_norm / (|g| pre' + 1e-6)), pre' = pre / denom[0] (denom: optional fp32 device scalar)."""
    assert denom is None or (denom.dtype == torch.float32 and denom.numel() >= 1)
    check(lib.emo_clip_coef(ptr(sumsq_t), max_norm, pre, ptr(denom), ptr(coef), stream()))


def adam_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, step, gscale):
    check(lib.emo_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), p.numel(), lr, beta1, beta2, eps, step, ptr(gscale), stream()))


def transpose_batch(desc, n, total_tiles):
    """desc: device int64 [n, 6] = {src ptr, dst ptr, rows, cols, first tile, tiles per row} (include/emo_hip.h); one launch."""
    check(lib.emo_transpose_batch(ptr(desc), n, total_tiles, stream()))


def add_bias2(x, b1, b2):
    """(x + b1, x + b2): x [M, D] (row-strided view allowed), fp32 biases of D elements (any shape); contiguous outputs in x's dtype, one launch."""
    M, D = x.shape
    assert b1.dtype == torch.float32 and b2.dtype == torch.float32 and b1.numel() == D and b2.numel() == D and b1.is_contiguous() and b2.is_contiguous()
    o1, o2 = torch.empty(M, D, device=x.device, dtype=x.dtype), torch.empty(M, D, device=x.device, dtype=x.dtype)
    check(lib.emo_add_bias2(ptr(x), _rows(x), ptr(b1), ptr(b2), ptr(o1), ptr(o2), dtype_code(x.dtype), M, D, stream()))
    return o1, o2


def cast(src, dst):
    assert src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous()
    check(lib.emo_cast(ptr(src), dtype_code(src.dtype), ptr(dst), dtype_code(dst.dtype), src.numel(), stream()))
    return dst
