"""Kernel-level parity (GPU): every C-ABI entry point against the oracle / a PyTorch fp32-fp64 CPU
reference of the same op.  Tolerances: EMO_F32 paths 2e-5 (exact-f32 MFMA, reassociation only);
EMO_BF16 paths 3e-2 relative to the tensor scale (bf16 storage, fp32 accumulate)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _ops():
    from emo_disentanger_amd import ops
    return ops


def _tol(dt):
    return 2e-5 if dt == torch.float32 else 3e-2


def _close(got, ref, dt, scale=None, mult=1.0):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    s = float(ref.abs().max()) if scale is None else scale
    err = float((got - ref).abs().max())
    assert err <= mult * _tol(dt) * max(s, 1e-6), 'max err %.3e vs scale %.3e (tol %.1e)' % (err, s, mult * _tol(dt))


def _r(*shape, seed=0, dt=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt)


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('a_trans,b_trans', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(70, 50, 37), (200, 136, 264), (129, 327, 512), (1024, 512, 512)])
def test_gemm_layouts(dt, a_trans, b_trans, M, N, K):
    ops = _ops()
    if dt == torch.bfloat16:   # bf16 path: leading dims must be multiples of 8 -> allocate padded, take views
        pad = lambda n: (n + 7) // 8 * 8
    else:
        pad = lambda n: n
    A = _r(*((K, pad(M)) if a_trans else (M, pad(K))), seed=1, dt=dt)
    Bm = _r(*((K, pad(N)) if b_trans else (N, pad(K))), seed=2, dt=dt)
    Av = A[:, :M] if a_trans else A[:, :K]
    Bv = Bm[:, :N] if b_trans else Bm[:, :K]
    ref = (Av.double().T if a_trans else Av.double()) @ (Bv.double() if b_trans else Bv.double().T)
    Ac, Bc = A.cuda(), Bm.cuda()            # slice AFTER the copy so that the padded row stride survives
    Ag = Ac[:, :M] if a_trans else Ac[:, :K]
    Bg = Bc[:, :N] if b_trans else Bc[:, :K]
    out = ops.gemm(Ag, Bg, a_trans=bool(a_trans), b_trans=bool(b_trans), out_dtype=torch.float32)
    _close(out, ref, dt, mult=1.0 if dt == torch.float32 else 0.3)


@pytest.mark.parametrize('a_trans,b_trans', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(128, 128, 16), (200, 136, 264), (129, 327, 509), (1024, 512, 512), (384, 2048, 96), (4096, 327, 512)])
def test_gemm_f32_128_tile_equals_the_64_tile_kernel(a_trans, b_trans, M, N, K, monkeypatch):
    """r06: fp32 products with M, N >= 128 run on a 128 x 128 x 16 tile kernel (16-byte operand loads, register-staged double buffer) instead of the
    64 x 64 element-wise one.  Same MFMA instruction, same k order inside a dot product => unsplit results are BIT-identical; every layout, odd sizes
    (scalar edge loads, K tails), unaligned row strides (views of padded tensors: vector loads off), epilogue."""
    ops = _ops()
    for pad in (0, 3):                                           # 3: row stride not a multiple of 4 -> the element-wise loader of the new kernel
        A = _r(*((K, M + pad) if a_trans else (M, K + pad)), seed=1).cuda()
        Bm = _r(*((K, N + pad) if b_trans else (N, K + pad)), seed=2).cuda()
        Ag = A[:, :M] if a_trans else A[:, :K]
        Bg = Bm[:, :N] if b_trans else Bm[:, :K]
        bias = _r(N, seed=3).cuda()
        kw = dict(a_trans=bool(a_trans), b_trans=bool(b_trans), bias=bias, act=ops.ACT_RELU)
        monkeypatch.setenv('EMO_GEMM_F32_TILE', '64')
        y64 = ops.gemm(Ag, Bg, **kw)
        monkeypatch.delenv('EMO_GEMM_F32_TILE')
        y = ops.gemm(Ag, Bg, **kw)
        assert torch.equal(y, y64), (pad, float((y - y64).abs().max()))
        ref = torch.relu((Ag.double().T if a_trans else Ag.double()) @ (Bg.double() if b_trans else Bg.double().T) + bias.double())
        _close(y, ref, torch.float32)


def test_gemm_f32_128_tile_split_k_weight_gradient():
    # the wgrad layout with a long reduction (split-K through the workspace, accumulate into C) on the 128-tile plan
    ops = _ops()
    Mtok, N, K = 8192, 512, 256
    dY, X = _r(Mtok, N, seed=5).cuda(), _r(Mtok, K, seed=6).cuda()
    C = _r(N, K, seed=7).cuda()
    C0 = C.clone()
    ops.gemm(dY, X, a_trans=True, b_trans=True, out=C, accumulate=True)
    ref = C0.double() + dY.double().T @ X.double()
    assert float((C.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    C2 = C0.clone()
    ops.gemm(dY, X, a_trans=True, b_trans=True, out=C2, accumulate=True)
    assert torch.equal(C, C2)                                    # workspace split-K: fixed summation order


@pytest.mark.parametrize('dt', DT)
def test_gemm_epilogue_bias_act_residual_aux(dt):
    ops = _ops()
    M, N, K = 300, 264, 128
    A, W = _r(M, K, seed=3, dt=dt), _r(N, K, seed=4, dt=dt, scale=0.2)
    bias, res = _r(N, seed=5), _r(M, N, seed=6, dt=dt)
    z = A.double() @ W.double().T + bias.double()
    for act, fn in ((ops.ACT_RELU, torch.relu), (ops.ACT_GELU_NEW, lambda x: 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))),
                    (ops.ACT_NONE, lambda x: x)):
        aux = torch.empty(M, N, device='cuda', dtype=dt)
        out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), act=act, residual=res.cuda(), aux_out=aux)
        _close(aux, z, dt)
        _close(out, fn(z) + res.double(), dt)


@pytest.mark.parametrize('dt', DT)
def test_gemm_epilogue_mul_modes_and_dropout(dt):
    ops = _ops()
    M, N, K = 256, 128, 64
    A, W = _r(M, K, seed=7, dt=dt), _r(N, K, seed=8, dt=dt, scale=0.3)
    aux = _r(M, N, seed=9, dt=dt)
    aux[aux.abs() < 0.5] = 0
    z = A.double() @ W.double().T
    out = ops.gemm(A.cuda(), W.cuda(), mul_aux=aux.cuda(), mul_mode=ops.MUL_NONZERO, mul_scale=1.25)
    _close(out, z * (aux.double() != 0) * 1.25, dt)
    x = aux.double()
    u = math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)
    dg = 0.5 * (1 + torch.tanh(u)) + 0.5 * x * (1 - torch.tanh(u) ** 2) * math.sqrt(2 / math.pi) * (1 + 3 * 0.044715 * x ** 2)
    out = ops.gemm(A.cuda(), W.cuda(), mul_aux=aux.cuda(), mul_mode=ops.MUL_DGELU_NEW)
    _close(out, z * dg, dt)
    # dropout: deterministic in (seed, offset), ~p zeros, survivors scaled by 1/(1-p), and consistent
    # with emo_dropout_apply / layernorm_bwd masks (same element indexing m*N+n)
    o1 = ops.gemm(A.cuda(), W.cuda(), p_drop=0.1, seed=11, offset=3)
    o2 = ops.gemm(A.cuda(), W.cuda(), p_drop=0.1, seed=11, offset=3)
    o3 = ops.gemm(A.cuda(), W.cuda(), p_drop=0.1, seed=11, offset=4)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    plain = ops.gemm(A.cuda(), W.cuda())
    masked = ops.dropout_apply(plain, 0.1, 11, 3)
    _close(o1, masked, dt, mult=2.0)
    frac = float((o1 == 0).float().mean())
    assert 0.08 < frac < 0.12
    keep = o1 != 0
    _close(o1[keep], (plain.double() / 0.9)[keep], dt, mult=2.0)


@pytest.mark.parametrize('M,N,K,b_trans', [(2048, 512, 2048, False), (2048, 512, 2048, True), (4096, 512, 1536, True), (2048, 1024, 1024, False)])
def test_gemm_small_grid_long_k_split_with_epilogue(M, N, K, b_trans):
    """bf16 products whose 128 x 128 tile grid leaves the chip empty (<= 256 tiles: stage 1, batch-size-4 steps) and whose reduction is long run
    split-K through the workspace + splitk_reduce_epi_kernel, which applies the whole fused epilogue: against fp64, dropout against
    emo_dropout_apply of the plain product (same element indexing), the multiplicative mask mode, and the pre-activation side output."""
    ops = _ops()
    dt = torch.bfloat16
    from emo_disentanger_amd._lib import lib, dtype_code
    assert lib.emo_gemm_workspace_bytes(M, N, K, dtype_code(dt), dtype_code(dt)) > 0          # this shape takes the split path
    A = _r(M, K, seed=21, dt=dt)
    W = _r(K, N, seed=22, dt=dt, scale=0.05) if b_trans else _r(N, K, seed=22, dt=dt, scale=0.05)
    bias, res = _r(N, seed=23), _r(M, N, seed=24, dt=dt)
    z = A.double() @ (W.double() if b_trans else W.double().T)
    Ac, Wc = A.cuda(), W.cuda()
    plain = ops.gemm(Ac, Wc, b_trans=b_trans)
    _close(plain, z, dt)
    aux = torch.empty(M, N, device='cuda', dtype=dt)
    out = ops.gemm(Ac, Wc, b_trans=b_trans, bias=bias.cuda(), act=ops.ACT_RELU, residual=res.cuda(), aux_out=aux)
    _close(aux, z + bias.double(), dt)
    _close(out, torch.relu(z + bias.double()) + res.double(), dt)
    o1 = ops.gemm(Ac, Wc, b_trans=b_trans, p_drop=0.1, seed=11, offset=3, residual=res.cuda())
    masked = ops.dropout_apply(plain, 0.1, 11, 3)
    _close(o1, masked.double().cpu() + res.double(), dt, mult=2.0)
    mask = _r(M, N, seed=25, dt=dt)
    mask[mask.abs() < 0.6] = 0
    o2 = ops.gemm(Ac, Wc, b_trans=b_trans, mul_aux=mask.cuda(), mul_mode=ops.MUL_NONZERO, mul_scale=1.25)
    _close(o2, z * (mask.double() != 0) * 1.25, dt)


def test_gemm_splitk_wgrad_accumulate():
    ops = _ops()
    for dt in DT:
        Mred, N, K = 4096, 136, 264          # dW[N,K] = dY[Mred,N]^T X[Mred,K]
        dY, X = _r(Mred, N, seed=12, dt=dt), _r(Mred, K, seed=13, dt=dt)
        ref = dY.double().T @ X.double()
        out = ops.gemm(dY.cuda(), X.cuda(), a_trans=True, b_trans=True, out_dtype=torch.float32)
        _close(out, ref, dt, mult=1.0 if dt == torch.float32 else 0.3)
        out2 = ops.gemm(dY.cuda(), X.cuda(), a_trans=True, b_trans=True, out=out.clone(), accumulate=True)
        _close(out2, 2 * ref, dt, mult=1.0 if dt == torch.float32 else 0.3)
        cs = ops.colsum(dY.cuda())
        _close(cs, dY.double().sum(0), dt, mult=1.0 if dt == torch.float32 else 0.3)


@pytest.mark.parametrize('splits', [None, 2, 4, 8, 24])
def test_gemm_splitk_workspace_is_deterministic_and_matches_atomics(splits, monkeypatch):
    # wgrad shapes: with the caller workspace the K-splits are summed in a fixed order (bitwise reproducible); without it they are
    # fp32 atomics into C.  Both must agree with the reference; odd sizes exercise tile edges and the ragged last split.
    ops = _ops()
    from emo_disentanger_amd._lib import lib
    if splits is not None:
        monkeypatch.setenv('EMO_GEMM_SPLITS', str(splits))
    for dt in DT:
        Mred, N, K = 8192 + 64, 264, 200
        assert lib.emo_gemm_workspace_bytes(N, K, Mred, 0 if dt == torch.float32 else 1, 0) > 0
        dY, X = _r(Mred, N, seed=12, dt=dt).cuda(), _r(Mred, K, seed=13, dt=dt).cuda()
        ref = dY.double().T @ X.double()
        base = _r(N, K, seed=14).cuda()
        o1 = ops.gemm(dY, X, a_trans=True, b_trans=True, out=base.clone(), accumulate=True)
        o2 = ops.gemm(dY, X, a_trans=True, b_trans=True, out=base.clone(), accumulate=True)
        assert torch.equal(o1, o2)
        _close(o1 - base, ref, dt, mult=1.0 if dt == torch.float32 else 0.3)
        o3 = ops.gemm(dY, X, a_trans=True, b_trans=True, out_dtype=torch.float32)
        _close(o3, ref, dt, mult=1.0 if dt == torch.float32 else 0.3)
        monkeypatch.setenv('EMO_GEMM_SPLIT_ATOMIC', '1')
        o4 = ops.gemm(dY, X, a_trans=True, b_trans=True, out_dtype=torch.float32)
        monkeypatch.delenv('EMO_GEMM_SPLIT_ATOMIC')
        _close(o4, o3, dt, scale=float(ref.abs().max()), mult=0.05)


@pytest.mark.parametrize('splits', [None, 3, 5, 13])
def test_gemm_splitk_workspace_stays_inside_its_bounds(splits, monkeypatch):
    # the split grid is rounded up to whole XCD rounds; surplus splits own no K range and must not touch the workspace
    # (regression: the fp32 kernel wrote zero tiles past an exactly-sized workspace)
    _ops()
    import ctypes
    from emo_disentanger_amd._lib import lib, ptr, check, Epilogue
    if splits is not None:
        monkeypatch.setenv('EMO_GEMM_SPLITS', str(splits))
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        Mred, N, K = 4096 + 128, 136, 72
        need = lib.emo_gemm_workspace_bytes(N, K, Mred, code, 0)
        assert need > 0
        dY, X = _r(Mred, N, seed=1, dt=dt).cuda(), _r(Mred, K, seed=2, dt=dt).cuda()
        guard = 1 << 20
        buf = torch.full((need + guard,), 0x5A, dtype=torch.uint8, device='cuda')
        out = torch.zeros(N, K, device='cuda')
        epi = Epilogue(mul_scale=1.0, workspace=ptr(buf), workspace_bytes=need)
        check(lib.emo_gemm(ptr(dY), 1, N, ptr(X), 1, K, ptr(out), K, N, K, Mred, code, 0, 0, ctypes.byref(epi), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert bool((buf[need:] == 0x5A).all()), 'workspace overrun'
        _close(out, dY.double().T @ X.double(), dt, mult=1.0 if dt == torch.float32 else 0.3)


def test_gemm_large_wgrad_matches_torch_and_is_deterministic():
    # 12 output tiles of 256^2, 32768 tokens: split-K with a workspace and the fused bias gradient (the 256^2 one-wave-per-SIMD tile)
    ops = _ops()
    M, N, K = 1024, 768, 32768
    dy, x = (_r(K, M, seed=1) * 0.5).to(torch.bfloat16).cuda(), (_r(K, N, seed=2) * 0.5).to(torch.bfloat16).cuda()
    outs = []
    for _ in range(3):
        dw, db = torch.zeros(M, N, device='cuda'), torch.zeros(M, device='cuda')
        ops.gemm(dy, x, a_trans=True, b_trans=True, out=dw, a_rowsum=db)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = dy.double().T @ x.double()
    assert float((outs[0].double() - ref).abs().max() / ref.abs().max()) < 1e-4
    want = dy.double().sum(0)
    assert float((db.double() - want).abs().max() / want.abs().max()) < 1e-4


@pytest.mark.parametrize('shape', [(512, 256, 320), (1024, 512, 2048), (2048, 768, 1024), (256, 1024, 4096)])
def test_gemm_256_tile_nt_matches_reference_and_128_tile(shape, monkeypatch):
    # emo_gemm_w128.hip: 256 x 256 tile, one wave per SIMD (default for K >= 1024 with >= 256 tiles; EMO_GEMM_W128=1 admits every eligible shape).
    # Against fp64 and BIT-identical to the 128 x 128 kernel (same k order, same dropout hash) for every epilogue it takes.
    ops = _ops()
    monkeypatch.setenv('EMO_GEMM_EPI_SPLIT', '0')        # (the 128 x 128 kernel unsplit: split-K sums in another order)
    M, N, K = shape
    A, W = _r(M, K, seed=1).to(torch.bfloat16).cuda(), _r(N, K, seed=2, scale=0.1).to(torch.bfloat16).cuda()
    bias, res = _r(N, seed=3).cuda(), _r(M, N, seed=4).to(torch.bfloat16).cuda()
    for kw in ({}, dict(bias=bias), dict(residual=res), dict(bias=bias, p_drop=0.1, seed=5, offset=7, residual=res)):
        monkeypatch.setenv('EMO_GEMM_W128', '1')
        y1, y2 = ops.gemm(A, W, **kw), ops.gemm(A, W, **kw)
        monkeypatch.setenv('EMO_GEMM_W128', '0')
        y0 = ops.gemm(A, W, **kw)
        assert torch.equal(y1, y2) and torch.equal(y1, y0), kw.keys()
        if 'p_drop' not in kw:
            ref = A.double() @ W.double().T + (bias.double() if 'bias' in kw else 0.0) + (res.double() if 'residual' in kw else 0.0)
            _close(y1, ref, torch.bfloat16, mult=1.0)


def _need_experimental():
    # the opt-in kernels of emo_gemm_p256.hip are only in a library built with `make EXTRA=-DEMO_EXPERIMENTAL` (emo_build_flags() & 1)
    if not (_ops().lib.emo_build_flags() & 1):
        pytest.skip('libemo_hip.so built without -DEMO_EXPERIMENTAL')


@pytest.mark.parametrize('shape', [(512, 256, 320), (1024, 512, 2048), (2048, 768, 1024), (256, 1024, 4096), (8192, 512, 64), (2304, 256, 96), (131072, 512, 1536)])
def test_gemm_persistent_256_tile_matches_reference(shape, monkeypatch):
    _need_experimental()
    # emo_gemm_p256.hip (r05, opt-in EMO_GEMM_P256=1): persistent tile walk, 32 x 32 x 16 MFMA, 32-deep slabs in a 5 + 5 ring.  Shapes cover
    # one tile per block (no walk), fewer blocks than CUs, several tiles per block with the operand stream crossing tile boundaries (2304 x 256 =
    # 9 tiles on 8 blocks; the last shape = the QKV dgrad of the benchmark: 4 tiles per CU), K = 64 (first slab is also the last), every
    # epilogue it takes; deterministic; the dropout mask is the one every other kernel applies.
    ops = _ops()
    monkeypatch.setenv('EMO_GEMM_EPI_SPLIT', '0')
    M, N, K = shape
    A, W = _r(M, K, seed=1).to(torch.bfloat16).cuda(), _r(N, K, seed=2, scale=0.1).to(torch.bfloat16).cuda()
    bias, res = _r(N, seed=3).cuda(), _r(M, N, seed=4).to(torch.bfloat16).cuda()
    for kw in ({}, dict(bias=bias), dict(residual=res), dict(bias=bias, p_drop=0.1, seed=5, offset=7, residual=res)):
        monkeypatch.setenv('EMO_GEMM_P256', '1')
        y1 = ops.gemm(A, W, **kw)
        assert ops.lib.emo_gemm_last_kernel() == 8
        y2 = ops.gemm(A, W, **kw)
        monkeypatch.setenv('EMO_GEMM_P256', '0')
        y0 = ops.gemm(A, W, **kw)
        assert ops.lib.emo_gemm_last_kernel() != 8
        assert torch.equal(y1, y2), kw.keys()
        if 'p_drop' not in kw:
            rows = slice(0, M) if M <= 8192 else slice(M - 4096, M)
            ref = A[rows].double() @ W.double().T + (bias.double() if 'bias' in kw else 0.0) + (res[rows].double() if 'residual' in kw else 0.0)
            _close(y1[rows], ref, torch.bfloat16, mult=1.0)
        # same products, another summation order inside a dot product: equal up to one bf16 rounding of the output; a dropped element is dropped in both
        assert float((y1.float() - y0.float()).abs().max()) <= 0.02 * float(y0.float().abs().max())
        if 'p_drop' in kw:
            base = ops.gemm(A, W, bias=bias).float()
            kept = (y1.float() - res.float()).abs() > 1e-6 * (1 + base.abs())
            kept0 = (y0.float() - res.float()).abs() > 1e-6 * (1 + base.abs())
            assert float((kept != kept0).float().mean()) < 1e-3


@pytest.mark.parametrize('shape', [(128, 512, 64), (1152, 512, 256), (256, 1024, 2048), (65536, 512, 1536)])
@pytest.mark.parametrize('persist', ['0', '1'])
def test_gemm_full_n_tile_matches_reference(shape, persist, monkeypatch):
    _need_experimental()
    # gemm_q512_kernel (emo_gemm_p256.hip, r05, opt-in EMO_GEMM_Q512=1): 128 x 512 tile — every A byte requested once, a lane's outputs complete
    # 128-B lines; A ring 8 slabs deep, B ring 3.  One tile per block and (EMO_Q512_PERSIST=1) blocks walking several tiles with the two operand
    # streams crossing tile boundaries at different times; K = 64 (first slab is also the last); N = 1024 (two column tiles per row panel).
    ops = _ops()
    monkeypatch.setenv('EMO_GEMM_EPI_SPLIT', '0')
    monkeypatch.setenv('EMO_Q512_PERSIST', persist)
    M, N, K = shape
    A, W = _r(M, K, seed=1).to(torch.bfloat16).cuda(), _r(N, K, seed=2, scale=0.1).to(torch.bfloat16).cuda()
    bias, res = _r(N, seed=3).cuda(), _r(M, N, seed=4).to(torch.bfloat16).cuda()
    for kw in ({}, dict(bias=bias, residual=res), dict(bias=bias, p_drop=0.1, seed=5, offset=7, residual=res)):
        monkeypatch.setenv('EMO_GEMM_Q512', '1')
        y1 = ops.gemm(A, W, **kw)
        assert ops.lib.emo_gemm_last_kernel() == 9
        monkeypatch.setenv('EMO_GEMM_Q512', '0')
        y0 = ops.gemm(A, W, **kw)
        assert ops.lib.emo_gemm_last_kernel() != 9
        if 'p_drop' not in kw:
            rows = slice(0, M) if M <= 8192 else slice(M - 4096, M)
            ref = A[rows].double() @ W.double().T + (bias.double() if 'bias' in kw else 0.0) + (res[rows].double() if 'residual' in kw else 0.0)
            _close(y1[rows], ref, torch.bfloat16, mult=1.0)
        assert float((y1.float() - y0.float()).abs().max()) <= 0.02 * float(y0.float().abs().max())
        if 'p_drop' in kw:
            base = ops.gemm(A, W, bias=bias).float()
            kept = (y1.float() - res.float()).abs() > 1e-6 * (1 + base.abs())
            kept0 = (y0.float() - res.float()).abs() > 1e-6 * (1 + base.abs())
            assert float((kept != kept0).float().mean()) < 1e-3


def test_gemm_persistent_256_tile_leaves_other_shapes_to_the_other_kernels(monkeypatch):
    _need_experimental()
    # EMO_GEMM_P256=1 only takes M, N multiples of 256, K a multiple of 32 (>= 64), bf16 outputs, bias / dropout / residual epilogues: everything
    # else must run (and be right) on the kernels it ran on before
    ops = _ops()
    monkeypatch.setenv('EMO_GEMM_P256', '1')
    for (M, N, K), kw in (((300, 256, 64), {}), ((256, 200, 64), {}), ((256, 256, 48), {}), ((256, 256, 32), {}), ((512, 256, 128), dict(act=ops.ACT_RELU)),
                          ((512, 256, 128), dict(out_dtype=torch.float32))):
        pad = lambda n: (n + 7) // 8 * 8
        A, W = _r(M, pad(K), seed=1).to(torch.bfloat16).cuda()[:, :K], _r(N, pad(K), seed=2, scale=0.1).to(torch.bfloat16).cuda()[:, :K]
        y = ops.gemm(A, W, **kw)
        assert ops.lib.emo_gemm_last_kernel() != 8, (M, N, K, kw)
        ref = A.double() @ W.double().T
        if kw.get('act') == ops.ACT_RELU:
            ref = ref.clamp_min(0)
        _close(y, ref, torch.bfloat16, mult=1.0)


@pytest.mark.parametrize('shape', [(8192, 2048, 512), (16384, 512, 2048), (8192, 1536, 512)])
def test_gemm_256_tile_wgrad_matches_reference_with_bias_gradient(shape, monkeypatch):
    # the wgrad instance of the same tile: token-major operands, split-K through the workspace, bias gradient by ones-MFMAs; with and
    # without the bias gradient the weight gradient must be the SAME bits (the r03 compiler trap: see the kernel source), and deterministic
    ops = _ops()
    K, M, N = shape
    dy, x = _r(K, M, seed=1).to(torch.bfloat16).cuda(), _r(K, N, seed=2).to(torch.bfloat16).cuda()
    ref, want = dy.double().T @ x.double(), dy.double().sum(0)
    for acc in (False, True):
        got = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('EMO_GEMM_W128_TN', mode)
            dw, db, dw2 = torch.full((M, N), 0.5, device='cuda'), torch.full((M,), 0.25, device='cuda'), torch.full((M, N), 0.5, device='cuda')
            ops.gemm(dy, x, a_trans=True, b_trans=True, out=dw, accumulate=acc, a_rowsum=db)
            ops.gemm(dy, x, a_trans=True, b_trans=True, out=dw2, accumulate=acc)
            got[mode] = (dw, db, dw2)
        dw, db, dw2 = got['1']
        assert torch.equal(dw, dw2)
        # the bias-gradient partials go through the workspace and are summed in a fixed order (r04): same bits on every run
        monkeypatch.setenv('EMO_GEMM_W128_TN', '1')
        for _ in range(3):
            db_r = torch.full((M,), 0.25, device='cuda')
            ops.gemm(dy, x, a_trans=True, b_trans=True, out=torch.full((M, N), 0.5, device='cuda'), accumulate=acc, a_rowsum=db_r)
            assert torch.equal(db_r, db)
        assert float((dw.double() - (ref + (0.5 if acc else 0.0))).abs().max() / ref.abs().max()) < 2e-5
        assert float((db.double() - (want + 0.25)).abs().max() / want.abs().max()) < 1e-4
        assert float((dw - got['0'][0]).abs().max() / ref.abs().max()) < 2e-5      # the 128 x 128 kernel (other split count: not the same bits)
    # Conv1D layout: the bias gradient is the column sum of the B operand
    monkeypatch.setenv('EMO_GEMM_W128_TN', '1')
    dw3, dbn = torch.zeros(M, N, device='cuda'), torch.full((N,), 0.25, device='cuda')
    ops.gemm(dy, x, a_trans=True, b_trans=True, out=dw3, b_rowsum=dbn)
    wantn = x.double().sum(0)
    dbn2 = torch.full((N,), 0.25, device='cuda')
    ops.gemm(dy, x, a_trans=True, b_trans=True, out=torch.zeros(M, N, device='cuda'), b_rowsum=dbn2)
    assert torch.equal(dbn, dbn2)
    assert float((dw3.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((dbn.double() - (wantn + 0.25)).abs().max() / wantn.abs().max()) < 1e-4


@pytest.mark.parametrize('M', [1, 7, 32])
def test_gemm_skinny_layernorm_folding(M):
    # decode path: y = LN(x) W^T + b and y2 = f W2^T + b2 + LN(x) computed WITHOUT a LayerNorm launch (statistics in-kernel / exported)
    ops = _ops()
    K, N = 512, 224
    x = (_r(M, K, seed=1) * 2.0 + 0.3).to(torch.bfloat16)
    W = _r(N, K, seed=2, scale=0.05)
    b, gamma, beta = _r(N, seed=3, scale=0.1), 1.0 + _r(K, seed=4, scale=0.1), _r(K, seed=5, scale=0.1)
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    ln = (xd - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    ref = torch.relu(ln @ W.double().T + b.double())
    Wg = (W * gamma[None, :]).to(torch.bfloat16).cuda()
    c1 = Wg.float().sum(1).contiguous()
    bias = (b + W @ beta).cuda()
    stats = torch.empty(M, 2, device='cuda')
    y = ops.gemm(x.cuda(), Wg, bias=bias, act=ops.ACT_RELU, ln_c1=c1, ln_stats_out=stats)
    _close(y, ref, torch.bfloat16, mult=2)
    _close(stats[:, 0], mean[:, 0], torch.float32, scale=1.0, mult=10)
    _close(stats[:, 1], 1.0 / torch.sqrt(var[:, 0] + 1e-5), torch.float32, scale=1.0, mult=10)
    # second GEMM (K2 = N): residual = LN(x)[:, :N2] rebuilt from the exported statistics (N2 = K so that x has the output's shape)
    f = _r(M, N, seed=6).to(torch.bfloat16)
    W2 = _r(K, N, seed=7, scale=0.05).to(torch.bfloat16)
    b2 = _r(K, seed=8, scale=0.1)
    ref2 = f.double() @ W2.double().T + b2.double() + ln
    y2 = ops.gemm(f.cuda(), W2.cuda(), bias=b2.cuda(), rln=(x.cuda(), stats, gamma.cuda(), beta.cuda()))
    _close(y2, ref2, torch.bfloat16, mult=2)
    with pytest.raises(Exception, match='skinny'):
        ops.gemm(_r(64, K, seed=1).to(torch.bfloat16).cuda(), Wg, ln_c1=c1)


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('Mred,N,K', [(4096 + 40, 328, 200), (1000, 136, 264), (8192, 1536, 512)])
def test_gemm_wgrad_carries_the_bias_gradient(dt, Mred, N, K):
    # dW += dY^T X with db += colsum(dY) taken inside the same launch (a_rowsum: nn.Linear layout; b_rowsum: HF Conv1D layout);
    # fp32 and the non-TN kernels fall back to the column-sum launch behind the same entry point
    ops = _ops()
    dY, X = _r(Mred, N, seed=12, dt=dt).cuda(), _r(Mred, K, seed=13, dt=dt).cuda()
    ref_w, ref_b = dY.double().T @ X.double(), dY.double().sum(0)
    tol = 1.0 if dt == torch.float32 else 0.3
    db0 = _r(N, seed=3).cuda()
    dw, db = torch.zeros(N, K, device='cuda'), db0.clone()
    ops.gemm(dY, X, a_trans=True, b_trans=True, out=dw, accumulate=True, a_rowsum=db)
    _close(dw, ref_w, dt, mult=tol)
    _close(db - db0, ref_b, dt, mult=tol)
    dw2, db2 = torch.zeros(K, N, device='cuda'), db0.clone()
    ops.gemm(X, dY, a_trans=True, b_trans=True, out=dw2, accumulate=True, b_rowsum=db2)
    _close(dw2, ref_w.T, dt, mult=tol)
    _close(db2 - db0, ref_b, dt, mult=tol)


def test_gemm_bf16_safe_and_tr_paths_agree(monkeypatch):
    # the transposed-operand fragments are fetched with ds_read_b64_tr_b16; EMO_GEMM_SAFE_TR=1 (read at first use)
    # selects a scalar-read variant of the same kernel; both are compared with the reference in the layout tests.
    ops = _ops()
    A, Bm = _r(264, 200, seed=20, dt=torch.bfloat16), _r(264, 136, seed=21, dt=torch.bfloat16)
    out = ops.gemm(A.cuda(), Bm.cuda(), a_trans=True, b_trans=True, out_dtype=torch.float32)
    _close(out, A.double().T @ Bm.double(), torch.bfloat16, mult=0.3)


# ------------------------------------------------------------------------------------------- relative-position attention (stage 1)
def _relattn_ref(q, k, v, R, u, vb):
    """q,k,v [B,T,H,dh] double; R [n_dist,H,dh] by distance; u, vb [H,dh].  The reference's AC + rel_shift(BD) written with explicit distances."""
    B, T, H, dh = q.shape
    AC = torch.einsum('bihd,bjhd->bhij', q + u, k)
    idx = (torch.arange(T)[:, None] - torch.arange(T)[None, :]).clamp(min=0)           # distance i - j (masked entries: any valid row)
    BD = torch.einsum('bihd,ijhd->bhij', q + vb, R[idx])
    sc = (AC + BD) / dh ** 0.5
    sc = sc.masked_fill(torch.triu(torch.ones(T, T), 1).bool(), -float('inf'))
    p = torch.softmax(sc, -1)
    p = p / (p.sum(-1, keepdim=True) + 1e-8)
    return torch.einsum('bhij,bjhd->bihd', p, v)


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('B,T,H,dh', [(2, 150, 2, 64), (1, 70, 3, 32), (2, 33, 2, 16), (1, 256, 1, 64), (1, 129, 2, 64)])
def test_relpos_attention_fwd_and_decode(dt, B, T, H, dh):
    ops = _ops()
    HD = H * dh
    qkv = _r(B * T, 3 * HD, seed=1, dt=dt, scale=0.7)
    R = _r(T + 5, HD, seed=2, dt=dt, scale=0.7)
    u, vb = _r(H, dh, seed=3, scale=0.3), _r(H, dh, seed=4, scale=0.3)
    q, k, v = [qkv[:, i * HD:(i + 1) * HD].double().view(B, T, H, dh) for i in range(3)]
    ref = _relattn_ref(q, k, v, R.double().view(-1, H, dh), u.double(), vb.double())
    qc = qkv.cuda()
    out, lse, zden = ops.relpos_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], R.cuda(), u.cuda(), vb.cuda(), B, T, H)
    _close(out.view(B, T, H, dh), ref, dt, mult=3)
    _close(zden, torch.full((B, H, T), 1.0), torch.float32, scale=1.0, mult=10)
    # one-token decode against a KV cache == the last row of the full attention (and the row before it, with the memory one shorter)
    T_max = T + 5
    kc = torch.zeros(B, T_max, HD, dtype=dt, device='cuda'); vc = torch.zeros(B, T_max, HD, dtype=dt, device='cuda')
    x3 = qc.view(B, T, 3 * HD)
    kc[:, :T - 1] = x3[:, :T - 1, HD:2 * HD]; vc[:, :T - 1] = x3[:, :T - 1, 2 * HD:]
    last = x3[:, T - 1].contiguous()
    lens = torch.full((B,), T, dtype=torch.long, device='cuda')
    o = ops.relpos_attn_decode(last[:, :HD], kc, vc, lens, H, R.cuda(), u.cuda(), vb.cuda(), k_new=last[:, HD:2 * HD], v_new=last[:, 2 * HD:])
    _close(o.view(B, H, dh), ref[:, T - 1], dt, mult=3)
    assert torch.equal(kc[:, T - 1], x3[:, T - 1, HD:2 * HD])                      # appended in-kernel
    # sliding memory: only the last mem_len keys (+ the token itself) are attended
    ml = 20
    if T > ml + 2:
        o2 = ops.relpos_attn_decode(last[:, :HD], kc, vc, lens, H, R.cuda(), u.cuda(), vb.cuda(), mem_len=ml)
        ref2 = _relattn_ref(q[:, T - 1 - ml:], k[:, T - 1 - ml:], v[:, T - 1 - ml:], R.double().view(-1, H, dh), u.double(), vb.double())[:, -1]
        _close(o2.view(B, H, dh), ref2, dt, mult=3)


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('B,T,H,dh', [(2, 150, 2, 64), (1, 70, 3, 32), (2, 33, 2, 16), (1, 128, 1, 64), (1, 330, 2, 64), (3, 64, 1, 32),
                                      (1, 2400, 1, 64)])     # the last one: tgt_len of the stage-1 YAMLs (38 tiles, 38 diagonals)
def test_relpos_attention_backward(dt, B, T, H, dh):
    ops = _ops()
    HD = H * dh
    qkv = _r(B * T, 3 * HD, seed=1, dt=dt, scale=0.7)
    R = _r(T, HD, seed=2, dt=dt, scale=0.7)
    u, vb = _r(H, dh, seed=3, scale=0.3), _r(H, dh, seed=4, scale=0.3)
    leaf = [t.double().requires_grad_(True) for t in (qkv, R, u, vb)]
    q, k, v = [leaf[0][:, i * HD:(i + 1) * HD].view(B, T, H, dh) for i in range(3)]
    ref = _relattn_ref(q, k, v, leaf[1].view(T, H, dh), leaf[2], leaf[3])
    dout = _r(B, T, H, dh, seed=5, dt=dt)
    ref.backward(dout.double())
    qc = qkv.cuda()
    out, lse, zden = ops.relpos_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], R.cuda(), u.cuda(), vb.cuda(), B, T, H)
    dqkv, dR, du, dvb = ops.relpos_attn_bwd(qc, R.cuda(), u.cuda(), vb.cuda(), out, dout.view(B * T, HD).cuda(), lse, zden, B, T, H)
    gs = float(leaf[0].grad.abs().max())
    _close(dqkv[:, 2 * HD:], leaf[0].grad[:, 2 * HD:], dt, scale=gs, mult=4)
    _close(dqkv[:, HD:2 * HD], leaf[0].grad[:, HD:2 * HD], dt, scale=gs, mult=4)
    _close(dqkv[:, :HD], leaf[0].grad[:, :HD], dt, scale=gs, mult=4)
    _close(dR, leaf[1].grad, dt, scale=float(leaf[1].grad.abs().max()), mult=6)
    _close(du, leaf[2].grad, dt, scale=float(leaf[2].grad.abs().max()), mult=8)
    _close(dvb, leaf[3].grad, dt, scale=float(leaf[3].grad.abs().max()), mult=8)


# ------------------------------------------------------------------------------------------- embedding / LN / xent
@pytest.mark.parametrize('dt', DT)
def test_embed_fwd_bwd(dt):
    ops = _ops()
    from oracle.weights import positional_encoding
    B, T, D, V = 3, 50, 64, 37
    E, S = _r(V, D, seed=1).requires_grad_(True), _r(2, D, seed=2).requires_grad_(True)
    pe = positional_encoding(D, 200)
    g = torch.Generator().manual_seed(0)
    tok, seg = torch.randint(0, V, (B, T), generator=g), torch.randint(0, 2, (B, T), generator=g)
    ref = (torch.nn.functional.embedding(tok, E) * 8.0 + torch.nn.functional.embedding(seg, S) * 8.0) + pe[:T].permute(1, 0, 2)
    out = ops.embed_fwd(tok.cuda(), seg.cuda(), E.detach().cuda(), S.detach().cuda(), pe.cuda(), dt, 8.0)
    _close(out, ref, dt)
    out5 = ops.embed_fwd(tok.cuda(), seg.cuda(), E.detach().cuda(), S.detach().cuda(), pe.cuda(), dt, 8.0, pos0=5)
    _close(out5, (torch.nn.functional.embedding(tok, E) * 8.0 + torch.nn.functional.embedding(seg, S) * 8.0) + pe[5:5 + T].permute(1, 0, 2), dt)
    dout = _r(B, T, D, seed=3, dt=dt)
    ref.backward(dout.float())
    dE, dS = torch.zeros(V, D, device='cuda'), torch.zeros(2, D, device='cuda')
    ops.embed_bwd(tok.cuda(), seg.cuda(), dout.cuda(), dE, dS, 8.0)
    _close(dE, E.grad, torch.float32, mult=5)
    _close(dS, S.grad, torch.float32, mult=5)
    # dropout consistency fwd/bwd: masked positions get no gradient
    o = ops.embed_fwd(tok.cuda(), None, E.detach().cuda(), None, pe.cuda(), torch.float32, 1.0, p_drop=0.5, seed=4, offset=1)
    dE2 = torch.zeros(V, D, device='cuda')
    ones = torch.ones(B, T, D, device='cuda')
    ops.embed_bwd(tok.cuda(), None, ones, dE2, None, 1.0, p_drop=0.5, seed=4, offset=1)
    mask = (o != 0).float() * 2.0
    ref2 = torch.zeros(V, D, device='cuda').index_add_(0, tok.cuda().view(-1), mask.view(-1, D))
    _close(dE2, ref2, torch.float32, mult=5)


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('D', [64, 256, 512])
def test_layernorm_fwd_bwd(dt, D):
    ops = _ops()
    M = 203
    x = _r(M, D, seed=1, dt=dt).float().requires_grad_(True)
    gm, bt = (_r(D, seed=2) * 0.1 + 1).requires_grad_(True), _r(D, seed=3).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x, (D,), gm, bt, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.detach().to(dt).cuda(), gm.detach().cuda(), bt.detach().cuda())
    _close(y, ref, dt)
    dy, dres = _r(M, D, seed=4, dt=dt), _r(M, D, seed=5, dt=dt)
    ref.backward(dy.float())
    dg, db = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
    dcol = torch.zeros(D, device='cuda')
    dx, dxd = ops.layernorm_bwd(dy.cuda(), x.detach().to(dt).cuda(), gm.detach().cuda(), mean, rstd, dg, db, dres=dres.cuda(), want_drop=True,
                                p_drop=0.25, seed=9, offset=2, dcol=dcol)
    _close(dcol, dxd.double().sum(0), dt, mult=4)
    _close(dx, x.grad + dres.float(), dt, mult=2)
    _close(dg, gm.grad, dt, mult=4)
    _close(db, bt.grad, dt, mult=4)
    _close(dxd, ops.dropout_apply(dx, 0.25, 9, 2), dt)


@pytest.mark.parametrize('M', [7, 2048, 8193, 40000])
def test_layernorm_bwd_column_sums_without_atomics(M, monkeypatch):
    # bf16 / D = 512: the blocks' partial dgamma / dbeta / dcol sums go through the caller's workspace and are added in block order:
    # += semantics kept, equal to the atomic path within fp32 summation order, and bit-identical from run to run
    ops = _ops()
    D, dt = 512, torch.bfloat16
    x, dy, dres = _r(M, D, seed=1, dt=dt).cuda(), _r(M, D, seed=4, dt=dt).cuda(), _r(M, D, seed=5, dt=dt).cuda()
    gm, bt = (_r(D, seed=2) * 0.1 + 1).cuda(), _r(D, seed=3).cuda()
    _, mean, rstd = ops.layernorm_fwd(x, gm, bt)

    def run():
        acc = [torch.full((D,), 0.5, device='cuda') for _ in range(3)]
        dx, dxd = ops.layernorm_bwd(dy, x, gm, mean, rstd, acc[0], acc[1], dres=dres, want_drop=True, p_drop=0.1, seed=3, offset=1, dcol=acc[2])
        return [dx, dxd] + acc
    assert ops.lib.emo_layernorm_bwd_workspace_bytes(ops.dtype_code(dt), M, D) > 0
    a, b = run(), run()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    monkeypatch.setenv('EMO_LN_BWD_ATOMIC', '1')
    assert ops.lib.emo_layernorm_bwd_workspace_bytes(ops.dtype_code(dt), M, D) == 0
    c = run()
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    for u, v in zip(a[2:], c[2:]):
        assert (u - v).abs().max().item() <= 2e-5 * max(1.0, v.abs().max().item())
    xf = x.float()
    xh = (xf - mean[:, None]) * rstd[:, None]
    assert (a[2] - 0.5 - (dy.float() * xh).sum(0)).abs().max().item() <= 2e-3 * (M ** 0.5)
    assert torch.allclose(a[4] - 0.5, a[1].float().sum(0), atol=1e-3 * (M ** 0.5), rtol=1e-4)


@pytest.mark.parametrize('M', [4096, 4101, 8193])
def test_layernorm_fwd_bf16_d512_fast_path(M):
    # >= 4096 rows of 512 bf16 columns take the one-16-B-load-per-lane kernel (two rows per wave step: odd row counts exercise the tail)
    ops = _ops()
    D = 512
    x = (_r(M, D, seed=1) * 1.7 + 0.3).to(torch.bfloat16)
    gm, bt = _r(D, seed=2) * 0.1 + 1, _r(D, seed=3)
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), gm.cuda(), bt.cuda())
    xd = x.double()
    ref = torch.nn.functional.layer_norm(xd, (D,), gm.double(), bt.double(), 1e-5)
    _close(y, ref, torch.bfloat16)
    assert float((mean.cpu().double() - xd.mean(1)).abs().max()) < 1e-5
    want_rstd = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-5)
    assert float(((rstd.cpu().double() - want_rstd) / want_rstd).abs().max()) < 1e-5


@pytest.mark.parametrize('M,V', [(301, 327), (2049, 327), (1027, 200), (1500, 450), (1100, 700), (1027, 512), (4099, 512)])
def test_xent_fwd_bwd_and_accuracy(M, V):
    # M >= 1024 and V <= 512 take the register-row kernels (two rows per wave step; odd M exercises the tail), V = 700 the generic ones,
    # V = 512 = the padded output projection
    ops = _ops()
    from oracle import host_ref
    logits = (_r(M, V, seed=1) * 3).requires_grad_(True)
    g = torch.Generator().manual_seed(1)
    tgt = torch.randint(0, V, (M,), generator=g)
    tgt[:40] = V - 1
    ref = torch.nn.functional.cross_entropy(logits, tgt, ignore_index=V - 1)
    ref.backward()
    lse, acc = ops.xent_fwd(logits.detach().cuda(), tgt.cuda(), V - 1)
    loss = acc[0] / acc[1]
    assert abs(float(loss) - float(ref)) < 1e-5
    gs = (1.0 / acc[1]).reshape(1)
    for dt in DT:
        dl = ops.xent_bwd(logits.detach().cuda(), tgt.cuda(), lse, gs, V - 1, dt)
        assert dl.shape[1] == (V + 7) // 8 * 8 and (dl.shape[1] == V or float(dl[:, V:].abs().max()) == 0.0)
        _close(dl[:, :V], logits.grad, dt)
    chord = (torch.rand(M, generator=g) < 0.2).long()
    melody = ((torch.rand(M, generator=g) < 0.3) & (chord == 0)).long()
    c = ops.accuracy_counts(logits.detach().cuda(), tgt.cuda(), chord.cuda(), melody.cuda(), V - 1).cpu().numpy()
    tot, ch, me, ot = host_ref.compute_accuracy(logits.detach().numpy()[None], tgt.numpy()[None], chord.numpy()[None], melody.numpy()[None], V - 1)
    assert abs(c[1] / c[0] - tot) < 1e-12 and abs(c[3] / c[2] - ch) < 1e-12 and abs(c[5] / c[4] - me) < 1e-12
    am = ops.argmax(logits.detach().cuda()).cpu()
    assert torch.equal(am, logits.detach().argmax(-1))


# ------------------------------------------------------------------------------------------- FAVOR+ attention
FAVOR_CASES = [(2, 150, 2, 64, 128), (1, 70, 3, 32, 64), (2, 33, 2, 16, 32), (1, 200, 2, 32, 128), (1, 64, 1, 16, 64)]


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('B,T,H,dh,nf', FAVOR_CASES)
def test_favor_attention_fwd_bwd_vs_oracle(dt, B, T, H, dh, nf):
    ops = _ops()
    from oracle import model_ref
    from oracle.weights import orthogonal_omega
    om = orthogonal_omega(dh, nf, np.random.default_rng(5))
    qkv = _r(B * T, 3 * H * dh, seed=1, dt=dt, scale=0.8)
    q, k, v = [qkv[:, i * H * dh:(i + 1) * H * dh].double().view(B, T, H, dh).requires_grad_(True) for i in range(3)]
    ref = model_ref.causal_linear_attention(q, k, v, om.double(), form='quadratic')
    dout = _r(B, T, H, dh, seed=2, dt=dt)
    ref.backward(dout.double())
    qc = qkv.cuda()
    HD = H * dh
    out, den, S, z = ops.favor_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), B, T, H, want_state=True)
    _close(out.view(B, T, H, dh), ref, dt, mult=3)
    # final state == sum_j phi(k_j) (x) v_j
    Kf = model_ref.favor_features(k.detach(), om.double())
    _close(S, torch.einsum('nlhf,nlhd->nhfd', Kf, v.detach()), dt, mult=3)
    _close(z, Kf.sum(1), dt, mult=3)
    dq, dk, dv = ops.favor_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), out, dout.view(B * T, HD).cuda(), den, B, T, H)
    gscale = max(float(q.grad.abs().max()), float(k.grad.abs().max()), float(v.grad.abs().max()))
    _close(dv.reshape(B, T, H, dh), v.grad, dt, scale=gscale, mult=4)
    _close(dq.reshape(B, T, H, dh), q.grad, dt, scale=gscale, mult=4)
    _close(dk.reshape(B, T, H, dh), k.grad, dt, scale=gscale, mult=4)


# the bf16 / d_head 64 / 128-feature "slice" kernels (emo_favor_fs.hip): against the oracle AND against the generic kernels
@pytest.mark.parametrize('B,T,H,segs', [(2, 256, 2, 1), (1, 32, 1, 1), (3, 96, 3, 1), (1, 1024, 4, 1), (2, 2048, 1, 1), (1, 1024, 2, 4), (2, 2048, 1, 3), (1, 2048, 2, 16)])
def test_favor_slice_kernels_vs_oracle_and_generic(B, T, H, segs, monkeypatch):
    ops = _ops()
    from oracle import model_ref
    from oracle.weights import orthogonal_omega
    dt, dh, nf = torch.bfloat16, 64, 128
    monkeypatch.setenv('EMO_FAVOR_SEGMENTS', str(segs))     # 1: single-segment scan (what B*H >= 256 gets by itself); > 1: segment-parallel, slice main passes
    om = orthogonal_omega(dh, nf, np.random.default_rng(5))
    qkv = _r(B * T, 3 * H * dh, seed=21, dt=dt, scale=0.8)
    q, k, v = [qkv[:, i * H * dh:(i + 1) * H * dh].double().view(B, T, H, dh).requires_grad_(True) for i in range(3)]
    ref = model_ref.causal_linear_attention(q, k, v, om.double(), form='quadratic')
    dout = _r(B, T, H, dh, seed=22, dt=dt)
    ref.backward(dout.double())
    qc, HD = qkv.cuda(), H * dh
    res = {}
    for mode in ('2', '0'):                                 # 2: slice kernels REQUIRED, forward and backward (the call fails if they do not run), 0: generic kernels
        monkeypatch.setenv('EMO_FAVOR_FS', mode)
        out, den, S, z = ops.favor_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), B, T, H, want_state=True)
        _close(out.view(B, T, H, dh), ref, dt, mult=3)
        Kf = model_ref.favor_features(k.detach(), om.double())
        _close(S, torch.einsum('nlhf,nlhd->nhfd', Kf, v.detach()), dt, mult=3)
        _close(z, Kf.sum(1), dt, mult=3)
        dq, dk, dv = ops.favor_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), out, dout.view(B * T, HD).cuda(), den, B, T, H)
        gscale = max(float(q.grad.abs().max()), float(k.grad.abs().max()), float(v.grad.abs().max()))
        _close(dv.reshape(B, T, H, dh), v.grad, dt, scale=gscale, mult=4)
        _close(dq.reshape(B, T, H, dh), q.grad, dt, scale=gscale, mult=4)
        _close(dk.reshape(B, T, H, dh), k.grad, dt, scale=gscale, mult=4)
        res[mode] = [x.float().cpu() for x in (out, den, dq, dk, dv)]
    for a, b_ in zip(res['2'], res['0']):                    # the two implementations round differently but compute the same thing
        assert float((a - b_).abs().max()) <= 3e-2 * max(float(b_.abs().max()), 1e-6)


# r06: the backward from a gradient that is already dN = dout / den (emo_favor_attn_bwd_dn; the out-projection dgrad divides in its epilogue:
# emo_epilogue_t.hdiv) against the fp64 quadratic form and against the plain form of the same kernels
@pytest.mark.parametrize('B,T,H', [(2, 256, 2), (1, 32, 1), (3, 96, 3), (2, 2048, 1), (32, 512, 8)])
def test_favor_backward_from_the_predivided_gradient(B, T, H, monkeypatch):
    ops = _ops()
    from oracle import model_ref
    from oracle.weights import orthogonal_omega
    dt, dh, nf = torch.bfloat16, 64, 128
    if B * H < 256:
        monkeypatch.setenv('EMO_FAVOR_SEGMENTS', '1')        # single-segment scan (what B*H >= 256 gets by itself)
    assert ops.favor_bwd_dn_ok(dt, B, T, H, dh, nf)
    om = orthogonal_omega(dh, nf, np.random.default_rng(5))
    qkv = _r(B * T, 3 * H * dh, seed=21, dt=dt, scale=0.8)
    dout = _r(B, T, H, dh, seed=22, dt=dt)
    qc, HD = qkv.cuda(), H * dh
    out, den = ops.favor_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), B, T, H)
    dq0, dk0, dv0 = [x.float().clone() for x in ops.favor_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), out, dout.view(B * T, HD).cuda(), den, B, T, H)]
    # dN exactly as the GEMM epilogue forms it: fp32 quotient, one rounding to bf16
    dn = (dout.view(B, T, H, dh).cuda().float() / den.permute(0, 2, 1).unsqueeze(-1)).to(dt).view(B * T, HD).contiguous()
    dq1, dk1, dv1 = [x.float() for x in ops.favor_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), out, dn, None, B, T, H, dn=True)]
    for a, b_ in ((dq1, dq0), (dk1, dk0), (dv1, dv0)):       # same arithmetic up to where the bf16 roundings of the operands sit
        assert float((a - b_).abs().max()) <= 2e-2 * float(b_.abs().max())
    if B * T * H <= 4096:                                    # fp64 autograd of the O(T^2) form
        q, k, v = [qkv[:, i * HD:(i + 1) * HD].double().view(B, T, H, dh).requires_grad_(True) for i in range(3)]
        model_ref.causal_linear_attention(q, k, v, om.double(), form='quadratic').backward(dout.double())
        gscale = max(float(q.grad.abs().max()), float(k.grad.abs().max()), float(v.grad.abs().max()))
        _close(dv1.reshape(B, T, H, dh), v.grad, dt, scale=gscale, mult=4)
        _close(dq1.reshape(B, T, H, dh), q.grad, dt, scale=gscale, mult=4)
        _close(dk1.reshape(B, T, H, dh), k.grad, dt, scale=gscale, mult=4)
    # outside the class (segmented scan) the entry point refuses instead of computing something else
    monkeypatch.setenv('EMO_FAVOR_SEGMENTS', '4')
    if T >= 512:
        assert not ops.favor_bwd_dn_ok(dt, B, T, H, dh, nf)
        with pytest.raises(Exception):
            ops.favor_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), out, dn, None, B, T, H, dn=True)


@pytest.mark.parametrize('M,T,N', [(4096, 512, 512), (32768, 2048, 512), (8192, 1024, 256)])
def test_gemm_row_block_divisor_epilogue(M, T, N):
    """emo_epilogue_t.hdiv: C[m][n] /= hdiv[m / T][n / 64][m % T] — the out-projection dgrad leaving dN = dout / den for emo_favor_attn_bwd_dn."""
    ops = _ops()
    A, W = _r(M, 512, seed=1).to(torch.bfloat16).cuda(), _r(N, 512, seed=2, scale=0.1).to(torch.bfloat16).cuda()
    den = (torch.rand(M // T, N // 64, T, generator=torch.Generator().manual_seed(3)) * 4 + 0.25).cuda()
    y = ops.gemm(A, W, hdiv=(den, T))
    assert ops.lib.emo_gemm_last_kernel() == 2
    ref = (A.double() @ W.double().T).view(M // T, T, N // 64, 64) / den.double().permute(0, 2, 1).unsqueeze(-1)
    _close(y.view(M // T, T, N // 64, 64), ref, torch.bfloat16, mult=1.0)
    y0 = ops.gemm(A, W).float().view(M // T, T, N // 64, 64) / den.permute(0, 2, 1).unsqueeze(-1)       # two roundings instead of one
    assert float((y.float().view_as(y0) - y0).abs().max()) <= 1e-2 * float(y0.abs().max())
    with pytest.raises(Exception):                              # not a plain epilogue / not the A-stationary class: refused
        ops.gemm(A, W, hdiv=(den, T), bias=torch.zeros(N, device='cuda'))
    with pytest.raises(Exception):
        ops.gemm(A[:1024], W, hdiv=(den[:1024 // T] if 1024 >= T else den[:1], T))


# segment-parallel scan (B*H < 256 workgroups): automatic segment count, forced counts, ragged last segment, empty tail segments
@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('B,T,H,dh,nf,segs', [(1, 700, 2, 64, 128, None), (1, 512, 2, 32, 64, 4), (2, 330, 1, 16, 32, 3), (1, 1000, 1, 32, 128, 16),
                                              (1, 257, 2, 64, 128, 2)])
def test_favor_attention_segmented_scan(dt, B, T, H, dh, nf, segs, monkeypatch):
    ops = _ops()
    from oracle import model_ref
    from oracle.weights import orthogonal_omega
    from emo_disentanger_amd._lib import lib
    if segs is not None:
        monkeypatch.setenv('EMO_FAVOR_SEGMENTS', str(segs))
    assert lib.emo_favor_attn_workspace_bytes(B, T, H, dh, nf) > 0          # these shapes do take the segmented path
    om = orthogonal_omega(dh, nf, np.random.default_rng(8))
    qkv = _r(B * T, 3 * H * dh, seed=11, dt=dt, scale=0.8)
    q, k, v = [qkv[:, i * H * dh:(i + 1) * H * dh].double().view(B, T, H, dh).requires_grad_(True) for i in range(3)]
    ref = model_ref.causal_linear_attention(q, k, v, om.double(), form='quadratic')
    dout = _r(B, T, H, dh, seed=12, dt=dt)
    ref.backward(dout.double())
    qc = qkv.cuda()
    HD = H * dh
    out, den, S, z = ops.favor_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), B, T, H, want_state=True)
    _close(out.view(B, T, H, dh), ref, dt, mult=3)
    Kf = model_ref.favor_features(k.detach(), om.double())
    _close(S, torch.einsum('nlhf,nlhd->nhfd', Kf, v.detach()), dt, mult=3)
    _close(z, Kf.sum(1), dt, mult=3)
    dq, dk, dv = ops.favor_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), out, dout.view(B * T, HD).cuda(), den, B, T, H)
    gscale = max(float(q.grad.abs().max()), float(k.grad.abs().max()), float(v.grad.abs().max()))
    _close(dv.reshape(B, T, H, dh), v.grad, dt, scale=gscale, mult=4)
    _close(dq.reshape(B, T, H, dh), q.grad, dt, scale=gscale, mult=4)
    _close(dk.reshape(B, T, H, dh), k.grad, dt, scale=gscale, mult=4)
    # and against the single-segment scan of the same kernels (fp32: only the summation order of the carried state differs)
    monkeypatch.setenv('EMO_FAVOR_SEGMENTS', '1')
    assert lib.emo_favor_attn_workspace_bytes(B, T, H, dh, nf) == 0
    out1, den1 = ops.favor_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], om.cuda(), B, T, H)
    _close(out, out1, dt, mult=3)
    _close(den, den1, dt, scale=float(den1.abs().max()), mult=3)


def test_favor_workspace_contract():
    ops = _ops()
    from emo_disentanger_amd._lib import lib, ptr, EmoError, check
    from oracle.weights import orthogonal_omega
    B, T, H, dh, nf = 1, 512, 2, 32, 64
    need = lib.emo_favor_attn_workspace_bytes(B, T, H, dh, nf)
    assert need > 0 and need % (4 * (nf * dh + nf)) == 0
    assert lib.emo_favor_attn_workspace_bytes(64, 2048, 8, 64, 128) == 0     # B*H >= 256: one segment, no scratch
    assert lib.emo_favor_attn_workspace_bytes(1, 100, 1, 64, 128) == 0       # too short to cut
    om = orthogonal_omega(dh, nf, np.random.default_rng(8)).cuda()
    qkv = _r(B * T, 3 * H * dh, seed=11, dt=torch.float32).cuda()
    HD = H * dh
    out = torch.empty(B * T, HD, device='cuda')
    den = torch.empty(B, H, T, device='cuda')
    small = torch.empty(need - 16, dtype=torch.uint8, device='cuda')
    with pytest.raises(EmoError, match='workspace'):
        check(lib.emo_favor_attn_fwd(ptr(qkv[:, :HD]), ptr(qkv[:, HD:2 * HD]), ptr(qkv[:, 2 * HD:]), 3 * HD, ptr(om), ptr(out), HD, ptr(den), None, None,
                                     0, B, T, H, dh, nf, 1e-6, ptr(small), small.numel(), torch.cuda.current_stream().cuda_stream))
    # NULL workspace = single-segment scan, same result as the wrapper's segmented call
    check(lib.emo_favor_attn_fwd(ptr(qkv[:, :HD]), ptr(qkv[:, HD:2 * HD]), ptr(qkv[:, 2 * HD:]), 3 * HD, ptr(om), ptr(out), HD, ptr(den), None, None,
                                 0, B, T, H, dh, nf, 1e-6, None, 0, torch.cuda.current_stream().cuda_stream))
    out2, den2 = ops.favor_attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], om, B, T, H)
    _close(out2, out, torch.float32, mult=3)


@pytest.mark.parametrize('dt', DT)
def test_favor_decode_step_matches_prefill(dt):
    ops = _ops()
    from oracle.weights import orthogonal_omega
    B, T, H, dh, nf = 3, 40, 2, 32, 64
    om = orthogonal_omega(dh, nf, np.random.default_rng(6)).cuda()
    qkv = _r(B * T, 3 * H * dh, seed=3, dt=dt, scale=0.8).cuda()
    HD = H * dh
    full, _ = ops.favor_attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], om, B, T, H)
    # prefill T-1 tokens, then one recurrent step for the last token of each stream
    x3 = qkv.view(B, T, 3 * HD)
    pre = x3[:, :T - 1].reshape(B * (T - 1), 3 * HD)
    _, _, S, z = ops.favor_attn_fwd(pre[:, :HD], pre[:, HD:2 * HD], pre[:, 2 * HD:], om, B, T - 1, H, want_state=True)
    last = x3[:, T - 1].contiguous()
    o = ops.favor_decode_step(last[:, :HD], last[:, HD:2 * HD], last[:, 2 * HD:], om, S, z, H)
    _close(o, full.view(B, T, HD)[:, T - 1], dt, mult=3)


# ------------------------------------------------------------------------------------------- softmax attention
@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('B,T,H,dh', [(2, 150, 2, 64), (1, 70, 3, 32), (2, 33, 2, 16), (1, 256, 1, 64)])
def test_softmax_attention_fwd_bwd(dt, B, T, H, dh):
    ops = _ops()
    HD = H * dh
    qkv = _r(B * T, 3 * HD, seed=4, dt=dt)
    q, k, v = [qkv[:, i * HD:(i + 1) * HD].double().view(B, T, H, dh).permute(0, 2, 1, 3).requires_grad_(True) for i in range(3)]
    w = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    w = w.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float('-inf')).softmax(-1)
    ref = (w @ v).permute(0, 2, 1, 3).reshape(B * T, HD)
    dout = _r(B * T, HD, seed=5, dt=dt)
    ref.backward(dout.double())
    qc = qkv.cuda()
    out, lse = ops.softmax_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], B, T, H)
    _close(out, ref, dt)
    dq, dk, dv = ops.softmax_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], out, dout.cuda(), lse, B, T, H)
    back = lambda g: g.permute(0, 2, 1, 3).reshape(B * T, HD)
    gs = max(float(q.grad.abs().max()), float(k.grad.abs().max()), float(v.grad.abs().max()))
    _close(dq, back(q.grad), dt, scale=gs, mult=3)
    _close(dk, back(k.grad), dt, scale=gs, mult=3)
    _close(dv, back(v.grad), dt, scale=gs, mult=3)
    # decode kernel: last query row against the cache of all T keys
    kc = qc[:, HD:2 * HD].reshape(B, T, HD).contiguous()
    vc = qc[:, 2 * HD:].reshape(B, T, HD).contiguous()
    ql = qc[:, :HD].reshape(B, T, HD)[:, -1].contiguous()
    od = ops.softmax_attn_decode(ql, kc, vc, torch.full((B,), T, dtype=torch.int64, device='cuda'), H)
    _close(od, ref.view(B, T, HD)[:, -1], dt)
    # head-major cache [B, H, T_max, dh] (r06, emo_softmax_attn_decode_layout) with the last key / value row appended by the kernel itself
    T_max = T + 5
    kh = torch.zeros(B, H, T_max, dh, device='cuda', dtype=dt)
    vh = torch.zeros(B, H, T_max, dh, device='cuda', dtype=dt)
    kh[:, :, :T - 1] = kc.view(B, T, H, dh)[:, :T - 1].permute(0, 2, 1, 3)
    vh[:, :, :T - 1] = vc.view(B, T, H, dh)[:, :T - 1].permute(0, 2, 1, 3)
    oh = ops.softmax_attn_decode(ql, kh, vh, torch.full((B,), T - 1, dtype=torch.int64, device='cuda'), H, lens_off=1,
                                 k_new=kc[:, -1].contiguous(), v_new=vc[:, -1].contiguous())
    assert torch.equal(oh, od)
    assert torch.equal(kh[:, :, T - 1], kc.view(B, T, H, dh)[:, -1]) and torch.equal(vh[:, :, T - 1], vc.view(B, T, H, dh)[:, -1])


@pytest.mark.parametrize('B,T,H,p', [(2, 256, 2, 0.0), (1, 128, 1, 0.0), (1, 640, 3, 0.0), (2, 384, 2, 0.2)])
def test_softmax_attention_32x32_kernels_vs_reference_and_generic(B, T, H, p, monkeypatch):
    """bf16 / d_head 64 / T % 128 == 0 runs on the 32 x 32 x 16 kernels (emo_softmax_attn32.hip): against the fp64 reference (p = 0) and against
    the generic kernels, with dropout too (same mask: the element -> hash mapping is shared), forward output, lse, and the backward fed with
    either forward's statistics."""
    ops = _ops()
    dt, dh = torch.bfloat16, 64
    HD = H * dh
    qkv = _r(B * T, 3 * HD, seed=41, dt=dt)
    dout = _r(B * T, HD, seed=42, dt=dt)
    qc = qkv.cuda()
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('EMO_SATTN32', mode)
        out, lse = ops.softmax_attn_fwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], B, T, H, p_drop=p, seed=3, offset=9)
        dq, dk, dv = ops.softmax_attn_bwd(qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:], out, dout.cuda(), lse, B, T, H, p_drop=p, seed=3, offset=9)
        res[mode] = [x.float().cpu() for x in (out, lse, dq, dk, dv)]
    for a, b_ in zip(res['1'], res['0']):
        assert float((a - b_).abs().max()) <= 3e-2 * max(float(b_.abs().max()), 1e-6)
    if p == 0.0:
        q, k, v = [qkv[:, i * HD:(i + 1) * HD].double().view(B, T, H, dh).permute(0, 2, 1, 3) for i in range(3)]
        wts = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        wts = wts.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float('-inf'))
        ref = (wts.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * T, HD)
        _close(res['1'][0], ref, dt)
        _close(res['1'][1].view(B, H, T), torch.logsumexp(wts, -1), dt, mult=0.3)


@pytest.mark.parametrize('B,T,H', [(1, 128, 1), (2, 384, 2), (1, 1024, 3)])
def test_softmax_attention_keep_bits_equal_the_hash(B, T, H, monkeypatch):
    """The forward's keep words (one bit per score at or below the diagonal) drive the dK/dV pass instead of the keyed hash: every output is
    BIT-identical to the calls without the buffer, the bits equal the mask read out of the kernel with V = identity, and a call the 32 x 32
    kernels do not serve reports no buffer instead of half-using one."""
    ops = _ops()
    dt, dh, p = torch.bfloat16, 64, 0.25
    HD = H * dh
    qc = _r(B * T, 3 * HD, seed=51, dt=dt).cuda()
    dout = _r(B * T, HD, seed=52, dt=dt).cuda()
    q, k, v = qc[:, :HD], qc[:, HD:2 * HD], qc[:, 2 * HD:]
    out0, lse0 = ops.softmax_attn_fwd(q, k, v, B, T, H, p_drop=p, seed=3, offset=9)
    g0 = ops.softmax_attn_bwd(q, k, v, out0, dout, lse0, B, T, H, p_drop=p, seed=3, offset=9)
    out1, lse1, keep = ops.softmax_attn_fwd(q, k, v, B, T, H, p_drop=p, seed=3, offset=9, want_keep=True)
    assert keep is not None and keep.numel() * 4 == ops.lib.emo_softmax_attn_keep_bytes(ops.dtype_code(dt), B, T, H, dh, p)
    g1 = ops.softmax_attn_bwd(q, k, v, out1, dout, lse1, B, T, H, p_drop=p, seed=3, offset=9, keep=keep)
    assert torch.equal(out0, out1) and torch.equal(lse0, lse1)
    for a, b_ in zip(g0, g1):
        assert torch.equal(a, b_)
    # decode the words: [bh][key tile][hi][row], bit i + 16 half <-> key 64 kt + 32 half + (i & 3) + 8 (i >> 2) + 4 hi
    w = keep.view(B * H, T // 64, 2, T).cpu().numpy().astype(np.uint32)
    from dropmask import site_multipliers
    m = site_multipliers((B * H * T, T), p, 3, 9).view(B * H, T, T).numpy() != 0
    for bh in range(B * H):
        for kt in range(T // 64):
            for hi in range(2):
                for bit in range(32):
                    i, half = bit & 15, bit >> 4
                    key = 64 * kt + 32 * half + (i & 3) + 8 * (i >> 2) + 4 * hi
                    rows = np.arange(64 * kt, T)                          # every wave that owns one of these rows swept this key tile
                    got = (w[bh, kt, hi, rows] >> bit) & 1
                    assert np.array_equal(got.astype(bool), m[bh, rows, key]), (bh, kt, hi, bit)
    # not served: fp32, d_head 32, T not a multiple of 128, dropout off
    assert ops.lib.emo_softmax_attn_keep_bytes(ops.dtype_code(torch.float32), B, T, H, dh, p) == 0
    assert ops.lib.emo_softmax_attn_keep_bytes(ops.dtype_code(dt), B, T, H, 32, p) == 0
    assert ops.lib.emo_softmax_attn_keep_bytes(ops.dtype_code(dt), B, T + 64, H, dh, p) == 0
    assert ops.lib.emo_softmax_attn_keep_bytes(ops.dtype_code(dt), B, T, H, dh, 0.0) == 0
    # the 32 x 32 dQ pass against the generic one (same mask, same statistics); delta feeds the dK / dV pass, which must not move
    monkeypatch.setenv('EMO_SATTN32_DQ', '0')
    g2 = ops.softmax_attn_bwd(q, k, v, out1, dout, lse1, B, T, H, p_drop=p, seed=3, offset=9, keep=keep)
    monkeypatch.delenv('EMO_SATTN32_DQ')
    assert float((g2[0].float() - g1[0].float()).abs().max()) <= 3e-2 * float(g2[0].float().abs().max())
    for a, b_ in zip(g2[1:], g1[1:]):
        assert float((a.float() - b_.float()).abs().max()) <= 1e-2 * float(b_.float().abs().max())
    monkeypatch.setenv('EMO_SATTN32', '0')
    assert ops.softmax_attn_fwd(q, k, v, B, T, H, p_drop=p, seed=3, offset=9, want_keep=True)[2] is None
    with pytest.raises(RuntimeError):
        ops.softmax_attn_bwd(q, k, v, out0, dout, lse0, B, T, H, p_drop=p, seed=3, offset=9, keep=keep)


def test_softmax_attention_dropout_consistency():
    # with dropout the backward must use the same mask as the forward: finite-difference-free check via linearity in v
    ops = _ops()
    B, T, H, dh = 1, 96, 2, 32
    HD = H * dh
    qkv = _r(B * T, 3 * HD, seed=6).cuda()
    out, lse = ops.softmax_attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], B, T, H, p_drop=0.3, seed=5, offset=7)
    dout = _r(B * T, HD, seed=7).cuda()
    dq, dk, dv = ops.softmax_attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, dout, lse, B, T, H, p_drop=0.3, seed=5, offset=7)
    # out is linear in v: <dout, out(v + e dv_dir)> - <dout, out(v)> = e <dv, dv_dir>
    dirv = _r(B * T, HD, seed=8).cuda()
    q2 = qkv.clone()
    q2[:, 2 * HD:] += dirv
    out2, _ = ops.softmax_attn_fwd(q2[:, :HD], q2[:, HD:2 * HD], q2[:, 2 * HD:], B, T, H, p_drop=0.3, seed=5, offset=7)
    lhs = float(((out2 - out).double() * dout.double()).sum())
    rhs = float((dv.double() * dirv.double()).sum())
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(rhs))
    out3, _ = ops.softmax_attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], B, T, H)
    assert not torch.allclose(out, out3)


# ------------------------------------------------------------------------------------------- sampling / optimizer
def test_nucleus_sampling_matches_host_reference():
    ops = _ops()
    from oracle import host_ref
    rng = np.random.default_rng(3)
    for V, scale, temp, p in ((327, 4.0, 1.1, 0.9), (327, 1.0, 1.2, 0.97), (40, 2.0, 1.1, 0.99), (370, 3.0, 1.0, 0.5)):
        logits = (rng.standard_normal((16, V)) * scale).astype(np.float32)
        u = rng.random(16).astype(np.float32)
        got = ops.sample_nucleus(torch.from_numpy(logits).cuda(), temp, p, torch.from_numpy(u).cuda()).cpu().numpy()
        for r in range(16):
            probs = host_ref.temperature(logits[r], temp)
            cand, pr = host_ref.nucleus_candidates(probs, p)
            cdf = np.cumsum(pr)
            cdf /= cdf[-1]
            exp = cand[np.searchsorted(cdf, u[r], side='right')]
            # ties/rounding at a cdf boundary may move the pick by one rank: accept a neighbour within 1e-5 of the boundary
            if got[r] != exp:
                i = int(np.searchsorted(cdf, u[r], side='right'))
                near = min(abs(cdf[i] - u[r]), abs(cdf[i - 1] - u[r]) if i > 0 else 1.0)
                assert near < 1e-5 and got[r] in cand, (V, r, got[r], exp)


def test_nucleus_step_keeps_loop_state_on_device():
    # emo_sample_nucleus_step: u[step[r], r], token -> out[r] and seq[r, col0 + step[r]], step[r] += 1; identical picks to emo_sample_nucleus
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    n, V, K, col0 = 5, 327, 4, 3
    U = torch.rand(K, n, generator=g).cuda()
    step = torch.tensor([0, 1, 0, 2, 1], dtype=torch.long).cuda()
    seq = torch.full((n, col0 + K + 2), -1, dtype=torch.long).cuda()
    exp_seq, exp_step = seq.clone(), step.clone()
    for it in range(2):
        logits = (torch.randn(n, V, generator=g) * 2.0).cuda()
        tok = ops.sample_nucleus_step(logits, 1.1, 0.9, U, step, seq=seq, col0=col0)
        u_rows = U[exp_step, torch.arange(n).cuda()].contiguous()
        ref = ops.sample_nucleus(logits, 1.1, 0.9, u_rows)
        assert torch.equal(tok, ref)
        exp_seq[torch.arange(n).cuda(), col0 + exp_step] = ref
        exp_step += 1
        assert torch.equal(step, exp_step) and torch.equal(seq, exp_seq)


def test_adam_and_clip_match_torch():
    ops = _ops()
    n = 10007
    p0, g = _r(n, seed=1), _r(n, seed=2)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3)
    pc, m, v = p0.clone().cuda(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    pb = torch.empty(n, device='cuda', dtype=torch.bfloat16)
    for step in range(1, 4):
        gi = g * step
        p.grad = gi.clone()
        total = torch.nn.utils.clip_grad_norm_([p], 0.5)
        opt.step()
        ss, coef = torch.zeros(ops.SUMSQ_FLOATS, device='cuda'), torch.zeros(1, device='cuda')
        ops.sumsq(gi.cuda(), ss)
        first = ss[0].clone()
        ops.sumsq(gi.cuda(), ss)                                   # scratch is reusable without re-zeroing; fixed summation order => same bits
        assert torch.equal(first, ss[0]) and float(ss[1025].view(torch.int32)) == 0
        assert abs(float(ss[0].sqrt()) - float(total)) < 1e-3 * float(total)
        ops.clip_coef(ss, 0.5, 1.0, coef)
        ops.adam_step(pc, gi.cuda(), m, v, pb, 1e-3, 0.9, 0.999, 1e-8, step, coef)
        _close(pc, p.detach(), torch.float32, mult=2)
    _close(pb, p.detach(), torch.bfloat16)


def test_favor_backward_reuses_the_forward_workspace():
    """Segment-parallel scan (B*H < 256): the backward normally recomputes the K-state increments the forward left in its workspace; with the
    forward's private workspace handed over (keep_ws / ws_saved) it skips that pass — same bits."""
    ops = _ops()
    B, T, H, dh, F = 2, 1024, 8, 64, 128
    HD = H * dh
    qkv = (_r(B * T, 3 * HD, seed=3) * 0.8).to(torch.bfloat16).cuda()
    om = _r(dh, F // 2, seed=4).cuda()
    dout = _r(B * T, HD, seed=5).to(torch.bfloat16).cuda()
    q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
    assert ops.lib.emo_favor_attn_workspace_bytes(B, T, H, dh, F) > 0
    out0, den0 = ops.favor_attn_fwd(q, k, v, om, B, T, H)
    g0 = ops.favor_attn_bwd(q, k, v, om, out0, dout, den0, B, T, H)
    out1, den1, ws = ops.favor_attn_fwd(q, k, v, om, B, T, H, keep_ws=True)
    assert ws is not None and torch.equal(out0, out1) and torch.equal(den0, den1)
    g1 = ops.favor_attn_bwd(q, k, v, om, out1, dout, den1, B, T, H, ws_saved=ws)
    for a, b_ in zip(g0, g1):
        assert torch.equal(a, b_)


def test_favor_omega_draw_is_orthogonal_with_row_norm_scaling():
    ops = _ops()
    for L, dh, nf in ((3, 64, 128), (2, 32, 128), (2, 16, 32)):
        cols = nf // 2
        nb = (cols + dh - 1) // dh
        g = torch.Generator().manual_seed(L)
        gauss = torch.randn(L, nb, dh, dh, generator=g)
        om = ops.favor_draw_omega(gauss.cuda(), torch.empty(L, dh, cols, device='cuda')).cpu().double()
        for l in range(L):
            for b in range(nb):
                G = gauss[l, b].double()
                blk = om[l][:, b * dh:(b + 1) * dh]
                w = blk.shape[1]
                norms = G.pow(2).sum(1).sqrt()[:w]
                gram = blk.T @ blk
                assert (gram - torch.diag(norms ** 2)).abs().max() < 1e-3 * float(norms.max() ** 2)   # orthogonal, |col j| = |row j of G|
                qref, _ = torch.linalg.qr(G)                                                       # same columns up to sign
                cosines = ((blk / norms) * qref[:, :w]).sum(0).abs()
                assert (cosines - 1).abs().max() < 1e-4


@pytest.mark.parametrize('a_trans,b_trans', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_large_shapes_all_layouts(a_trans, b_trans):
    """Large shapes through the tiled kernels (ragged M/N edges, split-K wgrad)."""
    ops = _ops()
    dt = torch.bfloat16
    for (M, N, K) in ((32768 + 40, 1024, 128), (512, 768, 8192)):
        A = _r(*((K, M) if a_trans else (M, K)), seed=1, dt=dt)
        Bm = _r(*((K, N) if b_trans else (N, K)), seed=2, dt=dt)
        ref = (A.double().T if a_trans else A.double()) @ (Bm.double() if b_trans else Bm.double().T)
        out = ops.gemm(A.cuda(), Bm.cuda(), a_trans=bool(a_trans), b_trans=bool(b_trans), out_dtype=torch.float32)
        _close(out, ref, dt, mult=0.3)
    # fused epilogue through the same kernel
    M, N, K = 32768, 1024, 128
    A, W = _r(M, K, seed=3, dt=dt), _r(N, K, seed=4, dt=dt, scale=0.2)
    bias, res = _r(N, seed=5), _r(M, N, seed=6, dt=dt)
    out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), act=ops.ACT_RELU, residual=res.cuda())
    _close(out, torch.relu(A.double() @ W.double().T + bias.double()) + res.double(), dt)


@pytest.mark.parametrize('M', [1, 7, 16, 32])
def test_gemm_skinny_decode_shapes(M):
    """M <= 32 (decode: n streams x 1 token) takes the register-only skinny kernel."""
    ops = _ops()
    dt = torch.bfloat16
    for N, K in ((512, 512), (1536, 512), (2048, 512), (512, 2048), (327, 512)):
        A, W = _r(M, K, seed=1, dt=dt), _r(N, K, seed=2, dt=dt, scale=0.1)
        bias, res = _r(N, seed=3), _r(M, N, seed=4, dt=dt)
        out = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), act=ops.ACT_RELU, residual=res.cuda())
        _close(out, torch.relu(A.double() @ W.double().T + bias.double()) + res.double(), dt)
        out32 = ops.gemm(A.cuda(), W.cuda(), bias=bias.cuda(), out_dtype=torch.float32)
        _close(out32, A.double() @ W.double().T + bias.double(), dt, mult=0.3)


# ------------------------------------------------------------------------------------------- A-stationary K = 512 GEMM
# (the shapes below 32768 rows: fewer 128-row panels than CUs — the reference YAML's batch size 4 is 8192 tokens — where the kernel splits the
# column tiles of a panel over blockIdx.y, r05: 4, 8 and 8 column blocks, 33 panels x 8)
@pytest.mark.parametrize('M,N', [(32768, 64), (32768 + 384, 512), (32768 + 1024, 1536), (32768 + 2048, 2048), (8192, 512), (8192, 1536), (8192, 2048),
                                 (4096 + 128, 2048)])
@pytest.mark.parametrize('out_dt', [torch.bfloat16, torch.float32])
def test_gemm_astat_k512_matches_reference_and_the_tiled_kernel(M, N, out_dt, monkeypatch):
    """emo_gemm_astat.hip (A stationary in registers, weights streamed through the LDS ring) against an fp64 product, and BIT-FOR-BIT
    against the 128^2 tiled kernel for every fused epilogue (both accumulate 16 MFMA 16x16x32 steps in k order; dropout masks are
    indexed by m*N+n): bias, ReLU + dropout, dropout + residual, the (aux != 0) mask multiply, aux_out."""
    ops = _ops()
    K = 512
    A, W = _r(M, K, seed=1, dt=torch.bfloat16).cuda(), _r(N, K, seed=2, dt=torch.bfloat16, scale=0.05).cuda()
    bias = _r(N, seed=3).cuda()
    res = _r(M, N, seed=4, dt=out_dt).cuda()
    mask = (_r(M, N, seed=5) > 0.3).to(out_dt).cuda() * 1.5
    ref = A.double() @ W.double().t()
    cases = [dict(), dict(bias=bias), dict(bias=bias, act=ops.ACT_RELU, p_drop=0.1, seed=7, offset=3),
             dict(bias=bias, p_drop=0.1, seed=9, offset=5, residual=res), dict(mul_aux=mask, mul_mode=ops.MUL_NONZERO, mul_scale=1.0 / 0.9),
             dict(bias=bias, act=ops.ACT_RELU, aux_out='alloc')]
    for kw in cases:
        outs = []
        for no_astat in ('', '1'):
            if no_astat:
                monkeypatch.setenv('EMO_GEMM_NO_ASTAT', '1')
            else:
                monkeypatch.delenv('EMO_GEMM_NO_ASTAT', raising=False)
            k2 = dict(kw)
            aux = None
            if k2.get('aux_out') == 'alloc':
                aux = k2['aux_out'] = torch.empty(M, N, device='cuda', dtype=out_dt)
            outs.append((ops.gemm(A, W, out_dtype=out_dt, **k2), aux))
        torch.cuda.synchronize()
        (new, new_aux), (old, old_aux) = outs
        if out_dt == torch.bfloat16 and 'bias' not in kw:
            assert torch.equal(new, old), (sorted(kw), float((new.float() - old.float()).abs().max()))
        elif out_dt == torch.bfloat16:        # the accumulators START from the bias here (one rounding order apart from adding it last): <= 1 bf16 ulp
            d = (new.float() - old.float()).abs()
            assert float((d / old.float().abs().clamp_min(0.25)).max()) <= 2 ** -7 and float((d > 0).float().mean()) < 0.02, sorted(kw)
        else:                                 # fp32 outputs of small shapes take the split-K / register-staged kernels: same math, other summation order
            _close(new, old, torch.float32, mult=10.0)
        if new_aux is not None:
            d = (new_aux.float() - old_aux.float()).abs()
            assert float((d / old_aux.float().abs().clamp_min(0.25)).max()) <= (2 ** -7 if out_dt == torch.bfloat16 else 2e-5)
            assert out_dt == torch.float32 or float((d > 0).float().mean()) < 0.02
        if not kw:
            _close(new, ref, torch.bfloat16 if out_dt == torch.bfloat16 else torch.float32, mult=1.0 if out_dt == torch.bfloat16 else 50.0)
    monkeypatch.delenv('EMO_GEMM_NO_ASTAT', raising=False)
    # a strided A view (the attention output / a column block of a fused projection) and a strided output view
    big = _r(M, 3 * K, seed=6, dt=torch.bfloat16).cuda()
    outb = torch.zeros(M, 2 * N, device='cuda', dtype=out_dt)
    ops.gemm(big[:, K:2 * K], W, out=outb[:, N:], bias=bias)
    _close(outb[:, N:], big[:, K:2 * K].double() @ W.double().t() + bias.double(), torch.bfloat16, mult=1.0)
    assert float(outb[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize('M', [32768 + 640, 8192])
def test_gemm_astat_bitmask_epilogue_equals_the_activation_mask(M):
    """mask_out packs (value after ReLU + dropout != 0) into 1 bit per output; EMO_MUL_BITMASK multiplies by it: the FFN2 dgrad through the
    bit mask equals, bit for bit, the dgrad through the [M, N] activation itself (EMO_MUL_NONZERO); elsewhere the two are refused loudly.
    (8192 rows: the column-split grid, every block writes / reads its own tiles of the mask.)"""
    ops = _ops()
    from emo_disentanger_amd._lib import EmoError
    N, K = 2048, 512
    h, W1, b1 = _r(M, K, seed=1, dt=torch.bfloat16).cuda(), _r(N, K, seed=2, dt=torch.bfloat16, scale=0.05).cuda(), _r(N, seed=3).cuda()
    fmask = torch.zeros(M, N // 8, device='cuda', dtype=torch.uint8)
    f = ops.gemm(h, W1, bias=b1, act=ops.ACT_RELU, p_drop=0.1, seed=5, offset=2, mask_out=fmask)
    f_plain = ops.gemm(h, W1, bias=b1, act=ops.ACT_RELU, p_drop=0.1, seed=5, offset=2)
    assert torch.equal(f, f_plain)                                  # writing the mask does not change the output
    bits = (ops.bitmask_rows(fmask, M, N)[:, :, None] >> torch.arange(8, device='cuda', dtype=torch.uint8)) & 1   # (the mask is stored tile by tile)
    assert torch.equal(bits.reshape(M, N).bool(), f != 0) and 0.3 < float((f != 0).float().mean()) < 0.6
    dy, W2t = _r(M, K, seed=6, dt=torch.bfloat16).cuda(), _r(N, K, seed=7, dt=torch.bfloat16, scale=0.05).cuda()
    a = ops.gemm(dy, W2t, mul_aux=fmask, mul_mode=ops.MUL_BITMASK, mul_scale=1.0 / 0.9)
    b = ops.gemm(dy, W2t, mul_aux=f, mul_mode=ops.MUL_NONZERO, mul_scale=1.0 / 0.9)
    assert torch.equal(a, b)
    with pytest.raises((EmoError, AssertionError)):                 # outside the A-stationary shape class: refused, never ignored
        ops.gemm(h[:100], W1, mask_out=fmask[:100])


@pytest.mark.parametrize('dt', DT)
def test_add_bias2_matches_torch(dt):
    # emo_add_bias2: the biased query copies q + r_w_bias, q + r_r_bias of the stage-1 attention backward in one launch (fp32 sum, rounded once)
    ops = _ops()
    M, D = 777, 512
    qkv = _r(M, 3 * D, seed=1, dt=dt).cuda()
    b1, b2 = _r(8, 64, seed=2).cuda(), _r(8, 64, seed=3).cuda()
    q = qkv[:, D:2 * D]                                              # a column block of the fused projection (row pitch 3 D)
    o1, o2 = ops.add_bias2(q, b1, b2)
    assert torch.equal(o1, (q.float() + b1.view(1, D)).to(dt)) and torch.equal(o2, (q.float() + b2.view(1, D)).to(dt))


@pytest.mark.parametrize('M,N', [(4096, 512), (8192, 2048), (32768 + 256, 1536)])
def test_gemm_layernorm_of_the_a_operand_matches_the_separate_kernels(M, N):
    # lna= (emo_hip.h lna_*, r05): the A-stationary kernel normalises its row panel in registers.  Against the standalone LayerNorm kernel + the same
    # product: the normalised rows agree to one bf16 rounding (another summation order for mean / variance), the statistics to 1e-6, and the
    # product of the kernel's OWN normalised rows is bit-identical to what it returns (the product reads exactly the rows it wrote).
    ops = _ops()
    K = 512
    x = (_r(M, K, seed=1) * 1.7 + 0.3).to(torch.bfloat16).cuda()
    W = _r(N, K, seed=2, scale=0.05).to(torch.bfloat16).cuda()
    bias, g, b = _r(N, seed=3).cuda(), (1.0 + 0.1 * _r(K, seed=4)).cuda(), (0.1 * _r(K, seed=5)).cuda()
    assert ops.gemm_lna_ok(M, N, K, torch.bfloat16)
    for kw in (dict(bias=bias), dict(bias=bias, act=ops.ACT_RELU, p_drop=0.1, seed=7, offset=3)):
        y, h, mean, rstd = ops.gemm(x, W, lna=(g, b, 1e-5), **kw)
        h_ref, m_ref, r_ref = ops.layernorm_fwd(x, g, b)
        xd = x.double()
        mu = xd.mean(1)
        var = ((xd - mu[:, None]) ** 2).mean(1)
        assert float((mean.double() - mu).abs().max()) <= 1e-5 and float((rstd.double() * (var + 1e-5).sqrt() - 1).abs().max()) <= 1e-5
        assert float((mean - m_ref).abs().max()) <= 1e-6 * max(1.0, float(m_ref.abs().max())) and float((rstd / r_ref - 1).abs().max()) <= 1e-5
        d = (h.float() - h_ref.float()).abs()
        assert float((d / h_ref.float().abs().clamp_min(0.25)).max()) <= 2 ** -7 and float((d > 0).float().mean()) < 0.01
        assert torch.equal(y, ops.gemm(h, W, **kw))


def test_gemm_layernorm_of_the_a_operand_rows_with_a_large_mean():
    # r05 advisor finding: the in-product LayerNorm takes the variance as E[x^2] - mean^2 from MFMA row sums (fp32), where the standalone kernel
    # centres first — rows whose |mean| is far above their spread lose variance digits to cancellation.  Bound measured and asserted here for
    # |mean| = 50 x std (E[x^2] ~ 2500 against a variance of ~1: the fp32 sum of 512 exact products carries ~1e-7 relative error, i.e. ~3e-4 of
    # the variance): rstd within 2e-3 of the fp64 value — below the bf16 rounding (2^-8) of the normalised rows it scales; the post-LN / pre-LN
    # streams of the two backbones have |mean| / std < 1 (test above).  Rows with |mean| > ~1000 x std would need the centred form.
    ops = _ops()
    M, N, K = 4096, 512, 512
    x = (_r(M, K, seed=11) + 50.0).to(torch.bfloat16).cuda()
    W = _r(N, K, seed=2, scale=0.05).to(torch.bfloat16).cuda()
    g, b = torch.ones(K, device='cuda'), torch.zeros(K, device='cuda')
    y, h, mean, rstd = ops.gemm(x, W, lna=(g, b, 1e-5))
    xd = x.double()
    mu = xd.mean(1)
    var = ((xd - mu[:, None]) ** 2).mean(1)
    rel = float((rstd.double() * (var + 1e-5).sqrt() - 1).abs().max())
    print('[lna large mean] max relative rstd error %.3g (mean / std = 50)' % rel)
    assert float((mean.double() - mu).abs().max()) <= 1e-4 and rel <= 2e-3
    h_ref, _, r_ref = ops.layernorm_fwd(x, g, b)
    assert float((h.float() - h_ref.float()).abs().max()) <= 2 ** -6 * float(h_ref.float().abs().max())


@pytest.mark.parametrize('M,p', [(32768, 0.0), (32768, 0.1), (65536, 0.1)])
def test_fused_feed_forward_block_equals_the_two_launch_form(M, p):
    # emo_ffn_fwd (r06): LN1 -> FFN1 -> relu -> dropout -> FFN2 -> dropout -> + residual in one launch.  Every output — normalised rows, statistics,
    # hidden activation, 1-bit mask, x2 — against the two emo_gemm launches it replaces (same products in the same order, same dropout hashes:
    # bit-identical), and x2 against fp64 at p = 0.
    ops = _ops()
    D, Hd = 512, 2048
    assert ops.ffn_fwd_ok(M, D, Hd, torch.bfloat16)
    x1 = (_r(M, D, seed=1) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    W1, W2 = _r(Hd, D, seed=2, scale=0.05).to(torch.bfloat16).cuda(), _r(D, Hd, seed=3, scale=0.03).to(torch.bfloat16).cuda()
    b1, b2 = _r(Hd, seed=4, scale=0.1).cuda(), _r(D, seed=5, scale=0.1).cuda()
    g, b = (1.0 + 0.1 * _r(D, seed=6)).cuda(), (0.1 * _r(D, seed=7)).cuda()
    f, h1, mean, rstd, mask, x2 = ops.ffn_fwd(x1, g, b, W1, b1, W2, b2, p_drop=p, seed=11, offset_f=5, offset_y=6)
    mask0 = torch.empty(M, Hd // 8, device='cuda', dtype=torch.uint8)
    f0, h0, m0, r0 = ops.gemm(x1, W1, bias=b1, act=ops.ACT_RELU, p_drop=p, seed=11, offset=5, mask_out=mask0, lna=(g, b, 1e-5))
    x20 = ops.gemm(f0, W2, bias=b2, p_drop=p, seed=11, offset=6, residual=h0)
    assert torch.equal(h1, h0) and torch.equal(mean, m0) and torch.equal(rstd, r0)
    assert torch.equal(f, f0) and torch.equal(mask, mask0)
    assert torch.equal(x2, x20), float((x2.float() - x20.float()).abs().max())
    if p == 0.0:
        rows = slice(M - 2048, M)
        hd = h1[rows].double()
        ref = hd + (torch.relu(hd @ W1.double().T + b1.double()).to(torch.bfloat16).double() @ W2.double().T + b2.double())
        _close(x2[rows], ref, torch.bfloat16, mult=1.0)
    f2, _, _, _, _, x22 = ops.ffn_fwd(x1, g, b, W1, b1, W2, b2, p_drop=p, seed=11, offset_f=5, offset_y=6)
    assert torch.equal(f, f2) and torch.equal(x2, x22)


def test_stream_wait_orders_a_side_stream_launch_behind_the_main_stream():
    # emo_stream_wait (fork / join of the weight-gradient stream without torch Stream contexts) + ops.gemm(stream=raw handle): a product launched
    # on a second stream must see operands that the main stream is still producing when the launch is queued, and the main stream must see its
    # result after the join.  The producer is made slow (a long chain of large copies) so that an unordered launch would read stale data.
    ops = _ops()
    from emo_disentanger_amd import _lib
    M, N, K = 4096, 512, 512
    side = torch.cuda.Stream()
    A_final = _r(M, K, seed=1).to(torch.bfloat16).cuda()
    W = _r(N, K, seed=2, scale=0.1).to(torch.bfloat16).cuda()
    big = torch.zeros(64 << 20, device='cuda', dtype=torch.float32)
    for rep in range(3):
        A = torch.zeros_like(A_final)
        out = torch.full((M, N), -7.0, device='cuda', dtype=torch.bfloat16)
        torch.cuda.synchronize()
        for _ in range(8):
            big.add_(1.0)                                            # ~2 ms of main-stream work queued in front of the producer
        A.copy_(A_final)                                            # the producer (main stream)
        _lib.check(_lib.lib.emo_stream_wait(side.cuda_stream, _lib.stream()))        # fork: side waits for everything queued on main so far
        A.record_stream(side)
        ops.gemm(A, W, out=out, stream=side.cuda_stream)
        _lib.check(_lib.lib.emo_stream_wait(_lib.stream(), side.cuda_stream))        # join: main continues behind the product
        res = out.float().clone()                                   # main stream
        torch.cuda.synchronize()
        _close(res, A_final.double() @ W.double().T, torch.bfloat16, mult=1.0)


def test_transpose_batch_matches_torch():
    # emo_transpose_batch: the transposed weight mirrors of one optimizer step in ONE launch (fast 16-B path and ragged shapes)
    ops = _ops()
    shapes = [(512, 512), (2048, 512), (1536, 512), (512, 2048), (72, 40), (130, 67), (8, 8)]
    srcs = [_r(r, c, seed=i).to(torch.bfloat16).cuda() for i, (r, c) in enumerate(shapes)]
    dsts = [torch.full((c, r), -7.0, device='cuda', dtype=torch.bfloat16) for r, c in shapes]
    rec, tile = [], 0
    for (r, c), a, b in zip(shapes, srcs, dsts):
        tpr = (c + 63) // 64
        rec.append([a.data_ptr(), b.data_ptr(), r, c, tile, tpr])
        tile += ((r + 63) // 64) * tpr
    desc = torch.tensor(rec, dtype=torch.int64, device='cuda')
    ops.transpose_batch(desc, len(rec), tile)
    for a, b in zip(srcs, dsts):
        assert torch.equal(b, a.t().contiguous())
