#!/usr/bin/env python3
"""Forced-mode check of the 256^2 four-phase GEMM (EMO_GEMM_G6=1, read once per process — hence a script that
tests/test_gpu_kernels.py::test_gemm_v6_forced_all_layouts runs in a child process): every operand layout, bf16 and fp32 outputs, the fused
epilogue, short and odd K-tile counts, split-K with a workspace — against fp64 torch, plus run-to-run bitwise stability (race screen)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
assert os.environ.get('EMO_GEMM_G6') == '1', 'run with EMO_GEMM_G6=1'
from emo_disentanger_amd import ops

bf = torch.bfloat16
g = torch.Generator().manual_seed(5)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(bf).cuda()
fails = []
for (M, N, K) in [(256, 256, 128), (512, 256, 192), (256, 768, 64 * 7), (1024, 512, 2048)]:
    for at in (False, True):
        for bt in (False, True):
            a = rnd(K, M) if at else rnd(M, K)
            b = rnd(K, N) if bt else rnd(N, K)
            A = (a.double().t() if at else a.double())
            B = (b.double() if bt else b.double().t())
            ref = A @ B
            for odt in (bf, torch.float32):
                outs = [ops.gemm(a, b, a_trans=at, b_trans=bt, out_dtype=odt) for _ in range(3)]
                tol = (2e-2 if odt == bf else 2e-3) * float(ref.abs().max())
                err = float((outs[0].double() - ref).abs().max())
                if err > tol or not all(torch.equal(outs[0], o) for o in outs[1:]):
                    fails.append(('plain', M, N, K, at, bt, str(odt), err, tol))
            if not at:      # fused forward epilogue: bias + relu + residual, pre-activation copy
                bias, res = torch.randn(N, generator=g).cuda(), rnd(M, N)
                aux = torch.empty(M, N, device='cuda', dtype=bf)
                y = ops.gemm(a, b, b_trans=bt, bias=bias, act=ops.ACT_RELU, residual=res, aux_out=aux)
                pre = ref + bias.double()
                want = torch.relu(pre) + res.double()
                e1 = float((y.double() - want).abs().max()); e2 = float((aux.double() - pre).abs().max())
                tol = 2e-2 * float(pre.abs().max())
                if e1 > tol or e2 > tol:
                    fails.append(('epilogue', M, N, K, at, bt, e1, e2, tol))
# split-K wgrad with bias gradients (both flavours) on the default heuristic shape class
for (M, N, K) in [(1024, 768, 32768), (512, 2048, 16384)]:
    dy, x = rnd(K, M, scale=0.5), rnd(K, N, scale=0.5)
    ref = dy.double().t() @ x.double()
    for which in ('a', 'b'):
        rs = torch.zeros(M if which == 'a' else N, device='cuda')
        out = torch.full((M, N), 7.0, device='cuda')
        kw = {'a_rowsum': rs} if which == 'a' else {'b_rowsum': rs}
        ops.gemm(dy, x, a_trans=True, b_trans=True, out=out, accumulate=True, **kw)
        want_rs = (dy if which == 'a' else x).double().sum(0)
        e = float((out.double() - 7.0 - ref).abs().max() / ref.abs().max())
        er = float((rs.double() - want_rs).abs().max() / want_rs.abs().max())
        if e > 2e-4 or er > 2e-4:
            fails.append(('wgrad', M, N, K, which, e, er))
print('FAILS', fails) if fails else print('v6 ok')
sys.exit(1 if fails else 0)
