#!/usr/bin/env python3
"""Run-to-run stability of the attention kernels and of a small training step: the FAVOR+ slice kernels (forward, backward plain / from dN, single-segment
and segmented) and the A-stationary hdiv product must be BIT-identical across repeats on identical inputs (a missing wait on an inline-asm LDS read
would show up here as a rare mismatch); the model step reports which parameter gradients differ between repeats and by how much."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops

N = int(os.environ.get('REPS', 200))
torch.manual_seed(0)
bad = 0
for (B, T, H, segs) in ((64, 512, 8, None), (2, 512, 8, None), (4, 2048, 8, None), (1, 1024, 2, '4')):
    if segs:
        os.environ['EMO_FAVOR_SEGMENTS'] = segs
    else:
        os.environ.pop('EMO_FAVOR_SEGMENTS', None)
    dh, F = 64, 128
    HD = H * dh
    qkv = (torch.randn(B * T, 3 * HD, device='cuda') * 0.8).to(torch.bfloat16)
    om = torch.randn(dh, F // 2, device='cuda')
    dout = torch.randn(B * T, HD, device='cuda').to(torch.bfloat16)
    q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
    out0, den0 = ops.favor_attn_fwd(q, k, v, om, B, T, H)
    g0 = [x.clone() for x in ops.favor_attn_bwd(q, k, v, om, out0, dout, den0, B, T, H)]
    dn_ok = ops.favor_bwd_dn_ok(torch.bfloat16, B, T, H, dh, F)
    if dn_ok:
        dn = (dout.view(B, T, H, dh).float() / den0.permute(0, 2, 1).unsqueeze(-1)).to(torch.bfloat16).view(B * T, HD).contiguous()
        h0 = [x.clone() for x in ops.favor_attn_bwd(q, k, v, om, out0, dn, None, B, T, H, dn=True)]
    mism = [0, 0, 0]
    for it in range(N):
        out, den = ops.favor_attn_fwd(q, k, v, om, B, T, H)
        mism[0] += int(not (torch.equal(out, out0) and torch.equal(den, den0)))
        g = ops.favor_attn_bwd(q, k, v, om, out0, dout, den0, B, T, H)
        mism[1] += int(not all(torch.equal(a, b) for a, b in zip(g, g0)))
        if dn_ok:
            h = ops.favor_attn_bwd(q, k, v, om, out0, dn, None, B, T, H, dn=True)
            mism[2] += int(not all(torch.equal(a, b) for a, b in zip(h, h0)))
    print('favor B=%d T=%d H=%d segs=%s: mismatching repeats fwd %d  bwd %d  bwd(dN) %d of %d' % (B, T, H, segs, mism[0], mism[1], mism[2], N))
    bad += sum(mism)
os.environ.pop('EMO_FAVOR_SEGMENTS', None)
# hdiv product
M, T_, Nn = 32768, 2048, 512
A = torch.randn(M, 512, device='cuda').to(torch.bfloat16)
W = (torch.randn(Nn, 512, device='cuda') * 0.1).to(torch.bfloat16)
den = torch.rand(M // T_, Nn // 64, T_, device='cuda') * 4 + 0.25
y0 = ops.gemm(A, W, hdiv=(den, T_))
mm = sum(int(not torch.equal(ops.gemm(A, W, hdiv=(den, T_)), y0)) for _ in range(N))
print('hdiv product: mismatching repeats %d of %d' % (mm, N))
bad += mm
# the small training step of tests/test_gpu_model.py::test_embedding_gradient_as_a_product_equals_the_scatter_kernel
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_model as tm
from oracle.weights import synthetic_batch
c = tm.PERF_CASES[1]
b = synthetic_batch(c['V'], 2, 512, seed=33)
x, seg, tgt = b['dec_input'].cuda(), b['track_mask'].cuda(), b['dec_target'].cuda()
ref, worst = None, {}
for it in range(int(os.environ.get('MODEL_REPS', 40))):
    torch.manual_seed(5)
    m, _ = tm._performer(c, 'bf16', dropout=0.0)
    m.train(); m.zero_grad()
    m.compute_loss(m(x, seg_inp=seg), tgt)['total_loss'].backward()
    g = {k_: p.grad.detach().float().clone() for k_, p in m.named_parameters()}
    if ref is None:
        ref = g
        continue
    for k_ in g:
        d = float((g[k_] - ref[k_]).abs().max()) / max(float(ref[k_].abs().max()), 1e-30)
        if d > worst.get(k_, 0.0):
            worst[k_] = d
top = sorted(worst.items(), key=lambda kv: -kv[1])[:8]
print('model step, largest relative run-to-run gradient differences:', [(k_, '%.2g' % v_) for k_, v_ in top if v_ > 0])
print('TOTAL bitwise mismatches in the deterministic kernels:', bad)
