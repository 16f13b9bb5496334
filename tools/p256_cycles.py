#!/usr/bin/env python3
"""-DEMO_DIAG build of csrc/emo_gemm_p256.hip only (hipcc ... -DEMO_DIAG -c emo_gemm_p256.hip, relink): where a wave of the persistent 256 x 256
kernel spends its cycles (s_memtime): prologue (first slabs), slab syncs (counted vmcnt + s_barrier), epilogues, the rest = k-steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
os.environ['EMO_GEMM_ABLATE'] = '8'
os.environ['EMO_GEMM_P256'] = '1'
M = 131072
for name, N, K, kw in (('FFN2 fwd', 512, 2048, dict(p_drop=0.1, seed=1, offset=2, res=True)), ('QKV dgrad', 512, 1536, dict(res=True)), ('plain K=2048', 512, 2048, {}),
                       ('plain K=4096', 512, 4096, {}), ('plain K=512 N=2048', 2048, 512, {}), ('plain K=2048 A from L2 (row stride 0)', 512, 2048, dict(l2=True))):
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    if kw.pop('l2', False):
        a = a[:1].expand(M, K)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda').to(torch.bfloat16) if kw.pop('res', False) else None
    o = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    diag = torch.zeros(8, device='cuda', dtype=torch.int64)
    for _ in range(3):
        ops.gemm(a, w, out=o, bias=b, residual=r, **kw)
    diag.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(a, w, out=o, bias=b, residual=r, rln=(None, diag.view(torch.float32), None, None), **kw)
    e1.record()
    torch.cuda.synchronize()
    d = diag.tolist()
    n, tiles = d[4], d[5] / d[4]
    tot, pro, sync, epi = d[1] / n, d[0] / n, d[2] / n, d[3] / n
    slabs = tiles * K / 32
    ks = tot - pro - sync - epi
    print('%-20s %.1f us; waves %d, %.1f tiles each; per wave: total %.0f cyc = prologue %.0f + syncs %.0f (%.0f per slab) + epilogues %.0f (%.0f per tile) + k-steps %.0f (%.0f per slab; 1024 = MFMA-bound)  => k-step share %.1f %%, MFMA-bound fraction of the whole %.3f'
          % (name, e0.elapsed_time(e1) * 1e3, n, tiles, tot, pro, sync, sync / slabs, epi, epi / tiles, ks, ks / slabs, 100 * ks / tot, slabs * 1024 / tot))
