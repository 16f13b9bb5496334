#!/usr/bin/env python3
"""-DEMO_DIAG build of csrc/emo_gemm_ffn.hip (hipcc ... -DEMO_DIAG -c emo_gemm_ffn.hip -o emo_gemm_ffn.o, relink): where a wave of the fused feed-forward
kernel spends its cycles.  Prints per-wave averages in thousands of shader cycles and the launch time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops, _lib

M, D, Hd = 131072, 512, 2048
x1 = (torch.randn(M, D, device='cuda') * 1.3).to(torch.bfloat16)
W1, W2 = (torch.randn(Hd, D, device='cuda') * 0.05).to(torch.bfloat16), (torch.randn(D, Hd, device='cuda') * 0.03).to(torch.bfloat16)
b1, b2 = torch.randn(Hd, device='cuda') * 0.1, torch.randn(D, device='cuda') * 0.1
g, b = torch.ones(D, device='cuda'), torch.zeros(D, device='cuda')
run = lambda: ops.ffn_fwd(x1, g, b, W1, b1, W2, b2, p_drop=0.1, seed=11, offset_f=5, offset_y=6)
for _ in range(3):
    run()
torch.cuda.synchronize()
fetch = getattr(_lib.lib, 'emo_ffn_diag_fetch', None)
buf = (ctypes.c_ulonglong * 16)()
if fetch is not None:
    fetch(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print('fused feed-forward forward: %.1f us per launch = %.0f TFLOP/s' % (ms * 1e3, 2 * 2.0 * M * D * Hd / ms / 1e9))
if fetch is not None:
    fetch(buf, 0)
    n = max(buf[15], 1)
    names = ['prologue (panel, LN, first stages)', 'phase 1 (FFN1 stages)', 'chunk epilogue', 'phase 2 (FFN2 stages)', '  of which vmcnt waits', '  of which barriers', 'final epilogue']
    tot = buf[14] / n
    for k, nm in enumerate(names):
        print('%-38s %8.1f k cycles per wave  %5.1f %%' % (nm, buf[k] / n / 1e3, 100.0 * buf[k] / n / tot))
    print('%-38s %8.1f k cycles per wave (%d waves)' % ('total', tot / 1e3, n))
