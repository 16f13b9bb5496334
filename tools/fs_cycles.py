#!/usr/bin/env python3
"""-DEMO_DIAG build only (make -C emo-disentanger_amd/csrc EXTRA=-DEMO_DIAG): where a wave of the FAVOR+ slice forward kernel spends its cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
B, T, H, dh, F = int(os.environ.get('BS', 64)), 2048, 8, 64, 128
HD = H * dh
qkv = (torch.randn(B * T, 3 * HD, device='cuda') * 0.8).to(torch.bfloat16)
om = torch.randn(dh, F // 2, device='cuda')
q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
import emo_disentanger_amd.ops as O
# call through ops with want_state: the DIAG kernel writes counters into the state buffer
out, den, S, z = ops.favor_attn_fwd(q, k, v, om, B, T, H, want_state=True)
S.zero_()
from emo_disentanger_amd._lib import lib, ptr, check, dtype_code
from emo_disentanger_amd.ops import stream
check(lib.emo_favor_attn_fwd(ptr(q), ptr(k), ptr(v), 3 * HD, ptr(om), ptr(out), HD, ptr(den), ptr(S), ptr(z), dtype_code(q.dtype), B, T, H, dh, F, 1e-6, None, 0, stream()))
torch.cuda.synchronize()
d = S.view(-1)[:20].view(torch.int64).tolist()
waves, nch = d[9], T // 32
names = ['A features', 'vmcnt wait', 'stores+barrier', 'B1c A^T mfma', 'B2 num', 'B3 S update', 'B1a dma issue', 'B1b lds reads']
d[6], d[8] = d[8], d[6]
d = d[:6] + [d[8], d[7]] + [d[6]]

tot = d[8]
print('waves %d, total %.0f cycles/wave, %.0f per chunk' % (waves, tot / waves, tot / waves / nch))
for n, x in zip(names, d[:8]):
    print('  %-16s %8.0f cycles/chunk  %5.1f %%' % (n, x / waves / nch, 100.0 * x / tot))
