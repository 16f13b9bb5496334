#!/usr/bin/env python3
"""-DEMO_DIAG build only: per-phase cycles of the FAVOR+ slice backward kernels (emo_fs_diag device counters)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
from emo_disentanger_amd._lib import lib
B, T, H, dh, F = int(os.environ.get('BS', 64)), 2048, 8, 64, 128
HD = H * dh
qkv = (torch.randn(B * T, 3 * HD, device='cuda') * 0.8).to(torch.bfloat16)
om = torch.randn(dh, F // 2, device='cuda')
dout = torch.randn(B * T, HD, device='cuda').to(torch.bfloat16)
q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
out, den = ops.favor_attn_fwd(q, k, v, om, B, T, H)
ops.favor_attn_bwd(q, k, v, om, out, dout, den, B, T, H)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib.emo_diag_fetch(buf, 64, 1)
ops.favor_attn_bwd(q, k, v, om, out, dout, den, B, T, H)
torch.cuda.synchronize()
lib.emo_diag_fetch(buf, 64, 1)
d = list(buf)
nch = T // 32
for name, base, names in (('dq', 16, ['A features', 'dN/dD', 'P^T', 'kfT+dPhi+Jac', 'state', 'vmcnt wait', 'stores+barrier', 'dma issue', 'C dq cols']),
                          ('dkv', 32, ['A1 features', 'dN/dD', 'P', 'barrier X', 'A tiles', 'dPhi+Jac', 'dV', 'states', 'vmcnt wait', 'stores+barrier Y', 'dma issue', 'C dk cols'])):
    waves, tot = d[base + 15], d[base + 14]
    print('%s: waves %d, %.0f cycles/wave, %.0f per chunk' % (name, waves, tot / waves, tot / waves / nch))
    for i, nm in enumerate(names):
        print('   %-18s %7.0f cycles/chunk %5.1f %%' % (nm, d[base + i] / waves / nch, 100.0 * d[base + i] / tot))
