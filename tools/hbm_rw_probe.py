#!/usr/bin/env python3
"""What the HBM of this box gives a streaming kernel: fill (write only), copy (read + write), sum (read only) of a 2-GB bf16 tensor through
torch's own elementwise kernels (yardstick only; not on the product path).  The K = 512 products of the layer write 0.4-0.54 GB per launch:
their floor is the WRITE rate, not the 8 TB/s read headline."""
import torch
n = 1 << 30
x = torch.empty(n, device='cuda', dtype=torch.bfloat16)
y = torch.empty(n, device='cuda', dtype=torch.bfloat16)
def t(f, it=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
b = n * 2
tf = t(lambda: x.fill_(1.0)); print('fill  (write %.2f GB): %.1f us  = %.2f TB/s written' % (b / 1e9, tf * 1e6, b / tf / 1e12))
tc = t(lambda: y.copy_(x)); print('copy  (read + write)  : %.1f us  = %.2f TB/s total (%.2f each way)' % (tc * 1e6, 2 * b / tc / 1e12, b / tc / 1e12))
xf = x.view(torch.float32)
ts = t(lambda: xf.sum()); print('sum   (read only, fp32 view): %.1f us  = %.2f TB/s read' % (ts * 1e6, b / ts / 1e12))
xs = x[: n // 4]
tf2 = t(lambda: xs.fill_(2.0)); print('fill 0.5 GB           : %.1f us  = %.2f TB/s written' % (tf2 * 1e6, b / 4 / tf2 / 1e12))
