import os, sys
sys.path.insert(0, '/root/repo')
import torch
from emo_disentanger_amd import ops
M, K, N = 131072, 512, 2048
a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
b = torch.randn(N, device='cuda')
o = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
mask = torch.empty(M, N // 8, device='cuda', dtype=torch.uint8)
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for name, kw in (('FFN1+mask', dict(act=ops.ACT_RELU, p_drop=0.1, seed=1, offset=2, mask_out=mask)), ('plain', {})):
    print(name, 'ablate %s: %.0f us' % (os.environ.get('EMO_GEMM_ABLATE', '0'), t(lambda: ops.gemm(a, w, out=o, bias=b, **kw))))
