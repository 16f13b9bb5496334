import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from emo_disentanger_amd import ops
M, d, f = 131072, 512, 2048
bf = torch.bfloat16
res = []
for name, n, k, rs in (('qkv', 3 * d, d, True), ('out', d, d, True), ('ffn1', f, d, True), ('ffn2', d, f, False)):
    xs = [torch.randn(M, k, device='cuda').to(bf) for _ in range(2)]
    dys = [torch.randn(M, n, device='cuda').to(bf) for _ in range(2)]
    dw = torch.zeros(n, k, device='cuda'); db = torch.zeros(n, device='cuda')
    def run(i):
        ops.gemm(dys[i % 2], xs[i % 2], a_trans=True, b_trans=True, out=dw, accumulate=True, a_rowsum=db if rs else None)
    for i in range(3): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): run(i)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res.append('%s %.1f us %.0f TF' % (name, ms * 1e3, 2.0 * M * n * k / ms / 1e9))
print(' | '.join(res))
