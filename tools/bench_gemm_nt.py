import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
def t(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
M, d, f = 131072, 512, 2048
bf = torch.bfloat16
rnd = lambda *s: torch.randn(*s, device='cuda').to(bf)
res = {}
for name, (m, n, k) in {'qkv': (M, 3*d, d), 'ffn1': (M, f, d), 'ffn2': (M, d, f)}.items():
    a, w, o = rnd(m, k), rnd(n, k), torch.empty(m, n, device='cuda', dtype=bf)
    ms = t(lambda: ops.gemm(a, w, out=o)); res[name] = (round(ms, 4), round(2*m*n*k/ms/1e9, 1))
print(os.environ.get('EMO_GEMM_ABLATE', '0'), json.dumps(res))
