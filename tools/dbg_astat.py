import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from emo_disentanger_amd import ops
torch.manual_seed(0)
M, K = 33792, 512
for N in (512, 1536):
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16); W = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    ref = (A.float() @ W.float().t() + bias) + res.float()
    for p in (0.0, 0.1):
        outs = {}
        for mode in ('astat', 'tiled'):
            if mode == 'tiled': os.environ['EMO_GEMM_NO_ASTAT'] = '1'
            else: os.environ.pop('EMO_GEMM_NO_ASTAT', None)
            outs[mode] = ops.gemm(A, W, bias=bias, p_drop=p, seed=9, offset=5, residual=res).float()
        torch.cuda.synchronize()
        for mode, o in outs.items():
            bad = ~torch.isfinite(o) | (o.abs() > 1e4)
            print('N', N, 'p', p, mode, 'finite-bad', int(bad.sum()), 'err vs ref (p=0 only)', float((o - ref).abs().max()) if p == 0 else '-')
        d = (outs['astat'] - outs['tiled']).abs()
        idx = (d > 0.5).nonzero()
        print('   mismatches > 0.5:', idx.shape[0], idx[:6].tolist(), [(float(outs['astat'][i, j]), float(outs['tiled'][i, j])) for i, j in idx[:4].tolist()])
