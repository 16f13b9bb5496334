#!/usr/bin/env python3
"""Workload for the vendor-vs-ours GEMM evidence (the vendor library is a YARDSTICK, never on the product path):
the long-reduction shapes of the training step — FFN2 forward / FFN1 dgrad (131072 x 512 x 2048), QKV dgrad (131072 x 512 x 1536) — and the
K = 512 shapes (N = 1536, 2048, 512), each run ITER times through torch.matmul (hipBLASLt) and through emo_gemm on the same operands.  Run it
under `rocprofv3 --kernel-trace --stats` and under the --pmc passes of tools/collect_yardstick.sh; tools/pmc_kernels.py prints one row per
kernel name (calls, average duration, VGPRs / LDS / workgroup size, MFMA-busy, HBM bytes, effective clock)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from emo_disentanger_amd import ops  # noqa: E402

ITER = int(os.environ.get('ITER', 12))
M = 131072
bf = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)


def rnd(*s):
    return (torch.randn(*s, device='cuda', generator=g) * 0.5).to(bf)


for name, (n, k) in {'long_2048': (512, 2048), 'long_1536': (512, 1536), 'k512_n1536': (1536, 512), 'k512_n2048': (2048, 512), 'k512_n512': (512, 512)}.items():
    a, w = rnd(M, k), rnd(n, k)
    o1, o2 = torch.empty(M, n, device='cuda', dtype=bf), torch.empty(M, n, device='cuda', dtype=bf)
    bias = torch.zeros(n, device='cuda')
    torch.cuda.synchronize()
    for _ in range(ITER):
        torch.matmul(a, w.t(), out=o1)
    for _ in range(ITER):
        ops.gemm(a, w, out=o2, bias=bias)
    torch.cuda.synchronize()
    err = float((o1.float() - o2.float()).abs().max())
    print('%-12s M=%d N=%d K=%d  max |vendor - ours| %.3g' % (name, M, n, k, err), flush=True)
