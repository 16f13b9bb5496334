#!/usr/bin/env python3
"""Per-phase cycle profile of the v6 GEMM (EMO_GEMM_ABLATE=8 stamps s_memtime at every barrier exit; block 0 reports cycles per K-tile
for the 8 barrier-delimited sections, per wave).  Needs a diagnostics build of the library (make CXXFLAGS+=-DG6_PROFILE=1); run with
EMO_GEMM_G6=1 EMO_GEMM_ABLATE=8 [ +1: no tile DMA inside the loop, +4: both waves of a SIMD in the same group ]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emo_disentanger_amd import ops
bf = torch.bfloat16
def rnd(*s): return torch.randn(*s, device='cuda').to(bf)
for name, (M, N, K, at, bt, odt) in {'w_ffn1': (2048, 512, 131072, True, True, torch.float32), 'ffn2_fwd': (131072, 512, 2048, False, False, bf),
                                     'sq4096_NT': (4096, 4096, 4096, False, False, bf), 'sq4096_TN': (4096, 4096, 4096, True, True, bf),
                                     'w_ffn1_bf16out_1split': (2048, 512, 8192, True, True, bf), 'tn_ld512': (512, 512, 65536, True, True, torch.float32)}.items():
    a = rnd(K, M) if at else rnd(M, K)
    b = rnd(K, N) if bt else rnd(N, K)
    c = torch.empty(M, N, device='cuda', dtype=odt)
    dbg = torch.zeros(M, N, device='cuda', dtype=odt)
    for _ in range(3):
        ops.gemm(a, b, a_trans=at, b_trans=bt, out=c, mul_aux=dbg, mul_mode=ops.MUL_NONE)
    torch.cuda.synchronize()
    raw = dbg.view(-1)[:72].float() if odt == torch.float32 else dbg.view(-1)[:144].view(torch.float32)     # (bf16 buffer: the kernel wrote fp32 words)
    v = raw[:64].view(8, 8)
    print('   SIMD id of waves 0..7:', [int(x) for x in raw[64:72].tolist()])
    print(name, 'cycles per K-tile section [R1|M1|R2|M2|R3|M3|R4|M4 as seen by group 0]:')
    for w in (0, 4):
        print('   wave', w, [int(x) for x in v[w].tolist()], 'sum', int(v[w].sum()))
