#!/usr/bin/env python3
"""Sustained (seconds-long, power-capped) rate of the training step's GEMM classes next to their first-burst rate, with package power and
shader clock sampled from the amdgpu hwmon files.  Inside the training step the package sits at its 1400 W limit during the GEMM phases and
the shader clock is pulled down to ~1.65 GHz: a 30-ms microbenchmark from a cool start overstates what a kernel holds, and two variants of a
kernel with the same instruction mix tie under the cap whatever their latency hiding.  `torch` rows = hipBLASLt through torch.matmul, the
vendor yardstick (never on the product path).
usage: python tools/sustained_gemm.py [seconds per case]"""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emo_disentanger_amd import ops  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
M = 131072
bf = torch.bfloat16


class Sampler(threading.Thread):
    """package power (W) and shader clock (MHz) from sysfs, ~50 Hz"""

    def __init__(self):
        super().__init__(daemon=True)
        self.hw = []                                              # every amdgpu hwmon the container can see; the busy card is the one drawing power
        for h in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
            pw = next((p for p in (h + '/power1_average', h + '/power1_input') if os.path.exists(p)), None)
            if pw and os.path.exists(h + '/freq1_input'):
                self.hw.append((pw, h + '/freq1_input'))
        self.rows, self.on = [], False
        self.alive = True

    def run(self):
        while self.alive:
            if self.on:
                best = None
                for pw, ck in self.hw:
                    try:
                        v = (int(open(pw).read()) / 1e6, int(open(ck).read()) / 1e6)
                    except Exception:
                        continue
                    if best is None or v[0] > best[0]:
                        best = v
                if best:
                    self.rows.append(best)
            time.sleep(0.02)

    def window(self):
        r = self.rows[len(self.rows) // 2:]                       # second half of the window: after the clock has settled
        self.rows = []
        if not r:
            return float('nan'), float('nan')
        return sum(x[0] for x in r) / len(r), sum(x[1] for x in r) / len(r)


smp = Sampler()
smp.start()


def run_case(name, fn, flops):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    time.sleep(1.5)                                               # cool start
    rates = []
    smp.on = True
    t_end = time.time() + secs
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        rates.append(flops * 50 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    smp.on = False
    pw, ck = smp.window()
    tail = rates[len(rates) // 2:]
    print('%-44s first burst %5.0f  sustained %5.0f TFLOP/s  (%4.0f W, %4.0f MHz)' % (name, rates[0], sum(tail) / len(tail), pw, ck), flush=True)


NB = 3


def bufs(r, c):
    return [torch.randn(r, c, device='cuda').to(bf) for _ in range(NB)]


def main():
    want = os.environ.get('CASES', '').split(',') if os.environ.get('CASES') else None
    cases = []
    # --- forward NT (nn.Linear): C[M,N] = A[M,K] W[N,K]^T
    for name, N, K in (('NT QKV fwd K=512 (astat)', 1536, 512), ('NT FFN1 fwd K=512 (astat)', 2048, 512), ('NT FFN2 fwd K=2048 (tiled)', 512, 2048)):
        A, W, O = bufs(M, K), (torch.randn(N, K, device='cuda') * 0.05).to(bf), [torch.empty(M, N, device='cuda', dtype=bf) for _ in range(NB)]
        bias = torch.randn(N, device='cuda')
        cases.append((name, lambda i, A=A, W=W, O=O, bias=bias: ops.gemm(A[i % NB], W, out=O[i % NB], bias=bias), 2.0 * M * N * K))
        cases.append((name.split(' (')[0] + ' (torch)', lambda i, A=A, W=W, O=O: torch.matmul(A[i % NB], W.t(), out=O[i % NB]), 2.0 * M * N * K))
    # --- dgrad NN: dX[M,K] = dY[M,N] W[N,K]
    for name, N, K in (('NN FFN1 dgrad (dY 2048 -> 512)', 2048, 512), ('NN QKV dgrad (dY 1536 -> 512)', 1536, 512)):
        dY, W, O = bufs(M, N), (torch.randn(N, K, device='cuda') * 0.05).to(bf), [torch.empty(M, K, device='cuda', dtype=bf) for _ in range(NB)]
        cases.append((name + ' (tiled)', lambda i, dY=dY, W=W, O=O: ops.gemm(dY[i % NB], W, b_trans=True, out=O[i % NB]), 2.0 * M * N * K))
        cases.append((name + ' (torch)', lambda i, dY=dY, W=W, O=O: torch.matmul(dY[i % NB], W, out=O[i % NB]), 2.0 * M * N * K))
    # --- wgrad TN: dW[N,K] = dY[M,N]^T X[M,K]
    for name, N, K in (('TN FFN1 wgrad (2048 x 512)', 2048, 512), ('TN FFN2 wgrad (512 x 2048)', 512, 2048)):
        dY, X = bufs(M, N), bufs(M, K)
        dW = torch.zeros(N, K, device='cuda')
        dWb = torch.zeros(N, K, device='cuda', dtype=bf)
        cases.append((name + ' (ours, fp32 out)', lambda i, dY=dY, X=X, dW=dW: ops.gemm(dY[i % NB], X[i % NB], a_trans=True, b_trans=True, out=dW, accumulate=True), 2.0 * M * N * K))
        cases.append((name + ' (torch, bf16 out)', lambda i, dY=dY, X=X, dWb=dWb: torch.matmul(dY[i % NB].t(), X[i % NB], out=dWb), 2.0 * M * N * K))
    # --- a compute-dense yardstick: 8192^3 through the vendor library
    a8, b8, c8 = torch.randn(8192, 8192, device='cuda').to(bf), torch.randn(8192, 8192, device='cuda').to(bf), torch.empty(8192, 8192, device='cuda', dtype=bf)
    cases.append(('8192^3 NT (torch)', lambda i: torch.matmul(a8, b8.t(), out=c8), 2.0 * 8192 ** 3))
    cases.append(('8192^3 NT (ours, tiled)', lambda i: ops.gemm(a8, b8, out=c8), 2.0 * 8192 ** 3))
    for name, fn, fl in cases:
        if want and not any(w in name for w in want):
            continue
        run_case(name, fn, fl)
    smp.alive = False


if __name__ == '__main__':
    main()
