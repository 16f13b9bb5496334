"""Per-phase time stamps of the one-launch decode step (group 0, all 32 members): where does a token step's time go?
wait = phase start -> all input granules gathered (includes the in-order return of the phase's weight loads), work = gathered -> published."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == '__main__':
    from emo_disentanger_amd import inference as inf
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    torch.cuda.set_device(0)
    C = bench.CFG
    torch.manual_seed(0)
    model = MusicPerformer(C['n_token'], C['n_layer'], C['n_head'], C['d_model'], C['d_ff'], C['d_model'], dropout=0.1, favor_feature_dims=C['n_feat'],
                           use_segment_emb=True, n_segment_types=2, compute_dtype='bf16').cuda().eval()
    n = 32
    eng = inf.make_engine(model, n)
    g = torch.Generator().manual_seed(1)
    ptok = torch.randint(0, 326, (n, 64), generator=g).cuda()
    eng.prefill(ptok, torch.ones_like(ptok))
    tok, seg = torch.randint(0, 326, (n,), generator=g).cuda(), torch.ones(n, dtype=torch.long).cuda()
    for _ in range(5):
        eng.step(tok, seg)
    diag = torch.zeros(32, 16, 8, 4, dtype=torch.int64, device='cuda')
    eng.persist['diag'] = diag
    torch.cuda.synchronize()
    eng.step(tok, seg)
    torch.cuda.synchronize()
    eng.check_persistent()
    d = diag.cpu().double()
    L = C['n_layer']
    t0, t1 = d[:, 15, 0, 0], d[:, 15, 0, 2]
    print('XCC id of the 32 members of group 0:', d[:, 15, 0, 1].int().tolist())
    print('kernel start spread %.2f us; member 0 start -> end %.1f us' % ((t0.max() - t0.min()) / 100, (t1[0] - t0.min()) / 100))
    cyc = d[0, 15, 1, 1] - d[0, 15, 1, 0]
    print('member 0: %.0f shader cycles in %.1f us = %.0f MHz effective shader clock' % (cyc, (t1[0] - t0[0]) / 100, cyc / ((t1[0] - t0[0]) / 100)))
    names = {1: 'P1 LN2+QKV', 2: 'P2 attention', 3: 'P3 out-proj', 4: 'P4 LN1+FFN1', 5: 'P5 FFN2'}
    tot = 0.0
    for ph in range(1, 6):
        w = (d[:, :L, ph, 1] - d[:, :L, ph, 0]) / 100
        k = (d[:, :L, ph, 2] - d[:, :L, ph, 1]) / 100
        sp = d[:, 1:L, ph, 3] if ph == 1 else d[:, :L, ph, 3]
        print('%-14s wait %.2f us (max over members %.2f)   work %.2f us (max %.2f)   poll passes %.1f' % (names[ph], w.mean(), w.max(1).values.mean() if False else w.mean(1).max(), k.mean(), k.mean(1).max(), sp.mean()))
        tot += (w.mean() + k.mean()) * L
    print('sum over phases x layers: %.1f us' % tot)
    print('first poll of a gather (us): E5 %.2f  E4 %.2f  E3 %.2f' % tuple((d[:, :L, 6, k].mean() / 100) for k in range(3)))
    print('barrier exit -> partial products written (us): P1 %.2f  P4 %.2f  P3 %.2f  P5 %.2f' % tuple((d[:, :L, 7, k].mean() / 100) for k in range(4)))
    # the dependency chain of member 0, layer 5
    for ph in range(1, 6):
        r = d[0, 5, ph]
        print('  member 0 layer 5 %-14s start %.2f gathered +%.2f published +%.2f' % (names[ph], (r[0] - t0.min()) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100))
