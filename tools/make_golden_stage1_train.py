#!/usr/bin/env python3
"""tests/golden/txl_trainloop.json: a trace of the REAL stage-1 train() loop (stage1_compose/train.py:19-115) on a tiny imported model with
synthetic batches on CPU (.cuda() neutralised).  Runs only in the build container."""
import json
import os
import pickle
import sys
import tempfile
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/stage1_compose'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def make_batches(c):
    rng = np.random.default_rng(c['batch_seed'])
    out = []
    for i in range(c['n_batches']):
        x = rng.integers(0, c['V'] - 1, size=(c['B'], c['T']), dtype=np.int64)
        tgt = np.concatenate([x[:, 1:], np.full((c['B'], 1), c['V'] - 2, dtype=np.int64)], 1)
        tgt[:, c['T'] - 5:] = c['V'] - 1
        chord = (rng.random((c['B'], c['T'])) < 0.2).astype(np.int64)
        melody = ((rng.random((c['B'], c['T'])) < 0.3) & (chord == 0)).astype(np.int64)
        chord[:, c['T'] - 5:] = 0
        melody[:, c['T'] - 5:] = 0
        out.append({'id': torch.arange(c['B']), 'n_seg': [1] * c['B'], 'dec_inp_0': torch.from_numpy(x), 'dec_tgt_0': torch.from_numpy(tgt),
                    'dec_seg_len_0': torch.full((c['B'],), c['T'], dtype=torch.long), 'inp_chord_0': torch.from_numpy(chord),
                    'inp_melody_0': torch.from_numpy(melody)})
    return out


def main():
    from oracle.txl_ref import make_state_dict_txl
    sys.modules['pickle5'] = pickle
    os.chdir(REF)
    sys.path.insert(0, REF)
    import train as rt
    from model.plain_transformer import PlainTransformer
    torch.Tensor.cuda = lambda self, *a, **k: self
    c = dict(V=40, L=2, H=4, d=64, dff=128, T=48, B=3, n_batches=7, batch_seed=77, seed=41, scale=6.0, warmup=3, max_lr=2e-3, eta_min=1e-4, T_max=10, log_interval=2)
    sd = make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    model = PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], 0, c['T'], dec_dropout=0.0, pre_lnorm=True)
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), lr=c['max_lr'])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=c['T_max'], eta_min=c['eta_min'])
    ck = tempfile.mkdtemp()
    rt.train_steps, rt.warmup_steps, rt.max_lr, rt.log_interval, rt.ckpt_dir, rt.log_file, rt.init_time = 0, c['warmup'], c['max_lr'], c['log_interval'], ck, 'log.txt', time.time()
    lrs, losses = [], []
    ostep, ocl = opt.step, model.compute_loss
    opt.step = lambda *a, **k: (lrs.append(opt.param_groups[0]['lr']), ostep(*a, **k))[1]

    def cl(*a, **k):
        o = ocl(*a, **k)
        losses.append(float(o['ce_loss']))
        return o
    model.compute_loss = cl
    accs = []
    oacc = rt.compute_accuracy

    def acc(*a, **k):
        r = oacc(*a, **k)
        accs.append([float(v) for v in r])
        return r
    rt.compute_accuracy = acc
    ep_loss, _ = rt.train(1, model, make_batches(c), opt, sched, c['V'] - 1)
    log_cols = [ln.split()[:3] for ln in open(os.path.join(ck, 'log.txt')).read().strip().split('\n')]
    out = dict(cfg=c, lrs_at_optim_step=lrs, losses=losses, accs=accs, ep_loss=float(ep_loss), final_lr=opt.param_groups[0]['lr'], log_cols=log_cols,
               final_param_sums={k: float(v.double().sum()) for k, v in model.state_dict().items()})
    json.dump(out, open(os.path.join(REPO, 'tests', 'golden', 'txl_trainloop.json'), 'w'))
    print('[golden stage1 train] steps', len(losses), 'losses', [round(l, 4) for l in losses], 'final lr', out['final_lr'])


if __name__ == '__main__':
    main()
