#!/usr/bin/env python3
"""Generate tests/golden/txl_*.npz by IMPORTING the real stage-1 reference (/root/reference/stage1_compose).  Runs only in the build
container; the fixtures hold inputs + expected outputs, the weights are regenerated from NumPy seeds by oracle/txl_ref.py.
Usage: PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_stage1.py"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/stage1_compose'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def main():
    from oracle.txl_ref import make_state_dict_txl
    os.chdir(REF)
    sys.path.insert(0, REF)
    from model.plain_transformer import PlainTransformer
    cases = [dict(name='txl_L2_d64_H4_T32_V50', V=50, L=2, H=4, d=64, dff=128, T=32, B=2, scale=8.0, seed=21, gen=6),
             dict(name='txl_L3_d128_H8_T96_V200', V=200, L=3, H=8, d=128, dff=256, T=96, B=3, scale=6.0, seed=22, gen=4),
             dict(name='txl_L1_d64_H2_T17_V30', V=30, L=1, H=2, d=64, dff=64, T=17, B=1, scale=5.0, seed=23, gen=5)]
    manifest = {}
    for c in cases:
        sd = make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
        model = PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], 0, c['T'], dec_dropout=0.0, pre_lnorm=True)
        msd = model.state_dict()
        assert list(msd.keys()) == list(sd.keys()), [k for k in msd if k not in sd] + [k for k in sd if k not in msd]
        assert all(tuple(msd[k].shape) == tuple(sd[k].shape) for k in sd)
        model.load_state_dict(sd)
        rng = np.random.default_rng(2000 + c['seed'])
        x = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], c['B']), dtype=np.int64))
        tgt = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], c['B']), dtype=np.int64))
        tgt[: c['T'] // 4] = c['V'] - 1                        # a pad span, ignored by the loss
        model.train()                                          # dropout 0 => deterministic
        logits, mems = model(x, tuple())
        loss = model.compute_loss(logits, tgt)['total_loss']
        loss.backward()
        names = [n for n, _ in model.named_parameters()]
        gn = np.array([float(p.grad.norm()) if p.grad is not None else 0.0 for _, p in model.named_parameters()])
        # generation with memory: the inference script builds the model with mem_len = tgt_len (stage1_compose/inference.py:173-184)
        gm = PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], c['T'], c['T'], dec_dropout=0.0, pre_lnorm=True)
        gm.load_state_dict(sd)
        gm.eval()
        gen_logits, mm = [], tuple()
        with torch.no_grad():
            prime = c['T'] // 2
            lg, mm = gm.generate(x[:prime, :1], mm)            # inference_utils.py:66-77: whole primer first, then one token at a time
            gen_logits.append(lg.numpy())
            for i in range(c['gen']):
                lg, mm = gm.generate(x[prime + i:prime + i + 1, :1], mm)
                gen_logits.append(lg.numpy())
            full, _ = gm(x[:prime + c['gen'], :1], tuple())     # teacher-forced logits of the same prefix (mem_len only caches)
        lg_d = logits.detach()
        np.savez_compressed(os.path.join(OUT, c['name'] + '.npz'), x=x.numpy(), tgt=tgt.numpy(), logits_head=lg_d[..., :8].numpy(),
                            logits_lse=torch.logsumexp(lg_d, -1).numpy(), logits_row0=lg_d[0].numpy(), logits_rowlast=lg_d[-1].numpy(),
                            argmax=lg_d.argmax(-1).numpy(), loss=np.float32(loss.item()), grad_names=np.array(names), grad_norms=gn,
                            gen_logits=np.stack(gen_logits), gen_full_last=full[-1, 0].numpy(), mem_len_after=np.int64(mm[0].shape[0]))
        manifest[c['name']] = {k: v for k, v in c.items() if k != 'name'}
        print('[golden stage1]', c['name'], 'loss', loss.item(), 'params', sum(p.numel() for p in model.parameters()))
    json.dump(manifest, open(os.path.join(OUT, 'txl_manifest.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
