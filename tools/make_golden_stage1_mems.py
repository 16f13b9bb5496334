#!/usr/bin/env python3
"""tests/golden/txl_mems_*.npz: Transformer-XL segment recurrence of the REAL stage-1 model (/root/reference/stage1_compose, imported on
CPU): two consecutive segments through PlainTransformer.forward with mem_len > 0, once with the shared memory update (dec_seg_len None)
and once with the per-sample one (optimus_txl_decoder.py:702-748), loss + backward on the second segment (gradients flow through the
current segment only; qkv_net / layer_norm also see the memory rows).  Runs only in the build container; weights are regenerated from
NumPy seeds by oracle/txl_ref.py.
Usage: PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_stage1_mems.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/stage1_compose'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

CASES = [dict(name='txl_mems_shared', V=40, L=2, H=4, d=64, dff=128, T=16, B=3, mem_len=24, seed=31, scale=6.0, seg_len=None),
         dict(name='txl_mems_persample', V=40, L=2, H=4, d=64, dff=128, T=16, B=3, mem_len=20, seed=32, scale=6.0, seg_len=[[16, 11, 7], [9, 16, 13]])]


def main():
    from oracle.txl_ref import make_state_dict_txl
    os.chdir(REF)
    sys.path.insert(0, REF)
    from model.plain_transformer import PlainTransformer
    for c in CASES:
        sd = make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
        model = PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], c['mem_len'], c['T'], dec_dropout=0.0, pre_lnorm=True)
        model.load_state_dict(sd)
        model.train()
        rng = np.random.default_rng(3000 + c['seed'])
        xs = [torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], c['B']), dtype=np.int64)) for _ in range(3)]
        tgt = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], c['B']), dtype=np.int64))
        tgt[-3:] = c['V'] - 1
        sl = [None] * 3 if c['seg_len'] is None else [torch.tensor(v) for v in c['seg_len']] + [None]
        mems, out = tuple(), {}
        for i in range(2):
            logits, mems = model(xs[i], mems, dec_seg_len=sl[i])
            out['logits%d' % i] = logits.detach().numpy()
            out['mem%d_shape' % i] = np.array(mems[0].shape)
            out['mem%d_first' % i] = mems[0].detach().numpy()
            out['mem%d_last' % i] = mems[-1].detach().numpy()
            out['mem%d_sums' % i] = np.array([float(m.double().sum()) for m in mems])
        model.zero_grad()
        logits, mems3 = model(xs[2], mems, dec_seg_len=None)
        loss = model.compute_loss(logits, tgt)['total_loss']
        loss.backward()
        names = [n for n, _ in model.named_parameters()]
        gn = np.array([float(p.grad.norm()) if p.grad is not None else 0.0 for _, p in model.named_parameters()])
        out.update(logits2=logits.detach().numpy(), mem2_shape=np.array(mems3[0].shape), loss=np.float32(loss.item()), grad_names=np.array(names),
                   grad_norms=gn, g_qkv0=model.decoder.layers[0].dec_attn.qkv_net.weight.grad.numpy()[:, :8],
                   g_ln0=model.decoder.layers[0].dec_attn.layer_norm.weight.grad.numpy(), g_rw=model.decoder.r_w_bias.grad.numpy(),
                   g_rr=model.decoder.r_r_bias.grad.numpy(), g_rnet1=model.decoder.layers[1].dec_attn.r_net.weight.grad.numpy()[:, :8],
                   x=np.stack([x.numpy() for x in xs]), tgt=tgt.numpy(), cfg=np.array([c[k] for k in ('V', 'L', 'H', 'd', 'dff', 'T', 'B', 'mem_len', 'seed')]),
                   scale=np.float32(c['scale']), seg_len=np.array(c['seg_len'] if c['seg_len'] is not None else []))
        np.savez_compressed(os.path.join(OUT, c['name'] + '.npz'), **out)
        print('[golden stage1 mems]', c['name'], 'loss', loss.item(), 'mem shapes', out['mem0_shape'], out['mem1_shape'], out['mem2_shape'])


if __name__ == '__main__':
    main()
