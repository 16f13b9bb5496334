#!/usr/bin/env python3
"""Generate tests/golden/txl_dropout_*.npz by IMPORTING the real stage-1 reference (/root/reference/stage1_compose) in TRAINING mode with
dropout 0.1 — the mode the stage-1 step trains (and is timed) in.  The run itself is the untouched reference (nn.Dropout / F.dropout on
torch's CPU generator).  The Bernoulli draws of that run are then replayed (same seed, same shapes, same order => same consumption of the
generator) and stored as packed keep-bits, so that the fixture holds INPUTS (tokens, targets, one keep-bit per dropout element) and the
reference's OUTPUTS (logits, loss, every parameter gradient's norm, a few full gradients); the script asserts that replaying the masks through
oracle/txl_ref.py reproduces the reference run before it writes anything.  Runs only in the build container.
Usage: PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_stage1_dropout.py"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/stage1_compose'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
P_DROP = 0.1


def replay_masks(seed, T, B, D, dff, H, L):
    """The multipliers the reference's forward draws after torch.manual_seed(seed), in ITS order (plain_transformer.py:64;
    optimus_txl_decoder.py:797, 803 [drop(zeros): the absent segment embedding — consumed, no effect], 804, per layer :361, :375, CoreNet
    dropouts :39-43, final :917), in the reference's time-major shapes."""
    torch.manual_seed(seed)
    draw = lambda *shape: F.dropout(torch.ones(*shape), P_DROP, True)
    m = {'emb': draw(T, B, D), 'emb2': draw(T, B, D)}
    draw(T, B, D)
    m['pos'] = draw(T, 1, D)
    for l in range(L):
        m['L%d.attn_prob' % l] = draw(T, T, B, H)
        m['L%d.attn_out' % l] = draw(T, B, D)
        m['L%d.ffn_hidden' % l] = draw(T, B, dff)
        m['L%d.ffn_out' % l] = draw(T, B, D)
    m['final'] = draw(T, B, D)
    return m


def main():
    from oracle import txl_ref
    os.chdir(REF)
    sys.path.insert(0, REF)
    from model.plain_transformer import PlainTransformer
    cases = [dict(name='txl_dropout_L2_d64_H4_T32_V50', V=50, L=2, H=4, d=64, dff=128, T=32, B=2, scale=8.0, seed=21, torch_seed=101),
             dict(name='txl_dropout_L3_d128_H8_T96_V200', V=200, L=3, H=8, d=128, dff=256, T=96, B=3, scale=6.0, seed=22, torch_seed=102)]
    manifest = {}
    for c in cases:
        sd = txl_ref.make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
        model = PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], 0, c['T'], dec_dropout=P_DROP, pre_lnorm=True)
        model.load_state_dict(sd)
        rng = np.random.default_rng(3000 + c['seed'])
        x = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], c['B']), dtype=np.int64))
        tgt = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], c['B']), dtype=np.int64))
        tgt[: c['T'] // 4] = c['V'] - 1
        model.train()
        torch.manual_seed(c['torch_seed'])
        logits, _ = model(x, tuple())                          # the reference, untouched, dropout ON
        loss = model.compute_loss(logits, tgt)['total_loss']
        loss.backward()
        names = [n for n, _ in model.named_parameters()]
        grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
        masks = replay_masks(c['torch_seed'], c['T'], c['B'], c['d'], c['dff'], c['H'], c['L'])
        # the replay is the same draw: the masked restatement must land on the reference's numbers
        ol, ologits, ograds = txl_ref.loss_and_grads(sd, x, tgt, c['L'], c['H'], masks=masks)
        assert float((ologits - logits.detach()).abs().max()) <= 2e-5 * float(logits.detach().abs().max()), 'replayed masks do not reproduce the reference'
        assert abs(float(ol) - float(loss)) < 1e-5
        for n in names:
            assert float((ograds[n] - grads[n]).abs().max()) <= 1e-4 * float(grads[n].abs().max() + 1e-12), n
        # and it is a different function from the dropout-off forward
        model.eval()
        with torch.no_grad():
            off, _ = model(x, tuple())
        assert float((off - logits.detach()).abs().max()) > 1e-2 * float(off.abs().max())
        bits = {('keep_' + k): np.packbits((v != 0).numpy().reshape(-1)) for k, v in masks.items()}
        full = ['decoder.r_w_bias', 'decoder.r_r_bias', 'decoder.layers.0.dec_attn.r_net.weight', 'decoder.layers.%d.dec_attn.qkv_net.weight' % (c['L'] - 1),
                'decoder.layers.0.pos_ff.CoreNet.3.bias', 'word_emb.emb_lookup.weight']
        np.savez_compressed(os.path.join(OUT, c['name'] + '.npz'), x=x.numpy(), tgt=tgt.numpy(), logits=logits.detach().numpy(),
                            loss=np.float32(loss.item()), grad_names=np.array(names), grad_norms=np.array([float(grads[n].norm()) for n in names]),
                            full_names=np.array(full), **{'grad_' + str(i): grads[n].numpy() for i, n in enumerate(full)}, **bits)
        manifest[c['name']] = {k: v for k, v in c.items() if k != 'name'}
        manifest[c['name']]['p'] = P_DROP
        print('[golden stage1 dropout]', c['name'], 'loss', loss.item(), 'max |logit on - off|', float((off - logits.detach()).abs().max()))
    json.dump(manifest, open(os.path.join(OUT, 'txl_dropout_manifest.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
