import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import dropmask
from oracle import txl_ref
from emo_disentanger_amd.model import plain_transformer as pt

def run(c, B, dtype, p, seed=77):
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], B), dtype=np.int64))
    tgt = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['T'], B), dtype=np.int64))
    sd = txl_ref.make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    m = pt.PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], 0, c['T'], dec_dropout=p, pre_lnorm=True, compute_dtype=dtype)
    m.load_state_dict(sd); m = m.cuda().train(); m.set_dropout_seed(seed)
    logits, _ = m(x.cuda(), tuple()); loss = m.compute_loss(logits, tgt.cuda())['total_loss']; loss.backward()
    T, D, H, L = c['T'], c['d'], c['H'], c['L']
    masks = None
    if p > 0:
        masks = dropmask.export_txl_masks(p, seed, 4096, B, T, D, c['dff'], H, L, lambda l: dropmask.site_multipliers((B, H, T, T), p, seed, 4096 + 8 * (l + 1) + 1))
    rl, rlg, rg = txl_ref.loss_and_grads(sd, x, tgt, L, H, masks=masks)
    gmax = max(float(g.abs().max()) for g in rg.values())
    errs = sorted(((float((prm.grad.cpu() - rg[k]).abs().max()) / gmax, k, float(rg[k].abs().max()) / gmax) for k, prm in m.named_parameters()), reverse=True)
    print(dtype, 'p', p, 'B', B, 'T', T, 'L', L, '|dloss| %.3g' % abs(float(loss) - float(rl)), 'dlogit %.3g' % float((logits.detach().float().cpu() - rlg).abs().max()), 'gmax %.3g' % gmax)
    for e in errs[:4]: print('    %.3g  %s (its own max %.3g of gmax)' % e)

bench = dict(V=200, L=12, H=8, d=512, dff=2048, T=512, seed=5, scale=2.0)
for p in (0.0, 0.1):
    run(bench, 4, 'fp32', p)
small = dict(V=50, L=2, H=4, d=64, dff=128, T=32, seed=21, scale=8.0)
mid = dict(V=200, L=3, H=8, d=128, dff=256, T=96, seed=22, scale=6.0)
for c, B in ((small, 2), (mid, 3)):
    for p in (0.0, 0.1):
        run(c, B, 'bf16', p)
        run(c, B, 'fp32', p)
