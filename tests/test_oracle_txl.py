"""CPU: the stage-1 (Transformer-XL lead-sheet LM) oracle restatement against fixtures recorded from the IMPORTED reference
(tools/make_golden_stage1.py).  Test infrastructure for the next hot-path row: no product code involved yet."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(json.load(open(os.path.join(G, 'txl_manifest.json'))).items())


@pytest.mark.parametrize('name,c', CASES)
def test_txl_oracle_forward_loss_and_grads_match_reference(name, c):
    from oracle import txl_ref
    g = np.load(os.path.join(G, name + '.npz'))
    sd = txl_ref.make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'decoder.pos_emb.inv_freq'}
    leaf['decoder.pos_emb.inv_freq'] = sd['decoder.pos_emb.inv_freq']
    x, tgt = torch.from_numpy(g['x']), torch.from_numpy(g['tgt'])
    logits, mems = txl_ref.forward(leaf, x, c['L'], c['H'])
    assert mems is None
    lg = logits.detach()
    np.testing.assert_allclose(lg[..., :8].numpy(), g['logits_head'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(torch.logsumexp(lg, -1).numpy(), g['logits_lse'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(lg[0].numpy(), g['logits_row0'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(lg[-1].numpy(), g['logits_rowlast'], rtol=0, atol=2e-5)
    loss = txl_ref.loss(leaf, logits, tgt)
    assert abs(float(loss.detach()) - float(g['loss'])) < 1e-5
    loss.backward()
    names = [str(n) for n in g['grad_names']]
    assert names == [k for k in sd if k != 'decoder.pos_emb.inv_freq']            # parameter registration order = optimizer state order
    for n, ref in zip(names, g['grad_norms']):
        got = float(leaf[n].grad.norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-3), (n, got, ref)
    # the embedding's padding row (index V-1) never appears in x here; F.embedding in the oracle has no padding_idx, the reference does:
    # equal gradients as long as the pad id is not an input token (stage-1 batches pad only the targets' tail)
    assert int((x == c['V'] - 1).sum()) == 0


@pytest.mark.parametrize('name,c', CASES)
def test_txl_oracle_generation_with_memory_matches_reference(name, c):
    from oracle import txl_ref
    g = np.load(os.path.join(G, name + '.npz'))
    sd = txl_ref.make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    x = torch.from_numpy(g['x'])
    prime, mem_len = c['T'] // 2, c['T']
    with torch.no_grad():
        lg, mems = txl_ref.forward(sd, x[:prime, :1], c['L'], c['H'], mems=None, mem_len=mem_len)
        outs = [lg[-1, 0].numpy()]
        for i in range(c['gen']):
            lg, mems = txl_ref.forward(sd, x[prime + i:prime + i + 1, :1], c['L'], c['H'], mems=mems, mem_len=mem_len)
            outs.append(lg[-1, 0].numpy())
    np.testing.assert_allclose(np.stack(outs), g['gen_logits'], rtol=0, atol=3e-5)
    assert mems[0].shape[0] == int(g['mem_len_after'])
    np.testing.assert_allclose(outs[-1], g['gen_full_last'], rtol=0, atol=5e-5)   # cached steps == full recompute of the prefix


@pytest.mark.parametrize('name', ['txl_mems_shared', 'txl_mems_persample'])
def test_oracle_segment_recurrence_matches_reference(name):
    """Two segments through forward() with mem_len > 0 (shared and per-sample memory update, optimus_txl_decoder.py:702-748), then loss and
    backward on a third one — fixtures made by tools/make_golden_stage1_mems.py from the imported reference."""
    from oracle import txl_ref
    g = np.load(os.path.join(G, name + '.npz'))
    V, L, H, d, dff, T, B, mem_len, seed = (int(v) for v in g['cfg'])
    sd = txl_ref.make_state_dict_txl(V, L, H, d, dff, seed=seed, scale=float(g['scale']))
    leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'inv_freq' not in k else v) for k, v in sd.items()}
    xs, seg = torch.from_numpy(g['x']), g['seg_len']
    mems = None
    for i in range(2):
        with torch.no_grad():
            lg, mems = txl_ref.forward(sd, xs[i], L, H, mems=mems, mem_len=mem_len, dec_seg_len=None if seg.size == 0 else torch.from_numpy(seg[i]))
        np.testing.assert_allclose(lg.numpy(), g['logits%d' % i], rtol=0, atol=2e-5)
        assert tuple(mems[0].shape) == tuple(g['mem%d_shape' % i])
        np.testing.assert_allclose(mems[0].numpy(), g['mem%d_first' % i], rtol=0, atol=1e-5)
        np.testing.assert_allclose(mems[-1].numpy(), g['mem%d_last' % i], rtol=0, atol=2e-5)
    lg, m3 = txl_ref.forward(leaf, xs[2], L, H, mems=mems, mem_len=mem_len)
    loss = txl_ref.loss(leaf, lg, torch.from_numpy(g['tgt']))
    loss.backward()
    assert abs(float(loss.detach()) - float(g['loss'])) < 2e-5 and tuple(m3[0].shape) == tuple(g['mem2_shape'])
    np.testing.assert_allclose(lg.detach().numpy(), g['logits2'], rtol=0, atol=2e-5)
    for n, want in zip(g['grad_names'], g['grad_norms']):
        got = float(leaf[str(n)].grad.norm()) if leaf[str(n)].grad is not None else 0.0
        assert abs(got - want) <= 1e-4 * max(1.0, want), n


DROP_CASES = sorted(json.load(open(os.path.join(G, 'txl_dropout_manifest.json'))).items())


@pytest.mark.parametrize('name,c', DROP_CASES)
def test_txl_oracle_with_dropout_masks_matches_reference_training_run(name, c):
    # the fixture is the IMPORTED reference in training mode (dropout 0.1, all eight sites incl. the attention-probability dropout followed
    # by the renormalisation p / (sum p + 1e-8), optimus_txl_decoder.py:361-363) with its Bernoulli draws stored as keep-bits: the masked
    # restatement must land on the reference's logits, loss and every gradient — this pins the oracle the dropout-ON GPU tests use
    from oracle import txl_ref
    from dropmask import load_txl_dropout_fixture
    g, masks = load_txl_dropout_fixture(os.path.join(G, name + '.npz'), c)
    sd = txl_ref.make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    x, tgt = torch.from_numpy(g['x']), torch.from_numpy(g['tgt'])
    loss, logits, grads = txl_ref.loss_and_grads(sd, x, tgt, c['L'], c['H'], masks=masks)
    scale = float(np.abs(g['logits']).max())
    np.testing.assert_allclose(logits.numpy(), g['logits'], rtol=0, atol=2e-5 * scale)
    assert abs(float(loss) - float(g['loss'])) < 1e-5
    names = [str(n) for n in g['grad_names']]
    assert names == list(grads.keys())
    for n, ref in zip(names, g['grad_norms']):
        got = float(grads[n].norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-3), (n, got, ref)
    for i, n in enumerate(str(k) for k in g['full_names']):
        ref = g['grad_%d' % i]
        np.testing.assert_allclose(grads[n].numpy(), ref, rtol=0, atol=1e-4 * float(np.abs(ref).max()))
    # sanity of the fixture itself: the keep rate is 1 - p, the masks matter, and dropping the renormalisation is detected
    keep = np.concatenate([(m != 0).numpy().reshape(-1) for m in masks.values()])
    assert abs(keep.mean() - (1 - c['p'])) < 0.01
    off, _ = txl_ref.forward(sd, x, c['L'], c['H'])
    assert float((off - logits).abs().max()) > 1e-2 * scale
