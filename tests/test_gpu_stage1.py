"""GPU: stage-1 lead-sheet LM (Transformer-XL) inference path against fixtures recorded from the IMPORTED reference
(tools/make_golden_stage1.py) and the pinned oracle (oracle/txl_ref.py): evaluation logits, validation loss, generation with memory."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(json.load(open(os.path.join(G, 'txl_manifest.json'))).items())


def _model(c, dtype, mem_len=0):
    from emo_disentanger_amd.model.plain_transformer import PlainTransformer
    from oracle.txl_ref import make_state_dict_txl
    sd = make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    m = PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], mem_len, c['T'], dec_dropout=0.1, pre_lnorm=True, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


@pytest.mark.parametrize('name,c', CASES)
def test_stage1_forward_and_loss_match_reference_fp32(name, c):
    from oracle import txl_ref
    g = np.load(os.path.join(G, name + '.npz'))
    m, sd = _model(c, 'fp32')
    x, tgt = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['tgt']).cuda()
    logits, mems = m(x, tuple())
    assert mems == [] and logits.shape == (c['T'], x.shape[1], c['V']) and logits.dtype == torch.float32
    lg = logits.cpu()
    scale = float(np.abs(g['logits_row0']).max())
    tol = 2e-4 * max(scale, 1.0)
    np.testing.assert_allclose(lg[..., :8].numpy(), g['logits_head'], rtol=0, atol=tol)
    np.testing.assert_allclose(torch.logsumexp(lg, -1).numpy(), g['logits_lse'], rtol=0, atol=tol)
    np.testing.assert_allclose(lg[0].numpy(), g['logits_row0'], rtol=0, atol=tol)
    np.testing.assert_allclose(lg[-1].numpy(), g['logits_rowlast'], rtol=0, atol=tol)
    with torch.no_grad():
        ref, _ = txl_ref.forward(sd, torch.from_numpy(g['x']), c['L'], c['H'])
    top2 = ref.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(lg.argmax(-1)[safe], torch.from_numpy(g['argmax'])[safe])                 # greedy ids bit-exact where the margin allows
    loss = m.compute_loss(logits, tgt)['total_loss']
    assert abs(float(loss) - float(g['loss'])) < 1e-4


@pytest.mark.parametrize('name,c', CASES)
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_stage1_generation_with_memory(name, c, dtype):
    g = np.load(os.path.join(G, name + '.npz'))
    m, _ = _model(c, dtype, mem_len=c['T'])
    x = torch.from_numpy(g['x']).cuda()
    prime = c['T'] // 2
    lg, mem = m.generate(x[:prime, :1], tuple())
    outs = [lg.cpu().numpy()]
    for i in range(c['gen']):
        lg, mem = m.generate(x[prime + i:prime + i + 1, :1], mem)
        outs.append(lg.cpu().numpy())
    got, ref = np.stack(outs), g['gen_logits']
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(got, ref, rtol=0, atol=(3e-4 if dtype == 'fp32' else 6e-2) * max(scale, 1.0))
    assert mem.len == int(g['mem_len_after'])
    if dtype == 'fp32':
        top2 = np.sort(ref, -1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 1e-3
        assert (got.argmax(-1)[safe] == ref.argmax(-1)[safe]).all()


def test_stage1_bf16_forward_close_and_training_path_is_refused():
    name, c = CASES[1]
    g = np.load(os.path.join(G, name + '.npz'))
    m, _ = _model(c, 'bf16')
    x = torch.from_numpy(g['x']).cuda()
    logits, _ = m(x, tuple())
    scale = float(np.abs(g['logits_row0']).max())
    assert float((logits[0].cpu() - torch.from_numpy(g['logits_row0'])).abs().max()) <= 6e-2 * scale
    assert float((logits[-1].cpu() - torch.from_numpy(g['logits_rowlast'])).abs().max()) <= 6e-2 * scale
    m.train()
    with pytest.raises(NotImplementedError, match='training path'):
        m(x, tuple())
