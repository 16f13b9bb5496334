"""GPU: the sampling loop and train loop against traces recorded from the IMPORTED reference, and the
decode engines (recurrent FAVOR+ state / KV cache) against full-prefix recompute."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def _tiny_gpt2(m, dtype='fp32', dropout=0.1):
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from oracle.weights import make_state_dict
    sd = make_state_dict('gpt2', m['V'], m['L'], m['H'], m['d'], m['dff'], seed=m['seed'], scale=m['scale'])
    mod = MusicGPT2(m['V'], m['L'], m['H'], m['d'], m['dff'], m['d'], dropout=dropout, use_segment_emb=True, n_segment_types=2, compute_dtype=dtype)
    mod.load_state_dict(sd)
    return mod.cuda().eval()


@pytest.mark.parametrize('use_cache', [True, False])
def test_generate_conditional_reproduces_reference_traces(use_cache):
    from emo_disentanger_amd import inference as inf
    g = json.load(open(os.path.join(G, 'generate.json')))
    e2i = {e: i for i, e in enumerate(g['events'])}
    i2e = {i: e for e, i in e2i.items()}
    model = _tiny_gpt2(g['model'])
    # greedy decode: bit-exact token ids (temperature-0 criterion of the north star)
    out = inf.generate_conditional(model, e2i, i2e, [list(b) for b in g['lead']], list(g['primer']), max_events=60, skip_check=True, temp=1.2,
                                   top_p=0.97, model_type='gpt2', use_cache=use_cache, sampler=lambda p: int(np.argmax(p)))
    assert out == g['greedy']
    # seeded nucleus runs: same NumPy RNG stream => same samples, rejections and final sequence
    for run in g['runs']:
        trace = []

        def spy(probs, run=run):
            w = inf.nucleus(probs, 0.97)
            trace.append(int(w))
            return w
        np.random.seed(run['seed'])
        out = inf.generate_conditional(model, e2i, i2e, [list(b) for b in g['lead']], list(g['primer']), max_events=200,
                                       skip_check=run['skip_check'], temp=1.2, top_p=0.97, model_type='gpt2', use_cache=use_cache, sampler=spy)
        assert trace == run['sampled']
        assert out == run['generated']


@pytest.mark.parametrize('kind', ['gpt2', 'performer'])
@pytest.mark.parametrize('skip_check', [False, True])
def test_generate_conditional_batch_equals_single_stream_runs(kind, skip_check):
    # SURVEY f-4: streams with DIFFERENT lead sheets / primers / RNGs in lock-step on one engine; every stream must reproduce
    # what the single-stream reference-shaped loop produces for it alone with the same sampler.
    from emo_disentanger_amd import inference as inf
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from oracle.weights import make_state_dict
    g = json.load(open(os.path.join(G, 'generate.json')))
    e2i = {e: i for i, e in enumerate(g['events'])}
    i2e = {i: e for e, i in e2i.items()}
    if kind == 'gpt2':
        model = _tiny_gpt2(g['model'])
    else:
        m = g['model']
        sd = make_state_dict('performer', m['V'], m['L'], m['H'], m['d'], m['dff'], favor_feature_dims=32, seed=m['seed'], scale=m['scale'])
        model = MusicPerformer(m['V'], m['L'], m['H'], m['d'], m['dff'], m['d'], favor_feature_dims=32, use_segment_emb=True, n_segment_types=2,
                               compute_dtype='fp32', redraw='fixed')
        model.load_state_dict(sd)
        model = model.cuda().eval()
    lead = [list(b) for b in g['lead']]
    leads = [lead, lead[::-1], lead[:2], [lead[1]] * 4, lead + lead]
    primers = [list(g['primer']), [1, 5, 6], list(g['primer']), [2, 4], [3, 5, 6]]
    seeds = [11, 12, 13, 14, 15]
    batch = inf.generate_conditional_batch(model, e2i, i2e, leads, primers, max_events=150, skip_check=skip_check, temp=1.2, top_p=0.97, seeds=seeds)
    assert len({tuple(b) for b in batch}) == len(leads)          # five different sequences (prompts of 4 different lengths)
    for i in range(len(leads)):
        rs = np.random.RandomState(seeds[i])
        single = inf.generate_conditional(model, e2i, i2e, leads[i], primers[i], max_events=150, skip_check=skip_check, temp=1.2, top_p=0.97,
                                          model_type=kind, sampler=lambda p, rs=rs: inf.nucleus(p, 0.97, rng=rs))
        assert batch[i] == single, i


@pytest.mark.parametrize('kind,dtype', [('performer', 'fp32'), ('performer', 'bf16'), ('gpt2', 'fp32'), ('gpt2', 'bf16')])
def test_decode_engine_equals_full_recompute(kind, dtype):
    from emo_disentanger_amd import inference as inf
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from oracle.weights import make_state_dict
    V, L, H, d, dff, nf, n, T0, Tn = 80, 2, 4, 128, 256, 64, 3, 37, 12
    sd = make_state_dict(kind, V, L, H, d, dff, favor_feature_dims=nf, seed=8, scale=3.0)
    if kind == 'performer':
        m = MusicPerformer(V, L, H, d, dff, d, favor_feature_dims=nf, use_segment_emb=True, n_segment_types=2, compute_dtype=dtype, redraw='fixed')
    else:
        m = MusicGPT2(V, L, H, d, dff, d, use_segment_emb=True, n_segment_types=2, compute_dtype=dtype)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    gen = torch.Generator().manual_seed(0)
    tok = torch.randint(0, V - 1, (n, T0 + Tn), generator=gen).cuda()
    seg = torch.randint(0, 2, (n, T0 + Tn), generator=gen).cuda()
    with torch.no_grad():
        full = m(tok, seg_inp=seg)                                  # [n, T, V] teacher-forced logits
    eng = inf.make_engine(m, n)
    lg = eng.append(tok[:, :T0], seg[:, :T0])
    tol = 2e-4 if dtype == 'fp32' else 6e-2
    scale = float(full.abs().max())
    assert float((lg - full[:, T0 - 1]).abs().max()) <= tol * scale
    for i in range(Tn):
        lg = eng.append(tok[:, T0 + i:T0 + i + 1], seg[:, T0 + i:T0 + i + 1])
        assert float((lg - full[:, T0 + i]).abs().max()) <= tol * scale, i
        if dtype == 'fp32':
            top2 = full[:, T0 + i].topk(2, -1).values
            safe = (top2[:, 0] - top2[:, 1]) > 1e-3
            assert torch.equal(lg.argmax(-1)[safe], full[:, T0 + i].argmax(-1)[safe])


def test_generate_streams_lockstep_greedy_and_nucleus():
    from emo_disentanger_amd import inference as inf
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from oracle.weights import make_state_dict
    V, L, H, d, dff, nf, n, T0, Tn = 327, 2, 8, 256, 512, 128, 4, 16, 24
    sd = make_state_dict('performer', V, L, H, d, dff, favor_feature_dims=nf, seed=9, scale=4.0)
    m = MusicPerformer(V, L, H, d, dff, d, favor_feature_dims=nf, use_segment_emb=True, n_segment_types=2, compute_dtype='fp32', redraw='fixed')
    m.load_state_dict(sd)
    m = m.cuda().eval()
    gen = torch.Generator().manual_seed(1)
    ptok = torch.randint(0, V - 1, (n, T0), generator=gen).cuda()
    pseg = torch.ones(n, T0, dtype=torch.long).cuda()
    out = inf.generate_streams(m, ptok, pseg, Tn, greedy=True)
    with torch.no_grad():
        full = m(out, seg_inp=torch.ones_like(out))
    top2 = full[:, T0 - 1:-1].topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(out[:, T0:][safe], full[:, T0 - 1:-1].argmax(-1)[safe])     # every greedy token is the argmax of its prefix
    eager = inf.generate_streams(m, ptok, pseg, Tn, greedy=True, use_graph=False)
    assert torch.equal(out, eager)                                                 # hipGraph replay == eager launches
    a = inf.generate_streams(m, ptok, pseg, Tn, temp=1.1, top_p=0.9, seed=3)
    b = inf.generate_streams(m, ptok, pseg, Tn, temp=1.1, top_p=0.9, seed=3, use_graph=False)
    c = inf.generate_streams(m, ptok, pseg, Tn, temp=1.1, top_p=0.9, seed=4)
    assert torch.equal(a, b) and not torch.equal(a, c) and int(a.max()) < V


def test_generate_streams_gpt2_kv_cache_graph():
    from emo_disentanger_amd import inference as inf
    g = json.load(open(os.path.join(G, 'generate.json')))
    m = _tiny_gpt2(g['model'])
    gen = torch.Generator().manual_seed(2)
    ptok = torch.randint(0, g['model']['V'] - 1, (3, 9), generator=gen).cuda()
    pseg = torch.ones(3, 9, dtype=torch.long).cuda()
    a = inf.generate_streams(m, ptok, pseg, 20, greedy=True)
    b = inf.generate_streams(m, ptok, pseg, 20, greedy=True, use_graph=False)
    assert torch.equal(a, b)
    with torch.no_grad():
        full = m(a, seg_inp=torch.ones_like(a))
    top2 = full[:, 8:-1].topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(a[:, 9:][safe], full[:, 8:-1].argmax(-1)[safe])


@pytest.mark.parametrize('accum', [1, 2])
def test_train_loop_matches_reference_trace(accum):
    from emo_disentanger_amd import train as tr
    from emo_disentanger_amd.data import synthetic_batch
    e = json.load(open(os.path.join(G, 'trainloop.json')))['accum%d' % accum]
    c = e['cfg']
    model = _tiny_gpt2(dict(V=c['V'], L=c['L'], H=c['H'], d=c['d'], dff=c['dff'], seed=c['seed'], scale=c['scale']), dropout=0.0)
    batches = [synthetic_batch(c['V'], c['B'], c['T'], seed=c['batch_seed0'] + i) for i in range(c['n_batches'])]
    for b in batches:
        b['dec_target'][:, :5] = c['V'] - 1
    cfg = tr.TrainConfig(warmup_steps=c['warmup'], max_lr=c['max_lr'], min_lr=c['eta_min'], lr_decay_steps=c['T_max'], accum_steps=accum,
                         log_interval=c['log_interval'], ckpt_dir=tempfile.mkdtemp(), verbose=False)
    opt = torch.optim.Adam(model.parameters(), lr=c['max_lr'])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, c['T_max'], eta_min=c['eta_min'])
    lrs, losses = [], []
    ostep, ocl = opt.step, model.compute_loss
    opt.step = lambda *a, **k: (lrs.append(opt.param_groups[0]['lr']), ostep(*a, **k))[1]

    def cl(*a, **k):
        o = ocl(*a, **k)
        losses.append(float(o['recons_loss']))
        return o
    model.compute_loss = cl
    ep_loss = tr.train_model(1, model, batches, opt, sched, c['V'] - 1, model_type='gpt2', cfg=cfg)
    np.testing.assert_allclose(lrs, e['lrs_at_optim_step'], rtol=1e-9)
    np.testing.assert_allclose(losses, e['losses'], rtol=0, atol=2e-4)        # incl. the F11 accumulation behaviour
    assert abs(ep_loss - e['ep_loss']) < 2e-4 and abs(opt.param_groups[0]['lr'] - e['final_lr']) < 1e-12
    cols = [ln.split()[:3] for ln in open(os.path.join(cfg.ckpt_dir, 'log.txt')).read().strip().split('\n')]
    assert [r[:2] for r in cols] == [r[:2] for r in e['log_cols']]
    for k, v in model.state_dict().items():
        if 'pe.pe' not in k:
            assert abs(float(v.double().sum()) - e['final_param_sums'][k]) <= 2e-3 * max(1.0, abs(e['final_param_sums'][k])), k


def test_fused_adam_equals_torch_adam_trajectory_and_state_dict():
    """Same gradients in => same parameters out as clip_grad_norm_(0.5) + torch.optim.Adam, step after step."""
    from emo_disentanger_amd.optim import FusedAdam
    from emo_disentanger_amd.data import synthetic_batch
    m = _tiny_gpt2(dict(V=40, L=2, H=4, d=64, dff=128, seed=9, scale=3.0), dropout=0.0).train()
    fused = FusedAdam(m, lr=1e-3, max_grad_norm=0.5)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in m.parameters()]
    o1 = torch.optim.Adam(ref, lr=1e-3)
    for i in range(4):
        b = synthetic_batch(40, 2, 32, seed=50 + i, device='cuda')
        fused.zero_grad()
        m.compute_loss(m(b['dec_input'], seg_inp=b['track_mask']), b['dec_target'])['total_loss'].backward()
        for r, p in zip(ref, m.parameters()):
            r.grad = p.grad.detach().clone()
        torch.nn.utils.clip_grad_norm_(ref, 0.5)
        o1.step()
        fused.step()
        for (k, p), r in zip(m.named_parameters(), ref):
            assert torch.allclose(p, r, rtol=1e-5, atol=2e-6), (i, k)
            r.data.copy_(p.data)                                   # stay in lock-step (Adam amplifies 1e-7 differences of ~0 gradients)
    s1, s2 = o1.state_dict(), fused.state_dict()
    assert s1['state'].keys() == s2['state'].keys()
    for i in s1['state']:
        assert torch.allclose(s1['state'][i]['exp_avg'], s2['state'][i]['exp_avg'].to(s1['state'][i]['exp_avg'].device), rtol=1e-3, atol=1e-7)
        assert torch.allclose(s1['state'][i]['exp_avg_sq'], s2['state'][i]['exp_avg_sq'].to(s1['state'][i]['exp_avg'].device), rtol=1e-3, atol=1e-10)
    o3 = FusedAdam(m, lr=1e-3)
    o3.load_state_dict(s1)                                    # a torch.optim.Adam checkpoint loads into the fused optimizer
    assert o3._step == 4
