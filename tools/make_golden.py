#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by IMPORTING the real reference from /root/reference.

Runs ONLY in the build container (the GPU box has no /root/reference).  Nothing of the
reference's source travels: the fixtures hold inputs + expected outputs only; weights are
regenerated from NumPy seeds by oracle/weights.py.

Recipe = SURVEY.md Appendix F:
  1. stub `pickle5` / `miditoolkit` (host-only deps missing here);
  2. shim transformers-5.x GPT2Block.forward to 4.28 behaviour (explicit causal mask, 1-tuple);
  3. chdir to stage2_accompaniment so `from model.music_gpt2 import MusicGPT2` resolves.
Usage: PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import json
import os
import pickle
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/stage2_accompaniment'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def install_reference():
    sys.modules['pickle5'] = pickle
    mt = types.ModuleType('miditoolkit')
    mt.midi = types.ModuleType('miditoolkit.midi')
    mt.midi.containers = types.ModuleType('miditoolkit.midi.containers')
    for n in ('Marker', 'Instrument', 'TempoChange', 'Note'):
        setattr(mt.midi.containers, n, object)
    mt.midi.parser = types.ModuleType('miditoolkit.midi.parser')
    mt.midi.parser.MidiFile = object
    sys.modules.update({'miditoolkit': mt, 'miditoolkit.midi': mt.midi,
                        'miditoolkit.midi.containers': mt.midi.containers,
                        'miditoolkit.midi.parser': mt.midi.parser})
    from transformers.models.gpt2 import modeling_gpt2 as mg
    orig = mg.GPT2Block.forward

    def fwd(self, h, *a, **kw):
        T = h.size(1)
        mask = torch.full((T, T), torch.finfo(h.dtype).min).triu(1)[None, None]
        out = orig(self, h, attention_mask=mask)
        return (out,) if torch.is_tensor(out) else out
    mg.GPT2Block.forward = fwd
    os.chdir(REF)
    sys.path.insert(0, REF)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_reference()
    from model.music_gpt2 import MusicGPT2
    import inference as ref_inf
    import train as ref_train
    from oracle.weights import make_state_dict

    torch.manual_seed(0)
    manifest = {}

    # ------------------------------------------------------------ gpt2 fwd / loss / grads
    cases = [
        dict(name='gpt2_L2_d64_H4_T16_V40', V=40, L=2, H=4, d=64, dff=128, T=16, B=2, scale=1.0, seed=1),
        dict(name='gpt2_L2_d64_H4_T128_V327', V=327, L=2, H=4, d=64, dff=256, T=128, B=2, scale=6.0, seed=2),
        dict(name='gpt2_L1_d256_H8_T64_V327', V=327, L=1, H=8, d=256, dff=512, T=64, B=1, scale=4.0, seed=3),
        dict(name='gpt2_L2_d128_H2_T96_V100_noseg', V=100, L=2, H=2, d=128, dff=256, T=96, B=3, scale=5.0, seed=4, noseg=True),
        # BASELINE configs[0]: GPT-2, B=1, T=512, d_model=256 (CPU plumbing case), 12 layers
        dict(name='gpt2_cfg0_L12_d256_H8_T512_V327', V=327, L=12, H=8, d=256, dff=2048, T=512, B=1, scale=3.0, seed=5),
    ]
    for c in cases:
        nseg = None if c.get('noseg') else 2
        sd = make_state_dict('gpt2', c['V'], c['L'], c['H'], c['d'], c['dff'], n_segment_types=nseg,
                             seed=c['seed'], scale=c['scale'])
        model = MusicGPT2(c['V'], c['L'], c['H'], c['d'], c['dff'], c['d'], dropout=0.0,
                          use_segment_emb=nseg is not None, n_segment_types=nseg)
        msd = model.state_dict()
        assert set(sd.keys()) == set(msd.keys()), (set(sd) ^ set(msd))
        model.load_state_dict(sd)
        rng = np.random.default_rng(1000 + c['seed'])
        x = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['B'], c['T']), dtype=np.int64))
        seg = torch.from_numpy(rng.integers(0, 2, size=(c['B'], c['T']), dtype=np.int64))
        tgt = torch.from_numpy(rng.integers(0, c['V'] - 1, size=(c['B'], c['T']), dtype=np.int64))
        tgt[:, : c['T'] // 4] = c['V'] - 1          # a pad span, ignored by the loss
        model.train()                               # dropout=0.0 => deterministic
        logits = model(x, seg_inp=None if nseg is None else seg)
        loss = model.compute_loss(logits, tgt)['total_loss']
        loss.backward()
        gnorm = {k: float(p.grad.norm()) for k, p in model.named_parameters()}
        model.eval()
        with torch.no_grad():
            last = model(x, seg_inp=None if nseg is None else seg, keep_last_only=True)
        lg = logits.detach()
        np.savez_compressed(
            os.path.join(OUT, c['name'] + '.npz'),
            x=x.numpy(), seg=seg.numpy(), tgt=tgt.numpy(),
            logits_head=lg[..., :8].numpy(), logits_lse=torch.logsumexp(lg, -1).numpy(),
            logits_row0=lg[:, 0].numpy(), logits_rowlast=lg[:, -1].numpy(), last=last.numpy(),
            argmax=lg.argmax(-1).numpy(), top2_margin=(lg.topk(2, -1).values[..., 0] - lg.topk(2, -1).values[..., 1]).numpy(),
            loss=np.float32(loss.item()),
            grad_names=np.array(list(gnorm.keys())), grad_norms=np.array(list(gnorm.values()), dtype=np.float64),
        )
        manifest[c['name']] = {k: v for k, v in c.items() if k != 'name'}
        print('[golden]', c['name'], 'loss', loss.item())

    # ------------------------------------------------------------ prologue (PE rows)
    from model.transformer_helpers import PositionalEncoding
    pe = PositionalEncoding(512).pe
    rows = [0, 1, 511, 2047, 3071, 11999]
    np.savez_compressed(os.path.join(OUT, 'pe_rows_d512.npz'), rows=np.array(rows), pe=pe[rows, 0].numpy())

    # ------------------------------------------------------------ sampling
    rng = np.random.default_rng(7)
    samp = {}
    logit_sets = {
        'flat': np.zeros(40, dtype=np.float32),
        'peaked': (rng.standard_normal(327) * 4).astype(np.float32),
        'mild': (rng.standard_normal(327) * 1.0).astype(np.float32),
        'ties': np.repeat(np.array([3., 1., 0., -1.], dtype=np.float32), 10),
        'overflow': (rng.standard_normal(60) * 50 + 200).astype(np.float32),
        'dominant': np.concatenate([[30.0], np.zeros(20)]).astype(np.float32),
    }
    for nm, lg in logit_sets.items():
        for temp, p in ((1.1, 0.99), (1.2, 0.97), (1.1, 0.9)):
            key = '%s_t%.1f_p%.2f' % (nm, temp, p)
            probs = ref_inf.temperature(lg.copy(), temp, inadmissibles=None)
            entry = {'logits': lg.tolist(), 'temp': temp, 'p': p, 'probs': np.asarray(probs, dtype=np.float64).tolist(),
                     'probs_dtype': str(np.asarray(probs).dtype)}
            # deterministic part: replicate the reference lines up to the draw via its own function + seeded RNG
            try:
                words = []
                for s in range(4):
                    np.random.seed(s)
                    words.append(int(ref_inf.nucleus(np.array(probs, copy=True), p)))
                entry['words_seed0_3'] = words
                # candidate set: union over many seeds is not exact; recover it by brute force below
                cand = set()
                for s in range(200):
                    np.random.seed(1000 + s)
                    cand.add(int(ref_inf.nucleus(np.array(probs, copy=True), p)))
                entry['observed_candidates'] = sorted(cand)
                entry['error'] = None
            except IndexError as e:
                entry['error'] = 'IndexError'
            samp[key] = entry
    json.dump(samp, open(os.path.join(OUT, 'sampling.json'), 'w'))
    print('[golden] sampling cases', len(samp), 'errors', [k for k, v in samp.items() if v['error']])

    # ------------------------------------------------------------ accuracy
    acc = {}
    for s in range(3):
        r = np.random.default_rng(50 + s)
        B, T, V = 2, 64, 30
        lg = torch.from_numpy(r.standard_normal((B, T, V)).astype(np.float32))
        tg = torch.from_numpy(r.integers(0, V, size=(B, T), dtype=np.int64))
        lg[0, :20] = torch.nn.functional.one_hot(tg[0, :20], V).float() * 10   # make some predictions right
        ch = torch.from_numpy((r.random((B, T)) < 0.2).astype(np.int64))
        me = torch.from_numpy(((r.random((B, T)) < 0.3) & (ch.numpy() == 0)).astype(np.int64))
        ch[tg == V - 1] = 0
        me[tg == V - 1] = 0
        out = ref_train.compute_accuracy(lg, tg, ch, me, V - 1)
        acc['case%d' % s] = {'logits': lg.numpy().tolist(), 'tgt': tg.numpy().tolist(), 'chord': ch.numpy().tolist(),
                             'melody': me.numpy().tolist(), 'pad': V - 1, 'out': [float(o) for o in out]}
    json.dump(acc, open(os.path.join(OUT, 'accuracy.json'), 'w'))

    # ------------------------------------------------------------ LR schedule + train loop (tiny gpt2, CPU-neutralised)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import tempfile
    tl = {}
    for accum in (1, 2):
        V, L, H, d, dff, T, B = 40, 2, 4, 64, 128, 32, 2
        sd = make_state_dict('gpt2', V, L, H, d, dff, seed=9, scale=3.0)
        model = MusicGPT2(V, L, H, d, dff, d, dropout=0.0, use_segment_emb=True, n_segment_types=2)
        model.load_state_dict(sd)
        from oracle.weights import synthetic_batch
        batches = [synthetic_batch(V, B, T, seed=300 + i, realistic_targets=False) for i in range(8)]
        for bt in batches:
            bt["dec_target"][:, :5] = V - 1      # a pad span
        ref_train.gpuid = 0
        ref_train.redraw_prob = 0.0
        ref_train.accum_steps = accum
        ref_train.warmup_steps = 3
        ref_train.max_lr = 1e-3
        ref_train.log_interval = 2
        ref_train.ckpt_dir = tempfile.mkdtemp()
        ref_train.train_steps = 0
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 6, eta_min=1e-4)
        lrs, losses = [], []
        orig_step = opt.step

        def step_spy(*a, **k):
            lrs.append(opt.param_groups[0]['lr'])
            return orig_step(*a, **k)
        opt.step = step_spy
        orig_cl = model.compute_loss

        def cl_spy(*a, **k):
            o = orig_cl(*a, **k)
            losses.append(float(o['recons_loss'].item()))
            return o
        model.compute_loss = cl_spy
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            ep_loss = ref_train.train_model(1, model, batches, opt, sched, V - 1, model_type='gpt2')
        log_txt = open(os.path.join(ref_train.ckpt_dir, 'log.txt')).read()
        final = {k: v.detach().numpy().astype(np.float64).sum().item() for k, v in model.state_dict().items() if 'pe.pe' not in k}
        tl['accum%d' % accum] = {'cfg': dict(V=V, L=L, H=H, d=d, dff=dff, T=T, B=B, seed=9, scale=3.0, warmup=3, max_lr=1e-3,
                                             eta_min=1e-4, T_max=6, n_batches=8, batch_seed0=300, log_interval=2),
                                 'lrs_at_optim_step': lrs, 'losses': losses, 'ep_loss': float(ep_loss),
                                 'final_lr': opt.param_groups[0]['lr'],
                                 'log_cols': [ln.split()[:3] for ln in log_txt.strip().split('\n')],
                                 'final_param_sums': final}
        print('[golden] trainloop accum', accum, 'losses', losses[:3], '...')
    json.dump(tl, open(os.path.join(OUT, 'trainloop.json'), 'w'))

    # ------------------------------------------------------------ generate_conditional traces (toy vocab + tiny gpt2)
    events = (['Emotion_Q%d' % i for i in range(1, 5)] + ['Key_C', 'Key_a', 'Tempo_110', 'Track_LeadSheet', 'Track_Full', 'Bar_None'] +
              ['Beat_%d' % i for i in range(16)] + ['Note_Octave_4', 'Note_Degree_1', 'Note_Duration_4', 'Note_Velocity_60',
                                                     'Chord_I_M', 'Chord_V_M', 'EOS_None', 'PAD_None'])
    event2idx = {e: i for i, e in enumerate(events)}
    idx2event = {i: e for e, i in event2idx.items()}
    V = len(events)
    sd = make_state_dict('gpt2', V, 2, 4, 64, 128, seed=11, scale=8.0)
    model = MusicGPT2(V, 2, 4, 64, 128, 64, dropout=0.1, use_segment_emb=True, n_segment_types=2)
    model.load_state_dict(sd)
    model.eval()
    lead = [[event2idx[e] for e in ['Bar_None', 'Beat_0', 'Chord_I_M', 'Note_Octave_4', 'Note_Degree_1', 'Note_Duration_4']],
            [event2idx[e] for e in ['Bar_None', 'Beat_0', 'Chord_V_M', 'Beat_8', 'Note_Octave_4', 'Note_Degree_1', 'Note_Duration_4']],
            [event2idx[e] for e in ['Bar_None', 'Beat_4', 'Chord_I_M', 'EOS_None']]]
    primer = [event2idx['Emotion_Q1'], event2idx['Key_C'], event2idx['Tempo_110']]
    gen = {'events': events, 'lead': lead, 'primer': primer, 'model': dict(V=V, L=2, H=4, d=64, dff=128, seed=11, scale=8.0), 'runs': []}
    import io
    import contextlib
    for seed, skip in ((0, False), (1, False), (2, True)):
        sampled = []
        orig_nuc = ref_inf.nucleus

        def nuc_spy(probs, p):
            w = orig_nuc(probs, p)
            sampled.append(int(w))
            return w
        ref_inf.nucleus = nuc_spy
        np.random.seed(seed)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out = ref_inf.generate_conditional(model, event2idx, idx2event, [list(b) for b in lead], list(primer),
                                               max_events=200, skip_check=skip, temp=1.2, top_p=0.97, model_type='gpt2')
        ref_inf.nucleus = orig_nuc
        gen['runs'].append({'seed': seed, 'skip_check': skip, 'sampled': sampled, 'generated': [int(t) for t in out]})
        print('[golden] generate seed', seed, 'len', len(out), 'samples', len(sampled))
    # greedy variant (argmax substituted for nucleus)
    ref_inf_nucleus = ref_inf.nucleus
    ref_inf.nucleus = lambda probs, p: int(np.argmax(probs))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = ref_inf.generate_conditional(model, event2idx, idx2event, [list(b) for b in lead], list(primer),
                                           max_events=60, skip_check=True, temp=1.2, top_p=0.97, model_type='gpt2')
    ref_inf.nucleus = ref_inf_nucleus
    gen['greedy'] = [int(t) for t in out]
    json.dump(gen, open(os.path.join(OUT, 'generate.json'), 'w'))

    json.dump(manifest, open(os.path.join(OUT, 'manifest.json'), 'w'), indent=1)
    print('[golden] done ->', OUT)


if __name__ == '__main__':
    main()
