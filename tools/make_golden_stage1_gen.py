#!/usr/bin/env python3
"""tests/golden/txl_generate.json: traces of the REAL stage-1 sampling loop (stage1_compose/inference_utils.py generate_plain_xl) on a tiny
imported PlainTransformer with a toy vocabulary, NumPy-seeded.  Runs only in the build container."""
import json
import os
import pickle
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/stage1_compose'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def main():
    from oracle.txl_ref import make_state_dict_txl
    sys.modules['pickle5'] = pickle                       # utils.py imports pickle5 (absent here)
    mt = types.ModuleType('miditoolkit')
    sys.modules.setdefault('miditoolkit', mt)
    os.chdir(REF)
    sys.path.insert(0, REF)
    from model.plain_transformer import PlainTransformer
    import inference_utils as ref
    events = (['Emotion_Q1', 'Emotion_Q2', 'Emotion_Positive', 'Emotion_Negative'] + ['Key_%s' % k for k in ['C', 'G', 'a', 'e']] + ['Bar_None'] +
              ['Beat_%d' % i for i in range(8)] + ['Chord_%s' % c for c in ['I_M', 'IV_M', 'V_M', 'vi_m', 'None_None']] +
              ['Note_Octave_4', 'Note_Degree_1', 'Note_Degree_3', 'Note_Degree_5', 'Note_Duration_2', 'Note_Duration_4'] + ['EOS_None', 'PAD_None'])
    e2i = {e: i for i, e in enumerate(events)}
    i2e = {i: e for e, i in e2i.items()}
    c = dict(V=len(events), L=2, H=4, d=64, dff=128, T=64, seed=31, scale=9.0)
    sd = make_state_dict_txl(c['V'], c['L'], c['H'], c['d'], c['dff'], seed=c['seed'], scale=c['scale'])
    m = PlainTransformer(c['d'], c['V'], c['L'], c['H'], c['d'], c['dff'], c['T'], c['T'], dec_dropout=0.1, pre_lnorm=True)
    m.load_state_dict(sd)
    m.eval()
    runs = []
    for seed, kw in [(0, dict(primer=['Emotion_Q1'], key_determine=None, max_bars=3, max_events=60)),
                     (1, dict(primer=['Emotion_Q2'], key_determine='rule', max_bars=4, max_events=50)),
                     (2, dict(primer=['Emotion_Positive', 'Key_C', 'Bar_None', 'Beat_0', 'Chord_I_M'], key_determine=None, max_bars=3, max_events=70, prompt_bars=1)),
                     (3, dict(primer=None, key_determine=None, max_bars=2, max_events=40, representation='remi'))]:
        sampled = []
        orig = ref.nucleus

        def spy(probs, p, orig=orig, sampled=sampled):
            w = orig(probs, p)
            sampled.append(int(w))
            return w
        ref.nucleus = spy
        np.random.seed(seed)
        try:
            with torch.no_grad():
                out, _ = ref.generate_plain_xl(m, e2i, i2e, temp=1.2, top_p=0.9, **kw)
            err = None
        except ValueError as e:
            out, err = None, str(e)
        ref.nucleus = orig
        runs.append(dict(seed=seed, kw=kw, generated=None if out is None else [int(t) for t in out], sampled=sampled, error=err))
        print('[golden stage1 gen] seed', seed, 'len', None if out is None else len(out), 'samples', len(sampled), 'err', err)
    json.dump(dict(events=events, model=c, runs=runs), open(os.path.join(REPO, 'tests', 'golden', 'txl_generate.json'), 'w'))


if __name__ == '__main__':
    main()
