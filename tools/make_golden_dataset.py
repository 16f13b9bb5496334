#!/usr/bin/env python3
"""Runs ONLY in the build container (needs /root/reference): writes three tiny synthetic pieces + a dictionary in the reference's on-disk
format under tests/golden/dataset/ and records what the REAL REMISkylineToMidiTransformerDataset (stage2_accompaniment/dataloader.py)
returns for them under seeded `random` — the fixture tests/test_host_logic.py::test_event_piece_dataset_matches_reference checks
emo-disentanger_amd/data.py::EventPieceDataset against."""
import json
import os
import pickle
import random
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden', 'dataset')
sys.modules['pickle5'] = pickle                      # the reference imports pickle5 (absent here); same API
sys.path.insert(0, '/root/reference/stage2_accompaniment')
import dataloader as ref                              # noqa: E402


def make_piece(rng, n_bars, dict_events):
    ev = [{'name': 'Emotion', 'value': 'Q1'}, {'name': 'Key', 'value': 'C'}]
    lead_pos, full_pos = [], []
    for b in range(n_bars):
        s = len(ev)
        ev.append({'name': 'Track', 'value': 'LeadSheet'})
        ev.append({'name': 'Bar', 'value': None})
        for _ in range(int(rng.integers(2, 6))):
            ev.append({'name': 'Beat', 'value': int(rng.integers(0, 16))})
            ev.append({'name': 'Chord', 'value': ['I_M', 'V_M', 'vi_m'][int(rng.integers(0, 3))]})
            ev.append({'name': 'Note', 'value': int(rng.integers(0, 6))})
        lead_pos.append((s, len(ev)))
        s = len(ev)
        ev.append({'name': 'Track', 'value': 'Full'})
        for _ in range(int(rng.integers(3, 9))):
            ev.append({'name': 'Beat', 'value': int(rng.integers(0, 16))})
            ev.append({'name': 'Note', 'value': int(rng.integers(0, 6))})
            ev.append({'name': 'Duration', 'value': int(rng.integers(1, 4))})
        full_pos.append((s, len(ev)))
    ev.append({'name': 'EOS', 'value': None})
    full_pos[-1] = (full_pos[-1][0], len(ev))
    for e in ev:
        dict_events.add('%s_%s' % (e['name'], e['value']))
    return lead_pos, full_pos, ev


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(5)
    names, pieces, vocab = ['p_long.pkl', 'p_mid.pkl', 'p_short.pkl'], [], set(['Bar_None', 'EOS_None'])
    for name, bars in zip(names, (14, 6, 2)):
        pieces.append(make_piece(rng, bars, vocab))
    event2idx = {e: i for i, e in enumerate(sorted(vocab))}
    idx2event = {i: e for e, i in event2idx.items()}
    pickle.dump((event2idx, idx2event), open(os.path.join(OUT, 'dictionary.pkl'), 'wb'), protocol=4)
    for name, p in zip(names, pieces):
        pickle.dump(p, open(os.path.join(OUT, name), 'wb'), protocol=4)
    pickle.dump(names[:2], open(os.path.join(OUT, 'train.pkl'), 'wb'), protocol=4)
    expect = {}
    for key, kw in {'w96': dict(model_dec_seqlen=96), 'w96_key': dict(model_dec_seqlen=96, predict_key=True), 'w400': dict(model_dec_seqlen=400)}.items():
        ds = ref.REMISkylineToMidiTransformerDataset(OUT, os.path.join(OUT, 'dictionary.pkl'), pieces=names, pad_to_same=True, **kw)
        runs = []
        for seed in (0, 1, 2):
            random.seed(seed)
            for i in range(len(ds)):
                s = ds[i]
                runs.append({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in s.items()})
        expect[key] = {'kw': kw, 'vocab_size': ds.vocab_size, 'pad_token': ds.pad_token, 'stbars': ds.piece_admissible_stbars, 'samples': runs}
    json.dump(expect, open(os.path.join(OUT, 'expected.json'), 'w'))
    print('wrote', OUT, {k: len(v['samples']) for k, v in expect.items()}, 'vocab', len(event2idx))


if __name__ == '__main__':
    main()
