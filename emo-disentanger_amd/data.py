"""Synthetic EMOPIA-shaped batches; dict keys / dtypes follow the reference dataset's
``__getitem__`` (/root/reference/stage2_accompaniment/dataloader.py:221-231) after default collation.
`EventPieceDataset` reads the reference's on-disk format (one pickle per piece + dictionary.pkl) and yields the same samples as
the reference's REMISkylineToMidiTransformerDataset (dataloader.py:41-231, pinned by tests/golden/dataset/); the event pickles
themselves are a download the repo does not ship, so the bench and the tests use `synthetic_batch`."""
import glob
import os
import pickle
import random

import numpy as np
import torch
from torch.utils.data import Dataset

N_TOKEN_FUNCTIONAL = 327   # 326 events + pad (derivation in SURVEY §8(d)); pad id = n_token-1, EOS = n_token-2


def synthetic_batch(n_token, B, T, seed=1234, realistic_targets=False, device=None):
    rng = np.random.default_rng(seed)
    pad, eos = n_token - 1, n_token - 2
    inp = rng.integers(0, n_token - 1, size=(B, T), dtype=np.int64)
    seg = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        t, cur = 0, 0
        while t < T:
            run = int(rng.integers(8, 65))
            seg[b, t:t + run] = cur
            cur ^= 1
            t += run
    tgt = np.empty_like(inp)
    tgt[:, :-1] = inp[:, 1:]
    tgt[:, -1] = eos
    if realistic_targets:
        tgt[seg == 0] = pad
    z = np.zeros((B, T), dtype=np.int64)
    out = {'id': torch.arange(B), 'dec_input': torch.from_numpy(inp), 'dec_target': torch.from_numpy(tgt),
           'track_mask': torch.from_numpy(seg), 'chord_idx': torch.from_numpy(z.copy()), 'melody_idx': torch.from_numpy(z.copy()),
           'length': torch.full((B,), T)}
    if device is not None:
        out = {k: v.to(device) for k, v in out.items()}
    return out



# ------------------------------------------------------------------------------------------------ on-disk event pieces
def load_vocab(vocab_file):
    """dictionary.pkl = (event2idx, idx2event).  The pad id is one past the dictionary (so n_token = len(dictionary) + 1)."""
    with open(vocab_file, 'rb') as f:
        event2idx, idx2event = pickle.load(f)[:2]
    return event2idx, idx2event, len(event2idx)


def _event_name(e):
    return '%s_%s' % (e['name'], e['value']) if isinstance(e, dict) else e


class EventPieceDataset(Dataset):
    """One sample per piece.  A piece pickle holds (lead_pos, full_pos, events): per bar the [start, end) event ranges of its lead-sheet
    part and of its full-arrangement part, and the event list (dicts {name, value} or 'Name_Value' strings); events before the first
    bar are the piece header (emotion / key tags).  A sample is header + the events from a random admissible start bar on, mapped to
    ids and padded to `model_dec_seqlen`; the target is the input shifted left by one INSIDE the full-arrangement spans (EOS closes the
    last one, pad everywhere else) and `track_mask` marks those spans — the model learns the accompaniment given the lead sheet.
    `chord_idx` / `melody_idx` flag targets that are Chord_* / Note_* events (accuracy break-down).  Start bars come from Python's global
    `random`, like the reference, so a seeded run reproduces its sampling."""

    def __init__(self, data_dir, vocab_file, model_dec_seqlen=10240, model_max_bars=None, pieces=[], pad_to_same=True, appoint_st_bar=None,
                 dec_end_pad_value=None, predict_key=None):
        self.vocab_file, self.data_dir = vocab_file, data_dir
        self.event2idx, self.idx2event, self.pad_token = load_vocab(vocab_file)
        self.vocab_size = self.pad_token + 1
        self.bar_token, self.eos_token = self.event2idx['Bar_None'], self.event2idx['EOS_None']
        self.model_dec_seqlen, self.model_max_bars = model_dec_seqlen, model_max_bars
        self.pad_to_same, self.predict_key, self.appoint_st_bar = pad_to_same, predict_key, appoint_st_bar
        self.dec_end_pad_value = self.eos_token if dec_end_pad_value == 'EOS' else self.pad_token
        self.pieces = sorted(os.path.join(data_dir, p) for p in pieces) if pieces else sorted(glob.glob(os.path.join(data_dir, '*.pkl')))
        kind = np.array([self.idx2event[i].split('_')[0] for i in range(self.pad_token)] + ['Pad'])
        self._is_chord, self._is_note = (kind == 'Chord').astype(np.int64), (kind == 'Note').astype(np.int64)
        self.piece_melody_pos, self.piece_chord_pos, self.piece_admissible_stbars = [], [], []
        for i, path in enumerate(self.pieces):
            lead_pos, full_pos, events = self._read(path)
            if i % 200 == 0:
                print('[data] indexing piece %d / %d' % (i, len(self.pieces)))
            self.piece_melody_pos.append(lead_pos)
            self.piece_chord_pos.append(full_pos)
            self.piece_admissible_stbars.append(self._start_bars(lead_pos, len(events)))

    @staticmethod
    def _read(path):
        with open(path, 'rb') as f:
            return pickle.load(f)[:3]

    def _start_bars(self, lead_pos, n_events):
        """Bars a sample may start at: bar 0 for a piece that fits; otherwise the leading run of bars that still leave at least half a
        window of events."""
        if n_events <= self.model_dec_seqlen:
            return [0]
        ok = []
        for bar, (start, _) in enumerate(lead_pos):
            if n_events - start < 0.5 * self.model_dec_seqlen:
                break
            ok.append(bar)
        return ok

    def __len__(self):
        return len(self.pieces)

    def _targets(self, inp, lead_pos, full_pos, st_bar):
        tgt = np.full_like(inp, self.pad_token)
        mask = np.zeros_like(inp)
        if self.predict_key:                                    # [emotion, key, ...]: the key is predicted from the emotion tag
            mask[0], mask[1], tgt[0] = 2, 3, inp[1]
        shift = lead_pos[0][0] - lead_pos[st_bar][0]           # the events between the header and the start bar were cut out
        last = len(lead_pos) - 1
        for bar in range(st_bar, len(lead_pos)):
            a, b = full_pos[bar][0] + shift, full_pos[bar][1] + shift
            mask[a:b] = 1
            if bar != last:
                tgt[a:b] = inp[a + 1:b + 1]
            else:
                tgt[a:b - 1] = inp[a + 1:b]
                tgt[b - 1] = self.eos_token
        return tgt, mask

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        events = self._read(self.pieces[idx])[2]
        lead_pos, full_pos = self.piece_melody_pos[idx], self.piece_chord_pos[idx]
        assert len(lead_pos) == len(full_pos)
        st_bar = random.choice(self.piece_admissible_stbars[idx])
        ids = [self.event2idx[_event_name(e)] for e in events[:lead_pos[0][0]] + events[lead_pos[st_bar][0]:]]
        length, W = len(ids), self.model_dec_seqlen
        if self.pad_to_same and length < W:
            ids = ids + [self.pad_token] * (W - length)
        inp = np.array(ids, dtype=int)
        tgt, mask = self._targets(inp, lead_pos, full_pos, st_bar)
        return {'id': idx, 'piece_id': os.path.basename(self.pieces[idx]).replace('.pkl', ''), 'dec_input': inp[:W], 'dec_target': tgt[:W],
                'chords_mhot': 0, 'track_mask': mask[:W], 'length': min(length, W), 'chord_idx': self._is_chord[tgt][:W],
                'melody_idx': self._is_note[tgt][:W]}


REMISkylineToMidiTransformerDataset = EventPieceDataset          # the reference's class name (dataloader.py:41), for drop-in imports


def load_split(path):
    """train.pkl / valid.pkl: a pickled list of piece file names."""
    with open(path, 'rb') as f:
        return pickle.load(f)
