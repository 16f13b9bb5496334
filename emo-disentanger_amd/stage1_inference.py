"""Stage-1 lead-sheet sampling on PlainTransformer.generate (K/V memory on the HIP relative-position decode kernel).

Behaviour follows /root/reference/stage1_compose/inference_utils.py (generate_plain_xl :51-134, match_emotion_key :137-142) as
pinned by traces of the real loop (tests/golden/txl_generate.json): the first sampled event of the functional / key
representations is the key (temperature 1.1, p 0.97; under key_determine='rule' a non-Key event is an error and a key whose mode
contradicts the emotion is rejected), Beat positions never decrease inside a bar, 256 consecutive rejected samples abort, PAD is
never appended, and a rejected sample makes the loop feed its previous input again — which appends that input to the model's
memory a second time (a reference quirk the traces contain, kept on purpose)."""
import time

import numpy as np
import torch

from . import sampling
from .sampling import beat_position, nucleus  # noqa: F401  (`nucleus` is looked up at call time: tests wrap it)

SHARP_NAMES = ('C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B')     # pitch-class spelling of convert_key.py:14-15
MAJOR_KEY = np.array(SHARP_NAMES)
MINOR_KEY = np.array([n.lower() for n in SHARP_NAMES])
_MAJOR_MOODS, _MINOR_MOODS = ('Q1', 'Q4', 'Positive'), ('Q2', 'Q3', 'Negative')


def temperature(logits, temperature):
    return sampling.temperature(logits, temperature, longdouble_softmax=True)


def match_emotion_key(emotion, key):
    """High-valence labels go with major keys (upper-case tonic), low-valence ones with minor keys (lower-case tonic)."""
    return (emotion in _MAJOR_MOODS and key in MAJOR_KEY) or (emotion in _MINOR_MOODS and key in MINOR_KEY)


class _LeadSheet:
    """Token list + grammar state of one lead sheet being written."""

    def __init__(self, event2idx, primer, prompt_bars, max_bars, max_events):
        self.tokens = [event2idx['Bar_None']] if primer is None else [event2idx[e] for e in primer]
        self.bars = 0 if (primer is None or prompt_bars is None) else prompt_bars
        self.max_bars, self.max_events = max_bars, max_events
        self.accepted, self.beat, self.rejected_in_a_row = 0, 0, 0
        self.finished, self.stuck = False, False

    def open(self):
        return not self.finished and self.bars < self.max_bars

    def offer(self, word, name):
        """True if `word` (event `name`) was appended."""
        if 'Beat' in name:
            pos = beat_position(name)
            if pos < self.beat:
                self.rejected_in_a_row += 1
                if self.rejected_in_a_row >= 256:
                    self.finished = self.stuck = True
                return False
            self.beat, self.rejected_in_a_row = pos, 0
        if 'Bar' in name:
            self.bars += 1
            self.beat = 0
        if name == 'PAD_None':
            return False
        self.tokens.append(int(word))
        self.accepted += 1
        if len(self.tokens) > self.max_events or name == 'EOS_None':
            self.finished = True
        return True


def generate_plain_xl(model, event2idx, idx2event, max_bars=160, max_events=2048, primer=None, temp=1.2, top_p=0.9, prompt_bars=None,
                      representation='functional', key_determine=None, sampler=None, verbose=False):
    """-> (token ids without the last one, seconds), or (None, seconds) when the model got stuck."""
    note = print if verbose else (lambda *a, **k: None)
    pick = (lambda probs, p: sampler(probs)) if sampler is not None else (lambda probs, p: nucleus(probs, p))
    sheet = _LeadSheet(event2idx, primer, prompt_bars, max_bars, max_events)
    dev = next(model.parameters()).device
    keyed = representation in ('functional', 'key')
    mems = tuple()
    t0 = time.time()
    while sheet.open():
        feed = sheet.tokens if sheet.accepted == 0 else sheet.tokens[-1:]
        logits, mems = model.generate(torch.tensor(feed, dtype=torch.long, device=dev).view(len(feed), 1), mems)
        logits = logits.cpu().numpy()
        if keyed and len(sheet.tokens) == 1:                     # the event after the emotion tag is the key
            word = pick(temperature(logits, 1.1), 0.97)
            name = idx2event[word]
            if key_determine == 'rule':
                kind, _, tonic = name.partition('_')
                if kind != 'Key':
                    raise ValueError('[info] key generation failed')
                if not match_emotion_key(idx2event[sheet.tokens[0]].split('_')[1], tonic):
                    continue
        else:
            word = pick(temperature(logits, temp), top_p)
            name = idx2event[word]
        took = sheet.offer(word, name)
        if sheet.stuck:
            note('[stage1 gen] stuck after 256 rejected samples')
            return None, time.time() - t0
        if took and ('Bar' in name or 'Key' in name):
            note('[stage1 gen] %s: %d bars, %d events' % (name, sheet.bars, len(sheet.tokens)))
    note('[stage1 gen] %d events in %.2f s' % (len(sheet.tokens), time.time() - t0))
    return sheet.tokens[:-1], time.time() - t0
