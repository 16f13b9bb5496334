"""ORACLE (test infrastructure, NOT product code).

NumPy restatement of the host-side pieces of the path: temperature / nucleus
sampling, the ``generate_conditional`` control flow, ``compute_accuracy`` and
the LR schedule.  All of these are PINNED by golden vectors produced from the
imported reference (tools/make_golden.py -> tests/golden/).

Reference citations (relative to /root/reference/stage2_accompaniment):
  temperature            inference.py:71-83
  nucleus                inference.py:86-100   (keeps the token that crosses p — F12)
  generate_conditional   inference.py:231-327
  compute_accuracy       train.py:184-193
  LR schedule            train.py:99-104  (+ torch CosineAnnealingLR closed form)
"""
import math

import numpy as np


def temperature(logits, temp, inadmissibles=None):
    logits = np.array(logits, copy=True)
    if inadmissibles is not None:
        logits[inadmissibles] -= np.inf
    with np.errstate(over='ignore', invalid='ignore'):
        e = np.exp(logits / temp)
        probs = e / np.sum(e)
    if np.count_nonzero(np.isnan(probs)) != 0:
        l128 = logits.astype(np.float128)
        e = np.exp(l128 / temp)
        probs = (e / np.sum(e)).astype(float)
    return probs


def nucleus_candidates(probs, p):
    """Deterministic part of nucleus(): returns (candidate index array, renormalised f64 probs).
    Raises IndexError exactly where the reference does (only the last sorted token crosses p)."""
    probs = np.array(probs, copy=True)
    probs /= sum(probs)                      # python sum: sequential accumulation in probs.dtype
    sorted_probs = np.sort(probs)[::-1]
    sorted_index = np.argsort(probs)[::-1]
    cusum = np.cumsum(sorted_probs)
    after = cusum > p
    if sum(after) > 0:
        last_index = np.where(after)[0][1]
        candi_index = sorted_index[:last_index]
    else:
        candi_index = sorted_index[:3]
    candi_probs = np.array([probs[i] for i in candi_index], dtype=np.float64)
    candi_probs /= sum(candi_probs)
    return candi_index, candi_probs


def nucleus(probs, p, rng=np.random):
    idx, pr = nucleus_candidates(probs, p)
    return rng.choice(idx, size=1, p=pr)[0]


def get_position_idx(event):
    return int(event.split('_')[-1])


def generate_conditional(logits_fn, event2idx, idx2event, lead_sheet_events, primer,
                         max_events=10000, skip_check=False, max_bars=None, temp=1.2, top_p=0.9,
                         inadmissibles=None, max_dec_inp_len=2048, sampler=None, trace=None):
    """Control-flow restatement.  ``logits_fn(tokens, segs) -> np.ndarray[V]`` is the
    last-position logits of the model over the (<=2048-token) window; ``sampler(probs)``
    defaults to nucleus with NumPy's global RNG (as the reference)."""
    generated = primer + [event2idx['Track_LeadSheet']] + lead_sheet_events[0] + [event2idx['Track_Full']]
    seg_inp = [0 for _ in range(len(generated))]
    seg_inp[-1] = 1
    target_bars, generated_bars = len(lead_sheet_events), 0
    if max_bars is not None:
        target_bars = min(max_bars, target_bars)
    steps, cur_pos, failed_cnt = 0, 0, 0
    while generated_bars < target_bars:
        assert len(generated) == len(seg_inp)
        logits = np.asarray(logits_fn(generated[-max_dec_inp_len:], seg_inp[-max_dec_inp_len:]))
        probs = temperature(logits, temp, inadmissibles=inadmissibles)
        word = int(sampler(probs) if sampler is not None else nucleus(probs, top_p))
        if trace is not None:
            trace.append(word)
        word_event = idx2event[word]
        if not skip_check:
            if 'Beat' in word_event:
                event_pos = get_position_idx(word_event)
                if not event_pos >= cur_pos:
                    failed_cnt += 1
                    if failed_cnt >= 256:
                        return generated
                    continue
                else:
                    cur_pos = event_pos
                    failed_cnt = 0
        if word_event == 'Track_LeadSheet':
            steps += 1
            generated.append(word)
            seg_inp.append(0)
            generated_bars += 1
            if generated_bars < target_bars:
                generated.extend(lead_sheet_events[generated_bars])
                seg_inp.extend([0 for _ in range(len(lead_sheet_events[generated_bars]))])
                generated.append(event2idx['Track_Full'])
                seg_inp.append(1)
                cur_pos = 0
            continue
        if word_event == 'PAD_None' or (word_event == 'EOS_None' and generated_bars < target_bars - 1):
            continue
        elif word_event == 'EOS_None' and generated_bars == target_bars - 1:
            generated.append(word)
            break
        generated.append(word)
        seg_inp.append(1)
        steps += 1
        if len(generated) > max_events:
            break
    return generated[:-1]


def compute_accuracy(dec_logits, dec_target, inp_chord, inp_melody, pad_token):
    """train.py:184-193 on NumPy arrays (NaN when a class is empty, as np.mean of empty)."""
    dec_logits, dec_target = np.asarray(dec_logits), np.asarray(dec_target)
    inp_chord, inp_melody = np.asarray(inp_chord), np.asarray(inp_melody)
    pred = np.argmax(dec_logits, axis=-1)
    nonpad = dec_target != pad_token
    with np.errstate(invalid='ignore', divide='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            total = np.mean(pred[nonpad] == dec_target[nonpad])
            chord = np.mean(pred[inp_chord == 1] == dec_target[inp_chord == 1])
            melody = np.mean(pred[inp_melody == 1] == dec_target[inp_melody == 1])
            n_t, n_c, n_m = nonpad.sum(), (inp_chord == 1).sum(), (inp_melody == 1).sum()
            others = (total * n_t - chord * n_c - melody * n_m) / (n_t - n_c - n_m)
    return total, chord, melody, others


def lr_at_step(train_steps, max_lr, eta_min, warmup_steps, T_max, accum_steps=1):
    """LR in effect AFTER the schedule update at the end of step `train_steps` (train.py:99-104)."""
    if (train_steps // accum_steps) < warmup_steps:
        return max_lr * train_steps / (warmup_steps * accum_steps)
    k = train_steps // accum_steps - warmup_steps
    return eta_min + (max_lr - eta_min) * (1 + math.cos(math.pi * k / T_max)) / 2
