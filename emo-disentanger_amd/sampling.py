"""Host-side token sampling with the reference's observable behaviour, written from its specification:

temperature  (stage2 inference.py:71-83, stage1 inference_utils.py:14-24)
    p = exp(l / t) / sum(exp(l / t)) evaluated in the logits' own dtype (float32 from the model); when that produces a NaN
    (overflow) the computation is repeated in 128-bit floats and returned as float64.  Optional `inadmissibles` get -inf first.
nucleus      (inference.py:86-100, inference_utils.py:27-41)
    renormalise by the left-to-right sum of the array (same dtype), sort descending, running sum; the candidate set ends ONE
    position after the first prefix whose mass exceeds p (the reference indexes the SECOND crossing: the crossing token is kept,
    and a distribution with exactly one crossing raises IndexError — tests/golden/sampling.json pins both); if no prefix exceeds
    p the top three are taken; the candidates are renormalised in float64 and one is drawn with NumPy's global RNG (or `rng`).

Bit-exactness notes: np.cumsum accumulates strictly left to right in the array's dtype, which is what Python's built-in sum()
over a NumPy array does, while np.sum() is pairwise; ties in the ranking come from np.argsort's default kind, like the reference."""
import numpy as np


def _sequential_total(a):
    return np.cumsum(a)[-1]


def temperature(logits, temperature, inadmissibles=None, longdouble_softmax=False):
    """longdouble_softmax: stage 1 finishes the overflow path with a max-shifted softmax (scipy.special.softmax there)."""
    if inadmissibles is not None:
        logits[inadmissibles] -= np.inf
    with np.errstate(over='ignore', invalid='ignore'):
        e = np.exp(logits / temperature)
        probs = e / np.sum(e)
    if np.isnan(probs).any():
        print('overflow detected, use 128-bit')
        z = logits.astype(np.float128) / temperature
        if longdouble_softmax:
            z = z - z.max()
        e = np.exp(z)
        probs = (e / np.sum(e)).astype(float)
        if longdouble_softmax:
            assert not np.isnan(probs).any()
    return probs


def nucleus_candidates(probs, p):
    """-> (candidate ids in rank order, their float64 weights).  `probs` is renormalised IN PLACE like the reference does."""
    probs /= _sequential_total(probs)
    order = np.argsort(probs)[::-1]
    mass = np.cumsum(np.sort(probs)[::-1])
    crossings = np.flatnonzero(mass > p)
    keep = crossings[1] if crossings.size else 3          # IndexError when there is exactly one crossing: reference behaviour (F12)
    ids = order[:keep]
    w = probs[ids].astype(np.float64)
    return ids, w / _sequential_total(w)


def nucleus(probs, p, rng=None):
    ids, w = nucleus_candidates(probs, p)
    return (np.random if rng is None else rng).choice(ids, size=1, p=w)[0]


def beat_position(event):
    """'Beat_7' -> 7"""
    return int(event.rsplit('_', 1)[1])
