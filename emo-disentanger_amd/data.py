"""Synthetic EMOPIA-shaped batches; dict keys / dtypes follow the reference dataset's
``__getitem__`` (/root/reference/stage2_accompaniment/dataloader.py:221-231) after default collation.
The real dataset class is out of scope (host-side pickles the repo does not ship)."""
import numpy as np
import torch

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
