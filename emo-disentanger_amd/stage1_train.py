"""Stage-1 training / validation loop — /root/reference/stage1_compose/train.py (train :19-115, validate :118-158, log_epoch :161-178,
compute_accuracy :181-190) around the HIP PlainTransformer.  The reference keeps its schedule state in module globals (train_steps,
warmup_steps, max_lr, log_interval, ckpt_dir, log_file, init_time); here they live in a Stage1Config / Stage1State pair.
One optimizer step per segment, gradient clip 0.5, linear warm-up of param_groups[0] then the closed-form CosineAnnealingLR via
sched.step(train_steps - warmup_steps) (:65-77).  Additions the reference lacks: arg-max accuracies counted on the device, the
data-parallel gradient exchange (dp.py) and the fused clip + Adam optimizer when `optim` is an optim.FusedAdam."""
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch


@dataclass
class Stage1Config:
    warmup_steps: int = 200
    max_lr: float = 1.0e-5
    log_interval: int = 50
    ckpt_dir: str = 'ckpt/stage1'
    log_file: str = 'log.txt'
    verbose: bool = True

    @classmethod
    def from_yaml(cls, config, representation, **kw):
        tr = config['training']
        start_epoch = 0 if tr['trained_epochs'] is None else tr['trained_epochs']
        return cls(warmup_steps=tr['warmup_steps'], max_lr=tr['max_lr'], log_interval=tr['log_interval'],
                   ckpt_dir=config['output']['ckpt_dir'].format(representation),
                   log_file='log.txt' if start_epoch == 0 else 'log_from_ep{:03d}.txt'.format(start_epoch), **kw)


@dataclass
class Stage1State:
    train_steps: int = 0
    init_time: float = field(default_factory=time.time)


def compute_accuracy(dec_logits, dec_target, inp_chord, inp_melody, pad_token):
    """(total, chord, melody, others) accuracy of the arg-max predictions, the quantities of stage1_compose/train.py:181-190, counted
    on the device by emo_accuracy_counts (six integers come back instead of the logits).  dec_logits [T, B, V], dec_target [T, B];
    inp_chord / inp_melody: [B, T] 0/1 masks as the dataloader hands them.  A class without tokens gives nan (mean of nothing)."""
    from . import ops
    T, B, V = dec_logits.shape
    dev = dec_logits.device

    def tb(mask):
        return torch.as_tensor(mask).to(dev).long().t().reshape(-1)
    n = ops.accuracy_counts(dec_logits.detach().reshape(T * B, V).float().contiguous(), dec_target.reshape(-1), tb(inp_chord), tb(inp_melody),
                            pad_token).cpu().numpy().astype(np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        total, chord, melody = n[1] / n[0], n[3] / n[2], n[5] / n[4]
        others = (total * n[0] - chord * n[2] - melody * n[4]) / (n[0] - n[2] - n[4])
    return total, chord, melody, others


def log_epoch(log_file, log_data, init_time, is_init=False):
    """Five left-aligned columns (widths 4/8/12/12/12): ep, steps, ce_loss (5 decimals), ep_time, total_time — the log.txt contract."""
    head = ('ep', 'steps', 'ce_loss', 'ep_time', 'total_time')
    row = (log_data['ep'], log_data['steps'], round(log_data['ce_loss'], 5), round(log_data['time'], 2), round(time.time() - init_time, 2))
    with open(log_file, 'w' if is_init else 'a') as f:
        if is_init:
            f.write('{:4} {:8} {:12} {:12} {:12}\n'.format(*head))
        f.write('{:<4} {:<8} {:<12} {:<12} {:<12}\n'.format(*row))


def setup_data_parallel(model):
    """BASELINE configs[4] trains stage 1 data-parallel (the reference has no distributed code): one process per GPU, rank 0's weights
    broadcast once, per-rank dropout streams; every rank iterates its own shard of the batches.  Returns (rank, world)."""
    from . import dp
    rank, _, world = dp.init_distributed()
    if world > 1:
        dp.sync_model_from_rank0(model)
    return rank, world


def _n_segments(batch, synced=False):
    """Segments of this batch.  The collate sets n_seg to the max over the batch's samples (stage1_compose/dataloader.py:195), so ranks
    holding different pieces would run different numbers of optimizer steps — and one gradient all-reduce belongs to each step.  Under
    data parallelism (synced=True) the count is the MAX over ranks (control plane); a rank without that segment runs a zero-token step."""
    from . import dp
    n = int(max(batch['n_seg']))
    if synced and dp.data_plane() is not None:
        n = int(dp.max_over_ranks(n))
    return n


def _segments(batch, dev, synced=False):
    """The dataloader packs a piece as n_seg segments: dec_inp_i / dec_tgt_i [B, T] (time-major on the device), dec_seg_len_i, masks.
    Yields None for a segment index this rank's batch does not have (only with synced=True, see _n_segments)."""
    for i in range(_n_segments(batch, synced)):
        if ('dec_inp_%d' % i) not in batch:
            yield None
            continue
        yield (batch['dec_inp_%d' % i].t().to(dev), batch['dec_tgt_%d' % i].t().to(dev), batch['dec_seg_len_%d' % i].to(dev),
               batch['inp_chord_%d' % i], batch['inp_melody_%d' % i])


def _backward_and_exchange(model, loss, dec_target, pad_token):
    """backward() + the one DP exchange, placed before the clip so that it sees the global gradient.  With more than one rank every
    rank back-propagates the SUM of its token losses; the non-pad count travels in the tail slot of the all-reduced buffer and the
    caller (or the fused optimizer) divides by it: the exact global token mean."""
    from . import dp
    if dp.data_plane() is None:
        loss.backward()
        return
    store = model._ensure_store()
    n_tok = 0 if dec_target is None else int((dec_target != pad_token).sum().item())
    if n_tok > 0:
        (loss * float(n_tok)).backward()
    else:
        # no segment left on this rank, or a segment that is all padding: the mean over zero tokens is NaN and NaN * 0 would poison
        # every replica through the sum — contribute an exactly-zero gradient and a zero token count instead
        store.ensure_grads()
        store.flat_grad.zero_()
    dp.allreduce_grads_(store, float(n_tok))


def train(epoch, model, dloader, optim, sched, pad_token, cfg, state):
    """One epoch: an optimizer step per segment (zero_grad, forward with the running memory, loss, backward, clip 0.5, step), linear
    warm-up to max_lr over warmup_steps then sched.step(steps - warmup_steps), a log line every log_interval steps.  Returns
    (sample-weighted mean ce_loss, seconds) — reference loop: stage1_compose/train.py:19-115, pinned by tests/golden/txl_trainloop.json."""
    from . import dp
    from .optim import FusedAdam
    model.train()
    fused = isinstance(optim, FusedAdam)
    dev = next(model.parameters()).device
    note = print if cfg.verbose else (lambda *a, **k: None)
    loss_sum, n_samples, t0 = 0.0, 0, time.time()
    accs = (float('nan'),) * 4
    for b_idx, batch in enumerate(dloader):
        mems = tuple()
        bsz = batch['id'].size(0)
        for seg in _segments(batch, dev, synced=True):
            optim.zero_grad() if fused else model.zero_grad()
            state.train_steps += 1
            if seg is not None:
                dec_input, dec_target, seg_len, chord, melody = seg
                dec_logits, mems = model(dec_input, mems, dec_seg_len=seg_len)
                losses = model.compute_loss(dec_logits, dec_target)
                accs = compute_accuracy(dec_logits, dec_target, chord, melody, pad_token)
                _backward_and_exchange(model, losses['total_loss'], dec_target, pad_token)
            else:                                                   # another rank still has a segment: take part in its step (same collectives)
                _backward_and_exchange(model, None, None, pad_token)
            if fused:
                optim.step()                                        # clip + 1/tokens (or 1/world) folded into the fused Adam
            else:
                if dp.data_plane() is not None:
                    st = model._ensure_store()
                    st.flat_grad.div_(st.flat_grad_ext[st.total].clamp_min(1.0))   # global count 0 -> the gradient is exactly zero anyway
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
                optim.step()
            if seg is not None:
                ce = losses['ce_loss'].item()
                if ce == ce or dp.data_plane() is None:             # (DP: an all-pad segment has no loss — nan, not counted)
                    loss_sum += bsz * ce
                    n_samples += bsz
            if state.train_steps < cfg.warmup_steps:
                optim.param_groups[0]['lr'] = cfg.max_lr * state.train_steps / cfg.warmup_steps
            else:
                sched.step(state.train_steps - cfg.warmup_steps)
            if state.train_steps % cfg.log_interval == 0:
                lf = os.path.join(cfg.ckpt_dir, cfg.log_file)
                log_epoch(lf, {'ep': epoch, 'steps': state.train_steps, 'ce_loss': loss_sum / max(n_samples, 1), 'time': time.time() - t0},
                          state.init_time, is_init=not os.path.exists(lf))
        note('[stage1 train] ep %d batch %d: ce %.4f | acc %.4f (chord %.4f, melody %.4f, others %.4f) | step %d | %.1f s'
             % ((epoch, b_idx, loss_sum / max(n_samples, 1)) + tuple(accs) + (state.train_steps, time.time() - t0)))
    return loss_sum / max(n_samples, 1), time.time() - t0


def validate(epoch, model, dloader, pad_token, rounds=1, verbose=True):
    """-> five lists (ce_loss, total / chord / melody / others accuracy), one entry per segment — train.py:118-158."""
    model.eval()
    dev = next(model.parameters()).device
    out = ([], [], [], [], [])
    if verbose:
        print('[stage1 validate] ep %d' % epoch)
    with torch.no_grad():
        for _ in range(rounds):
            for batch in dloader:
                mems = tuple()
                for dec_input, dec_target, seg_len, chord, melody in _segments(batch, dev):
                    dec_logits, mems = model(dec_input, mems, dec_seg_len=seg_len)
                    vals = (model.compute_loss(dec_logits, dec_target)['ce_loss'].item(),) + tuple(compute_accuracy(dec_logits, dec_target, chord, melody, pad_token))
                    for lst, v in zip(out, vals):
                        lst.append(v)
    return out
