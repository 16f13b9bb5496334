"""Stage-1 training / validation loop — /root/reference/stage1_compose/train.py (train :19-115, validate :118-158, log_epoch :161-178,
compute_accuracy :181-190) around the HIP PlainTransformer.  The reference keeps its schedule state in module globals (train_steps,
warmup_steps, max_lr, log_interval, ckpt_dir, log_file, init_time); here they live in a Stage1Config / Stage1State pair.
One optimizer step per segment, gradient clip 0.5, linear warm-up of param_groups[0] then the closed-form CosineAnnealingLR via
sched.step(train_steps - warmup_steps), exactly as :65-77."""
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
    """train.py:181-190.  dec_logits [T, B, V], dec_target [T, B]; inp_chord / inp_melody [B, T] masks (as the dataloader hands them)."""
    dec_pred = torch.argmax(dec_logits, dim=-1).permute(1, 0).cpu()
    dec_target = dec_target.permute(1, 0).cpu()
    inp_chord, inp_melody = torch.as_tensor(inp_chord).cpu(), torch.as_tensor(inp_melody).cpu()
    with np.errstate(invalid='ignore', divide='ignore'):
        total_acc = np.mean(np.array((dec_pred[dec_target != pad_token] == dec_target[dec_target != pad_token])))
        chord_acc = np.mean(np.array((dec_pred[inp_chord == 1] == dec_target[inp_chord == 1])))
        melody_acc = np.mean(np.array((dec_pred[inp_melody == 1] == dec_target[inp_melody == 1])))
        n_tot, n_ch, n_me = len(dec_target[dec_target != pad_token]), len(dec_target[inp_chord == 1]), len(dec_target[inp_melody == 1])
        others_acc = (total_acc * n_tot - chord_acc * n_ch - melody_acc * n_me) / (n_tot - n_ch - n_me)
    return total_acc, chord_acc, melody_acc, others_acc


def log_epoch(log_file, log_data, init_time, is_init=False):
    if is_init:
        with open(log_file, 'w') as f:
            f.write('{:4} {:8} {:12} {:12} {:12}\n'.format('ep', 'steps', 'ce_loss', 'ep_time', 'total_time'))
    with open(log_file, 'a') as f:
        f.write('{:<4} {:<8} {:<12} {:<12} {:<12}\n'.format(log_data['ep'], log_data['steps'], round(log_data['ce_loss'], 5), round(log_data['time'], 2),
                                                            round(time.time() - init_time, 2)))


def setup_data_parallel(model):
    """BASELINE configs[4] trains stage 1 data-parallel (the reference itself has no distributed code): one process per GPU, rank 0's weights
    broadcast once, per-rank dropout streams; every rank iterates its own shard of the batches (the caller's sampler).  Returns (rank, world)."""
    from . import dp
    rank, _, world = dp.init_distributed()
    if world > 1:
        dp.sync_model_from_rank0(model)
    return rank, world


def _average_gradients(model):
    """The one exchange of the stage-1 DP step: sum all-reduce of the flat fp32 gradient buffer (the parameters' .grad are views into it),
    scaled by 1 / world, placed between backward() and the clip so that the clip sees the global gradient (train.py:60-61)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        from . import dp
        flat = model._ensure_store().flat_grad
        dp.allreduce_sum_(flat)
        flat.mul_(1.0 / dist.get_world_size())


def train(epoch, model, dloader, optim, sched, pad_token, cfg, state):
    model.train()
    recons_loss_rec, accum_samples = 0., 0
    say = print if cfg.verbose else (lambda *a, **k: None)
    say('[train] epoch %d' % epoch)
    st = time.time()
    dev = next(model.parameters()).device
    for batch_idx, batch_samples in enumerate(dloader):
        mems = tuple()
        for segment in range(max(batch_samples['n_seg'])):
            model.zero_grad()
            dec_input = batch_samples['dec_inp_{}'.format(segment)].permute(1, 0).to(dev)
            dec_target = batch_samples['dec_tgt_{}'.format(segment)].permute(1, 0).to(dev)
            dec_seg_len = batch_samples['dec_seg_len_{}'.format(segment)].to(dev)
            inp_chord, inp_melody = batch_samples['inp_chord_{}'.format(segment)], batch_samples['inp_melody_{}'.format(segment)]
            state.train_steps += 1
            dec_logits, mems = model(dec_input, mems, dec_seg_len=dec_seg_len)
            losses = model.compute_loss(dec_logits, dec_target)
            total_acc, chord_acc, melody_acc, others_acc = compute_accuracy(dec_logits.detach(), dec_target, inp_chord, inp_melody, pad_token)
            losses['total_loss'].backward()
            _average_gradients(model)
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
            optim.step()
            recons_loss_rec += batch_samples['id'].size(0) * losses['ce_loss'].item()
            accum_samples += batch_samples['id'].size(0)
            if state.train_steps < cfg.warmup_steps:
                optim.param_groups[0]['lr'] = cfg.max_lr * state.train_steps / cfg.warmup_steps
            else:
                sched.step(state.train_steps - cfg.warmup_steps)
            if not state.train_steps % cfg.log_interval:
                lf = os.path.join(cfg.ckpt_dir, cfg.log_file)
                log_epoch(lf, {'ep': epoch, 'steps': state.train_steps, 'ce_loss': recons_loss_rec / accum_samples, 'time': time.time() - st},
                          state.init_time, is_init=not os.path.exists(lf))
        say('[train] epoch %d batch %d  ce %.4f  acc all/chord/melody/other %.4f %.4f %.4f %.4f  step %d  %.1f s' %
            (epoch, batch_idx, recons_loss_rec / accum_samples, total_acc, chord_acc, melody_acc, others_acc, state.train_steps, time.time() - st))
    return recons_loss_rec / accum_samples, time.time() - st


def validate(epoch, model, dloader, pad_token, rounds=1, verbose=True):
    model.eval()
    rec = [[], [], [], [], []]
    dev = next(model.parameters()).device
    if verbose:
        print('[validate] epoch %d' % epoch)
    with torch.no_grad():
        for r in range(rounds):
            for batch_idx, batch_samples in enumerate(dloader):
                mems = tuple()
                for segment in range(max(batch_samples['n_seg'])):
                    dec_input = batch_samples['dec_inp_{}'.format(segment)].permute(1, 0).to(dev)
                    dec_target = batch_samples['dec_tgt_{}'.format(segment)].permute(1, 0).to(dev)
                    dec_seg_len = batch_samples['dec_seg_len_{}'.format(segment)].to(dev)
                    dec_logits, mems = model(dec_input, mems, dec_seg_len=dec_seg_len)
                    losses = model.compute_loss(dec_logits, dec_target)
                    accs = compute_accuracy(dec_logits, dec_target, batch_samples['inp_chord_{}'.format(segment)],
                                            batch_samples['inp_melody_{}'.format(segment)], pad_token)
                    rec[0].append(losses['ce_loss'].item())
                    for i, a in enumerate(accs):
                        rec[i + 1].append(a)
    return tuple(rec)
