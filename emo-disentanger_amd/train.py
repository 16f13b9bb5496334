"""Training harness — same step, log formats, checkpoint naming and YAML keys as
/root/reference/stage2_accompaniment/train.py (train_model :32-131, validate :134-181,
compute_accuracy :184-193, __main__ :196-356), with the module globals of the reference gathered in
``TrainConfig`` and two additions the reference lacks: the data-parallel gradient all-reduce between
backward() and the clip and device-side accuracy (no full-logits D2H, SURVEY f-2).

Reference quirk kept on purpose: ``model.zero_grad()`` runs at the top of EVERY
micro-batch, so with accum_steps > 1 only the last micro-batch of a window (scaled 1/accum) is
applied.  ``TrainConfig.faithful_accum=False`` gives real accumulation instead.
"""
import argparse
import math
import os
import random
import shutil
import time

import numpy as np
import torch
import yaml

from . import dp, ops
from .optim import FusedAdam


class TrainConfig:
    def __init__(self, gpuid=0, warmup_steps=200, max_lr=1e-4, min_lr=1e-5, lr_decay_steps=500000, redraw_prob=0.0, accum_steps=1,
                 log_interval=50, ckpt_dir='ckpt', ckpt_interval=10, max_epochs=1000, world_size=1, faithful_accum=True, verbose=True):
        self.__dict__.update(locals())
        del self.__dict__['self']
        self.train_steps = 0
        self.window_tokens = None         # token count of the open accumulation window (kept across epochs)
        self._rng = None

    def shared_rng(self):
        """The FAVOR+ redraw decision (`random() > feat_redraw_prob`, train.py:61) must fall the same way on every rank, or the
        replicas' omega generators — seeded identically by dp.sync_model_from_rank0 — drift apart.  Single process: the `random`
        module, as in the reference; data parallel: one generator per process seeded with a value rank 0 broadcasts."""
        if self._rng is None:
            if self.world_size > 1 and dp.data_plane() is not None:
                import torch.distributed as dist
                seed = torch.tensor([random.getrandbits(62)], dtype=torch.int64)
                dist.broadcast(seed, src=0)                           # control plane (gloo)
                self._rng = random.Random(int(seed))
            else:
                self._rng = random
        return self._rng

    @classmethod
    def from_yaml(cls, conf, representation='functional', **kw):
        t = conf['training']
        return cls(gpuid=t['gpuid'], warmup_steps=t['warmup_steps'], max_lr=t['lr'], min_lr=t['lr_scheduler']['eta_min'],
                   lr_decay_steps=t['lr_scheduler']['T_max'], redraw_prob=t.get('feat_redraw_prob', 0.0), accum_steps=t.get('accum_steps', 1),
                   log_interval=t['log_interval'], ckpt_dir=t['ckpt_dir'].format(representation), ckpt_interval=t['ckpt_interval'],
                   max_epochs=t['num_epochs'], **kw)


def log_epoch(log_file, log_data, is_init=False):
    if is_init:
        with open(log_file, 'w') as f:
            f.write('{:4} {:8} {:12} {:12}\n'.format('ep', 'steps', 'recons_loss', 'ep_time'))
    with open(log_file, 'a') as f:
        f.write('{:<4} {:<8} {:<12} {:<12}\n'.format(log_data['ep'], log_data['steps'], round(log_data['recons_loss'], 5), round(log_data['time'], 2)))


def lr_after_step(train_steps, cfg):
    """LR in effect after the schedule update that ends step `train_steps` (train.py:99-104; the cosine branch is the
    closed form that CosineAnnealingLR.step(epoch) evaluates)."""
    if (train_steps // cfg.accum_steps) < cfg.warmup_steps:
        return cfg.max_lr * train_steps / (cfg.warmup_steps * cfg.accum_steps)
    k = train_steps // cfg.accum_steps - cfg.warmup_steps
    return cfg.min_lr + (cfg.max_lr - cfg.min_lr) * (1 + math.cos(math.pi * k / cfg.lr_decay_steps)) / 2


def compute_accuracy(dec_logits, dec_target, inp_chord, inp_melody, pad_token):
    """train.py:184-193 on the device: argmax + masked compares in one kernel, 6 counters to the host.
    Returns (total_acc, chord_acc, melody_acc, others_acc); empty classes give nan like np.mean([])."""
    V = dec_logits.shape[-1]
    from .engine import padded_logits
    lp = padded_logits(dec_logits.detach()) if dec_logits.dtype == torch.float32 else None     # the padded projection's buffer: pad columns never win the argmax
    c = ops.accuracy_counts(lp if lp is not None else dec_logits.detach().reshape(-1, V).float().contiguous(), dec_target.reshape(-1), inp_chord.reshape(-1),
                            inp_melody.reshape(-1), pad_token).cpu().numpy().astype(np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        total, chord, melody = c[1] / c[0], c[3] / c[2], c[5] / c[4]
        others = (total * c[0] - chord * c[2] - melody * c[4]) / (c[0] - c[2] - c[4])
    return total, chord, melody, others


def _to_dev(t, dev):
    return t.to(dev, non_blocking=True) if torch.is_tensor(t) else t


def train_model(epoch, model, dloader, optim, sched, pad_token, model_type="performer", cfg=None):
    cfg = cfg or TrainConfig()
    model.train()
    dev = next(model.parameters()).device
    # running loss sum: a DEVICE scalar (fp64) — read back only where the reference prints / logs it, so that a quiet run
    # (verbose=False) has no host sync inside the step (r02: .item() + a 6-counter .cpu() per optimizer step)
    recons_loss_rec, accum_samples = torch.zeros((), device=dev, dtype=torch.float64), 0
    say = print if cfg.verbose else (lambda *a, **k: None)
    say('[epoch {:03d}] training ...'.format(epoch))
    say('[epoch {:03d}] # batches = {}'.format(epoch, len(dloader)))
    st = time.time()
    fused = isinstance(optim, FusedAdam)
    exchange = None
    rng = cfg.shared_rng()
    for batch_idx, batch_samples in enumerate(dloader):
        if cfg.faithful_accum or (cfg.train_steps % cfg.accum_steps) == 0:
            optim.zero_grad() if fused else model.zero_grad()
        batch_dec_inp = _to_dev(batch_samples['dec_input'], dev)
        batch_dec_tgt = _to_dev(batch_samples['dec_target'], dev)
        batch_track_mask = _to_dev(batch_samples['track_mask'], dev)
        batch_inp_lens = batch_samples['length']
        batch_chord_idx = _to_dev(batch_samples['chord_idx'], dev)
        batch_melody_idx = _to_dev(batch_samples['melody_idx'], dev)
        cfg.train_steps += 1
        train_steps = cfg.train_steps
        if model_type == "performer":
            omit_feature_map_draw = rng.random() > cfg.redraw_prob
            dec_logits = model(batch_dec_inp, seg_inp=batch_track_mask, chord_inp=None, attn_kwargs={'omit_feature_map_draw': omit_feature_map_draw})
        else:
            omit_feature_map_draw = True
            dec_logits = model(batch_dec_inp, seg_inp=batch_track_mask, chord_inp=None)
        losses = model.compute_loss(dec_logits, batch_dec_tgt)
        total_loss = losses['total_loss'] / cfg.accum_steps if cfg.accum_steps > 1 else losses['total_loss']
        if cfg.world_size > 1:
            # data parallel: back-propagate the SUM of this rank's token losses; the non-pad count travels with the gradient and
            # the optimizer divides by the all-reduced count => the exact global mean even when ranks hold different numbers of
            # non-pad targets (the reference's loss is a mean over non-pad tokens, music_performer.py:72-76)
            n_tok = (batch_dec_tgt != pad_token).sum().to(torch.float32)
            if exchange is None:
                exchange = dp.GradExchange(model)
            if (train_steps % cfg.accum_steps) == 0:
                exchange.arm()                                        # the late layers' all-reduce starts inside this backward
            if cfg.faithful_accum or cfg.accum_steps == 1:            # (reference quirk F11: only the window's last micro-batch survives, scaled 1/accum)
                cfg.window_tokens = n_tok
                (total_loss * n_tok).backward()
            else:                                                     # real accumulation: token-weighted mean over the whole window
                cfg.window_tokens = n_tok if (train_steps - 1) % cfg.accum_steps == 0 or cfg.window_tokens is None else cfg.window_tokens + n_tok
                (losses['total_loss'] * n_tok).backward()
        else:
            total_loss.backward()
        if (train_steps % cfg.accum_steps) == 0:
            if cfg.world_size > 1:
                exchange.finish(cfg.window_tokens)                       # the one exchange per optimizer step, in up to 3 pieces
            if fused:
                optim.step()                                         # clip(0.5) + 1/sum(tokens) folded into the fused Adam
            else:
                if cfg.world_size > 1:
                    st_ = model._store
                    st_.flat_grad.div_(st_.flat_grad_ext[st_.total])
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
                optim.step()
                optim.zero_grad()
            n_b = batch_samples['id'].size(0)
            # (the reference's bookkeeping, quirks included: recons_loss was divided in place by accum_steps when accum>1)
            recons = losses['recons_loss'].detach().double() / (cfg.accum_steps if cfg.accum_steps > 1 else 1)
            recons_loss_rec += n_b * recons * cfg.accum_steps * cfg.accum_steps
            accum_samples += n_b * cfg.accum_steps
            if cfg.verbose:
                total_acc, chord_acc, melody_acc, others_acc = compute_accuracy(dec_logits, batch_dec_tgt, batch_chord_idx, batch_melody_idx, pad_token)
                say(' -- epoch {:03d} | batch {:03d}/{:03d}: len: {}\n   * loss = {:.4f}, total_acc = {:.4f}, chord_acc = {:.4f}, '
                    'melody_acc = {:.4f}, others_acc = {:.4f}, step = {}, time_elapsed = {:.2f} secs | redraw: {}'.format(
                        epoch, batch_idx + 1, len(dloader), batch_inp_lens, float(recons_loss_rec) / accum_samples, total_acc, chord_acc, melody_acc,
                        others_acc, train_steps, time.time() - st, (not omit_feature_map_draw)))
        if (train_steps // cfg.accum_steps) < cfg.warmup_steps:
            optim.param_groups[0]['lr'] = cfg.max_lr * train_steps / (cfg.warmup_steps * cfg.accum_steps)
        elif sched is not None:
            sched.step((train_steps // cfg.accum_steps - cfg.warmup_steps))
        else:
            optim.param_groups[0]['lr'] = lr_after_step(train_steps, cfg)
        if not train_steps % cfg.log_interval:
            log_data = {'ep': epoch, 'steps': train_steps, 'recons_loss': float(recons_loss_rec) / accum_samples, 'time': time.time() - st}
            lf = os.path.join(cfg.ckpt_dir, 'log.txt')
            log_epoch(lf, log_data, is_init=not os.path.exists(lf))
    ep_loss = float(recons_loss_rec) / accum_samples
    say('[epoch {:03d}] training completed\n  -- loss = {:.4f}\n  -- time elapsed = {:.2f} secs.'.format(epoch, ep_loss, time.time() - st))
    log_data = {'ep': epoch, 'steps': cfg.train_steps, 'recons_loss': ep_loss, 'time': time.time() - st}
    lf = os.path.join(cfg.ckpt_dir, 'log.txt')
    log_epoch(lf, log_data, is_init=not os.path.exists(lf))
    return ep_loss


def validate(model, dloader, pad_token, rounds=1, model_type="performer", cfg=None):
    cfg = cfg or TrainConfig()
    model.eval()
    dev = next(model.parameters()).device
    loss_rec, total_acc_rec, chord_acc_rec, melody_acc_rec, others_acc_rec = [], [], [], [], []
    with torch.no_grad():
        for r in range(rounds):
            for batch_idx, bs in enumerate(dloader):
                inp, tgt, seg = _to_dev(bs['dec_input'], dev), _to_dev(bs['dec_target'], dev), _to_dev(bs['track_mask'], dev)
                kw = {'attn_kwargs': {'omit_feature_map_draw': cfg.shared_rng().random() > cfg.redraw_prob}} if model_type == 'performer' else {}
                dec_logits = model(inp, seg_inp=seg, chord_inp=None, **kw)
                losses = model.compute_loss(dec_logits, tgt)
                loss_rec.append(losses['recons_loss'].item())
                a = compute_accuracy(dec_logits, tgt, _to_dev(bs['chord_idx'], dev), _to_dev(bs['melody_idx'], dev), pad_token)
                total_acc_rec.append(a[0]); chord_acc_rec.append(a[1]); melody_acc_rec.append(a[2]); others_acc_rec.append(a[3])
    return loss_rec, total_acc_rec, chord_acc_rec, melody_acc_rec, others_acc_rec


class SyntheticLoader:
    """Stand-in for DataLoader(REMISkylineToMidiTransformerDataset): n_batches synthetic batches per epoch (sharded by rank)."""

    def __init__(self, n_token, batch_size, seq_len, n_batches, seed=1234, rank=0, realistic_targets=True):
        from .data import synthetic_batch
        self.batches = [synthetic_batch(n_token, batch_size, seq_len, seed=dp.shard_seed(seed, rank) + 1000 * i, realistic_targets=realistic_targets)
                        for i in range(n_batches)]

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def build_model(model_type, vocab_size, model_conf, compute_dtype=None):
    """train.py:288-302 constructor calls."""
    if model_type == 'performer':
        from .model.music_performer import MusicPerformer
        return MusicPerformer(vocab_size, model_conf['n_layer'], model_conf['n_head'], model_conf['d_model'], model_conf['d_ff'], model_conf['d_embed'],
                              use_segment_emb=model_conf['use_segemb'], n_segment_types=2, favor_feature_dims=model_conf['feature_map']['n_dims'],
                              use_chord_mhot_emb=False, compute_dtype=compute_dtype)
    if model_type == 'gpt2':
        from .model.music_gpt2 import MusicGPT2
        return MusicGPT2(vocab_size, model_conf['n_layer'], model_conf['n_head'], model_conf['d_model'], model_conf['d_ff'], model_conf['d_embed'],
                         use_segment_emb=model_conf['use_segemb'], n_segment_types=2, use_chord_mhot_emb=False, compute_dtype=compute_dtype)
    raise NotImplementedError("Unsuppported model:", model_type)


def load_pretrained(model, path):
    """train.py:304-311: drop feature_map.omega, update, strict load."""
    pretrained = {k: v for k, v in torch.load(path, map_location='cpu').items() if 'feature_map.omega' not in k}
    sd = model.state_dict()
    sd.update(pretrained)
    model.load_state_dict(sd)


def main(argv=None):
    parser = argparse.ArgumentParser(description='stage-2 training on MI355X (same flags as the reference train.py)')
    req = parser.add_argument_group('required arguments')
    req.add_argument('-m', '--model_type', choices=['performer', 'gpt2'], required=True)
    req.add_argument('-c', '--configuration', required=True, help='one of the four stage-2 YAMLs (same keys)')
    req.add_argument('-r', '--representation', choices=['remi', 'functional'], required=True)
    parser.add_argument('--synthetic', type=int, default=0, help='if >0: batches per epoch of synthetic EMOPIA-shaped data instead of the event pickles')
    parser.add_argument('--dtype', default=None, choices=[None, 'bf16', 'fp32'])
    parser.add_argument('--epochs', type=int, default=None)
    parser.add_argument('--workers', type=int, default=8, help='DataLoader worker processes (reference: 8)')
    args = parser.parse_args(argv)
    train_conf = yaml.load(open(args.configuration, 'r'), Loader=yaml.FullLoader)
    rank, local_rank, world = dp.init_distributed()
    cfg = TrainConfig.from_yaml(train_conf, args.representation, world_size=world, verbose=rank == 0)
    torch.cuda.set_device(local_rank if world > 1 else cfg.gpuid)
    model_conf, dl = train_conf['model'], train_conf['data_loader']
    if args.synthetic:
        vocab = 327 if args.representation == 'functional' else 370
        dloader = SyntheticLoader(vocab, dl['batch_size'], model_conf['max_len'], args.synthetic, rank=rank)
        val_dloader = SyntheticLoader(vocab, dl['batch_size'], model_conf['max_len'], max(1, args.synthetic // 8), seed=99, rank=rank)
    else:                                                         # the reference's event pickles (train.py:252-285)
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        from .data import EventPieceDataset, load_split
        data_path, vocab_path = dl['data_path'].format(args.representation), dl['vocab_path'].format(args.representation)
        sets = [EventPieceDataset(data_dir=data_path, vocab_file=vocab_path, model_dec_seqlen=model_conf['max_len'], pieces=load_split(dl[k]),
                                  pad_to_same=True, predict_key=False) for k in ('train_split', 'val_split')]
        vocab = sets[0].vocab_size                                # n_token and the pad id come from the dictionary
        print('[info] # training pieces:', len(sets[0].pieces))
        samplers = [DistributedSampler(d, num_replicas=world, rank=rank, shuffle=True) if world > 1 else None for d in sets]
        dloader, val_dloader = [DataLoader(d, batch_size=dl['batch_size'], shuffle=sm is None, sampler=sm, num_workers=args.workers, pin_memory=True)
                                for d, sm in zip(sets, samplers)]
    model = build_model(args.model_type, vocab, model_conf, args.dtype).cuda()
    if train_conf['training']['trained_params']:
        load_pretrained(model, train_conf['training']['trained_params'])
    if world > 1:
        dp.sync_model_from_rank0(model)
    model.train()
    print('# params:', sum(p.numel() for p in model.parameters() if p.requires_grad))
    print('segemb:', model.segemb)
    optimizer = FusedAdam(model, lr=cfg.max_lr, max_grad_norm=0.5, world_size=world, token_weighted=True)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, cfg.lr_decay_steps, eta_min=cfg.min_lr)
    if train_conf['training']['trained_optim']:
        optimizer.load_state_dict(torch.load(train_conf['training']['trained_optim'], map_location='cpu'))
    params_dir, optimizer_dir = os.path.join(cfg.ckpt_dir, 'params/'), os.path.join(cfg.ckpt_dir, 'optim/')
    if rank == 0:
        for d in (cfg.ckpt_dir, params_dir, optimizer_dir):
            os.makedirs(d, exist_ok=True)
        shutil.copy(args.configuration, os.path.join(cfg.ckpt_dir, 'config.yaml'))
    for ep in range(args.epochs or cfg.max_epochs):
        if getattr(dloader, 'sampler', None) is not None and hasattr(dloader.sampler, 'set_epoch'):
            dloader.sampler.set_epoch(ep)
        loss = train_model(ep + 1, model, dloader, optimizer, scheduler, vocab - 1, model_type=args.model_type, cfg=cfg)
        if rank == 0 and not (ep + 1) % cfg.ckpt_interval:
            torch.save(model.state_dict(), os.path.join(params_dir, 'ep{:03d}_loss{:.3f}_params.pt'.format(ep + 1, loss)))
            torch.save(optimizer.state_dict(), os.path.join(optimizer_dir, 'ep{:03d}_loss{:.3f}_optim.pt'.format(ep + 1, loss)))
        val_losses, ta, ca, ma, oa = validate(model, val_dloader, vocab - 1, model_type=args.model_type, cfg=cfg)
        if rank == 0:
            with open(os.path.join(cfg.ckpt_dir, 'valloss.txt'), 'a') as f:
                f.write("ep{:03d} | loss: {:.3f} | valloss: {:.3f} (±{:.3f}) | total_acc: {:.3f} | chord_acc: {:.3f} | melody_acc: {:.3f} | "
                        "others_acc: {:.3f}\n".format(ep + 1, loss, np.mean(val_losses), np.std(val_losses), np.mean(ta), np.mean(ca), np.mean(ma), np.mean(oa)))


if __name__ == '__main__':
    main()
