#!/usr/bin/env python3
"""bench.py — training tokens/s of the stage-2 Performer (d512, L12, H8, F128, seq 2048, bf16) on
synthetic EMOPIA-shaped batches: BASELINE.json configs[1] at N=1, configs[2] (DP over RCCL) at N>1.

One "step" = zero_grad + forward + CE loss + backward + grad all-reduce (N>1) + clip(0.5) + Adam +
LR schedule + device-side accuracy counts, with dropout 0.1 active and omega redrawn every forward
(reference-faithful).  Prints ONE JSON line on rank 0 (contract in the task statement) with
`roofline` (dominant kernel, HIP-event timed on the launch stream) and `cpu_baseline` (the oracle's
CPU training step timed on this box's host cores; N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

CFG = dict(n_token=327, n_layer=12, n_head=8, d_model=512, d_ff=2048, n_feat=128, seq=2048)
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def gemm_flops_per_token():
    d, f, L, V = CFG['d_model'], CFG['d_ff'], CFG['n_layer'], CFG['n_token']
    return 3 * (L * 2 * (4 * d * d + 2 * d * f) + 2 * d * V)      # fwd + dgrad + wgrad = 227.5 MFLOP


def time_kernel(fn, iters=10, warm=3):
    """Average duration (ms) of `fn` (launches on torch's current stream) measured with HIP events on that stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


GEMM_KERNELS = {   # layout class of ops.gemm -> the kernel instance it launches at the bench shapes (names as in the rocprofv3 summary)
    'TN': ('gemm_bf16_kernel<false,false,false,float,64,RS> (RS=1: with the bias gradient, RS=0: without) + splitk_reduce_kernel',
           'wgrad dW = dY^T X, reduction over the B*T tokens'),
    'NN': ('gemm_bf16_glds_kernel<true,false,bf16,32,2>', 'dgrad dX = dY W'),
    'NT': ('gemm_bf16_glds_kernel<true,true,bf16,32,3>', 'forward Y = X W^T + fused epilogue, K = 512 (shapes outside the A-stationary class)'),
    'NT/K=512': ('gemm_astat_kernel<bf16,BITS> (A stationary in registers, weights through the LDS ring)',
                 'K = 512 products: QKV / out-projection / FFN1 forward, FFN2 / out-projection dgrad against transposed weight mirrors'),
    'NT/K>1024': ('gemm_bf16_glds_kernel<true,true,bf16,64,2>', 'forward Y = X W^T + fused epilogue, K = 2048'),
}


def dominant_kernel_roofline(step_fn, B, T, n_steps=2):
    """Every GEMM launch inside `n_steps` real training steps is bracketed by HIP events on its launch stream (in situ: same data,
    same cache state as the timed region).  The four GEMM kernel instances (wgrad TN, dgrad NN, forward NT at K=512 and K=2048) take ~65 % of the
    step; `roofline` is the class with the largest total time (the dominant kernel of the rocprofv3 summary under profiles/),
    the others are listed in `roofline_others`.  achieved = sum(algorithmic FLOPs = 2*M*N*K) / sum(durations).

    Kernel durations only mean something when kernels do not share the CUs: the optional second HIP stream for wgrad
    (EMO_WGRAD_STREAM=1, off by default) is forced off for the instrumented steps; when it is enabled for the timed region the
    stretched in-step average is reported next to the serialized one."""
    from emo_disentanger_amd import engine, ops

    extra = {}

    def instrumented(side_on):
        was = engine._SIDE['on']
        engine._SIDE['on'] = side_on and was
        ops.GEMM_TIMING, ops.KERNEL_TIMING = [], {}
        try:
            for _ in range(n_steps):
                step_fn()
            torch.cuda.synchronize()
        finally:
            rec, ops.GEMM_TIMING = ops.GEMM_TIMING, None
            other, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
            engine._SIDE['on'] = was
        extra.clear()
        extra.update(other)
        by = {}
        for kind, e0, e1, fl, by_ in (r[:5] for r in rec):
            d = by.setdefault(kind, {'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'n': 0})
            d['ms'] += e0.elapsed_time(e1); d['flops'] += fl; d['bytes'] += by_; d['n'] += 1
        return by

    serial = instrumented(False)
    overlapped = instrumented(True) if engine._SIDE['on'] else {}
    pmc_files = {'TN': os.path.join(ROOT, 'profiles', 'r01_pmc_wgrad.json'), 'NN': os.path.join(ROOT, 'profiles', 'r01_pmc_dgrad.json'),
                 'NT/K=512': os.path.join(ROOT, 'profiles', 'r02_pmc_astat.json')}

    def entry(kind):
        d = serial[kind]
        achieved = d['flops'] / d['ms'] / 1e9
        traffic = None      # HBM bytes per launch from rocprofv3 PMC passes (collected offline, committed under profiles/)
        pmc = pmc_files.get(kind)
        if pmc and os.path.exists(pmc) and B * T == 131072:
            j = json.load(open(pmc))
            traffic = j.get('traffic_bytes_per_launch')
            if 'ratio' in j:         # counters were collected on ONE shape of the class: scale the class's algorithmic bytes by its measured ratio
                traffic = round(j['ratio'] * d['bytes'] / d['n'])
        return {'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
                'kernel': '%s (%s)' % GEMM_KERNELS.get(kind, (kind, 'other GEMM')), 'launches_timed': d['n'], 'avg_launch_ms': round(d['ms'] / d['n'], 4),
                'total_ms_per_step': round(d['ms'] / n_steps, 2),
                'avg_launch_ms_overlapped_in_step': round(overlapped[kind]['ms'] / overlapped[kind]['n'], 4) if kind in overlapped else None,
                'algorithmic_flops_per_launch': round(d['flops'] / d['n']), 'algorithmic_bytes_per_launch': round(d['bytes'] / d['n'])}

    order = sorted(serial, key=lambda k: -serial[k]['ms'])
    roof = entry(order[0])
    roof['timing'] = 'HIP events on the launch stream around every GEMM launch in %d real training steps (kernels serialized on one stream)' % n_steps
    roof['rocprof_summary'] = ('profiles/r02_bench_train_rocprof_stats.txt = rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3 --no-cpu-baseline '
                               '--no-gen --no-stage1 --no-gpt2 --no-step0-check` (training kernels only); PMC traffic: profiles/r02_pmc_astat.json, r01_pmc_dgrad.json, r01_pmc_wgrad.json')
    roof['roofline_others'] = [entry(k) for k in order[1:]] + [attn_entry(k, v, n_steps) for k, v in sorted(extra.items())]
    return roof


ATTN_KERNELS = {'favor_fwd': ('favor_fwd_kernel', 'hbm', 'FAVOR+ causal linear attention forward: features + chunked prefix-sum scan; bytes = q, k, v read + out written (4*512*e per token*layer)'),
                'favor_bwd': ('favor_bwd_dq_kernel + favor_bwd_dkv_kernel', 'hbm', 'FAVOR+ backward (forward sweep dq, reverse sweep dk/dv); bytes = 7*512*e per token*layer'),
                'sattn_fwd': ('sattn_fwd_kernel', 'mfma', 'GPT-2 causal softmax attention forward (flash tiles): 2 matmuls, causal half'),
                'sattn_bwd': ('sattn_bwd_dq_kernel + sattn_bwd_dkv_kernel', 'mfma', 'GPT-2 attention backward: 7 matmuls, causal half')}


def attn_entry(kind, rec, n_steps):
    """roofline entry of an attention kernel class from in-situ HIP-event brackets (ops._timed)."""
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec)
    fl, by = sum(r[2] for r in rec), sum(r[3] for r in rec)
    name, bound, what = ATTN_KERNELS[kind]
    if bound == 'hbm':
        ach, peak, unit = by / ms / 1e6, PEAK_HBM_GBS, 'GB/s'
    else:
        ach, peak, unit = fl / ms / 1e9, PEAK_BF16_TFLOPS, 'TFLOP/s'
    return {'bound': bound, 'achieved': round(ach, 1), 'peak': peak, 'unit': unit, 'frac': round(ach / peak, 4), 'traffic': None, 'kernel': '%s (%s)' % (name, what),
            'launches_timed': len(rec), 'avg_launch_ms': round(ms / len(rec), 4), 'total_ms_per_step': round(ms / n_steps, 2),
            'algorithmic_flops_per_launch': round(fl / len(rec)), 'algorithmic_bytes_per_launch': round(by / len(rec))}


def gpt2_bench(n_steps=6, B=16, T=2048):
    """Secondary line: the GPT-2 backbone of the same hot path (north_star: masked-softmax attention) at the pop1k7_pretrain_gpt2 shape
    (d512 / L12 / H8), B=16 x T=2048, bf16, dropout 0.1, fwd + bwd + clip + fused Adam on synthetic tokens; the flash-attention kernels are
    timed in situ with HIP events."""
    import contextlib
    from emo_disentanger_amd import ops
    from emo_disentanger_amd.data import synthetic_batch
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from emo_disentanger_amd.optim import FusedAdam
    with contextlib.redirect_stdout(sys.stderr):
        m = MusicGPT2(CFG['n_token'], 12, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2, dropout=0.1, compute_dtype='bf16').cuda().train()
    opt = FusedAdam(m, lr=1e-5, max_grad_norm=0.5)
    b = synthetic_batch(CFG['n_token'], B, T, device='cuda')

    def step():
        opt.zero_grad()
        loss = m.compute_loss(m(b['dec_input'], seg_inp=b['track_mask']), b['dec_target'])['total_loss']
        loss.backward()
        opt.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    ops.KERNEL_TIMING = {}
    try:
        step()
        step()
        torch.cuda.synchronize()
    finally:
        rec, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
    return {'metric': 'GPT-2 backbone train tokens/sec', 'value': round(B * T / dt, 1), 'unit': 'tokens/s', 'ms_per_step': round(dt * 1e3, 3),
            'config': {'workload': 'stage2 GPT-2 d512 L12 H8 d_ff2048, B=%d x T=%d, bf16, dropout 0.1 (attention-probability dropout in-kernel), fused Adam' % (B, T)},
            'loss': round(float(loss.detach()), 4), 'roofline': [attn_entry(k, v, 2) for k, v in sorted(rec.items())]}


def generation_bench(model, n_streams=32, prompt=64, n_new=2048 - 64, top_p=0.9, temp=1.1):
    """BASELINE configs[3]: 32 parallel streams, 64-token prompt, generate to 2048 tokens, nucleus p=0.9, recurrent FAVOR+ state in HBM,
    hipGraph-replayed decode step; tokens/s = streams * new tokens / wall time (engine set-up, prefill of the prompt and graph capture included)."""
    from emo_disentanger_amd import inference as inf
    dev = next(model.parameters()).device
    g = torch.Generator().manual_seed(7)
    ptok = torch.randint(0, CFG['n_token'] - 1, (n_streams, prompt), generator=g).to(dev)
    pseg = torch.ones(n_streams, prompt, dtype=torch.long, device=dev)
    model.eval()
    inf.generate_streams(model, ptok, pseg, 8, temp=temp, top_p=top_p, seed=1)          # warm-up (kernels, graph machinery)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = inf.generate_streams(model, ptok, pseg, n_new, temp=temp, top_p=top_p, seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.train()
    assert out.shape == (n_streams, prompt + n_new) and int(out.max()) < CFG['n_token']
    return {'metric': 'AR gen tokens/sec, stage2 Performer d512 L12, %d streams, nucleus p=%.2f' % (n_streams, top_p),
            'value': round(n_streams * n_new / dt, 1), 'unit': 'tokens/s', 'streams': n_streams, 'prompt': prompt, 'new_tokens': n_new,
            'ms_per_token_step': round(1000 * dt / n_new, 3), 'engine': 'FAVOR+ recurrent state in HBM + hipGraph replay of the decode step'}


def stage1_bench(n_steps=10, B=4, T=512, V=200):
    """Secondary line, BASELINE configs[4] (single GPU): the stage-1 lead-sheet LM (Transformer-XL decoder, emopia_finetune.yaml shape:
    d512 / L12 / H8 / d_ff 2048, tgt_len 512, batch 4), bf16, dropout 0.1, fwd + bwd + clip + fused Adam on synthetic tokens."""
    from emo_disentanger_amd.model.plain_transformer import PlainTransformer
    from emo_disentanger_amd.optim import FusedAdam
    m = PlainTransformer(512, V, 12, 8, 512, 2048, 0, T, dec_dropout=0.1, pre_lnorm=True, compute_dtype='bf16').cuda().train()
    opt = FusedAdam(m, lr=1e-5, max_grad_norm=0.5)
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randint(0, V - 1, (T, B), generator=g).cuda(), torch.randint(0, V - 1, (T, B), generator=g).cuda()

    def step():
        opt.zero_grad()
        loss = m.compute_loss(m(x, tuple())[0], tgt)['total_loss']
        loss.backward()
        opt.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    return {'metric': 'stage1 (Transformer-XL lead-sheet LM) train tokens/sec', 'value': round(B * T / dt, 1), 'unit': 'tokens/s', 'ms_per_step': round(dt * 1e3, 3),
            'config': {'workload': 'BASELINE configs[4] shape, 1 GPU: d512 L12 H8 d_ff2048 tgt_len=%d batch=%d V=%d, bf16, dropout 0.1' % (T, B, V)},
            'loss': round(float(loss), 4)}


def step0_check(model, batch):
    """Loss of the bench's OWN weights on the first sequence of its own batch, HIP path vs the CPU oracle (omega fixed, dropout 0, outside the
    timed region; SURVEY §8(d) "Timing protocol").  bf16 = the timed compute mode, fp32 = parity mode (north_star: |dloss| <= 1e-4)."""
    from oracle import model_ref
    x, seg, tgt = batch['dec_input'][:1], batch['track_mask'][:1], batch['dec_target'][:1]
    redraw, training = model.redraw, model.training
    model.redraw = 'fixed'
    model.eval()
    out = {}
    try:
        with torch.no_grad():
            for dt in ('fp32', 'bf16'):                         # (ends on bf16: the store of the timed run is rebuilt once)
                model.set_compute_dtype(dt)
                out['loss_hip_' + dt] = float(model.compute_loss(model(x, seg_inp=seg), tgt)['total_loss'])
            sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
            t0 = time.time()
            ref = model_ref.compute_loss(model_ref.forward('performer', sd, x.cpu(), seg.cpu(), CFG['n_layer'], CFG['n_head'], CFG['d_model']),
                                         tgt.cpu(), CFG['n_token'])
            out['loss_oracle'] = float(ref)
            out['oracle_seconds'] = round(time.time() - t0, 2)
    finally:
        model.redraw = redraw
        model.train(training)
    out['abs_err_fp32'] = abs(out['loss_hip_fp32'] - out['loss_oracle'])
    out['abs_err_bf16'] = abs(out['loss_hip_bf16'] - out['loss_oracle'])
    out['abs_err'] = out['abs_err_bf16']
    out['ok'] = bool(out['abs_err_fp32'] <= 1e-4 and out['abs_err_bf16'] <= 5e-3)
    out['sample'] = 'first sequence (B=1 x T=%d) of the timed batch, the timed weights, omega fixed, dropout 0' % x.shape[1]
    return out


def cpu_generation_baseline(ctx=256, n_tok=4):
    """Reference-style AR step on the CPU oracle: full-prefix recompute per token (inference.py:252-272), 1 stream."""
    from oracle import model_ref
    from oracle.weights import make_state_dict
    sd = make_state_dict('performer', CFG['n_token'], CFG['n_layer'], CFG['n_head'], CFG['d_model'], CFG['d_ff'], favor_feature_dims=CFG['n_feat'], seed=0)
    x = torch.randint(0, CFG['n_token'] - 1, (1, ctx))
    seg = torch.ones(1, ctx, dtype=torch.long)
    with torch.no_grad():
        model_ref.forward('performer', sd, x, seg, CFG['n_layer'], CFG['n_head'], CFG['d_model'], keep_last_only=True)
        t0 = time.time()
        for _ in range(n_tok):
            model_ref.forward('performer', sd, x, seg, CFG['n_layer'], CFG['n_head'], CFG['d_model'], keep_last_only=True)
        dt = (time.time() - t0) / n_tok
    return {'value': round(1.0 / dt, 2), 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'oracle full-prefix recompute at context %d, 1 stream, %d tokens' % (ctx, n_tok)}


def cpu_baseline(T, steps=2):
    """Oracle (CPU restatement of the reference path: torch fp32 eager + the C causal-product) timed on the host cores: forward + loss +
    backward + clip(0.5) + Adam on B=1 x T tokens, all cores (`value`) and 8 threads (`value_8_threads`, the survey container's core count)."""
    from oracle import model_ref
    from oracle.weights import make_state_dict, synthetic_batch
    sd = make_state_dict('performer', CFG['n_token'], CFG['n_layer'], CFG['n_head'], CFG['d_model'], CFG['d_ff'], favor_feature_dims=CFG['n_feat'], seed=0)
    b = synthetic_batch(CFG['n_token'], 1, T, seed=1234)
    cores = torch.get_num_threads()
    args = ('performer', sd, b, CFG['n_token'], CFG['n_layer'], CFG['n_head'], CFG['d_model'])
    params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if v.is_floating_point() and not k.endswith('pe.pe') and 'omega' not in k}
    opt = torch.optim.Adam(params.values(), lr=1e-4)

    def one_step():
        _, _, grads = model_ref.loss_and_grads(*args, p_drop=0.1, training=True)
        for k, p in params.items():
            p.grad = grads[k]
        torch.nn.utils.clip_grad_norm_(params.values(), 0.5)
        opt.step()
    one_step()                                                        # warm-up (also builds the C kernel)
    t0 = time.time()
    for _ in range(steps):
        one_step()
    dt = (time.time() - t0) / steps
    out = {'value': round(T / dt, 1), 'unit': 'tokens/s', 'cores': cores, 'kind': 'port',
           'sample': 'oracle Performer L12 d512 fwd + bwd + clip + Adam, B=1 x T=%d, %d timed steps, torch fp32 eager + C causal product' % (T, steps)}
    if cores > 8:
        torch.set_num_threads(8)
        os.environ['OMP_NUM_THREADS'] = '8'
        try:
            t0 = time.time()
            one_step()
            out['value_8_threads'] = round(T / (time.time() - t0), 1)
        finally:
            torch.set_num_threads(cores)
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run and exit with its
    status.  Fails instead of shrinking the job when the node has fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        sys.exit('bench.py: --gpus %d requested but %d GPU(s) visible: refusing to run (one process per GPU, no oversubscription)' % (n, have))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='sequences per GPU (BASELINE.md: B=64)')
    ap.add_argument('--seq', type=int, default=CFG['seq'])
    ap.add_argument('--redraw', default='every_forward')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-gen', action='store_true')
    ap.add_argument('--no-stage1', action='store_true')
    ap.add_argument('--no-gpt2', action='store_true')
    ap.add_argument('--no-step0-check', action='store_true')
    args = ap.parse_args()

    want = max(args.gpus, 1)
    if 'WORLD_SIZE' not in os.environ and want > 1:
        launch_ranks(want)                                   # never returns
    from emo_disentanger_amd import dp, ops
    from emo_disentanger_amd.data import synthetic_batch
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from emo_disentanger_amd.optim import FusedAdam
    rank, local_rank, world = dp.env_world()
    if world != want:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus %d` (it starts the ranks itself) or '
                 '`python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d`' % (want, world, want, want, want))
    if torch.cuda.device_count() < (local_rank + 1 if world > 1 else 1):
        sys.exit('bench.py: rank %d needs GPU %d but only %d visible: one process per GPU, no sharing' % (rank, local_rank, torch.cuda.device_count()))
    dp.init_distributed()
    local_dev = local_rank if world > 1 else 0
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    torch.manual_seed(0)
    B, T = args.batch, args.seq
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):             # the constructor prints '[info] model init completed' like the reference's; stdout carries ONE JSON line
        model = MusicPerformer(CFG['n_token'], CFG['n_layer'], CFG['n_head'], CFG['d_model'], CFG['d_ff'], CFG['d_model'],
                               favor_feature_dims=CFG['n_feat'], use_segment_emb=True, n_segment_types=2, dropout=0.1,
                               compute_dtype='bf16', redraw=args.redraw).to(dev)
    model.train()
    if world > 1:
        dp.sync_model_from_rank0(model)
    max_lr, eta_min, warmup_steps, T_max = 1e-4, 1e-5, 200, 500000      # pop1k7_pretrain.yaml
    opt = FusedAdam(model, lr=max_lr, max_grad_norm=0.5, world_size=world, token_weighted=True)
    batches = [synthetic_batch(CFG['n_token'], B, T, seed=dp.shard_seed(1234, rank) + 100 * i, device=dev) for i in range(2)]
    for b in batches:                                        # non-pad target count of the rank's batch (token-weighted DP mean)
        b['n_tok'] = (b['dec_target'] != CFG['n_token'] - 1).sum().to(torch.float32)
    s0 = step0_check(model, batches[1]) if (world == 1 and not args.no_step0_check) else None     # batches[1] = the batch of step 1
    ps = model._ensure_store()
    counts = torch.zeros(6, device=dev, dtype=torch.int64)
    loss_acc = torch.zeros((), device=dev)
    state = {'step': 0}
    exchange = dp.GradExchange(model, ps) if world > 1 else None

    def step():
        state['step'] += 1
        b = batches[state['step'] % len(batches)]
        opt.zero_grad()
        logits = model(b['dec_input'], seg_inp=b['track_mask'], attn_kwargs={'omit_feature_map_draw': False})
        losses = model.compute_loss(logits, b['dec_target'])
        if world > 1:
            exchange.arm()                                   # the late layers' all-reduce overlaps the rest of the backward (dp.GradExchange)
            (losses['total_loss'] * b['n_tok']).backward()
            exchange.finish(b['n_tok'])                      # flat fp32 gradient + token count in up to 3 pieces (emo_comm_allreduce)
        else:
            losses['total_loss'].backward()
        opt.step()
        loss_acc.add_(losses['recons_loss'].detach())
        counts.add_(ops.accuracy_counts(logits.detach().view(-1, logits.shape[-1]), b['dec_target'].view(-1), b['chord_idx'].view(-1),
                                        b['melody_idx'].view(-1), CFG['n_token'] - 1))
        s = state['step']
        opt.param_groups[0]['lr'] = max_lr * s / warmup_steps if s < warmup_steps else \
            eta_min + (max_lr - eta_min) * (1 + math.cos(math.pi * (s - warmup_steps) / T_max)) / 2

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dp.barrier()
    loss_acc.zero_()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dp.barrier()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0)
    mean_loss = float(loss_acc) / max(args.steps, 1)
    tokens = world * B * T * args.steps
    value = tokens / elapsed
    out = {'metric': 'train tokens/sec (+ AR gen tokens/sec in "gen"), stage2 Performer d512 L12 seq%d' % T, 'value': round(value, 1), 'unit': 'tokens/s', 'n_gpus': world,
           'rccl_ranks': (ops.lib.emo_comm_world() if dp.data_plane() == 'rccl' else world if dp.data_plane() == 'nccl' else 0), 'comm': dp.data_plane() or 'none',
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000 * elapsed / args.steps, 3), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': 'BASELINE configs[%d]: stage2 Performer d_model=512 n_layer=12 n_head=8 favor_dims=128 seq=%d, B=%d/GPU, '
                                  'dropout 0.1, omega redraw %s, fwd+bwd+allreduce+clip+Adam' % (1 if world == 1 else 2, T, B, args.redraw),
                      'global_batch': world * B, 'seq_len': T, 'parallelism': 'dp%d' % world, 'n_token': CFG['n_token']},
           'mean_loss': round(mean_loss, 4), 'gemm_tflops_model': round(value * gemm_flops_per_token() / 1e12, 1)}
    if not args.no_roofline:                 # every rank runs the instrumented steps (they contain the all-reduce)
        roof = dominant_kernel_roofline(step, B, T)
        if rank == 0:
            out['roofline'] = roof
    if rank == 0:
        if s0 is not None:
            out['step0_check'] = s0
        if world == 1 and not args.no_gen:
            out['gen'] = generation_bench(model)
            if not args.no_cpu_baseline:
                out['gen']['cpu_baseline'] = cpu_generation_baseline()
        if world == 1 and not args.no_stage1:
            out['stage1'] = stage1_bench()
        if world == 1 and not args.no_gpt2:
            del model, opt                                   # (the Performer's 21 GB of saved activations are not needed any more)
            torch.cuda.empty_cache()
            out['gpt2'] = gpt2_bench()
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(min(T, 2048))
        print(json.dumps(out), flush=True)
    if world > 1:
        dp.barrier()
        dp.shutdown()


if __name__ == '__main__':
    main()
