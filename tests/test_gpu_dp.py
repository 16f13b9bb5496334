"""GPU: the data-parallel path (BASELINE configs[2] / configs[4]) on the ONE GPU a test box has.

* the C-ABI exchange (emo_comm_*: RCCL bound with dlopen) initialises, all-reduces and broadcasts on device buffers (world 1 —
  RCCL refuses two ranks on one device, so the exchange itself cannot be exercised with RCCL here);
* a REAL 2-rank training step — two processes sharing the GPU, host-staged gloo data plane — through train.train_model +
  FusedAdam(world_size=2, token_weighted=True) equals the 1-rank step on the concatenated batch: all-reduced gradient and
  parameters after the optimizer step, with unequal non-pad token counts per rank (fp32 parity mode);
* bench.py --gpus 2 on a 1-GPU box refuses instead of silently benchmarking one rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMM_SCRIPT = r'''
import ctypes, os, sys, torch
sys.path.insert(0, os.environ["EMO_ROOT"])
from emo_disentanger_amd._lib import lib, check, F32, I64
torch.cuda.set_device(0)
uid = (ctypes.c_char * 128)()
check(lib.emo_comm_unique_id(uid))
assert lib.emo_comm_world() == 0 and lib.emo_comm_rank() == -1
check(lib.emo_comm_init(uid, 0, 1))
assert lib.emo_comm_world() == 1 and lib.emo_comm_rank() == 0
s = torch.cuda.current_stream().cuda_stream
flat = torch.arange(1 << 22, device="cuda", dtype=torch.float32)
ref = flat.clone()
check(lib.emo_comm_allreduce(flat.data_ptr(), flat.numel(), F32, s))
check(lib.emo_comm_broadcast(flat.data_ptr(), flat.numel(), F32, 0, s))
cnt = torch.tensor([7, 9], device="cuda")
check(lib.emo_comm_allreduce(cnt.data_ptr(), 2, I64, s))
torch.cuda.synchronize()
assert torch.equal(flat, ref) and cnt.tolist() == [7, 9]
assert lib.emo_comm_init(uid, 0, 1) != 0 and b"already initialised" in lib.emo_last_error()
check(lib.emo_comm_destroy())
assert lib.emo_comm_allreduce(flat.data_ptr(), 4, F32, s) != 0        # loud after destroy
print("emo_comm ok")
'''

STEP_SCRIPT = r'''
import os, sys, tempfile, numpy as np, torch
sys.path.insert(0, os.environ["EMO_ROOT"])
from emo_disentanger_amd import dp, train as tr
from emo_disentanger_amd.data import synthetic_batch
from emo_disentanger_amd.model.music_performer import MusicPerformer
from emo_disentanger_amd.optim import FusedAdam
rank, _, world = dp.init_distributed()
torch.cuda.set_device(0)
torch.manual_seed(0)
V, B, T = 60, 4, 96
m = MusicPerformer(V, 2, 4, 64, 128, 64, dropout=0.0, favor_feature_dims=32, use_segment_emb=True, n_segment_types=2,
                   compute_dtype="fp32", redraw="fixed").cuda()
if world > 1:
    if rank == 1:                                   # replicas must come out identical anyway: rank 0's weights and omega win
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
            for n, b in m.named_buffers():
                if "omega" in n:
                    b.mul_(-1.0)
    dp.sync_model_from_rank0(m)
    assert dp.data_plane() == "gloo"
full = synthetic_batch(V, B, T, seed=3, realistic_targets=True)      # pad targets wherever track_mask == 0: unequal counts per shard
per = B // world
shard = {k: v[rank * per:(rank + 1) * per] for k, v in full.items()}
n_tok = int((shard["dec_target"] != V - 1).sum())
opt = FusedAdam(m, lr=1e-3, max_grad_norm=0.5, world_size=world, token_weighted=True)
cfg = tr.TrainConfig(world_size=world, verbose=False, warmup_steps=1, max_lr=1e-3, log_interval=10 ** 9, ckpt_dir=tempfile.mkdtemp())
p0 = m._ensure_store().flat32.clone()
loss = tr.train_model(1, m, [shard], opt, None, V - 1, cfg=cfg)
st = m._store
torch.cuda.synchronize()
g = st.flat_grad.clone()
if world > 1:
    g = g / st.flat_grad_ext[st.total]
    assert float(st.flat_grad_ext[st.total]) == float(int((full["dec_target"] != V - 1).sum()))
if world > 1:
    assert dp._STATE['bucketed_steps'] == (1 if os.environ.get("EMO_DP_BUCKETS") == "force" else 0)     # late-layer bucket + rest, or one all-reduce
np.savez(os.environ["EMO_OUT"] + ".rank%d.npz" % rank, grad=g.cpu().numpy(), before=p0.cpu().numpy(), after=st.flat32.cpu().numpy(), n_tok=n_tok, loss=loss)
dp.barrier()
dp.shutdown()
'''


STAGE1_SCRIPT = r'''
import os, sys, tempfile, numpy as np, torch
sys.path.insert(0, os.environ["EMO_ROOT"])
from emo_disentanger_amd import dp, stage1_train as st
from emo_disentanger_amd.model.plain_transformer import PlainTransformer
from emo_disentanger_amd.optim import FusedAdam
torch.cuda.set_device(0)
torch.manual_seed(0)
V, B, T = 40, 4, 48
m = PlainTransformer(64, V, 2, 4, 64, 128, 0, T, dec_dropout=0.0, pre_lnorm=True, compute_dtype="fp32").cuda()
rank, world = st.setup_data_parallel(m)              # rank 0's weights broadcast (both ranks built the same ones here)
g = np.random.default_rng(5)
x = g.integers(0, V - 1, size=(B, T), dtype=np.int64)
tgt = np.concatenate([x[:, 1:], np.full((B, 1), V - 2, dtype=np.int64)], 1)
for b in range(B):
    tgt[b, T - 3 - 4 * b:] = V - 1                   # a different number of pad targets per sequence -> unequal counts per rank
per = B // world
sl = slice(rank * per, (rank + 1) * per)
z = np.zeros((per, T), dtype=np.int64)
batch = {"id": torch.arange(per), "n_seg": [1] * per, "dec_inp_0": torch.from_numpy(x[sl]), "dec_tgt_0": torch.from_numpy(tgt[sl]),
         "dec_seg_len_0": torch.full((per,), T, dtype=torch.long), "inp_chord_0": torch.from_numpy(z), "inp_melody_0": torch.from_numpy(z.copy())}
opt = FusedAdam(m, lr=1e-3, max_grad_norm=0.5, world_size=world, token_weighted=True) if os.environ["EMO_S1_OPT"] == "fused" else torch.optim.Adam(m.parameters(), lr=1e-3)
cfg = st.Stage1Config(warmup_steps=10 ** 6, max_lr=1e-3, log_interval=10 ** 9, ckpt_dir=tempfile.mkdtemp(), verbose=False)
opt.param_groups[0]["lr"] = 1e-3
p0 = m._ensure_store().flat32.clone()
st.train(1, m, [batch], opt, None, V - 1, cfg, st.Stage1State())
ps = m._store
torch.cuda.synchronize()
np.savez(os.environ["EMO_OUT"] + ".rank%d.npz" % rank, before=p0.cpu().numpy(), after=ps.flat32.cpu().numpy())
dp.barrier()
dp.shutdown()
'''


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _env(**kw):
    env = dict(os.environ, EMO_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update({k: str(v) for k, v in kw.items()})
    return env


def test_emo_comm_c_abi_single_rank():
    r = subprocess.run([sys.executable, '-c', COMM_SCRIPT], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'emo_comm ok' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize('buckets', ['1', 'force'])
def test_two_rank_training_step_equals_one_rank_on_concatenated_batch(tmp_path, buckets):
    # buckets = force: dp.GradExchange splits the exchange (late layers' gradients during the backward, the rest after it) on the gloo plane too
    out1, out2 = str(tmp_path / 'w1'), str(tmp_path / 'w2')
    r = subprocess.run([sys.executable, '-c', STEP_SCRIPT], env=_env(EMO_OUT=out1, WORLD_SIZE=1, RANK=0, LOCAL_RANK=0), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, '-c', STEP_SCRIPT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=_env(EMO_OUT=out2, WORLD_SIZE=2, RANK=rk, LOCAL_RANK=0, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, EMO_COMM='gloo',
                                       EMO_DP_BUCKETS=buckets))
             for rk in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(l[-3000:] for l in logs)
    one = np.load(out1 + '.rank0.npz')
    two = [np.load(out2 + '.rank%d.npz' % rk) for rk in range(2)]
    assert int(two[0]['n_tok']) != int(two[1]['n_tok']) and int(two[0]['n_tok']) + int(two[1]['n_tok']) == int(one['n_tok'])
    np.testing.assert_array_equal(two[0]['before'], one['before'])            # rank 0's weights are the single-rank weights ...
    np.testing.assert_array_equal(two[1]['before'], two[0]['before'])         # ... and rank 1 was overwritten by the broadcast
    gmax = np.abs(one['grad']).max()
    for t in two:
        err = np.abs(t['grad'] - one['grad']).max() / gmax
        assert err <= 2e-6, 'all-reduced token-weighted gradient differs from the 1-rank gradient: %.3g of max|g|' % err
        # Adam's first update is lr * g / (|g| + 1e-8): elements whose gradient is a cancelling sum near 1e-8 amplify the fp32
        # reassociation of the two-shard sum, so the bound is on the 99.9th percentile (and a loose one on the maximum)
        step = np.abs(one['after'] - one['before']).max()
        d = np.abs(t['after'] - one['after']) / step
        assert step > 0 and np.quantile(d, 0.999) <= 1e-3 and d.max() <= 0.5, \
            'parameters after the fused Adam step differ: p99.9 %.3g, max %.3g of the largest update' % (np.quantile(d, 0.999), d.max())
    np.testing.assert_array_equal(two[0]['after'], two[1]['after'])           # replicas stay bit-identical


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('box has >= 2 GPUs')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and 'refusing' in r.stderr and '"value"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=_env(WORLD_SIZE=1, RANK=0, LOCAL_RANK=0), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and '"value"' not in r.stdout


def test_bench_two_rank_line_on_one_gpu():
    """The N > 1 leg of bench.py end to end (rank launch, strict plane check, replica broadcast, rank-stamped exchange self-test, timed product
    loop with the overlapped exchange, per-rank gather, ONE JSON line from rank 0) with two ranks sharing the test GPU on the host-staged gloo
    plane (EMO_BENCH_SHARE_GPU is test-only; RCCL refuses two ranks on one device)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--no-roofline'],
                       env=_env(EMO_BENCH_SHARE_GPU='1', EMO_COMM='gloo'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['comm'] == 'gloo' and d['scaling'] == 'weak' and d['config']['global_batch'] == 8 and d['config']['parallelism'] == 'dp2'
    assert d['dp_selftest']['ok'] and d['dp_selftest']['bad_elements'] == 0 and d['dp_selftest']['split'] and len(d['per_rank_ms_per_step']) == 2
    assert abs(d['value'] - 2 * 4 * d['config']['seq_len'] * d['steps'] / (d['ms_per_step'] * d['steps'] / 1e3)) <= 1e-3 * d['value']   # whole-job tokens / max-over-ranks time
    assert 3.0 < d['mean_loss'] < 8.0


@pytest.mark.parametrize('opt', ['fused', 'adam'])
def test_stage1_two_rank_step_equals_one_rank(tmp_path, opt):
    """BASELINE configs[4] (stage-1 lead-sheet LM, data parallel): stage1_train.train with 2 ranks (token-count-weighted exchange, fused
    clip + Adam or the reference's torch.optim.Adam behind clip_grad_norm_) equals the 1-rank step on the concatenated batch."""
    out1, out2 = str(tmp_path / 'w1'), str(tmp_path / 'w2')
    r = subprocess.run([sys.executable, '-c', STAGE1_SCRIPT], env=_env(EMO_OUT=out1, WORLD_SIZE=1, RANK=0, LOCAL_RANK=0, EMO_S1_OPT=opt), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, '-c', STAGE1_SCRIPT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=_env(EMO_OUT=out2, WORLD_SIZE=2, RANK=rk, LOCAL_RANK=0, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, EMO_COMM='gloo',
                                       EMO_S1_OPT=opt)) for rk in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(l[-3000:] for l in logs)
    one = np.load(out1 + '.rank0.npz')
    two = [np.load(out2 + '.rank%d.npz' % rk) for rk in range(2)]
    step = np.abs(one['after'] - one['before']).max()
    assert step > 0
    for t in two:
        np.testing.assert_array_equal(t['before'], one['before'])
        d = np.abs(t['after'] - one['after']) / step
        assert np.quantile(d, 0.999) <= 1e-3 and d.max() <= 0.5, (np.quantile(d, 0.999), d.max())
    np.testing.assert_array_equal(two[0]['after'], two[1]['after'])
