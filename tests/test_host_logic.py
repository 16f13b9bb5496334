"""CPU-side tests of the product's host logic (no GPU compute): sampling helpers against the golden
vectors of the imported reference, LR schedule / log format, and the data-parallel plumbing over
gloo with world_size 2."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(__file__), 'golden')
SAMP = json.load(open(os.path.join(G, 'sampling.json')))


@pytest.mark.parametrize('key', sorted(SAMP))
def test_product_temperature_and_nucleus_match_reference(key):
    from emo_disentanger_amd import inference as inf
    e = SAMP[key]
    lg = np.array(e['logits'], dtype=np.float32)
    probs = inf.temperature(lg.copy(), e['temp'], inadmissibles=None)
    np.testing.assert_array_equal(np.asarray(probs, dtype=np.float64), np.array(e['probs']))
    if e['error'] == 'IndexError':
        with pytest.raises(IndexError):                       # the reference's F12 edge is preserved, not papered over
            inf.nucleus(np.array(probs, copy=True), e['p'])
        return
    words = []
    for s in range(4):
        np.random.seed(s)
        words.append(int(inf.nucleus(np.array(probs, copy=True), e['p'])))
    assert words == e['words_seed0_3']


def test_lr_schedule_and_log_format_match_reference():
    from emo_disentanger_amd import train as tr
    tl = json.load(open(os.path.join(G, 'trainloop.json')))
    for key, e in tl.items():
        c, accum = e['cfg'], int(key[-1])
        cfg = tr.TrainConfig(warmup_steps=c['warmup'], max_lr=c['max_lr'], min_lr=c['eta_min'], lr_decay_steps=c['T_max'], accum_steps=accum)
        exp = [c['max_lr'] if s == 1 else tr.lr_after_step(s - 1, cfg) for s in range(1, c['n_batches'] + 1) if s % accum == 0]
        np.testing.assert_allclose(exp, e['lrs_at_optim_step'], rtol=1e-12)
        assert abs(tr.lr_after_step(c['n_batches'], cfg) - e['final_lr']) < 1e-15
    d = tempfile.mkdtemp()
    f = os.path.join(d, 'log.txt')
    tr.log_epoch(f, {'ep': 1, 'steps': 50, 'recons_loss': 3.1234567, 'time': 12.3456}, is_init=True)
    lines = open(f).read().split('\n')
    assert lines[0] == 'ep   steps    recons_loss  ep_time     ' and lines[1] == '1    50       3.12346      12.35       '


def test_train_config_reads_reference_yaml_keys():
    import yaml
    from emo_disentanger_amd import train as tr
    cfg_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), 'emo-disentanger_amd', 'config')
    for name, accum in (('pop1k7_pretrain.yaml', 1), ('emopia_finetune_gpt2.yaml', 2)):
        conf = yaml.load(open(os.path.join(cfg_dir, name)), Loader=yaml.FullLoader)
        c = tr.TrainConfig.from_yaml(conf, 'functional')
        assert c.accum_steps == accum and c.warmup_steps == 200 and 'functional' in c.ckpt_dir
        assert conf['model']['n_layer'] == 12 and conf['model']['feature_map']['n_dims'] == 128


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from emo_disentanger_amd import dp
    r, lr, w = dp.init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(dp.shard_seed(1234, rank))
    flat = torch.randn(1000, generator=g)
    mine = flat.clone()
    dp.allreduce_sum_(flat)
    params = torch.full((10,), float(rank))
    dp.broadcast_([params])
    mx = dp.max_over_ranks(1.0 + rank, 'cpu')
    q.put((rank, mine.numpy().copy(), flat.numpy().copy(), params.numpy().copy(), mx))   # by value (tensor FD passing races the exit)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_plumbing_gloo_world2():
    import socket
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    with socket.socket() as sk:                     # a free port (fixed pid-derived ports collided across test runs)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, m0, f0, p0, x0), (_, m1, f1, p1, x1) = res
    assert not np.array_equal(m0, m1)                      # each rank drew its own shard (weak scaling)
    assert np.allclose(f0, m0 + m1) and np.array_equal(f0, f1)   # grad all-reduce = sum; 1/world is folded into the clip coef
    assert not p0.any() and not p1.any()                   # replicas start from rank 0's weights
    assert x0 == x1 == 2.0                                 # bench timing = max over ranks
