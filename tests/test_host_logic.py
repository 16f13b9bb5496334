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
    # the optimizer-step exchange (dp.allreduce_grads_): every rank holds the gradient of the SUM of its token losses and its non-pad
    # token count; one collective moves both, and dividing by the summed count gives the global token mean (not the mean of means)
    class _Store:
        total = 8
        flat_grad_ext = torch.zeros(16)
        flat_grad = flat_grad_ext[:8]
    n_tok = (3.0, 5.0)[rank]                                   # unequal counts on purpose
    g_mean = torch.full((8,), 1.0 + 2.0 * rank)                # this rank's mean-loss gradient
    _Store.flat_grad.copy_(g_mean * n_tok)
    dp.allreduce_grads_(_Store, torch.tensor(n_tok))
    assert float(_Store.flat_grad_ext[8]) == 8.0 and float(_Store.flat_grad_ext[9:].abs().sum()) == 0.0
    assert torch.allclose(_Store.flat_grad / _Store.flat_grad_ext[8], torch.full((8,), (3.0 * 1.0 + 5.0 * 3.0) / 8.0))
    assert dp.data_plane() == 'gloo'
    # the overlapped exchange (dp.GradExchange): the late layers' contiguous range goes first, from the backward hook; the rest + the
    # token-count tail after the backward.  Flat layout like ParamStore's: emb | out-proj | layer 0..3 | segemb, 8-element aligned.
    class _Model:
        n_layer = 4
        _bwd_hook = None
        def _layer_prefix(self, l):
            return 'dec.%d.' % l
    class _Store2:
        device = 'cpu'
        offsets = {'emb.w': 0, 'out.w': 16, 'dec.0.a': 24, 'dec.0.b': 32, 'dec.1.a': 40, 'dec.2.a': 56, 'dec.2.b': 64, 'dec.3.a': 72, 'seg.w': 88}
        total = 96
        flat_grad_ext = torch.zeros(104)
        flat_grad = flat_grad_ext[:96]
    os.environ['EMO_DP_BUCKETS'] = 'force'
    m2 = _Model()
    ex = dp.GradExchange(m2, _Store2)
    assert ex.range == (56, 88, 2)                              # layers 2..3
    _Store2.flat_grad.copy_(torch.arange(96.) * (rank + 1))
    ex.arm()
    assert m2._bwd_hook is not None
    m2._bwd_hook(3)                                             # nothing yet: layer 3 done, layer 2 still running
    assert float(_Store2.flat_grad[60]) == 60.0 * (rank + 1)
    m2._bwd_hook(2)                                             # layers 2..3 complete: their range is summed now, the rest is untouched
    assert float(_Store2.flat_grad[60]) == 60.0 * 3 and float(_Store2.flat_grad[10]) == 10.0 * (rank + 1) and float(_Store2.flat_grad[90]) == 90.0 * (rank + 1)
    ex.finish(torch.tensor(n_tok))
    assert m2._bwd_hook is None and torch.equal(_Store2.flat_grad, torch.arange(96.) * 3) and float(_Store2.flat_grad_ext[96]) == 8.0
    _Store2.flat_grad.copy_(torch.arange(96.) * (rank + 1))    # a step whose hook never fires (stage-1 model): one all-reduce in finish()
    ex.arm()
    ex.finish(torch.tensor(n_tok))
    assert torch.equal(_Store2.flat_grad, torch.arange(96.) * 3) and float(_Store2.flat_grad_ext[96]) == 8.0
    _Store2.offsets['stray'] = 60                               # a foreign parameter inside the range: no split
    assert dp.GradExchange(m2, _Store2).range is None
    q.put((rank, mine.numpy().copy(), flat.numpy().copy(), params.numpy().copy(), mx))   # by value (tensor FD passing races the exit)
    dp.barrier()
    dp.shutdown()


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
    assert np.allclose(f0, m0 + m1) and np.array_equal(f0, f1)   # grad all-reduce = sum; the 1/world (or 1/tokens) scale is folded into the clip coef
    assert not p0.any() and not p1.any()                   # replicas start from rank 0's weights
    assert x0 == x1 == 2.0                                 # bench timing = max over ranks


def _rccl_init_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from emo_disentanger_amd import dp
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    try:
        dp._init_rccl(rank, world)
        q.put((rank, 'ok', ''))
    except RuntimeError as e:
        q.put((rank, 'raised', str(e)))
    finally:
        dist.barrier()                                     # every rank comes back out of the bring-up: nobody is left inside a collective
        dist.destroy_process_group()


def test_rccl_bringup_agrees_on_a_local_failure_world2():
    """dp._init_rccl on a box without a GPU: binding RCCL or touching the device fails LOCALLY on the ranks — the steps in front of the
    collective ncclCommInitRank end in agreements over the gloo control plane, so both ranks raise (the same outcome) and neither hangs
    (a rank that failed before emo_comm_init used to leave the others blocked inside it)."""
    import socket
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU (here the bring-up would succeed or fail inside RCCL)')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_rccl_init_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ['raised', 'raised'], res
    assert all('RCCL' in r[2] for r in res), res


def test_checkpoint_contract_param_order_and_init_rule():
    """The reference resumes optimizers by PARAMETER ORDER (stock Adam.state_dict(), train.py:318-326) and loads flat state dicts
    (train.py:304-311): names, order and shapes must match the imported reference (fixture: named_parameters() of the real MusicGPT2),
    and fresh models must follow weights_init (transformer_helpers.py:24-40): Linear/Embedding N(0, 0.01), bias 0, LayerNorm weight
    N(1, 0.01); HF Conv1D keeps its own N(0, 0.02)."""
    import json
    import os
    import numpy as np
    import torch
    from emo_disentanger_amd.model.music_gpt2 import MusicGPT2
    from emo_disentanger_amd.model.music_performer import MusicPerformer
    from oracle.weights import make_state_dict
    G = os.path.join(os.path.dirname(__file__), 'golden')
    man = json.load(open(os.path.join(G, 'manifest.json')))
    for name, c in man.items():
        ref_names = [str(n) for n in np.load(os.path.join(G, name + '.npz'))['grad_names']]
        nseg = None if c.get('noseg') else 2
        m = MusicGPT2(c['V'], c['L'], c['H'], c['d'], c['dff'], c['d'], use_segment_emb=nseg is not None, n_segment_types=nseg)
        assert [n for n, _ in m.named_parameters()] == ref_names, name
        sd = make_state_dict('gpt2', c['V'], c['L'], c['H'], c['d'], c['dff'], n_segment_types=nseg, seed=1)      # keys asserted == reference's at fixture time
        msd = m.state_dict()
        assert list(msd.keys()) == list(sd.keys()) and all(tuple(msd[k].shape) == tuple(sd[k].shape) for k in sd), name
    torch.manual_seed(0)
    g = MusicGPT2(327, 2, 8, 512, 2048, 512, use_segment_emb=True, n_segment_types=2).state_dict()
    p = MusicPerformer(327, 2, 8, 512, 2048, 512, favor_feature_dims=128, use_segment_emb=True, n_segment_types=2).state_dict()

    def stat(t):
        return float(t.float().mean()), float(t.float().std())
    for k, (mu, sd_) in {'token_emb.emb_lookup.weight': (0.0, 0.01), 'dec_out_proj.weight': (0.0, 0.01), 'transformer_decoder.0.ln_1.weight': (1.0, 0.01),
                         'transformer_decoder.0.attn.c_attn.weight': (0.0, 0.02), 'transformer_decoder.0.mlp.c_proj.weight': (0.0, 0.02)}.items():
        m_, s_ = stat(g[k])
        assert abs(m_ - mu) < 3e-3 and abs(s_ - sd_) < 0.15 * sd_, (k, m_, s_)
    assert float(g['dec_out_proj.bias'].abs().max()) == 0.0 and float(g['transformer_decoder.0.ln_1.bias'].abs().max()) == 0.0
    lp = 'transformer_decoder.decoder_layers.0.'
    for k, (mu, sd_) in {lp + 'attention.query_projection.weight': (0.0, 0.01), lp + 'linear1.weight': (0.0, 0.01), lp + 'norm1.weight': (1.0, 0.01),
                         'segemb.emb_lookup.weight': (0.0, 0.01)}.items():
        m_, s_ = stat(p[k])
        assert abs(m_ - mu) < 3e-3 and abs(s_ - sd_) < 0.15 * sd_, (k, m_, s_)
    assert float(p[lp + 'linear1.bias'].abs().max()) == 0.0
    # loaders filter 'feature_map.omega' (train.py:306, inference.py:411): the buffer is in the state dict under that name
    assert any(k.endswith('inner_attention.feature_map.omega') for k in p)


def test_product_positional_encoding_rows_match_reference_fixture():
    """a2: the PRODUCT's PositionalEncoding buffer (state-dict entry pe.pe, read by emo_embed_fwd) is bit-identical to rows dumped from the
    imported reference (transformer_helpers.py:43-63), without going through load_state_dict."""
    from emo_disentanger_amd.model.transformer_helpers import PositionalEncoding
    z = np.load(os.path.join(G, 'pe_rows_d512.npz'))
    pe = PositionalEncoding(512)
    assert tuple(pe.pe.shape) == (12000, 1, 512) and pe.pe.dtype == torch.float32
    assert np.array_equal(pe.pe[torch.from_numpy(z['rows']), 0].numpy(), z['pe'])
    assert torch.equal(pe(7), pe.pe[:7]) and tuple(pe(7, bsz=3).shape) == (7, 3, 512)


def test_event_piece_dataset_matches_reference():
    """data.EventPieceDataset on the reference's on-disk format (piece pickles + dictionary.pkl) returns, sample by sample, what the REAL
    REMISkylineToMidiTransformerDataset returned for the same files under the same `random` seeds (fixture: tools/make_golden_dataset.py):
    start-bar choice, padding / truncation to the window, shifted targets + EOS inside the full-arrangement spans, track mask, the
    predict_key variant, Chord / Note flags, vocabulary size and pad id."""
    import random
    from emo_disentanger_amd import data
    D = os.path.join(G, 'dataset')
    exp = json.load(open(os.path.join(D, 'expected.json')))
    names = ['p_long.pkl', 'p_mid.pkl', 'p_short.pkl']
    assert data.REMISkylineToMidiTransformerDataset is data.EventPieceDataset and data.load_split(os.path.join(D, 'train.pkl')) == names[:2]
    for key, e in exp.items():
        ds = data.EventPieceDataset(D, os.path.join(D, 'dictionary.pkl'), pieces=names, pad_to_same=True, **e['kw'])
        assert (ds.vocab_size, ds.pad_token) == (e['vocab_size'], e['pad_token']) and ds.piece_admissible_stbars == e['stbars']
        it = iter(e['samples'])
        for seed in (0, 1, 2):
            random.seed(seed)
            for i in range(len(ds)):
                got, want = ds[i], next(it)
                assert set(got) == set(want)
                for k, v in want.items():
                    g = got[k].tolist() if isinstance(got[k], np.ndarray) else got[k]
                    assert g == v, (key, seed, i, k)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3)))           # default collation gives the train loop's [B, T] int64 tensors
    assert batch['dec_input'].shape == (3, 400) and batch['dec_input'].dtype == torch.int64 and batch['track_mask'].max() == 1


def test_lead_sheet_reader_and_emotion_candidates(tmp_path):
    """Host helpers of the generation command line (reference inference.py:149-165, 431-447): bars are split at Bar_None, a leading Key_* line
    is the key (default Key_C), the emotion candidates come from the file name, unknown names raise ValueError('wrong emotion label')."""
    from emo_disentanger_amd import inference as inf
    e2i = {e: i for i, e in enumerate(['Key_G', 'Bar_None', 'Beat_0', 'Chord_I_M', 'Note_Pitch_60', 'Beat_8'])}
    f = tmp_path / 'samp_00_Positive_roman.txt'
    f.write_text('\n'.join(['Key_G', 'Bar_None', 'Beat_0', 'Chord_I_M', 'Bar_None', 'Beat_8', 'Note_Pitch_60']) + '\n')
    key, bars = inf.read_lead_sheet(str(f), e2i)
    assert key == 'Key_G' and bars == [[1, 2, 3], [1, 5, 4]]
    g = tmp_path / 'x_Q3.txt'
    g.write_text('Bar_None\nBeat_0\n')
    assert inf.read_lead_sheet(str(g), e2i) == ('Key_C', [[1, 2]])
    assert inf.emotions_of('samp_00_Positive_roman.txt') == ['Q1', 'Q4'] and inf.emotions_of('a_Negative.txt') == ['Q2', 'Q3']
    assert inf.emotions_of('a_Q2_x.txt') == ['Q2'] and inf.emotions_of('a_None.txt') == ['None']
    with pytest.raises(ValueError, match='wrong emotion label'):
        inf.emotions_of('plain.txt')
