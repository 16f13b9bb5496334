"""Data-parallel plumbing: one process per GPU.  The reference has NO distributed code; the build
adds ONE exchange per optimizer step — a sum all-reduce of the flat fp32 gradient buffer (152.7 MB at the perf config) placed
between backward() and the clip (/root/reference/stage2_accompaniment/train.py:76-81) — plus one broadcast of parameters and
FAVOR+ omega from rank 0 at start-up.

Two planes:
  * control plane  — a torch.distributed "gloo" group on host memory: rendezvous, barriers, the max-over-ranks of the bench
    timing, and the side channel that ships RCCL's 128-byte unique id from rank 0 to the other ranks;
  * data plane     — `emo_comm_allreduce` / `emo_comm_broadcast` of libemo_hip.so (include/emo_hip.h): RCCL over xGMI on the
    caller's HIP stream, in place on the flat buffers.  EMO_COMM selects it: "rccl" (default on a GPU), "nccl"
    (torch.distributed's RCCL binding, kept as the escape hatch when the direct binding cannot initialise), "gloo"
    (host-staged; the CPU tests, and N processes sharing ONE GPU where RCCL refuses duplicate devices).

Exact global mean with unequal token counts (SURVEY §7 "DP exactness"): each rank back-propagates the SUM of its token losses
(`loss * n_tokens`), the token count rides in the last slot of the all-reduced buffer, and the fused optimizer divides by the
all-reduced count (optim.FusedAdam(token_weighted=True)); with equal counts this equals the plain 1/world average."""
import ctypes
import os
import sys

import torch
import torch.distributed as dist

_STATE = {'plane': None, 'nccl_group': None, 'bucketed_steps': 0}


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def data_plane():
    """'rccl' | 'nccl' | 'gloo' | None (single process)."""
    return _STATE['plane']


def _init_rccl(rank, world):
    """Three steps, each ending in an agreement over the gloo control plane, so that every rank raises the SAME outcome and none is left
    blocked in a collective the others never enter: (1) LOCAL — every rank dlopen()s RCCL and resolves its symbols (emo_comm_bind: no
    unique id is drawn, so no bootstrap listener thread / socket is started on the ranks that do not need one); (2) only if ALL ranks
    could bind: rank 0 draws the id, it is shipped, and the collective ncclCommInitRank + a probe all-reduce run; (3) the ranks agree on
    the outcome of (2) — an init error, a probe mismatch or an asynchronous RCCL error on ONE rank is raised on ALL of them."""
    from ._lib import I64, check, lib
    bound = lib.emo_comm_bind() == 0
    ok = torch.tensor([1 if bound else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                         # control plane (gloo, host memory)
    if int(ok) == 0:
        raise RuntimeError('RCCL could not be bound on every rank (rank %d: %s)' % (rank, 'ok' if bound else lib.emo_last_error().decode()))
    # local steps that can fail on ONE rank (the device touch) happen BEFORE the collective init and are agreed on, and the id travels with a status
    # byte: a rank never enters ncclCommInitRank alone, and a failed draw is not left to RCCL's handling of an all-zero bootstrap address
    local_err = None
    try:
        torch.empty(1, device='cuda')                                 # the HIP context of this thread is on the rank's device before RCCL binds it
        torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001 — reported after the agreement below
        local_err = e
    msg = torch.zeros(129, dtype=torch.uint8)                         # [0:128] = the id, [128] = 1 when rank 0 could draw it
    if rank == 0:
        buf = (ctypes.c_char * 128)()
        if lib.emo_comm_unique_id(buf) == 0:
            msg[:128] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
            msg[128] = 1
    dist.broadcast(msg, src=0)
    ok = torch.tensor([1 if (local_err is None and int(msg[128]) == 1) else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok) == 0:
        raise RuntimeError('RCCL plane not started (rank %d: %s)' % (rank, local_err if local_err is not None else
                                                                     ('rank 0 could not draw the unique id' if int(msg[128]) == 0 else 'ok here, failed on another rank')))
    msg = msg[:128]
    err = None
    try:
        check(lib.emo_comm_init(ctypes.c_char_p(bytes(msg.numpy().tobytes())), rank, world))
        probe = torch.full((4,), rank + 1, device='cuda', dtype=torch.int64)
        check(lib.emo_comm_allreduce(probe.data_ptr(), 4, I64, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()                                      # an asynchronous RCCL error surfaces here, not as a wrong sum
        if int(probe[0].item()) != world * (world + 1) // 2:
            raise RuntimeError('emo_comm self-check failed: all-reduce of rank+1 gave %d for world %d' % (int(probe[0]), world))
    except Exception as e:   # noqa: BLE001 — reported after the agreement below
        err = e
    ok = torch.tensor([0 if err is not None else 1])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok) == 0:
        lib.emo_comm_destroy()
        raise RuntimeError('RCCL plane failed to initialise (rank %d: %s)' % (rank, err if err is not None else 'ok here, failed on another rank'))


def init_distributed(backend=None, strict=None):
    """Idempotent; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract).  `backend` (or EMO_COMM /
    EMO_DIST_BACKEND) picks the data plane; the control plane is always gloo.  strict (or EMO_COMM_STRICT=1): no fall-back from the
    C-ABI RCCL plane to torch.distributed's binding — the failure is raised (bench.py: a run that lands on another plane would not be
    measuring emo_comm_*)."""
    if strict is None:
        strict = os.environ.get('EMO_COMM_STRICT', '0') == '1'
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        plane = backend or os.environ.get('EMO_COMM') or os.environ.get('EMO_DIST_BACKEND') or ('rccl' if torch.cuda.is_available() else 'gloo')
        if plane not in ('rccl', 'nccl', 'gloo'):
            raise ValueError('EMO_COMM must be rccl | nccl | gloo, got %r' % plane)
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        if plane == 'rccl':
            try:
                _init_rccl(rank, world)
            except Exception as e:   # noqa: BLE001 — stay up on the torch binding of the same RCCL rather than lose the run
                if strict:
                    raise
                print('[emo dp] rank %d: direct RCCL binding failed (%s); falling back to torch.distributed nccl' % (rank, e), file=sys.stderr, flush=True)
                plane = 'nccl'
            ok = torch.tensor([1 if plane == 'rccl' else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)                 # all ranks agree on one plane
            if int(ok) == 0 and plane == 'rccl':
                from ._lib import lib
                lib.emo_comm_destroy()
                plane = 'nccl'
        if plane == 'nccl':
            _STATE['nccl_group'] = dist.new_group(backend='nccl')
        _STATE['plane'] = plane
    return rank, local_rank, world


def shutdown():
    if _STATE['plane'] == 'rccl':
        from ._lib import lib
        lib.emo_comm_destroy()
    _STATE['plane'], _STATE['nccl_group'] = None, None
    if dist.is_initialized():
        dist.destroy_process_group()


def _comm_dtype(t):
    from ._lib import BF16, F32, I64
    return {torch.float32: F32, torch.bfloat16: BF16, torch.int64: I64}[t.dtype]


def allreduce_sum_(flat):
    """In-place sum over ranks of one contiguous buffer (no bucketing: a single large message suits the point-to-point xGMI
    links; DESIGN.md §6 has the cost model)."""
    plane = _STATE['plane']
    if plane is None:
        return flat
    assert flat.is_contiguous()
    if plane == 'rccl' and flat.is_cuda:
        from ._lib import check, lib
        check(lib.emo_comm_allreduce(flat.data_ptr(), flat.numel(), _comm_dtype(flat), torch.cuda.current_stream().cuda_stream))
    elif plane == 'nccl' and flat.is_cuda:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_STATE['nccl_group'])
    elif flat.is_cuda:                                                # gloo plane with device buffers: staged through host memory
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_grads_(store, n_tokens=None):
    """The one exchange per optimizer step: sum of the flat gradient buffer over ranks.  n_tokens (device scalar or number): this
    rank's non-pad target count, carried in the buffer's tail slot so that ONE collective moves both (token-weighted mean)."""
    if _STATE['plane'] is None:
        return
    store.tail_tokens = n_tokens is not None                          # FusedAdam.step checks this against its token_weighted mode
    if n_tokens is None:
        allreduce_sum_(store.flat_grad)
        return
    tail = store.flat_grad_ext[store.total:]
    tail.zero_()
    tail[0:1].copy_(torch.as_tensor(n_tokens, dtype=torch.float32).reshape(1))
    allreduce_sum_(store.flat_grad_ext)


def _late_layer_range(model, store):
    """(a, b, split): the flat-buffer element range [a, b) that holds exactly the parameters of decoder layers split .. L-1 (the layers
    whose gradients are complete first in the backward sweep), or None when the layout does not allow it."""
    if not hasattr(model, '_layer_prefix') or not hasattr(model, 'n_layer') or model.n_layer < 2:
        return None
    L = model.n_layer
    split = L // 2
    try:
        late = tuple(model._layer_prefix(l) for l in range(split, L))
    except NotImplementedError:
        return None
    ends = sorted(store.offsets.values()) + [store.total]
    nxt = {o: e for o, e in zip(ends[:-1], ends[1:])}                 # offset -> offset of the next parameter (alignment padding included)
    inside = [n for n in store.offsets if n.startswith(late)]
    if not inside:
        return None
    a, b = min(store.offsets[n] for n in inside), max(nxt[store.offsets[n]] for n in inside)
    if any(a <= o < b for n, o in store.offsets.items() if not n.startswith(late)):
        return None                                                   # not contiguous: keep the single exchange
    return a, b, split


class GradExchange:
    """The gradient exchange of one optimizer step, overlapped with the backward sweep: the gradients of the late
    decoder layers (L/2 .. L-1, complete half-way through the backward) are all-reduced on a communication stream while the early layers
    are still being back-propagated; the rest of the flat buffer (embeddings, output projection, early layers, the token-count tail)
    follows when the backward is done.  All collectives of a step go to ONE stream in the same order on every rank.  Planes without
    asynchronous collectives (gloo) and models without the layer hook fall back to the single all-reduce of allreduce_grads_.
    EMO_DP_BUCKETS=0 disables the split, =force enables it on the gloo plane too (tests)."""

    def __init__(self, model, store=None):
        self.model, self.store = model, store or model._ensure_store()
        self.range = _late_layer_range(model, self.store)
        self.stream, self._fired = None, False

    def _enabled(self):
        mode = os.environ.get('EMO_DP_BUCKETS', '1')
        return self.range is not None and mode != '0' and (_STATE['plane'] in ('rccl', 'nccl') or (mode == 'force' and _STATE['plane'] is not None))

    def arm(self):
        """Call before backward() of the micro-step that ends the accumulation window."""
        self._fired = False
        if self._enabled():
            self.model._bwd_hook = self._on_layer

    def _comm(self):
        """The communication stream (None for host buffers: the gloo collectives are synchronous there)."""
        if self.stream is None and torch.device(self.store.device).type == 'cuda':
            self.stream = torch.cuda.Stream(device=self.store.device)
        return self.stream

    @staticmethod
    def _on(cs):
        import contextlib
        return torch.cuda.stream(cs) if cs is not None else contextlib.nullcontext()

    def _on_layer(self, l):
        if self._fired or l != self.range[2]:
            return
        cs = self._comm()
        if cs is not None:
            from .engine import join_side_stream
            join_side_stream()                                        # the weight gradients of layers >= split run on the side stream
            cs.wait_stream(torch.cuda.current_stream())
        with self._on(cs):
            allreduce_sum_(self.store.flat_grad_ext[self.range[0]:self.range[1]])
        self._fired = True

    def finish(self, n_tokens=None):
        """Call after backward(): exchanges what is left; on return the current stream sees the summed gradients."""
        self.model._bwd_hook = None
        if _STATE['plane'] is None:
            return
        if not self._fired:
            allreduce_grads_(self.store, n_tokens)
            return
        st = self.store
        st.tail_tokens = n_tokens is not None
        a, b = self.range[0], self.range[1]
        end = st.total
        if n_tokens is not None:
            tail = st.flat_grad_ext[st.total:]
            tail.zero_()
            tail[0:1].copy_(torch.as_tensor(n_tokens, dtype=torch.float32).reshape(1))
            end = st.flat_grad_ext.numel()
        cs = self._comm()
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream())
        with self._on(cs):
            if a > 0:
                allreduce_sum_(st.flat_grad_ext[:a])
            if end > b:
                allreduce_sum_(st.flat_grad_ext[b:end])
        if cs is not None:
            torch.cuda.current_stream().wait_stream(cs)
        self._fired = False
        _STATE['bucketed_steps'] += 1


def broadcast_(tensors, src=0):
    plane = _STATE['plane']
    if plane is None:
        return
    for t in tensors:
        assert t.is_contiguous()
        if plane == 'rccl' and t.is_cuda:
            from ._lib import check, lib
            check(lib.emo_comm_broadcast(t.data_ptr(), t.numel(), _comm_dtype(t), src, torch.cuda.current_stream().cuda_stream))
        elif plane == 'nccl' and t.is_cuda:
            dist.broadcast(t, src=src, group=_STATE['nccl_group'])
        elif t.is_cuda:
            host = t.cpu()
            dist.broadcast(host, src=src)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def sync_model_from_rank0(model):
    """Identical replicas: rank 0's weights, omega buffers AND the seed of the omega generator (so that the per-forward FAVOR+
    redraws stay identical on every rank); dropout streams are independent per rank."""
    ps = model._ensure_store()
    bufs = [ps.flat32] + [b for n, b in model.named_buffers() if 'omega' in n]
    broadcast_(bufs)
    ps.invalidate_mirror()
    rank = dist.get_rank() if dist.is_initialized() else 0
    seed = torch.tensor([int(getattr(model, '_seed', 0))], dtype=torch.int64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(seed, src=0)
    if hasattr(model, 'set_omega_seed'):
        model.set_omega_seed(int(seed) ^ 0x5DEECE66D)
    model.set_dropout_seed(int(seed) + 7919 * rank)


def shard_seed(base_seed, rank):
    """Weak scaling: every rank draws its own B sequences (default_rng(1234 + rank))."""
    return base_seed + rank


def max_over_ranks(value, device=None):
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # control plane
    return float(t.item())
