"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI
on ROCm; "gloo" for the CPU tests).  The reference has NO distributed code; the build
adds ONE exchange per optimizer step — a sum all-reduce of the flat fp32 gradient buffer (152.7 MB
at the perf config) placed between backward() and the clip (train.py:76-80) — plus an initial
broadcast of parameters and FAVOR+ omega from rank 0.  The 1/world scaling is folded into the clip
coefficient of the fused Adam step (optim.FusedAdam)."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_distributed(backend=None):
    """Idempotent; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = backend or os.environ.get('EMO_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def allreduce_sum_(flat):
    """In-place sum over ranks of one flat buffer (no bucketing: a single large message suits the
    point-to-point xGMI links; see DESIGN.md §multi-GPU for the cost model)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def broadcast_(tensors, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src=src)


def sync_model_from_rank0(model):
    """Replicate rank 0's weights and omega buffers (identical replicas; independent dropout streams per rank)."""
    ps = model._ensure_store()
    bufs = [ps.flat32] + [b for n, b in model.named_buffers() if 'omega' in n]
    broadcast_(bufs)
    ps.flat32.add_(0)          # bump the version counter => bf16 mirror refresh on next forward
    rank = dist.get_rank() if dist.is_initialized() else 0
    model.set_dropout_seed(model._seed + 7919 * rank)


def shard_seed(base_seed, rank):
    """Weak scaling: every rank draws its own B sequences (default_rng(1234 + rank))."""
    return base_seed + rank


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
