"""GPU: the RCCL leg of the data-parallel path on the one GPU a test box has.  World size 1 cannot exercise the exchange itself (the
gloo world-2 test in test_host_logic.py does, on CPU), but it proves that torch.distributed's "nccl" backend (= RCCL) initialises in this
image with the environment dp.init_distributed sets and that the flat-gradient all-reduce / parameter broadcast / max-over-ranks calls
run on device tensors on the bench's stream."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["EMO_ROOT"])
from emo_disentanger_amd import dp
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
flat = torch.arange(1 << 22, device="cuda", dtype=torch.float32)
ref = flat.clone()
dist.all_reduce(flat, op=dist.ReduceOp.SUM)          # what dp.allreduce_sum_ issues when world > 1
dist.broadcast(flat, src=0)                          # dp.broadcast_
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(flat, ref)
assert dp.allreduce_sum_(flat) is flat and dp.max_over_ranks(1.25, flat.device) == 1.25
assert dp.shard_seed(1234, 3) == 1237
dist.destroy_process_group()
print("rccl ok")
'''


def test_rccl_single_rank_collectives_run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMO_ROOT=root, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'rccl ok' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
