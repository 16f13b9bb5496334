"""emo-disentanger_amd — MI355X-native implementation of EMO-Disentanger's stage-2 causal-LM hot path.

Python host code (this package) mirrors the reference's object contract
(stage2_accompaniment/model/{music_performer,music_gpt2}.py, train.py step, inference.py loop)
and calls hand-written gfx950 HIP kernels through the C-ABI of ``libemo_hip.so``
(include/emo_hip.h) with ctypes.  PyTorch-ROCm is used only for device memory, streams,
autograd glue and torch.distributed (RCCL).  There is NO CPU fallback: every op raises if the
HIP library or a GPU is missing.
"""
__version__ = '0.1.0'

from . import _lib  # noqa: F401  (fails loudly if libemo_hip.so is missing)
