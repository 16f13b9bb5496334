"""ctypes binding of libemo_hip.so (C-ABI declared in include/emo_hip.h).

The library is built IN-TREE by ``__graft_entry__.build()`` / ``make -C emo-disentanger_amd/csrc``.
Missing library => ImportError (the product path has no CPU / eager fallback)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libemo_hip.so')

F32, BF16, I64 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU_NEW, ACT_GELU = 0, 1, 2, 3
MUL_NONE, MUL_NONZERO, MUL_DGELU_NEW, MUL_BITMASK, MUL_DGELU = 0, 1, 2, 3, 4

if not os.path.exists(LIB_PATH):
    raise ImportError('libemo_hip.so not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                      'or `make -C emo-disentanger_amd/csrc` (hipcc --offload-arch=gfx950)')
lib = ctypes.CDLL(LIB_PATH)

c_p, c_i, c_l, c_f, c_u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64


class Epilogue(ctypes.Structure):
    _fields_ = [('bias', c_p), ('act', c_i), ('aux_out', c_p), ('mul_aux', c_p), ('mul_mode', c_i), ('mul_scale', c_f),
                ('p_drop', c_f), ('seed', c_u64), ('offset', c_u64), ('residual', c_p),
                ('ln_c1', c_p), ('ln_eps', c_f), ('ln_stats_out', c_p), ('rln_x', c_p), ('rln_stats', c_p), ('rln_gamma', c_p), ('rln_beta', c_p), ('a_rowsum', c_p), ('b_rowsum', c_p),
                ('mask_out', c_p), ('workspace', c_p), ('workspace_bytes', c_l),
                ('lna_gamma', c_p), ('lna_beta', c_p), ('lna_out', c_p), ('lna_mean', c_p), ('lna_rstd', c_p), ('hdiv', c_p), ('hdiv_T', c_l)]


_SIG = {
    'emo_version': (c_i, []),
    'emo_build_flags': (c_i, []),
    'emo_last_error': (ctypes.c_char_p, []),
    'emo_device_cus': (c_i, []),
    'emo_gemm': (c_i, [c_p, c_i, c_l, c_p, c_i, c_l, c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_i, ctypes.POINTER(Epilogue), c_p]),
    'emo_gemm_last_kernel': (c_i, []),
    'emo_epilogue_size': (c_i, []),
    'emo_gemm_workspace_bytes': (c_l, [c_l, c_l, c_l, c_i, c_i]),
    'emo_ffn_fwd_supported': (c_i, [c_i, c_l, c_l, c_l]),
    'emo_ffn_fwd': (c_i, [c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_i, c_f, c_u64, c_u64, c_u64, c_p]),
    'emo_colsum': (c_i, [c_p, c_i, c_l, c_l, c_l, c_p, c_i, c_p]),
    'emo_embed_fwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_l, c_l, c_l, c_l, c_l, c_p, c_f, c_f, c_u64, c_u64, c_p]),
    'emo_embed_bwd': (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_l, c_l, c_l, c_l, c_f, c_f, c_u64, c_u64, c_p]),
    'emo_layernorm_fwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_l, c_f, c_p]),
    'emo_layernorm_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_layernorm_bwd_workspace_bytes': (c_l, [c_i, c_l, c_l]),
    'emo_layernorm_bwd_ws': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_l, c_f, c_u64, c_u64, c_p, c_l, c_p]),
    'emo_dropout_apply': (c_i, [c_p, c_p, c_i, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_favor_attn_workspace_bytes': (c_l, [c_l, c_l, c_l, c_l, c_l]),
    'emo_favor_attn_fwd': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_p, c_i, c_l, c_l, c_l, c_l, c_l, c_f, c_p, c_l, c_p]),
    'emo_favor_attn_bwd': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_l, c_f, c_p, c_l, c_p]),
    'emo_favor_attn_bwd_kstate': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_l, c_f, c_p, c_l, c_i, c_p]),
    'emo_favor_attn_bwd_dn_supported': (c_i, [c_i, c_l, c_l, c_l, c_l, c_l]),
    'emo_favor_attn_bwd_dn': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_l, c_f, c_p]),
    'emo_favor_decode_step': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_f, c_p]),
    'emo_performer_decode_step_workspace_bytes': (c_l, []),
    'emo_performer_decode_step_supported': (c_i, []),
    'emo_performer_decode_step': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_f, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_l, c_p, c_l, c_f, c_f, c_p, c_p]),
    'emo_performer_decode_step_sampled': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_f, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_l, c_l, c_p, c_l, c_f, c_f,
                                                c_f, c_f, c_p, c_p, c_p, c_l, c_l, c_p, c_p]),
    'emo_gpt2_decode_step_supported': (c_i, []),
    'emo_gpt2_decode_step': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_f, c_l, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_p, c_l, c_f, c_p, c_p]),
    'emo_gpt2_decode_step_sampled': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_f, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_l, c_p, c_l, c_f,
                                           c_f, c_f, c_p, c_p, c_p, c_l, c_l, c_p, c_p]),
    'emo_favor_draw_omega': (c_i, [c_p, c_p, c_l, c_l, c_l, c_p]),
    'emo_softmax_attn_fwd': (c_i, [c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_softmax_attn_bwd': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_softmax_attn_keep_bytes': (c_l, [c_i, c_l, c_l, c_l, c_l, c_f]),
    'emo_softmax_attn_fwd_keep': (c_i, [c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p, c_l, c_p]),
    'emo_softmax_attn_bwd_keep': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p, c_l, c_p]),
    'emo_relpos_attn_fwd': (c_i, [c_p, c_p, c_p, c_l, c_p, c_l, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_relpos_attn_bwd_kv': (c_i, [c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_relpos_attn_bwd': (c_i, [c_p, c_p, c_p, c_l, c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_relpos_attn_bwd_r': (c_i, [c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_f, c_u64, c_u64, c_p]),
    'emo_relpos_attn_bwd_r_workspace_bytes': (c_l, [c_l, c_l, c_l, c_l]),
    'emo_relpos_attn_decode': (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_p, c_p, c_l, c_p, c_l, c_l, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_p]),
    'emo_softmax_attn_decode': (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_l, c_l, c_l, c_p]),
    'emo_softmax_attn_decode_layout': (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_l, c_l, c_l, c_i, c_p]),
    'emo_xent_fwd': (c_i, [c_p, c_p, c_l, c_l, c_l, c_p, c_p, c_p]),
    'emo_xent_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_l, c_l, c_l, c_p]),
    'emo_argmax': (c_i, [c_p, c_l, c_l, c_p, c_p]),
    'emo_sample_nucleus': (c_i, [c_p, c_l, c_l, c_f, c_f, c_p, c_p, c_p]),
    'emo_sample_nucleus_step': (c_i, [c_p, c_l, c_l, c_f, c_f, c_p, c_p, c_p, c_l, c_l, c_p, c_p]),
    'emo_accuracy_counts': (c_i, [c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_p, c_p]),
    'emo_sumsq': (c_i, [c_p, c_l, c_p, c_p]),
    'emo_clip_coef': (c_i, [c_p, c_f, c_f, c_p, c_p, c_p]),
    'emo_adam_step': (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_l, c_p, c_p]),
    'emo_cast': (c_i, [c_p, c_i, c_p, c_i, c_l, c_p]),
    'emo_add_bias2': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_i, c_l, c_l, c_p]),
    'emo_transpose_batch': (c_i, [c_p, c_i, c_l, c_p]),
    'emo_stream_wait': (c_i, [c_p, c_p]),
    'emo_comm_bind': (c_i, []),
    'emo_comm_unique_id': (c_i, [c_p]),
    'emo_comm_init': (c_i, [c_p, c_i, c_i]),
    'emo_comm_world': (c_i, []),
    'emo_comm_rank': (c_i, []),
    'emo_comm_allreduce': (c_i, [c_p, c_l, c_i, c_p]),
    'emo_comm_broadcast': (c_i, [c_p, c_l, c_i, c_i, c_p]),
    'emo_comm_destroy': (c_i, []),
}
for _name, (_res, _args) in _SIG.items():
    _fn = getattr(lib, _name)          # AttributeError here = header/library mismatch
    _fn.restype, _fn.argtypes = _res, _args
if lib.emo_epilogue_size() != ctypes.sizeof(Epilogue):      # a stale libemo_hip.so (or a stale mirror above) would read garbage pointers
    raise ImportError('libemo_hip.so was built with an emo_epilogue_t of %d bytes, this binding has %d: rebuild (python -c "import __graft_entry__ as g; g.build()")'
                      % (lib.emo_epilogue_size(), ctypes.sizeof(Epilogue)))


class EmoError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise EmoError('libemo_hip error %d: %s' % (rc, lib.emo_last_error().decode()))


def dtype_code(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise EmoError('unsupported dtype %s (float32 / bfloat16 only)' % t)


def ptr(t):
    """device pointer of a tensor (None -> NULL). Tensors must live on the GPU: no CPU fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EmoError('emo-disentanger_amd ops need GPU tensors (the HIP path has no CPU fallback)')
    return t.data_ptr()


def stream():
    """hipStream_t of torch's current stream on the current device.  (The raw getter: `torch.cuda.current_stream().cuda_stream` builds a Stream
    object through four layers of Python per call — 8.5 us, ~200 calls per training step, a quarter of the host time that BOUNDS the step at the
    reference YAML's batch size 4; tools/b4_host_profile.py, r05.)"""
    return _raw_stream(_cur_device())


# (private torch entry points, present in every torch with a CUDA / ROCm build this package supports; the public form is the fall-back)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)
if _raw_stream is None or _cur_device is None:
    def stream():                                                  # noqa: F811
        return torch.cuda.current_stream().cuda_stream
