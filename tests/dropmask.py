"""TEST INFRASTRUCTURE: read the dropout multipliers the HIP path uses back out of the library, so that the oracle can be run with the SAME
draw (oracle.model_ref `masks=`) and a dropout-ON forward / backward can be compared element for element.

Every product kernel regenerates its keep decision from (seed, per-site offset, flat element index) — include/emo_hip.h, "dropout".  The
multipliers of a site are what `emo_dropout_apply` writes for an all-ones tensor of the site's shape with the site's (seed, offset); the site
offsets are the engine's (emo-disentanger_amd/engine.py: embedding = base, layer l = base + 8 (l + 1) + {1, 2, 3}).  That the fused kernels
(GEMM epilogues, LayerNorm backward, embedding, attention probabilities) index elements the way this read-out assumes is exactly what the
parity tests that consume these masks check: a kernel with a different element -> hash mapping fails them.  The attention-probability site is
additionally read out of the attention kernels themselves (tests/test_gpu_dropout_parity.py::attention_keep_from_kernel)."""
import torch


def site_multipliers(shape, p, seed, offset):
    """[shape] fp32 on the CPU: keep / (1 - p) of every element of the site (seed, offset)."""
    from emo_disentanger_amd import ops
    return ops.dropout_apply(torch.ones(shape, device='cuda', dtype=torch.float32), p, seed, offset).cpu()


def export_masks(kind, p, seed, base, B, T, D, d_ff, H, L):
    """Multipliers of every dropout site of ONE forward of MusicPerformer / MusicGPT2 (kind = 'performer' | 'gpt2') whose dropout base is
    `base` (the model's n-th forward after set_dropout_seed(seed) has base = 4096 n), keyed by the oracle's site names."""
    m = {'emb': site_multipliers((B, T, D), p, seed, base)}
    for l in range(L):
        off = base + 8 * (l + 1)
        if kind == 'performer':
            m['L%d.attn_out' % l] = site_multipliers((B, T, D), p, seed, off + 1)
            m['L%d.ffn_hidden' % l] = site_multipliers((B, T, d_ff), p, seed, off + 2)
            m['L%d.ffn_out' % l] = site_multipliers((B, T, D), p, seed, off + 3)
        else:
            m['L%d.attn_prob' % l] = site_multipliers((B, H, T, T), p, seed, off + 1)
            m['L%d.attn_out' % l] = site_multipliers((B, T, D), p, seed, off + 2)
            m['L%d.mlp_out' % l] = site_multipliers((B, T, D), p, seed, off + 3)
    return m


def slice_batch(masks, b):
    """The masks of sequence b alone (batch dimension kept, size 1)."""
    return {k: v[b:b + 1] for k, v in masks.items()}


# ---------------------------------------------------------------------------------------------------- stage 1 (Transformer-XL)
def load_txl_dropout_fixture(path, c):
    """tests/golden/txl_dropout_*.npz (tools/make_golden_stage1_dropout.py: the IMPORTED reference run in training mode, torch's Bernoulli
    draws stored as packed keep-bits) -> (npz, masks) with masks = the oracle's `masks=` dict (multipliers, reference time-major shapes)."""
    import numpy as np
    g = np.load(path)
    T, B, D, dff, H, L, p = c['T'], c['B'], c['d'], c['dff'], c['H'], c['L'], c['p']
    shapes = {'emb': (T, B, D), 'emb2': (T, B, D), 'pos': (T, 1, D), 'final': (T, B, D)}
    for l in range(L):
        shapes.update({'L%d.attn_prob' % l: (T, T, B, H), 'L%d.attn_out' % l: (T, B, D), 'L%d.ffn_hidden' % l: (T, B, dff), 'L%d.ffn_out' % l: (T, B, D)})
    inv = float(torch.tensor(1.0) / (1.0 - p))
    masks = {}
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        masks[k] = torch.from_numpy(np.unpackbits(g['keep_' + k])[:n].astype(np.float32).reshape(shp)) * inv
    return g, masks


def export_txl_masks(p, seed, base, B, T, D, d_ff, H, L, attn_keep):
    """Multipliers of every dropout site of ONE training forward of the product's PlainTransformer (model/plain_transformer.py: TXLStackFn —
    embedding = base, decoder.drop on it = base + 1, final = base + 2, pos_emb = base + 3, layer l = base + 8 (l + 1) + {1 attention
    probabilities, 2 o_net output, 3 CoreNet hidden, 4 CoreNet output}), converted to the oracle's (= the reference's) time-major shapes.
    The product indexes batch-major ([B, T, ...]) and keeps pos_emb by DISTANCE (row d = distance d; the reference's row i is distance
    klen - 1 - i).  attn_keep(l) -> [B, H, T, T] multipliers of layer l's attention probabilities, read out of the attention kernel itself."""
    tm = lambda shape, off: site_multipliers(shape, p, seed, off).transpose(0, 1).contiguous()          # [B, T, X] -> [T, B, X]
    m = {'emb': tm((B, T, D), base), 'emb2': tm((B, T, D), base + 1), 'final': tm((B, T, D), base + 2),
         'pos': site_multipliers((T, D), p, seed, base + 3).flip(0)[:, None, :].contiguous()}
    for l in range(L):
        off = base + 8 * (l + 1)
        m['L%d.attn_prob' % l] = attn_keep(l).permute(2, 3, 0, 1).contiguous()                          # [B, H, Tq, Tk] -> [Tq, Tk, B, H]
        m['L%d.attn_out' % l] = tm((B, T, D), off + 2)
        m['L%d.ffn_hidden' % l] = tm((B, T, d_ff), off + 3)
        m['L%d.ffn_out' % l] = tm((B, T, D), off + 4)
    return m
