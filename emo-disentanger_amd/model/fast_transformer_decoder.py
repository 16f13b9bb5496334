"""Parameter containers of the Performer decoder stack — mirrors
/root/reference/stage2_accompaniment/model/fast_transformer_decoder.py:13-74 and the module tree of
pytorch-fast-transformers it instantiates (TransformerEncoderLayer > AttentionLayer >
CausalLinearAttention > Favor), so that state_dict keys and parameter registration order are the
reference's.  Compute happens in emo_disentanger_amd.engine."""
import torch
from torch import nn


class _Favor(nn.Module):
    def __init__(self, d_head, n_dims):
        super().__init__()
        self.n_dims = n_dims
        self.register_buffer('omega', torch.zeros(d_head, n_dims // 2))


class _CausalLinearAttention(nn.Module):
    def __init__(self, d_head, n_dims, eps=1e-6):
        super().__init__()
        self.feature_map = _Favor(d_head, n_dims)
        self.eps = eps


class AttentionLayer(nn.Module):
    def __init__(self, d_model, n_heads, n_dims):
        super().__init__()
        self.inner_attention = _CausalLinearAttention(d_model // n_heads, n_dims)
        self.query_projection = nn.Linear(d_model, d_model)
        self.key_projection = nn.Linear(d_model, d_model)
        self.value_projection = nn.Linear(d_model, d_model)
        self.out_projection = nn.Linear(d_model, d_model)
        self.n_heads = n_heads


class TransformerEncoderLayer(nn.Module):
    def __init__(self, attention, d_model, d_ff, dropout=0.1, activation='relu'):
        super().__init__()
        if activation not in ('relu', 'gelu'):         # upstream: F.relu if activation == "relu" else F.gelu (fast_transformers/transformers.py)
            raise ValueError('activation must be "relu" or "gelu", got %r' % (activation,))
        self.activation = activation
        self.attention = attention
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)


class FastTransformerDecoder(nn.Module):
    def __init__(self, n_layer, n_head, d_model, d_ff, dropout=0.1, activation='relu', favor_feature_dims=None):
        super().__init__()
        self.n_layer, self.n_head, self.d_model, self.d_ff = n_layer, n_head, d_model, d_ff
        self.dropout, self.activation = dropout, activation
        self.favor_feature_dims = 2 * d_model // n_head if favor_feature_dims is None else favor_feature_dims
        self.attention_layers = [AttentionLayer(d_model, n_head, self.favor_feature_dims) for _ in range(n_layer)]
        self.decoder_layers = nn.ModuleList(
            [TransformerEncoderLayer(self.attention_layers[l], d_model, d_ff, dropout, activation) for l in range(n_layer)])

    def forward(self, x, lengths=None, attn_kwargs=None):
        raise RuntimeError('FastTransformerDecoder is a parameter container; call MusicPerformer.forward')
