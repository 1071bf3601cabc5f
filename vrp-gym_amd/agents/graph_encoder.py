"""Encoder parameter containers + HIP forward (reference: agents/graph_encoder.py).

The torch modules below exist for three reasons only: (1) identical
`state_dict()` keys/shapes so reference checkpoints load, (2) identical
initialisation (the same modules are constructed in the same order, so a seed
yields the reference's initial weights), (3) an optimiser can own the
parameters.  Their `forward` never runs torch math: it hands the parameter
pointers to `vrp_encoder_forward` (include/vrpgym_hip.h).
"""
import torch
import torch.nn as nn

from . import runtime


class BatchNorm(nn.Module):
    """Holder of one nn.BatchNorm1d (key prefix `.norm.`); graph_encoder.py:141-154."""

    def __init__(self, feature_dim):
        super().__init__()
        self.norm = nn.BatchNorm1d(feature_dim)


class MultiHeadAttentionLayer(nn.Module):
    """graph_encoder.py:157-198 as a container: attention_layer, bn1, bn2, ff.{0,2}."""

    def __init__(self, embedding_dim, hidden_dim, num_heads):
        super().__init__()
        self.attention_layer = nn.MultiheadAttention(embedding_dim, num_heads, batch_first=True)
        self.bn1 = BatchNorm(embedding_dim)
        self.bn2 = BatchNorm(embedding_dim)
        self.ff = nn.Sequential(nn.Linear(embedding_dim, hidden_dim), nn.ReLU(),
                                nn.Linear(hidden_dim, embedding_dim))


class GraphEncoder(nn.Module):
    def __init__(self, node_input_dim, embedding_dim=128, hidden_dim=512,
                 num_attention_layers=3, num_heads=8):
        super().__init__()
        self.node_embed = nn.Linear(node_input_dim, embedding_dim)
        self.attention_layers = nn.ModuleList(
            MultiHeadAttentionLayer(embedding_dim, hidden_dim, num_heads)
            for _ in range(num_attention_layers))
        self._dims = (node_input_dim, embedding_dim, hidden_dim, num_heads)
        runtime.check_supported_dims(embedding_dim, num_heads, hidden_dim)

    def _apply(self, fn, *a, **k):  # .to()/.cuda() re-allocate parameters
        runtime.invalidate(self)
        return super()._apply(fn, *a, **k)

    def forward(self, x, depot_mask=None):
        """x (B,N,F) -> (B,N,128) on this module's device (graph_encoder.py:41-58)."""
        return runtime.encoder_forward(self, x, None, self.training)


class GraphDemandEncoder(GraphEncoder):
    def __init__(self, depot_input_dim, node_input_dim, embedding_dim=128, hidden_dim=512,
                 num_attention_layers=3, num_heads=8):
        super().__init__(node_input_dim, embedding_dim, hidden_dim, num_attention_layers,
                         num_heads)
        self.node_f_dim, self.depot_f_dim, self.emb_dim = node_input_dim, depot_input_dim, embedding_dim
        self.depot_embed = nn.Linear(depot_input_dim, embedding_dim)

    def forward(self, x, depot_mask):
        """graph_encoder.py:95-138: depot rows use depot_embed, the rest node_embed."""
        return runtime.encoder_forward(self, x, depot_mask, self.training)
