"""Decoder parameter container + HIP decode step (reference: agents/graph_decoder.py)."""
import torch
import torch.nn as nn

from . import runtime


class GraphDecoder(nn.Module):
    def __init__(self, emb_dim=128, num_heads=8, v_dim=128, k_dim=128):
        super().__init__()
        # creation order == RNG order of the reference (graph_decoder.py:29-44)
        self._first_node = nn.Parameter(torch.rand(1, 1, emb_dim))
        self._last_node = nn.Parameter(torch.rand(1, 1, emb_dim))
        self.attention = nn.MultiheadAttention(3 * emb_dim, num_heads, kdim=k_dim, vdim=v_dim,
                                               batch_first=True)
        self._kp = nn.Linear(emb_dim, emb_dim, bias=False)
        self._att_output = nn.Linear(emb_dim * 3, emb_dim, bias=False)
        self._context_proj = nn.Linear(emb_dim * 2 + 1, emb_dim * 3, bias=False)
        self.num_heads = num_heads
        self._episode = None
        self.step_flags = 0   # VRP_STEP_* kernel-selection bits for forward() (tests, A/B)
        # other sizes can be constructed (state_dict compatibility) but the HIP kernels are
        # specialised for the reference's architecture: running them raises in runtime.py
        self.hip_supported = runtime.check_supported_dims(emb_dim, num_heads, None, decoder=True)

    def _apply(self, fn, *a, **k):
        runtime.invalidate(self)
        return super()._apply(fn, *a, **k)

    @property
    def first_step(self):
        return self._episode is None or self._episode.t == 0

    def forward(self, node_embs, mask=None, load=None, C=10, rollout=False):
        """One decoding step (graph_decoder.py:51-115) -> (idx (B,1) int64, log_prob).
        Stateful like the reference: the first call of an episode uses the learned
        placeholders, later calls the first/last chosen nodes; `reset()` ends it."""
        return runtime.decoder_step(self, node_embs, mask, load, greedy=bool(rollout), clip=float(C))

    def reset(self):
        """graph_decoder.py:117-124."""
        self._episode = None
