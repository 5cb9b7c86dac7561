"""KV-cache bookkeeping with the reference's API (models/drafters/kv_cache.py:4-155).

Layout contract kept: one slab per device `[2*L, B, Hkv, max_position_embeddings, head_dim]`,
per-layer K/V views, `current_length` as a CPU int64 tensor `[2L]` (the reference keeps it on the
host "for quick access", kv_cache.py:124-128).  `KVCache.copy` -- the accepted-path move -- runs
the HIP gather (`lantern_kv_gather`) instead of index_select + copy_.
"""
from __future__ import annotations

import torch


class KVCache:
    def __init__(self, data: torch.Tensor, current_length: torch.Tensor):
        self.data = data
        self.current_length = current_length

    @property
    def shape(self):
        return (self.data.shape[0], self.data.shape[1], self.current_length.item(), self.data.shape[3])

    def copy(self, indices: torch.Tensor, prev_length: int, dim: int = 2):
        from .. import ops
        assert dim == 2, "KV slabs gather along the sequence axis"
        n = indices.numel()
        dev = self.data.device
        retrieve = (indices.to(torch.int64) - prev_length).reshape(1, n).to(dev)
        best = torch.zeros(1, dtype=torch.int32, device=dev)
        alen = torch.full((1,), n - 1, dtype=torch.int32, device=dev)
        ops.kv_gather([self.data], torch.zeros(1, dtype=torch.int32, device=dev),
                      torch.tensor([prev_length], dtype=torch.int64, device=dev), retrieve, best, alen)
        self.current_length.fill_(prev_length + n)

    def cat(self, tensor: torch.Tensor, dim: int = 2):
        dst = self.data.narrow(dim, int(self.current_length), tensor.shape[dim])
        dst.copy_(tensor)
        self.current_length.add_(tensor.shape[dim])
        return torch.narrow(self.data, 2, 0, int(self.current_length))


def initialize_past_key_values(model, batch_size: int = 1):
    """Same return triple as the reference: (past_key_values, [slabs], current_length_data).
    Single-device form (the multi-device split of kv_cache.py:88-122 belongs to the out-of-scope
    model-parallel loader)."""
    config = model.config
    try:
        device = model.model.layers[0].self_attn.q_proj.weight.device
    except AttributeError:
        device = next(model.parameters()).device
    L = config.num_hidden_layers
    data = torch.zeros(L * 2, batch_size, config.num_key_value_heads, config.max_position_embeddings,
                       config.hidden_size // config.num_attention_heads, device=device, dtype=model.dtype)
    current_length_data = torch.zeros(L * 2, dtype=torch.long, device="cpu")
    past_key_values = [[KVCache(data[2 * i + j], current_length_data[2 * i + j]) for j in range(2)] for i in range(L)]
    return past_key_values, [data], current_length_data
