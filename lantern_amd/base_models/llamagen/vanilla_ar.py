"""BASELINE.json config C1: LlamaGen-B 256x256 class-conditional, vanilla autoregressive decode on the CPU -- plumbing only
(no GPU, no drafter, no HIP): tokens out of generate(), statistics entry out of the driver.

The reference's vanilla `LlamaForCausalLM.generate` (models/kv_variants/modeling_llamagen_kv.py:1377-1444) is text-conditional
(T5); its class-conditional side exists as the `LabelEmbedder` (:120-150) that the drafter's `c2i` switch points at
(cnets_llamagen.py:562-565).  SURVEY 8d therefore defines C1 as the build's own minimal counterpart: a randomly initialised
Llama-style decoder of LlamaGen-B's size (hidden 768, 12 layers, 12 heads, vocabulary 16384, 16x16 = 256 tokens) whose prefix
is ONE class-label embedding, decoded with the reference's loop shape: cond / uncond as a batch of 2 per image, the pre-allocated
KV cache of models/drafters/kv_cache.py, `cfg_logit_process`, temperature / top-k / top-p sampling, and the reference's return
triple (tokens, mean accept length = 1.0, seconds).  Not a performance path."""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...drafters.kv_cache import initialize_past_key_values


@dataclass
class LlamaGenBConfig:
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    num_key_value_heads: int = 12
    vocab_size: int = 16384
    num_classes: int = 1000
    max_position_embeddings: int = 16 * 16 + 1
    intermediate_size: int = 2048
    seed: int = 0


class _Block(nn.Module):
    def __init__(self, c: LlamaGenBConfig):
        super().__init__()
        H = c.hidden_size
        self.n_heads, self.dh = c.num_attention_heads, H // c.num_attention_heads
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(H, H, bias=False) for _ in range(4))
        self.gate, self.up, self.down = nn.Linear(H, c.intermediate_size, bias=False), nn.Linear(H, c.intermediate_size, bias=False), \
            nn.Linear(c.intermediate_size, H, bias=False)
        self.n1, self.n2 = nn.RMSNorm(H), nn.RMSNorm(H)
        object.__setattr__(self, "self_attn", self)      # (not a child module) kv_cache.initialize_past_key_values reads layers[i].self_attn.q_proj.weight.device

    def forward(self, x, kv):
        B, T, H = x.shape
        h = self.n1(x)
        q = self.q_proj(h).view(B, T, self.n_heads, self.dh).transpose(1, 2)
        k = kv[0].cat(self.k_proj(h).view(B, T, self.n_heads, self.dh).transpose(1, 2), dim=2)
        v = kv[1].cat(self.v_proj(h).view(B, T, self.n_heads, self.dh).transpose(1, 2), dim=2)
        S = k.shape[2]
        mask = torch.ones(T, S, dtype=torch.bool).tril(diagonal=S - T)
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        x = x + self.o_proj(a.transpose(1, 2).reshape(B, T, H))
        h = self.n2(x)
        return x + self.down(F.silu(self.gate(h)) * self.up(h))


class ClassCondLlamaGen(nn.Module):
    def __init__(self, config: Optional[LlamaGenBConfig] = None):
        super().__init__()
        c = self.config = config or LlamaGenBConfig()
        g = torch.Generator().manual_seed(c.seed)
        self.tok_embeddings = nn.Embedding(c.vocab_size, c.hidden_size)
        self.cls_embedding = nn.Embedding(c.num_classes + 1, c.hidden_size)      # last row = the unconditional ("dropped") label
        self.pos_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.model = nn.Module()
        self.model.layers = nn.ModuleList(_Block(c) for _ in range(c.num_hidden_layers))
        self.norm = nn.RMSNorm(c.hidden_size)
        self.lm_head = nn.Linear(c.hidden_size, c.vocab_size, bias=False)
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(p.shape[-1])))
        self.dtype = torch.float32

    def forward(self, x, past_key_values, pos0: int):
        x = x + self.pos_embeddings(torch.arange(pos0, pos0 + x.shape[1]))[None]
        for layer, kv in zip(self.model.layers, past_key_values):
            x = layer(x, kv)
        return self.lm_head(self.norm(x))[:, -1]

    @torch.no_grad()
    def generate(self, class_labels, max_length: int = 256, temperature: float = 1.0, top_k: int = 2000, top_p: float = 1.0,
                 cfg: Optional[float] = 4.0, generator: Optional[torch.Generator] = None):
        """-> (tokens [B, max_length] i64, mean accept length (1.0: vanilla decoding accepts one token per step), seconds)."""
        labels = torch.as_tensor(class_labels, dtype=torch.long).reshape(-1)
        B = labels.shape[0]
        st = time.time()
        rows = 2 * B if cfg is not None else B
        if getattr(self, "_kv_rows", None) != rows:
            self.past_key_values, self.past_key_values_data, self.current_length_data = initialize_past_key_values(self, rows)
            self._kv_rows = rows
        self.current_length_data.zero_()
        cond = self.cls_embedding(labels)[:, None]
        if cfg is not None:
            cond = torch.cat([cond, self.cls_embedding(torch.full_like(labels, self.config.num_classes))[:, None]])
        seq = torch.empty((B, max_length), dtype=torch.long)
        logits = self(cond, self.past_key_values, 0)
        for i in range(max_length):
            tok = _sample(_cfg_logit_process(logits, cfg), temperature, top_k, top_p, generator)
            seq[:, i] = tok
            if i + 1 < max_length:
                x = self.tok_embeddings(torch.cat([tok, tok]) if cfg is not None else tok)[:, None]
                logits = self(x, self.past_key_values, i + 1)
        return seq, 1.0, time.time() - st


def _cfg_logit_process(combined_logits, cfg):
    if cfg is None:
        return combined_logits
    cond, uncond = torch.split(combined_logits, combined_logits.shape[0] // 2, dim=0)
    return uncond + (cond - uncond) * cfg


def _sample(logits, temperature, top_k, top_p, generator):
    if temperature <= 1e-5:
        return logits.argmax(-1)
    logits = logits / temperature
    if top_k and top_k < logits.shape[-1]:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if 0.0 < top_p < 1.0:
        srt, idx = torch.sort(logits, descending=False)
        remove = srt.softmax(-1).cumsum(-1) <= (1 - top_p)
        remove[..., -1:] = False
        logits = logits.masked_fill(remove.scatter(-1, idx, remove), float("-inf"))
    return torch.multinomial(logits.softmax(-1), 1, generator=generator).squeeze(1)
