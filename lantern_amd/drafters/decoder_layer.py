"""The drafter's single decoder layer on the HIP skinny GEMM (second part).

Reference: models/drafters/cnets_lumina_mgpt.py -- ChameleonDecoderLayer (:769-843) = ChameleonRMSNorm (:209-223) ->
ChameleonAttention (:411-541: q/k/v projections, per-head ChameleonLayerNorm :375-396, rotary :228-357, KV concat, eager
softmax attention under the additive drafter mask, o_proj) -> residual -> ChameleonRMSNorm -> ChameleonMLP (:359-373) -> residual.

During drafting the layer sees M = 2 x top_k = 20 rows per call, six calls per verify cycle: its cost is the 404 MB of
projection weights it streams (7B: 4 x 4096^2 + 3 x 4096 x 11008 bf16), not arithmetic.  The seven projections therefore go
through `lantern_linear_rows` (drafter_fc.hip: `linear_rows_kernel`, MFMA 32x32x16 fed straight from the [out, in] weight rows,
M <= 128) with q/k/v and gate/up each fused into ONE launch over a concatenated weight; norms, head norm + rotary are single HIP kernels and
the attention is lantern_tree_attention when the caller hands the tree block over as ancestor words (cnets.Model.forward does).  CPU tensors
and other dtypes (the f32 reference vectors) and M > 128 (the drafter's prefill) take torch.nn.functional.linear; the drafting shape on the
device either runs the fused HIP path or raises (no silent switch) -- the layer is a drop-in nn.Module with the reference's parameter names
(`self_attn.q_proj.weight`, `mlp.gate_proj.weight`, `input_layernorm.weight`, ...), so a reference drafter checkpoint loads into it unchanged.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _hip_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.shape[-1] % 16 == 0


def skinny_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [..., K] @ weight[N, K].T (+ bias): the HIP weight-streaming kernel (128-row slices) for bf16 device tensors, torch for anything else
    (f32 / CPU: the composition the golden vectors are checked on)."""
    rows = x.numel() // x.shape[-1]
    if rows > 0 and _hip_ok(x, weight):
        out = ops.linear_rows(x.reshape(rows, x.shape[-1]), weight, 0, weight.shape[0], bias=bias)
        return out.reshape(*x.shape[:-1], weight.shape[0])
    if x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.shape[-1] % 16 != 0:
        # the product's dtype says "HIP kernel" but its shape does not fit one: say so instead of sliding onto torch's GEMM unnoticed
        raise ops._lib.LanternError(f"skinny_linear: bf16 device product with K = {x.shape[-1]} (K % 16 != 0) has no HIP kernel; pad K or run the layer in f32")
    return F.linear(x, weight, bias)          # f32 / CPU: the composition the golden vectors are checked on


def _mm(A, weight, epilogue: int = 0, **kw):
    """The layer's GEMMs on the packed weights: the stream-K kernel at the drafting shape (<= 32 rows), the row-blocked one for prefills."""
    if A.shape[0] <= 32:
        return ops.linear_rows_streamk(A, weight, epilogue, **kw)
    kw.pop("pair_rows", None)
    return ops.linear_rows_packed(A, weight, epilogue, **kw)



def additive_mask_words(attention_mask, B: int, T: int, S: int):
    """An additive attention mask [1|B, 1, T, >= S] (0 = visible, finfo.min = hidden: what _prepare_decoder_attention_mask builds) restated as the
    HIP kernels' description of visibility: one first-visible-key index per batch row for the S - T cached keys (left padding) + one ancestor
    word per (row, new token) for the T new keys.  Returns (ok, kv_start [B] i64, words [B, T] i64); ok is a 0-dim bool tensor: False when the
    mask cannot be said that way (queries that disagree about the prefix, holes behind the first visible key)."""
    m = attention_mask
    if m.dim() == 4:
        m = m[:, 0]
    vis = (m[:, -T:, :S] > torch.finfo(m.dtype).min / 2) if m.is_floating_point() else m[:, -T:, :S].bool()
    if vis.shape[0] != B:
        vis = vis.expand(B, -1, -1)
    pre = vis[:, :, :S - T]
    p0 = pre[:, 0].to(torch.int64)
    ok = (pre == pre[:, :1]).all()
    if S > T:
        ok = ok & (p0.cummax(dim=1).values == p0).all()
        kv_start = torch.where(p0.any(dim=1), p0.argmax(dim=1), torch.full((B,), S - T, dtype=torch.int64, device=m.device))
    else:
        kv_start = torch.zeros(B, dtype=torch.int64, device=m.device)
    w = torch.ones(T, dtype=torch.int64, device=m.device) << torch.arange(T, dtype=torch.int64, device=m.device)
    words = (vis[:, :, S - T:].to(torch.int64) * w).sum(-1)
    return ok, kv_start, words


def _attend_without_tree(q, k, v, attention_mask, kv_start, past, T, B, H, nq, nk):
    """The branch of the fused layers for callers that bring neither ancestor words nor the causal hint (cnets.Model.forward always brings
    one).  Still the HIP attention, never torch's SDPA: no mask = every new token sees every key; an additive mask is restated as
    ancestor words + a first-visible-key index (additive_mask_words; one host read to check that it can be) for T <= 64 new tokens, and
    refused where it cannot."""
    S = past + T
    if T > 64:
        raise ops._lib.LanternError(
            f"DecoderLayer (fused HIP path): {T} new tokens with an additive mask and no `causal=True` hint: tree attention takes at most 64 query rows -- "
            "pass causal=True (+ kv_start) for a left-padded causal prefill, or set layer.fused = False for the torch composition")
    if attention_mask is None:
        words = torch.full((T,), (1 << T) - 1 if T < 64 else -1, dtype=torch.int64, device=q.device)
    else:
        ok, start, words = additive_mask_words(attention_mask, B, T, S)
        if not bool(ok):
            raise ops._lib.LanternError(
                "DecoderLayer (fused HIP path): this additive attention_mask is not (left padding) x (a block over the new tokens): no HIP kernel "
                "describes it -- set layer.fused = False for the torch composition")
        kv_start = start if kv_start is None else kv_start
    return ops.tree_attention(q.transpose(1, 2), k, v, words, kv_start=kv_start, max_kv_len=S).reshape(B * T, H)

_CAUSAL_WORDS = {}


def causal_block_attention(q, k, v, kv_start, past: int):
    """Causal attention of T new tokens behind `past` cached ones under left padding, on lantern_tree_attention: q [B, nq, T, d] (the head stage's
    layout), k / v [B, nk, >= past + T, d] with the new keys already in place; query i of row b sees keys kv_start[b] .. past + i (and always
    itself: a query inside the padding keeps a finite row, as the reference's additive mask leaves it).  Blocks of <= 64 queries: a block's own
    keys are its "tree" keys with lower-triangular ancestor words, everything in front of the block is the visible prefix.  -> [B, T, nq * d]."""
    B, nq, T, d = q.shape
    dev = q.device
    qn = q.transpose(1, 2)
    out = torch.empty((B, T, nq * d), dtype=q.dtype, device=dev)
    ks = torch.zeros(B, dtype=torch.int64, device=dev) if kv_start is None else kv_start.to(device=dev, dtype=torch.int64)
    one = torch.ones((), dtype=torch.int64, device=dev)
    for b0 in range(0, T, 64):
        n = min(64, T - b0)
        p0 = past + b0
        words = _CAUSAL_WORDS.get((n, dev))
        if words is None:
            i = torch.arange(n, dtype=torch.int64, device=dev)
            tri = torch.where(i >= 63, -one, (one << (i + 1).clamp(max=63)) - 1)
            words = _CAUSAL_WORDS[(n, dev)] = (tri, one << i)
        tri, diag = words
        npad = (ks - p0).clamp(0, 64)
        padw = torch.where(npad >= 64, -one, (one << npad.clamp(max=63)) - 1)
        bits = (tri[None] & ~padw[:, None]) | diag[None]
        ops.tree_attention(qn[:, b0:b0 + n], k, v, bits, kv_start=ks, max_kv_len=p0 + n, out=out[:, b0:b0 + n])
    return out


class RMSNorm(nn.Module):
    """cnets_lumina_mgpt.py:209-223 (statistics in f32, the weight applied in the input dtype)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        xf = x.to(torch.float32)
        xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * xf.to(x.dtype)


class HeadLayerNorm(nn.Module):
    """cnets_lumina_mgpt.py:375-396: layer norm over head_dim with one (gamma, beta) row per model-parallel shard, repeated over the
    shard's heads.  Parameters keep nn.LayerNorm's names and the [model_parallel_size, head_dim] shape."""

    def __init__(self, head_dim: int, model_parallel_size: int, n_heads_per_mp: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(model_parallel_size, head_dim))
        self.bias = nn.Parameter(torch.zeros(model_parallel_size, head_dim))
        self.head_dim, self.n_heads_per_mp, self.eps = head_dim, n_heads_per_mp, eps

    def forward(self, x):                                  # [tokens, heads, head_dim]
        x = F.layer_norm(x, (self.head_dim,), None, None, eps=1e-5)
        return x * self.weight.repeat_interleave(self.n_heads_per_mp, dim=0) + self.bias.repeat_interleave(self.n_heads_per_mp, dim=0)


class Rotary(nn.Module):
    """cnets_lumina_mgpt.py:228-299 (cos / sin tables of the concatenated-halves convention)."""

    def __init__(self, dim: int, max_position_embeddings: int = 2048, base: float = 10000.0):
        super().__init__()
        self.dim, self.base = dim, base
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)), persistent=False)
        self._build(max_position_embeddings, self.inv_freq.device, torch.get_default_dtype())

    def _build(self, seq_len, device, dtype):
        self.max_seq_len_cached = seq_len
        t = torch.arange(seq_len, device=device, dtype=self.inv_freq.dtype)
        emb = torch.outer(t, self.inv_freq.to(device))
        emb = torch.cat((emb, emb), dim=-1)
        self.register_buffer("cos_cached", emb.cos().to(dtype), persistent=False)
        self.register_buffer("sin_cached", emb.sin().to(dtype), persistent=False)

    def tables_bf16(self, device, seq_len: int):
        """The whole cos / sin tables as bf16 on `device`, converted once (the fused head kernel indexes them by position)."""
        if seq_len > self.max_seq_len_cached:
            self._build(seq_len, device, torch.get_default_dtype())
            self._bf16 = None
        c = getattr(self, "_bf16", None)
        if c is None or c[0].device != device:
            self._bf16 = c = (self.cos_cached.to(device=device, dtype=torch.bfloat16).contiguous(), self.sin_cached.to(device=device, dtype=torch.bfloat16).contiguous())
        return c

    def forward(self, x, seq_len: int):
        if seq_len > self.max_seq_len_cached:
            self._build(seq_len, x.device, x.dtype)
        return self.cos_cached[:seq_len].to(dtype=x.dtype, device=x.device), self.sin_cached[:seq_len].to(dtype=x.dtype, device=x.device)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class Attention(nn.Module):
    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        mp = getattr(config, "model_parallel_size", 1)
        bias = getattr(config, "attention_bias", False)
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=bias)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.o_proj = nn.Linear(self.hidden_size, self.hidden_size, bias=bias)
        self.q_norm = HeadLayerNorm(self.head_dim, mp, self.num_heads // mp)
        self.k_norm = HeadLayerNorm(self.head_dim, mp, self.num_key_value_heads // mp)
        self.rotary_emb = Rotary(self.head_dim, getattr(config, "max_position_embeddings", 2048), getattr(config, "rope_theta", 10000.0))
        self._qkv = None          # (versions, fused weight, fused bias): q/k/v stream as one launch

    def _fused_qkv(self):
        ws = (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight)
        key = tuple((w.data_ptr(), w._version) for w in ws)
        if self._qkv is None or self._qkv[0] != key:
            b = None if self.q_proj.bias is None else torch.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]).contiguous()
            self._qkv = (key, torch.cat(ws, dim=0).contiguous(), b)
        return self._qkv[1], self._qkv[2]

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False, use_cache=False, **kw):
        bsz, q_len, _ = hidden_states.shape
        nq, nk, hd = self.num_heads, self.num_key_value_heads, self.head_dim
        if bsz * q_len <= 128 and _hip_ok(hidden_states, self.q_proj.weight):
            w, b = self._fused_qkv()
            qkv = skinny_linear(hidden_states, w, b)
            q, k, v = qkv.split((nq * hd, nk * hd, nk * hd), dim=-1)
        else:
            q, k, v = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
        q = self.q_norm(q.reshape(-1, nq, hd)).reshape(bsz, q_len, nq, hd).transpose(1, 2)
        k = self.k_norm(k.reshape(-1, nk, hd)).reshape(bsz, q_len, nk, hd).transpose(1, 2)
        v = v.reshape(bsz, q_len, nk, hd).transpose(1, 2)
        kv_len = q_len + (past_key_value[0].shape[-2] if past_key_value is not None else 0)
        cos, sin = self.rotary_emb(v, kv_len)
        cos, sin = cos[position_ids].unsqueeze(1), sin[position_ids].unsqueeze(1)
        q, k = q * cos + _rotate_half(q) * sin, k * cos + _rotate_half(k) * sin
        if past_key_value is not None:
            k, v = torch.cat([past_key_value[0], k], dim=2), torch.cat([past_key_value[1], v], dim=2)
        present = (k, v) if use_cache else None
        if self.num_key_value_groups > 1:
            k = k[:, :, None].expand(bsz, nk, self.num_key_value_groups, kv_len, hd).reshape(bsz, nq, kv_len, hd)
            v = v[:, :, None].expand(bsz, nk, self.num_key_value_groups, kv_len, hd).reshape(bsz, nq, kv_len, hd)
        w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
        if attention_mask is not None:
            w = w + attention_mask[:, :, :, :kv_len]
        w = F.softmax(w, dim=-1).to(q.dtype)
        out = torch.matmul(w, v).transpose(1, 2).reshape(bsz, q_len, self.hidden_size)
        out = skinny_linear(out, self.o_proj.weight, self.o_proj.bias)
        return out, (w if output_attentions else None), present


class MLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        bias = getattr(config, "mlp_bias", False)
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=bias)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=bias)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=bias)
        act = getattr(config, "hidden_act", "silu")
        if act not in ("silu", "swish", "gelu"):
            raise ValueError(f"hidden_act={act}: the drafter layers of the reference use silu")
        self.act_fn = F.gelu if act == "gelu" else F.silu
        self._gu = None

    def _fused_gate_up(self):
        ws = (self.gate_proj.weight, self.up_proj.weight)
        key = tuple((w.data_ptr(), w._version) for w in ws)
        if self._gu is None or self._gu[0] != key:
            b = None if self.gate_proj.bias is None else torch.cat([self.gate_proj.bias, self.up_proj.bias]).contiguous()
            self._gu = (key, torch.cat(ws, dim=0).contiguous(), b)
        return self._gu[1], self._gu[2]

    def forward(self, x):
        rows = x.numel() // x.shape[-1]
        if rows <= 128 and _hip_ok(x, self.gate_proj.weight):
            w, b = self._fused_gate_up()
            g, u = skinny_linear(x, w, b).split(self.gate_proj.out_features, dim=-1)
            return skinny_linear(self.act_fn(g) * u, self.down_proj.weight, self.down_proj.bias)
        return self.down_proj(self.act_fn(self.gate_proj(x)) * self.up_proj(x))


class DecoderLayer(nn.Module):
    """Drop-in for the reference's ChameleonDecoderLayer in the drafter (same call signature and return tuple)."""

    def __init__(self, config, layer_idx: int = 0):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = Attention(config, layer_idx)
        self.mlp = MLP(config)
        eps = getattr(config, "rms_norm_eps", 1e-6)
        self.input_layernorm = RMSNorm(config.hidden_size, eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps)

    def _packed(self, name, weight, pair_rows=0):
        """The weight in the layout the stream-K kernel streams (ops.pack_linear_weight), made once per weight version; weights whose K is
        not a multiple of 64 stay row-major (the kernel takes both)."""
        if weight.shape[1] % 64:
            return weight
        cache = self.__dict__.setdefault("_packed_weights", {})
        key = (weight.data_ptr(), weight._version)
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = cache[name] = (key, ops.pack_linear_weight(weight, pair_rows))
        return hit[1]

    def _packed_fused(self, name, mods, pair_rows=0):
        """(packed weight, fused bias) of projections that share one launch -- q / k / v, gate / up: their rows concatenated, packed, and the row-major
        concatenation dropped again (the layer keeps the nn.Linear weights and the packed bricks, not a third copy: 280 MB at 7B size).  Keyed by the
        source weights' (data_ptr, version): load_state_dict and in-place optimiser steps re-pack; `repack()` after a write through `.data`."""
        ws = tuple(m.weight for m in mods)
        if ws[0].shape[1] % 64:
            cat = torch.cat(ws, dim=0).contiguous()
            return cat, (None if mods[0].bias is None else torch.cat([m.bias for m in mods]).contiguous())
        cache = self.__dict__.setdefault("_packed_weights", {})
        key = tuple((w.data_ptr(), w._version) for w in ws) + tuple((m.bias.data_ptr(), m.bias._version) for m in mods if m.bias is not None)
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            cat = torch.cat(ws, dim=0).contiguous()
            b = None if mods[0].bias is None else torch.cat([m.bias for m in mods]).contiguous()
            hit = cache[name] = (key, ops.pack_linear_weight(cat, pair_rows), b)
            del cat
        return hit[1], hit[2]

    def repack(self):
        """Forget every packed / fused copy of the weights (they are rebuilt on the next call).  load_state_dict does this by itself; call it after
        writing weights in a way autograd's version counter does not see (`weight.data.copy_`, `set_`)."""
        self.__dict__.pop("_packed_weights", None)
        for m in (self.self_attn, self.mlp):
            for attr in ("_qkv", "_gu"):
                if hasattr(m, attr):
                    setattr(m, attr, None)

    def _load_from_state_dict(self, *a, **k):
        self.repack()
        return super()._load_from_state_dict(*a, **k)

    def _fast(self, x, attention_mask, position_ids, past_key_value, use_cache, tree_bits=None, tree_keys=0, kv_start=None, causal=False):
        """Decode shape on the device (<= 32 bf16 rows): rmsnorm, fused q/k/v GEMM, head norm + rotary (+ cache append), one attention call,
        o_proj + residual, rmsnorm, gate/up GEMM with silu * up in its epilogue, down_proj + residual -- the four GEMMs in stream-K form
        (lantern_linear_rows_streamk: one launch each, the weight split into equal contiguous shares over 2 workgroups per CU).
        With `tree_bits` (ops.drafter_tree_bits: the tree block of the drafter's mask as ancestor words) the attention is
        lantern_tree_attention over the cache in place -- no additive mask, no repeat_kv, no [T, S] scores."""
        at, mlp = self.self_attn, self.mlp
        B, T, H = x.shape
        nq, nk, d = at.num_heads, at.num_key_value_heads, at.head_dim
        x2 = x.reshape(B * T, H)
        xn = ops.rmsnorm_rows(x2, self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
        wq, b = self._packed_fused("qkv", (at.q_proj, at.k_proj, at.v_proj))
        qkv = _mm(xn, wq, bias=b)          # stream-K: every workgroup streams an equal, contiguous share of the weight (prefills: row blocks)
        kv_len = T + (past_key_value[0].shape[-2] if past_key_value is not None else 0)
        cos, sin = at.rotary_emb.tables_bf16(x.device, kv_len)
        past = 0 if past_key_value is None else past_key_value[0].shape[-2]
        if self.inplace_cache and use_cache:
            # the cache lives in a slab of this layer and grows in place: `present` is a view of it; a `past` that is such a view (what the
            # drafter hands back: the previous call's present, or a shorter prefix of it after a rollback) costs no copy at all
            ks, vs = self._cache_slab(B, nk, d, kv_len, x.device, past_key_value)
            q, _, _ = ops.qk_norm_rope(qkv, B, T, nq, nk, d, at.q_norm.weight, at.q_norm.bias, at.k_norm.weight, at.k_norm.bias, cos, sin, position_ids,
                                       k_slab=ks, v_slab=vs, row0=past)
            k, v = ks[:, :, :kv_len], vs[:, :, :kv_len]
        else:
            q, k, v = ops.qk_norm_rope(qkv, B, T, nq, nk, d, at.q_norm.weight, at.q_norm.bias, at.k_norm.weight, at.k_norm.bias, cos, sin, position_ids)
            if past_key_value is not None:
                k, v = torch.cat([past_key_value[0], k], dim=2), torch.cat([past_key_value[1], v], dim=2)
        present = (k, v) if use_cache else None
        if tree_bits is not None:
            # the T new tokens are the last T of the `tree_keys` tree keys; the kernel takes as many query rows as tree keys: the
            # rows in front are placeholders (zero queries that see themselves)
            t1 = int(tree_keys)
            qn = q.transpose(1, 2)                                  # [B, T, nq, d]
            if t1 > T:
                qp = torch.zeros((B, t1, nq, d), dtype=q.dtype, device=q.device)
                qp[:, t1 - T:] = qn
            else:
                qp = qn
            o = ops.tree_attention(qp, k, v, tree_bits, kv_start=kv_start, max_kv_len=kv_len)[:, t1 - T:].reshape(B * T, H)
        elif causal:
            # a prefill: causal among the new tokens behind the left padding, block by block on the same kernel (no [T, S] mask, no eager softmax)
            o = causal_block_attention(q, k, v, kv_start, past).reshape(B * T, H)
        else:
            o = _attend_without_tree(q, k, v, attention_mask, kv_start, past, T, B, H, nq, nk)
        h1 = _mm(o, self._packed("o", at.o_proj.weight), ops.EPI_RESIDUAL, bias=at.o_proj.bias, residual=x2)
        hn = ops.rmsnorm_rows(h1, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon)
        inter = mlp.gate_proj.out_features
        wg, bg = self._packed_fused("gate_up", (mlp.gate_proj, mlp.up_proj), inter)
        act = _mm(hn, wg, ops.EPI_SILU_MUL, bias=bg, pair_rows=inter)
        out = _mm(act, self._packed("down", mlp.down_proj.weight), ops.EPI_RESIDUAL, bias=mlp.down_proj.bias, residual=h1)
        return out.reshape(B, T, H), present

    def _fast_ok(self, x, position_ids, output_attentions, causal=False):
        at = self.self_attn
        rows = x.shape[0] * x.shape[1] if x.dim() == 3 else 0
        # (more than 32 rows = a prefill: the row-blocked GEMM on the packed weights and the block-causal attention, when the caller says the mask
        # is causal + left padding -- cnets.Model.forward does)
        shape_ok = rows <= 32 or (causal and x.shape[-1] % 64 == 0 and self.mlp.gate_proj.out_features % 64 == 0)
        return (self.fused and not output_attentions and x.dim() == 3 and shape_ok and _hip_ok(x, at.q_proj.weight)
                and at.head_dim in (64, 128) and position_ids is not None and self.mlp.act_fn is F.silu
                and self.input_layernorm.weight.dtype == torch.bfloat16 and at.q_norm.weight.dtype == torch.bfloat16)

    inplace_cache = False          # True: `present` is a growing view of a layer-owned slab (no torch.cat of the whole cache per call).  Only for callers
                                   # that, like the drafter, never use an OLDER, LONGER cache again after appending to a shorter prefix of it.

    def _cache_slab(self, B, nk, d, need, device, past):
        s = getattr(self, "_slab", None)
        mine = (s is not None and past is not None and past[0].data_ptr() == s[0].data_ptr() and past[1].data_ptr() == s[1].data_ptr()
                and past[0].shape[0] == B and past[0].stride() == s[0][:, :, :past[0].shape[2]].stride())
        if s is None or s[0].shape[0] != B or s[0].shape[2] < need or s[0].device != device:
            cap = max(256, 1 << (need - 1).bit_length())
            ns = (torch.empty((B, nk, cap, d), dtype=torch.bfloat16, device=device), torch.empty((B, nk, cap, d), dtype=torch.bfloat16, device=device))
            if past is not None:
                ns[0][:, :, :past[0].shape[2]].copy_(past[0])
                ns[1][:, :, :past[1].shape[2]].copy_(past[1])
            self._slab = s = ns
        elif past is not None and not mine:          # a cache that is not (a prefix of) the slab: bring it in once
            s[0][:, :, :past[0].shape[2]].copy_(past[0])
            s[1][:, :, :past[1].shape[2]].copy_(past[1])
        return s

    fused = True          # False: every projection its own launch through skinny_linear / torch (the composition the fast path is tested against)

    supports_tree_bits = True      # Model.forward hands the tree block over as ancestor words (tree_bits / tree_keys / kv_start) when it has one

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False, use_cache=False,
                tree_bits=None, tree_keys=0, kv_start=None, causal=False, **kw) -> Tuple[torch.Tensor, ...]:
        if (self.fused and not output_attentions and hidden_states.dim() == 3 and hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16
                and hidden_states.shape[0] * hidden_states.shape[1] <= 32 and _hip_ok(hidden_states, self.self_attn.q_proj.weight)
                and not self._fast_ok(hidden_states, position_ids, output_attentions)):
            # the drafting shape must not slide onto torch's ops unnoticed: say which precondition of the fused HIP path is missing
            at = self.self_attn
            raise ops._lib.LanternError(
                f"DecoderLayer: the drafting shape ({tuple(hidden_states.shape)}, bf16, device) takes the fused HIP path, which needs head_dim 64 or 128 "
                f"(got {at.head_dim}), silu, bf16 weights / norms, hidden % 16 == 0 and position_ids; set `layer.fused = False` for the per-projection "
                "path (skinny GEMM under torch's element-wise ops)")
        if self._fast_ok(hidden_states, position_ids, output_attentions, causal):
            if tree_bits is not None and not (0 < hidden_states.shape[1] <= tree_keys <= 64):
                tree_bits = None
            y, present = self._fast(hidden_states, attention_mask, position_ids, past_key_value, use_cache, tree_bits, tree_keys, kv_start,
                                    causal and tree_bits is None)
            return (y, present) if use_cache else (y,)
        a, w, present = self.self_attn(self.input_layernorm(hidden_states), attention_mask=attention_mask, position_ids=position_ids,
                                       past_key_value=past_key_value, output_attentions=output_attentions, use_cache=use_cache)
        hidden_states = hidden_states + a
        hidden_states = hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))
        out = (hidden_states,)
        if output_attentions:
            out += (w,)
        if use_cache:
            out += (present,)
        return out


# ----------------------------------------------------------------------------- the LlamaGen drafter's layer
def apply_rotary_pairs(x, freqs_cis):
    """cnets_llamagen.py:67-77: x [B, T, heads, d], freqs_cis [T, d/2, 2] (cos, sin) -- rotation of adjacent pairs in f32, cast back."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    fc = freqs_cis.to(xs.device).view(1, xs.size(1), 1, xs.size(3), 2)
    out = torch.stack([xs[..., 0] * fc[..., 0] - xs[..., 1] * fc[..., 1], xs[..., 1] * fc[..., 0] + xs[..., 0] * fc[..., 1]], dim=-1)
    return out.flatten(3).type_as(x)


def precompute_freqs_cis_2d(grid_size: int, n_elem: int, base: float = 10000.0, cls_token_num: int = 120):
    """LlamaGen's 2-D rotary table (cnets_llamagen.py:47-64): [cls_token_num + grid_size^2 + 10, n_elem / 2, 2] f32 -- zeros for the condition
    tokens, (cos, sin) of the row frequencies then the column frequencies for the grid tokens, ten zero rows behind."""
    half = n_elem // 2
    freqs = 1.0 / (base ** (torch.arange(0, half, 2)[: (half // 2)].float() / half))
    f = torch.outer(torch.arange(grid_size), freqs)
    grid = torch.cat([f[:, None, :].expand(-1, grid_size, -1), f[None, :, :].expand(grid_size, -1, -1)], dim=-1)
    cache = torch.stack([torch.cos(grid), torch.sin(grid)], dim=-1).flatten(0, 1)
    if cls_token_num > 0:
        cache = torch.cat([torch.zeros(cls_token_num, n_elem // 2, 2), cache])
    return torch.cat([cache, torch.zeros(10, n_elem // 2, 2)])


class LlamaAttention(nn.Module):
    """cnets_llamagen.py:224-375 (LlamaAttention as the LlamaGen drafter uses it: rotary through `freqs_cis`, eager softmax in f32)."""

    def __init__(self, config):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = getattr(config, "num_key_value_heads", self.num_heads)
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size} and `num_heads`: {self.num_heads}).")
        qkv_bias = bool(getattr(config, "qkv_bias", False))
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=qkv_bias)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=qkv_bias)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=qkv_bias)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        self._qkv = None

    _fused_qkv = Attention._fused_qkv

    def forward(self, hidden_states, attention_mask=None, position_ids=None, freqs_cis=None, past_key_value=None, output_attentions=False, use_cache=False):
        bsz, q_len, _ = hidden_states.shape
        nq, nk, hd = self.num_heads, self.num_key_value_heads, self.head_dim
        q = self.q_proj(hidden_states).view(bsz, q_len, nq, hd)
        k = self.k_proj(hidden_states).view(bsz, q_len, nk, hd)
        v = self.v_proj(hidden_states).view(bsz, q_len, nk, hd).transpose(1, 2)
        q = apply_rotary_pairs(q, freqs_cis).transpose(1, 2)
        k = apply_rotary_pairs(k, freqs_cis).transpose(1, 2)
        if past_key_value is not None:
            k, v = torch.cat([past_key_value[0], k], dim=2), torch.cat([past_key_value[1], v], dim=2)
        present = (k, v) if use_cache else None
        kv_len = k.shape[2]
        if self.num_key_value_groups > 1:
            k = k[:, :, None].expand(bsz, nk, self.num_key_value_groups, kv_len, hd).reshape(bsz, nq, kv_len, hd)
            v = v[:, :, None].expand(bsz, nk, self.num_key_value_groups, kv_len, hd).reshape(bsz, nq, kv_len, hd)
        w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
        if attention_mask is not None:
            w = w + attention_mask
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        out = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(bsz, q_len, self.hidden_size)
        return self.o_proj(out), (w if output_attentions else None), present


class LlamaDecoderLayer(DecoderLayer):
    """Drop-in for the reference's LlamaDecoderLayer in the LlamaGen drafter (cnets_llamagen.py:428-494; same call signature incl. `freqs_cis`,
    same return tuple, same parameter names -- `self_attn.q_proj.weight`, `mlp.gate_proj.weight`, `post_attention_layernorm.weight`, and
    `input_layernorm.weight` for index != 0: EAGLE drops layer 0's input norm).  At the drafting shape on the device (<= 32 bf16 rows, head_dim 64 or
    128, hidden % 64 == 0) it runs the same HIP kernels as the Chameleon layer -- stream-K GEMMs on packed weights with the residual / silu * up
    epilogues, lantern_rmsnorm_rows, lantern_tree_attention -- with lantern_qk_rope_pairs as the head stage (no per-head norm, pair rotary from the
    gathered `freqs_cis` rows)."""

    def __init__(self, config, index: int = 0):
        nn.Module.__init__(self)
        self.hidden_size = config.hidden_size
        self.index = index
        self.self_attn = LlamaAttention(config)
        self.mlp = MLP(config)
        eps = getattr(config, "rms_norm_eps", 1e-6)
        if index != 0:
            self.input_layernorm = RMSNorm(config.hidden_size, eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps)

    def _fast_ok(self, x, position_ids, output_attentions, freqs_cis=None, causal=False):
        at = self.self_attn
        rows = x.shape[0] * x.shape[1] if x.dim() == 3 else 0
        shape_ok = rows <= 32 or (causal and self.mlp.gate_proj.out_features % 64 == 0)          # > 32 rows: a prefill (see DecoderLayer._fast_ok)
        return (self.fused and not output_attentions and x.dim() == 3 and shape_ok and _hip_ok(x, at.q_proj.weight)
                and at.head_dim in (64, 128) and x.shape[-1] % 64 == 0 and freqs_cis is not None and self.mlp.act_fn is F.silu
                and self.post_attention_layernorm.weight.dtype == torch.bfloat16)

    def _fast(self, x, attention_mask, freqs_cis, past_key_value, use_cache, tree_bits=None, tree_keys=0, kv_start=None, causal=False):
        at, mlp = self.self_attn, self.mlp
        B, T, H = x.shape
        nq, nk, d = at.num_heads, at.num_key_value_heads, at.head_dim
        x2 = x.reshape(B * T, H)
        xn = x2 if self.index == 0 else ops.rmsnorm_rows(x2, self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
        wq, b = self._packed_fused("qkv", (at.q_proj, at.k_proj, at.v_proj))
        qkv = _mm(xn, wq, bias=b)
        past = 0 if past_key_value is None else past_key_value[0].shape[-2]
        kv_len = T + past
        fr = freqs_cis.reshape(-1, d // 2, 2)                      # the reference hands over the rows of its table at this call's positions: [T, d/2, 2]
        pos = self.__dict__.get("_arange")
        if pos is None or pos.numel() < fr.shape[0] or pos.device != x.device:
            pos = self.__dict__["_arange"] = torch.arange(max(64, fr.shape[0]), dtype=torch.int64, device=x.device)
        if fr.shape[0] not in (T, B * T):
            raise ops._lib.LanternError(f"LlamaDecoderLayer: freqs_cis holds {fr.shape[0]} rows for {T} tokens")
        if self.inplace_cache and use_cache:
            ks, vs = self._cache_slab(B, nk, d, kv_len, x.device, past_key_value)
            q, _, _ = ops.qk_rope_pairs(qkv, B, T, nq, nk, d, fr, pos[:fr.shape[0]], k_slab=ks, v_slab=vs, row0=past)
            k, v = ks[:, :, :kv_len], vs[:, :, :kv_len]
        else:
            q, k, v = ops.qk_rope_pairs(qkv, B, T, nq, nk, d, fr, pos[:fr.shape[0]])
            if past_key_value is not None:
                k, v = torch.cat([past_key_value[0], k], dim=2), torch.cat([past_key_value[1], v], dim=2)
        present = (k, v) if use_cache else None
        if tree_bits is not None:
            t1 = int(tree_keys)
            qn = q.transpose(1, 2)
            if t1 > T:
                qp = torch.zeros((B, t1, nq, d), dtype=q.dtype, device=q.device)
                qp[:, t1 - T:] = qn
            else:
                qp = qn
            o = ops.tree_attention(qp, k, v, tree_bits, kv_start=kv_start, max_kv_len=kv_len)[:, t1 - T:].reshape(B * T, H)
        elif causal:
            o = causal_block_attention(q, k, v, kv_start, past).reshape(B * T, H)
        else:
            o = _attend_without_tree(q, k, v, attention_mask, kv_start, past, T, B, H, nq, nk)
        h1 = _mm(o, self._packed("o", at.o_proj.weight), ops.EPI_RESIDUAL, residual=x2)
        hn = ops.rmsnorm_rows(h1, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon)
        inter = mlp.gate_proj.out_features
        wg, bg = self._packed_fused("gate_up", (mlp.gate_proj, mlp.up_proj), inter)
        act = _mm(hn, wg, ops.EPI_SILU_MUL, bias=bg, pair_rows=inter)
        out = _mm(act, self._packed("down", mlp.down_proj.weight), ops.EPI_RESIDUAL, bias=mlp.down_proj.bias, residual=h1)
        return out.reshape(B, T, H), present

    def forward(self, hidden_states, attention_mask=None, position_ids=None, freqs_cis=None, past_key_value=None, output_attentions=False,
                use_cache=False, tree_bits=None, tree_keys=0, kv_start=None, causal=False, **kw) -> Tuple[torch.Tensor, ...]:
        if self._fast_ok(hidden_states, position_ids, output_attentions, freqs_cis, causal):
            if tree_bits is not None and not (0 < hidden_states.shape[1] <= tree_keys <= 64):
                tree_bits = None
            y, present = self._fast(hidden_states, attention_mask, freqs_cis, past_key_value, use_cache, tree_bits, tree_keys, kv_start,
                                    causal and tree_bits is None)
            return (y, present) if use_cache else (y,)
        if (self.fused and not output_attentions and hidden_states.dim() == 3 and hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16
                and hidden_states.shape[0] * hidden_states.shape[1] <= 32 and _hip_ok(hidden_states, self.self_attn.q_proj.weight)):
            at = self.self_attn
            raise ops._lib.LanternError(
                f"LlamaDecoderLayer: the drafting shape ({tuple(hidden_states.shape)}, bf16, device) takes the fused HIP path, which needs head_dim 64 or "
                f"128 (got {at.head_dim}), hidden % 64 == 0, silu, bf16 weights / norms and freqs_cis; set `layer.fused = False` for torch's ops")
        residual = hidden_states
        x = hidden_states if self.index == 0 else self.input_layernorm(hidden_states)
        a, w, present = self.self_attn(x, attention_mask=attention_mask, position_ids=position_ids, freqs_cis=freqs_cis, past_key_value=past_key_value,
                                       output_attentions=output_attentions, use_cache=use_cache)
        hidden_states = residual + a
        hidden_states = hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))
        out = (hidden_states,)
        if output_attentions:
            out += (w,)
        if use_cache:
            out += (present,)
        return out
