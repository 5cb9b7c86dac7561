"""Host mirror of the reference's drafter `Model` (models/drafters/cnets_lumina_mgpt.py:957-1392,
cnets_llamagen.py:509-1023, cnets_anole.py) for the verify/accept path: same method names, argument meaning and
return values -- `init_tree`, `init_tree_v1`, `reset`, `reset_kv`, `forward`, `_prepare_decoder_attention_mask`,
`repeat_hidden`, `sample`, `topK_genrate` / `topK_genrate_v1` (LlamaGen / Anole calling convention, the reference's
spelling) and `topK_generate(tree_type=...)` (Lumina-mGPT calling convention).

What runs where:
  * input stage `fc(cat(embed[ids] * upscale, hidden))`            -> lantern_drafter_fc (MFMA, bf16)
  * additive attention mask (causal + padding + tree)              -> lantern_drafter_attention_mask
  * CFG combine + model mask + HF processors on the head's logits  -> lantern_cfg_mask_topk_window
  * log-softmax / top-k / cumulative scores / best-10-of-100       -> lantern_expand_dynamic
  * top-(N-1) + tree mask / positions / retrieve rows              -> lantern_tree_dynamic_finalize
  * static drafter buffers (masks, tree_indices, repeat_nums)      -> lantern_tree_drafter_build
  * conditional probabilities of the k samples w/o replacement     -> lantern_sample_static
The decoder layer(s) between the input stage and the head are HIP too at the drafting shape: `layers=None` builds
`lantern_amd.drafters.decoder_layer.DecoderLayer` (Lumina-mGPT, and Anole with one head-norm row per head) or `LlamaDecoderLayer` (LlamaGen:
no head norm, pair rotary from the 2-D freqs_cis table, which the Model then owns like the reference's) -- drop-ins with the reference's parameter
names, so a reference drafter checkpoint loads unchanged (`Model.from_reference`).  A config outside the kernels (head_dim other than 64 / 128,
hidden not a multiple of 64: the reduced-size test configs) raises unless `allow_torch_layers=True` asks for the plain-torch stand-in
(`TorchDecoderLayer`); any module with the reference layer's call signature can still be injected (`layers=[...]`).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
from torch import nn

from .. import ops

TOPK = 10


# ----------------------------------------------------------------------------- default decoder layer (plain torch)
class _RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        dt = x.dtype
        x = x.float()
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)
        return self.weight * x.to(dt)


def _rope(x, pos, base):
    d = x.shape[-1]
    inv = 1.0 / (base ** (torch.arange(0, d, 2, device=x.device, dtype=torch.float32) / d))
    ang = pos[..., None].float() * inv                      # [B,T,d/2]
    cos, sin = torch.cat((ang, ang), -1).cos()[:, None], torch.cat((ang, ang), -1).sin()[:, None]
    x1, x2 = x[..., :d // 2], x[..., d // 2:]
    return (x.float() * cos + torch.cat((-x2, x1), -1).float() * sin).to(x.dtype)


class TorchDecoderLayer(nn.Module):
    """Llama-style pre-norm decoder layer (EAGLE drops the input norm of layer 0); HF parameter names."""

    def __init__(self, config, index: int = 0):
        super().__init__()
        H = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.num_kv = getattr(config, "num_key_value_heads", self.num_heads)
        self.head_dim = H // self.num_heads
        self.rope_theta = getattr(config, "rope_theta", 10000.0)
        self.index = index
        attn = nn.Module()
        attn.q_proj = nn.Linear(H, self.num_heads * self.head_dim, bias=False)
        attn.k_proj = nn.Linear(H, self.num_kv * self.head_dim, bias=False)
        attn.v_proj = nn.Linear(H, self.num_kv * self.head_dim, bias=False)
        attn.o_proj = nn.Linear(self.num_heads * self.head_dim, H, bias=False)
        self.self_attn = attn
        mlp = nn.Module()
        I = config.intermediate_size
        mlp.gate_proj, mlp.up_proj, mlp.down_proj = nn.Linear(H, I, bias=False), nn.Linear(H, I, bias=False), nn.Linear(I, H, bias=False)
        self.mlp = mlp
        if index != 0:
            self.input_layernorm = _RMSNorm(H, getattr(config, "rms_norm_eps", 1e-6))
        self.post_attention_layernorm = _RMSNorm(H, getattr(config, "rms_norm_eps", 1e-6))

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                use_cache=False, **_):
        B, T, H = hidden_states.shape
        x = self.input_layernorm(hidden_states) if self.index != 0 else hidden_states
        a = self.self_attn
        q = a.q_proj(x).view(B, T, self.num_heads, self.head_dim).transpose(1, 2)
        k = a.k_proj(x).view(B, T, self.num_kv, self.head_dim).transpose(1, 2)
        v = a.v_proj(x).view(B, T, self.num_kv, self.head_dim).transpose(1, 2)
        pos = position_ids.expand(B, T) if position_ids.dim() == 2 else position_ids[None].expand(B, T)
        q, k = _rope(q, pos, self.rope_theta), _rope(k, pos, self.rope_theta)
        if past_key_value is not None:
            k, v = torch.cat([past_key_value[0], k], dim=2), torch.cat([past_key_value[1], v], dim=2)
        present = (k, v) if use_cache else None
        rep = self.num_heads // self.num_kv
        kk, vv = (k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)) if rep > 1 else (k, v)
        w = torch.matmul(q, kk.transpose(2, 3)) / math.sqrt(self.head_dim)
        if attention_mask is not None:
            w = w + attention_mask
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(w, vv).transpose(1, 2).reshape(B, T, H)
        hidden_states = hidden_states + a.o_proj(o)
        y = self.post_attention_layernorm(hidden_states)
        m = self.mlp
        hidden_states = hidden_states + m.down_proj(torch.nn.functional.silu(m.gate_proj(y)) * m.up_proj(y))
        out = (hidden_states,)
        if use_cache:
            out += (present,)
        return out


# ----------------------------------------------------------------------------- one C call per drafting depth
class DraftPlan:
    """The depth loop of the EAGLE-2 drafter on lantern_draft_depth: every buffer of a drafting depth allocated once per (model, head), the argument
    block filled once per drafting call, ONE ctypes call per depth (input stage, the decoder layer, the fused head expansion and the next depth's
    inputs are enqueued by the library).  Built by Model._depth_plan when the model runs the HIP layers; the Python loop stays for everything else."""

    ROWS = 64

    def __init__(self, model, head, k: int, hx: dict, T: Optional[int] = None, depth: Optional[int] = None):
        import ctypes as C
        from .. import _lib
        from .decoder_layer import LlamaDecoderLayer
        self._C, self._L = C, _lib.lib()
        layer = model.layers[0]
        at, mlp = layer.self_attn, layer.mlp
        dev = model.fc.weight.device
        bf = torch.bfloat16
        self.model, self.layer, self.k, self.depth, self.dev = model, layer, k, int(model.depth if depth is None else depth), dev
        B, T, H = 2, (k if T is None else int(T)), at.hidden_size          # T: rows per depth (EAGLE-2: top_k; a static tree: its widest level)
        nq, nk, d, inter = at.num_heads, at.num_key_value_heads, at.head_dim, mlp.gate_proj.out_features
        self.B, self.T, self.H, self.nq, self.nk, self.d = B, T, H, nq, nk, d
        self.llama = isinstance(layer, LlamaDecoderLayer)
        a = self.args = _lib.DraftDepthArgs()
        a.layer_kind, a.B, a.T, a.H, a.n_q_heads, a.n_kv_heads, a.head_dim, a.inter, a.vocab = int(self.llama), B, T, H, nq, nk, d, inter, model.vocab_size
        a.eps2 = float(layer.post_attention_layernorm.variance_epsilon)
        a.embed_scale = float(model.embed_upscale) if model.embed_upscale > 1.0 else 1.0
        keep = self._keep = []

        def ptr(t):
            if t is None:
                return None
            keep.append(t)
            return t.data_ptr()
        emb = model.embed_tokens.weight
        a.embed = ptr(emb if emb.dtype == bf else emb.to(bf))
        a.fc_w, a.fc_b, a.fc_packed = ptr(model._packed_weight("fc", model.fc.weight).data), ptr(model.fc.bias), 1
        if hasattr(layer, "input_layernorm"):
            a.ln1_w, a.eps1 = ptr(layer.input_layernorm.weight), float(layer.input_layernorm.variance_epsilon)
        wq, b = layer._packed_fused("qkv", (at.q_proj, at.k_proj, at.v_proj))
        a.qkv_w, a.qkv_b = ptr(wq.data), ptr(b)
        a.o_w, a.o_b = ptr(layer._packed("o", at.o_proj.weight).data), ptr(at.o_proj.bias)
        a.ln2_w = ptr(layer.post_attention_layernorm.weight)
        wg, bg = layer._packed_fused("gate_up", (mlp.gate_proj, mlp.up_proj), inter)
        a.gate_up_w, a.gate_up_b = ptr(wg.data), ptr(bg)
        a.down_w, a.down_b, a.layer_packed = ptr(layer._packed("down", mlp.down_proj.weight).data), ptr(mlp.down_proj.bias), 1
        if self.llama:
            self.freqs = model.freqs_cis.to(device=dev, dtype=torch.float32).contiguous()
            a.freqs, a.table_rows = ptr(self.freqs), self.freqs.shape[0]
        else:
            cos, sin = at.rotary_emb.tables_bf16(dev, at.rotary_emb.max_seq_len_cached)
            a.qn_w, a.qn_b, a.kn_w, a.kn_b = ptr(at.q_norm.weight), ptr(at.q_norm.bias), ptr(at.k_norm.weight), ptr(at.k_norm.bias)
            a.model_parallel, a.cos_table, a.sin_table, a.table_rows = at.q_norm.weight.shape[0], ptr(cos), ptr(sin), cos.shape[0]
        # head
        a.head_w, a.head_b, a.head_packed = ptr(hx["packed"].data if hx["packed"] is not None else head.weight), ptr(head.bias), int(hx["packed"] is not None)
        a.row_lo, a.n_cols, a.model, a.cfg = hx["row_lo"], hx["n_cols"], hx["model"], float(model.cfg_scale)
        a.pos_base, a.w_latent, a.h_latent, a.newline_id, a.eos_id = 2, 48, 48, 8803, 8196
        a.top_k_filter, a.top_k = hx["top_k_filter"], k
        # state / logs / work buffers
        D = self.depth
        z = lambda *shape, dt=bf: torch.zeros(shape, dtype=dt, device=dev)
        self.hidden = z(2, B, T, H)                      # ping-pong: depth i reads [i & 1], writes [(i + 1) & 1]
        self.ids = z(D + 1, B * T, dt=torch.int64)
        # the score / token / parent lists of a drafting call, contiguous with the first expansion's entries in front: what tree_dynamic_finalize takes is a view
        self.all_par = z(1 + (D + 1) * T, dt=torch.int64)
        self.parents = self.all_par[1:].view(D + 1, T)
        self.tree_bits = z(self.ROWS, dt=torch.int64)
        self.all_ti = z(k + D * T * k, dt=torch.int64)
        self.all_cu = z(k + D * T * k, dt=torch.float32)
        self.log_ti = self.all_ti[k:].view(D, T, k)
        self.log_cu = self.all_cu[k:].view(D, T, k)
        self.cs = z(D, k, dt=torch.int64)
        self.sc = z(D + 1, k, dt=torch.float32)         # [0]: the first expansion's scores
        self.pos = z(D, B, T, dt=torch.int64)
        self.head_pos = z(D, T, dt=torch.int64)
        self.kv_start = z(B, dt=torch.int64)
        self._kv_start_zero = True          # (freshly zeroed; begin / draft re-zero it only after a call that set it)
        self.bits0 = (torch.ones(T, dtype=torch.int64, device=dev) << torch.arange(T, dtype=torch.int64, device=dev))
        self.par0 = torch.arange(T, dtype=torch.int64, device=dev) + 1
        self.steps = torch.arange(D, dtype=torch.int64, device=dev)
        self.steps_full = self.steps[:, None, None].expand(D, B, T).contiguous()          # depth index of every (depth, batch row, token) position slot
        self.steps1_row = (self.steps[:, None] + 1).expand(D, T).contiguous()              # + 1: the head's positions
        self.parents[0].copy_(self.par0)
        self.tree_bits[:T].copy_(self.bits0)
        nqkv = (nq + 2 * nk) * d
        self.work = dict(x=z(B * T, H), xn=z(B * T, H), qkv=z(B * T, nqkv), q=z(B, nq, self.ROWS, d), attn=z(B, self.ROWS, H), h1=z(B * T, H), hn=z(B * T, H),
                         act=z(B * T, inter), out=z(B * T, H), head_ws=z(T, hx["n_cols"]))
        for n, t in self.work.items():
            setattr(a, n, t.data_ptr())
        self.ta = None          # (the stream-K workspace is taken per run(): it belongs to the stream the launches go to, "one launch at a time per workspace")

    def begin(self, pkv, hidden0, ids0, scores0, positions, head_positions=None, kv_start=None, len_posi=None, position_diff=None, head_len=None):
        """Start a drafting call: pkv the prefix cache (the layer's slab grows in place), hidden0 [2, T, H] / ids0 [T] / scores0 [T] the first
        expansion's outputs, positions(i) [2, T] or [T] int64 per depth as a [depth, ...] tensor -- or positions None and len_posi (an int, or a [B, 1] /
        [B] device tensor: the streams' lengths; depth i sits at len_posi + i; position_diff: the unconditional row's offset, cnets_anole.py:858-862)
        and head_len (the stream whose positions drive the Lumina grammar rows: the head sees head_len + i + 1), filled here with one launch each."""
        a, B, T, D = self.args, self.B, self.T, self.depth
        past = pkv[0][0].shape[2]
        ks, vs = self.layer._cache_slab(B, self.nk, self.d, past + D * T, self.dev, pkv[0])
        a.k_slab, a.v_slab, a.kv_rows = ks.data_ptr(), vs.data_ptr(), ks.shape[2]
        self._slab, self.past = (ks, vs), past
        need = int(self._L.lantern_tree_attention_workspace(B, self.nq, self.ROWS, self.d, self._C.c_int64(ks.shape[2])))
        if self.ta is None or self.ta.numel() < need:
            self.ta = torch.empty(max(need, 16), dtype=torch.uint8, device=self.dev)
        a.ta_ws, a.ta_ws_bytes = self.ta.data_ptr(), self.ta.numel()
        # (hidden0 / ids0 may arrive as broadcast views -- `last_hidden[:, None].expand(B, T, H)`, `ids.reshape(1, T).expand(B, T)` -- and are materialised by
        # these copies, one launch each; parents[0] and the first T ancestor words are constants no depth writes: set once in __init__)
        self.hidden[0].copy_(hidden0)
        self.ids[0].view(B, T).copy_(ids0.reshape(1, T).expand(B, T))
        self.sc[0].copy_(scores0.reshape(-1))
        if positions is not None:
            self.pos.copy_(positions.reshape(D, -1, T).expand(D, B, T) if positions.numel() != D * B * T else positions.reshape(D, B, T))
        elif torch.is_tensor(len_posi):
            torch.add(self.steps_full, len_posi.reshape(1, B, 1), out=self.pos)
        else:
            torch.add(self.steps_full, int(len_posi), out=self.pos)
            if position_diff is not None:
                self.pos[:, 1].sub_(position_diff.reshape(()))
        if head_positions is not None:
            self.head_pos.copy_(head_positions)
        elif head_len is not None:
            torch.add(self.steps1_row, head_len.reshape(()), out=self.head_pos)
            head_positions = self.head_pos
        if kv_start is None:
            if not self._kv_start_zero:
                self.kv_start.zero_()
                self._kv_start_zero = True
        else:
            self.kv_start.copy_(kv_start)
            self._kv_start_zero = False
        a.kv_start = self.kv_start.data_ptr()
        a.positions_per_batch_row = 1
        a.cfg = float(self.model.cfg_scale)
        self._lumina_grammar = head_positions is not None

    def run(self, i: int):
        """Depth i: one C call."""
        a, T, D = self.args, self.T, self.depth
        k = self.k
        a.stream = torch.cuda.current_stream().cuda_stream
        sk = ops._sk_workspace(self.dev)          # the CURRENT stream's workspace: a plan cached across generate() calls may run on another stream
        a.sk_ws, a.sk_ws_bytes = sk.data_ptr(), sk.numel()
        a.ids, a.hidden_in = self.ids[i].data_ptr(), self.hidden[i & 1].data_ptr()
        a.position_ids = self.pos[i].data_ptr()
        a.head_pos = self.head_pos[i].data_ptr() if self._lumina_grammar else None
        a.kv_row0, a.t1 = self.past + i * T, (i + 1) * T
        a.tree_bits = self.tree_bits.data_ptr()
        a.scores_in = self.sc[i].data_ptr()
        a.topk_index, a.cu_scores = self.log_ti[i].data_ptr(), self.log_cu[i].data_ptr()
        a.topk_cs_index, a.scores_out = self.cs[i].data_ptr(), self.sc[i + 1].data_ptr()
        if i + 1 < D:
            a.hidden_next, a.ids_next, a.parents_next = self.hidden[(i + 1) & 1].data_ptr(), self.ids[i + 1].data_ptr(), self.parents[i + 1].data_ptr()
            a.parent_bias_next = 1 + k * k * max(0, i) + k          # the bias of iteration i + 1 (cnets_llamagen.py:798-801)
        else:
            a.hidden_next = a.ids_next = a.parents_next = None
        ops.check(self._L.lantern_draft_depth(self._C.byref(a)), "draft_depth")

    def finalize_inputs(self, cu0, ti0):
        """(scores [1, n], tokens [1, n], parents [1, m]) of the whole drafting call for tree_dynamic_finalize -- the first expansion's k entries copied in
        front of the depth loop's logs (which the depth calls wrote in place): no torch.cat."""
        k, D, T = self.k, self.depth, self.T
        self.all_cu[:k].copy_(cu0.reshape(-1))
        self.all_ti[:k].copy_(ti0.reshape(-1))
        return self.all_cu[None], self.all_ti[None], self.all_par[None, :1 + D * T]

    def lists(self):
        """(scores_list, ss_token, parents_list) entries of the depth loop, as the Python loop appends them."""
        D = self.depth
        return ([self.log_cu[i].reshape(-1) for i in range(D)], [self.log_ti[i].reshape(-1) for i in range(D)], [self.parents[i] for i in range(D)])


class StaticDraftPlan(DraftPlan):
    """The static-tree loops (EAGLE v1: topK_generate(tree_type="static"), topK_genrate_v1) on the same call: per depth ONE lantern_draft_depth with
    n_draw > 0 -- input stage, the decoder layer on the level's T_i rows, head + CFG + processors + softmax + Model.sample's draws without replacement
    (lantern_head_sample), the next level's tokens / hidden rows through the tree's tables -- behind ONE lantern_head_sample for the root row.  The
    per-level Python of the reference (sample -> torch.cat -> repeat_hidden -> forward -> head -> processors: cnets_lumina_mgpt.py:1245-1328,
    cnets_llamagen.py:944-1023) is gone from the host.  Draws: injected uniforms (`Model.static_draw_uniforms`, default torch.rand on the device, one
    call per drafting cycle) or injected indices (`Model.static_draw_idx`), see include/lantern_hip.h lantern_head_sample."""

    def __init__(self, model, head, k: int, hx: dict, tb: dict):
        levels = [int(len(t)) for t in tb["tree_indices"]]
        super().__init__(model, head, k, hx, T=max(levels), depth=len(levels))
        dev, V = self.dev, int(model.vocab_size)
        self.levels = levels
        self.row_off = [1 + sum(levels[:i]) for i in range(len(levels) + 1)]          # ss rows: 0 = the root's, then level by level
        self.R = self.row_off[-1]
        self.key_off = [sum(levels[:i]) for i in range(len(levels) + 1)]             # tree keys in front of level i
        i32 = lambda x: torch.as_tensor(x, dtype=torch.int32, device=dev).contiguous()
        # level i's tokens = the draws of the rows one level up, flattened, at tree_indices[i]; its hidden rows = the parents' output rows
        self.gather = [i32(t) for t in tb["tree_indices"]]
        self.rep = [i32(torch.repeat_interleave(torch.arange(len(r)), torch.as_tensor(r))) for r in tb["repeat_nums"]]
        bits = torch.zeros(self.ROWS, dtype=torch.int64)
        for i, m in enumerate(tb["attn_mask"]):
            m2 = m.reshape(m.shape[-2], m.shape[-1]).cpu()
            w = torch.ones(m2.shape[1], dtype=torch.int64) << torch.arange(m2.shape[1], dtype=torch.int64)
            bits[self.key_off[i]:self.key_off[i + 1]] = ((m2 != 0).to(torch.int64) * w).sum(-1)
        self.tree_bits.copy_(bits.to(dev))
        f32, i64, f64 = torch.float32, torch.int64, torch.float64
        z = lambda *shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        # double-buffered results: the previous drafting call's (ss_token, ss_prob, ss_op) stay intact while the next call fills the other set
        self.res = [dict(probs=z(self.R, V, dt=f32), tok=z(self.R, k, dt=i64), prob=z(self.R, k, dt=f32)) for _ in range(2)]
        self.parity = 0
        self.draw_u = z(self.R, k, dt=f64)
        self.draw_idx = z(self.R, k, dt=i64)
        # positions of all levels in ONE buffer, level blocks [B, T_i] back to back, filled by one gather + one add per drafting call (the per-level
        # add / expand / copy pairs were 16 tiny launches per call); the head positions (root row first, then level by level) by one add
        n_pos = self.B * sum(levels)
        self.pos_all = z(n_pos, dt=i64)
        self.pos_l, pb, pi, o = [], [], [], 0
        for i, t in enumerate(levels):
            self.pos_l.append(self.pos_all[o:o + self.B * t].view(self.B, t))
            pb += [b for b in range(self.B) for _ in range(t)]
            pi += [i] * (self.B * t)
            o += self.B * t
        self.pos_b, self.pos_i = torch.as_tensor(pb, dtype=i64, device=dev), torch.as_tensor(pi, dtype=i64, device=dev)
        self.pos_unc = (self.pos_b == 1).to(i64)
        self.head_all = z(1 + sum(levels), dt=i64)
        self.head_off = torch.as_tensor([0] + [i + 1 for i, t in enumerate(levels) for _ in range(t)], dtype=i64, device=dev)
        self.head_pos0, self.head_pos_l, o = self.head_all[0:1], [], 1
        for t in levels:
            self.head_pos_l.append(self.head_all[o:o + t])
            o += t
        self.last = z(self.B, 1, self.H, dt=torch.bfloat16)
        self.args.n_draw = k

    def draft(self, pkv, last_hidden, len_posi, head_len=None, kv_start=None, position_diff=None):
        """One drafting call behind the prefill: last_hidden [2, H]; len_posi: [2, 1] device tensor (Lumina: the cond / uncond stream lengths) or an int
        (LlamaGen / Anole: + position_diff for the uncond row, cnets_anole.py:858-862); head_len: the stream whose positions drive the Lumina grammar mask
        (len_posi[1]) or None.  Returns (ss_token [R, k], ss_prob [R, k], [probs rows per level])."""
        a, C, L, B, k, D = self.args, self._C, self._L, self.B, self.k, self.depth
        mdl = self.model
        self.parity ^= 1
        res = self.res[self.parity]
        past = pkv[0][0].shape[2]
        ks, vs = self.layer._cache_slab(B, self.nk, self.d, past + self.key_off[-1], self.dev, pkv[0])
        a.k_slab, a.v_slab, a.kv_rows = ks.data_ptr(), vs.data_ptr(), ks.shape[2]
        self._slab, self.past = (ks, vs), past
        need = int(L.lantern_tree_attention_workspace(B, self.nq, self.ROWS, self.d, C.c_int64(ks.shape[2])))
        if self.ta is None or self.ta.numel() < need:
            self.ta = torch.empty(max(need, 16), dtype=torch.uint8, device=self.dev)
        a.ta_ws, a.ta_ws_bytes = self.ta.data_ptr(), self.ta.numel()
        if kv_start is None:
            if not self._kv_start_zero:
                self.kv_start.zero_()
                self._kv_start_zero = True
        else:
            self.kv_start.copy_(kv_start)
            self._kv_start_zero = False
        a.kv_start, a.positions_per_batch_row, a.cfg = self.kv_start.data_ptr(), 1, float(mdl.cfg_scale)
        a.tree_bits = self.tree_bits.data_ptr()
        # the draws of this call: injected indices, injected uniforms, or fresh uniforms (ONE torch.rand per drafting call)
        idx = getattr(mdl, "static_draw_idx", None)
        if idx is not None:
            self.draw_idx.copy_(idx.reshape(self.R, k))
            du, di = None, self.draw_idx
        else:
            src = getattr(mdl, "static_draw_uniforms", None)
            if src is not None:
                self.draw_u.copy_(src(self.R, k))
            else:
                torch.rand((self.R, k), dtype=torch.float64, device=self.dev, out=self.draw_u)
            du, di = self.draw_u, None
        self._du, self._di = du, di
        # positions of every level (len_posi advances by one per level; the tree's own position offsets are zero: utils_c.py:100-179)
        lumina = head_len is not None
        if torch.is_tensor(len_posi):
            torch.index_select(len_posi.reshape(B), 0, self.pos_b, out=self.pos_all)
            self.pos_all.add_(self.pos_i)
        else:
            torch.add(self.pos_i, int(len_posi), out=self.pos_all)
            if position_diff is not None:          # the unconditional row's positions (cnets_anole.py:858-862), every level at once
                self.pos_all.addcmul_(self.pos_unc, position_diff.reshape(1).to(torch.int64), value=-1)
        self._lumina = lumina
        if lumina:
            torch.add(self.head_off, head_len.reshape(1), out=self.head_all)
        # ---- the root row: head + sample on the prefill's last hidden row (cond, uncond)
        stream = torch.cuda.current_stream().cuda_stream
        sk = ops._sk_workspace(self.dev)
        self.last.copy_(last_hidden.reshape(B, 1, self.H))
        u0 = None if du is None else du.data_ptr()
        i0 = None if di is None else di.data_ptr()
        ops.check(L.lantern_head_sample(C.c_void_p(self.last.data_ptr()), C.c_void_p(a.head_w), C.c_void_p(a.head_b), 1, self.H, a.row_lo, a.n_cols, a.vocab,
                                        C.c_float(a.cfg), a.model, C.c_void_p(self.head_pos0.data_ptr() if lumina else None), C.c_int64(a.pos_base), a.w_latent,
                                        a.h_latent, a.newline_id, a.eos_id, a.top_k_filter, k, C.c_void_p(u0), C.c_void_p(i0), C.c_void_p(a.head_ws),
                                        C.c_void_p(res["probs"].data_ptr()), C.c_void_p(res["tok"].data_ptr()), C.c_void_p(res["prob"].data_ptr()),
                                        a.head_packed, C.c_void_p(sk.data_ptr()), C.c_size_t(sk.numel()), C.c_void_p(stream)), "head_sample")
        # ---- the levels: each level's input stage reads its tokens (the draws one level up, through tree_indices) and its hidden rows (the parents' output
        # rows, repeat_hidden) through the tree's tables -- level 0 from the root's draws and the last hidden row; nothing is copied between levels
        for i in range(D):
            self.run_level(i, res, stream, sk)
        return res["tok"], res["prob"], [res["probs"][0:1]] + [res["probs"][self.row_off[i]:self.row_off[i + 1]] for i in range(D)]

    def run_level(self, i: int, res, stream, sk):
        """Level i: ONE C call."""
        a, k, D = self.args, self.k, self.depth
        T = self.levels[i]
        r0 = self.row_off[i]
        a.stream = stream
        a.sk_ws, a.sk_ws_bytes = sk.data_ptr(), sk.numel()
        a.T = T
        # tokens: the flat draws of the rows one level up ([T_prev, k]; level 0: the root row's); hidden: the rows one level up (level 0: the last hidden row)
        rp = self.row_off[i - 1] if i > 0 else 0
        n_prev = self.levels[i - 1] if i > 0 else 1
        a.ids = res["tok"][rp:].data_ptr()
        a.hidden_in = self.work["out"].data_ptr() if i > 0 else self.last.data_ptr()
        a.in_gather, a.in_rep, a.in_src_T, a.in_n_flat = self.gather[i].data_ptr(), self.rep[i].data_ptr(), n_prev, n_prev * k
        a.position_ids = self.pos_l[i].data_ptr()
        a.head_pos = self.head_pos_l[i].data_ptr() if self._lumina else None
        a.kv_row0, a.t1 = self.past + self.key_off[i], self.key_off[i + 1]
        a.draw_u = None if self._du is None else self._du[r0:].data_ptr()
        a.draw_idx = None if self._di is None else self._di[r0:].data_ptr()
        a.probs_out, a.ss_token, a.ss_prob = res["probs"][r0:].data_ptr(), res["tok"][r0:].data_ptr(), res["prob"][r0:].data_ptr()
        a.T_next = 0          # (the next level reads through the tables: no next-inputs launch)
        a.next_gather = a.next_rep = a.hidden_next = a.ids_next = None
        ops.check(self._L.lantern_draft_depth(self._C.byref(a)), "draft_depth")


# ----------------------------------------------------------------------------- the drafter
class Model(nn.Module):
    """`model_type`: "lumina_mgpt" | "llamagen" | "anole" selects the logit post-processing of the head's output
    (SURVEY 8a-bis): Lumina -> position-dependent MultiModalLogitsProcessor + top-k; LlamaGen -> HF processors; Anole ->
    non-image ids to finfo.min, then HF processors."""

    def __init__(self, config, layers: Optional[List[nn.Module]] = None, bias=True, total_tokens=63, depth=5, top_k=8, threshold=1.0,
                 embed_upscale=1.0, model_type="lumina_mgpt", image_lo=4, image_hi=8196, allow_torch_layers=False):
        super().__init__()
        self.config = config
        self.padding_idx = getattr(config, "pad_token_id", None)
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.top_k = top_k
        self.total_tokens = total_tokens - 1
        self.depth = depth
        self.threshold = math.log(threshold)
        self.embed_upscale = embed_upscale
        n_layers = getattr(config, "num_hidden_layers", 1)
        self.freqs_cis = None
        if layers is None:
            layers = self._default_layers(config, n_layers, model_type, allow_torch_layers)
        self.layers = nn.ModuleList(layers)
        self.fc = nn.Linear(2 * config.hidden_size, config.hidden_size, bias=bias)
        self.logsoftmax = nn.LogSoftmax(dim=-1)
        self.model_type, self.image_lo, self.image_hi = model_type, image_lo, image_hi
        self.cfg_scale = 3.0
        self.tree_mask = None
        self.stable_kv = None
        self.layer_kwargs = None          # optional callable(position_ids) -> extra kwargs for injected layers
        if self.freqs_cis is not None:    # LlamaGen's layers take the rows of the 2-D rotary table at the call's positions (cnets_llamagen.py:661-663)
            self.layer_kwargs = self._freqs_kwargs
        for p in self.embed_tokens.parameters():
            p.requires_grad = False

    def _default_layers(self, config, n_layers, model_type, allow_torch_layers):
        """The HIP decoder layers of the model family when the config fits their kernels; otherwise an error -- or, asked for explicitly, the
        plain-torch stand-in.  (The drafting shape never slides onto torch's ops unnoticed.)"""
        from .decoder_layer import DecoderLayer, LlamaDecoderLayer, precompute_freqs_cis_2d
        H, nh = config.hidden_size, config.num_attention_heads
        fits = H % nh == 0 and (H // nh) in (64, 128) and H % 64 == 0 and getattr(config, "intermediate_size", 0) % 64 == 0
        if not fits:
            if allow_torch_layers:
                return [TorchDecoderLayer(config, i) for i in range(n_layers)]
            raise ops._lib.LanternError(
                f"cnets.Model: hidden_size={H}, heads={nh} is outside the HIP decoder layer (head_dim 64 or 128, hidden and intermediate sizes multiples "
                "of 64); pass layers=[...] (any module with the reference layer's signature) or allow_torch_layers=True for the plain-torch stand-in")
        if model_type == "llamagen":
            # block size / class-token count by input type, as the reference sets them (cnets_llamagen.py:561-580)
            block, cls = {"c2i": (576, 0), "t2i": (256, 119), "t2i2": (1024, 119)}[getattr(config, "input_type", "t2i")]
            grid = int(block ** 0.5)
            table = precompute_freqs_cis_2d(grid, H // nh, float(getattr(config, "rope_base", 10000)), cls)
            self.freqs_cis = torch.cat([table, torch.zeros_like(table[:10])], dim=0)
            layers = [LlamaDecoderLayer(config, i) for i in range(n_layers)]
            for l in layers:
                l.inplace_cache = True          # the drafter never returns to an older, longer cache: `present` grows in a layer-owned slab
            return layers
        if model_type == "anole":          # one head-norm row per head (cnets_anole.py:317-332, :363-364)
            import copy
            config = copy.copy(config)
            config.model_parallel_size = nh
            if getattr(config, "num_key_value_heads", nh) != nh:
                raise ops._lib.LanternError("cnets.Model: the Anole drafter layer's per-head norm rows are built for num_key_value_heads == num_attention_heads")
        layers = [DecoderLayer(config, i) for i in range(n_layers)]
        for l in layers:
            l.inplace_cache = True
        return layers

    def _freqs_kwargs(self, position_ids):
        if self.freqs_cis.device != position_ids.device:
            self.freqs_cis = self.freqs_cis.to(position_ids.device)
        return dict(freqs_cis=self.freqs_cis[position_ids].squeeze(0))

    @classmethod
    def from_reference(cls, ref, model_type: str, **kw):
        """The reference's drafter `Model` (cnets_lumina_mgpt / cnets_llamagen / cnets_anole, loaded by its own `from_pretrained`) re-hosted on this
        class: same config, same tree parameters, its state_dict loaded as is (the parameter names are the reference's), HIP decoder layers."""
        # (the reference's drafter Model does not keep its config; its attention modules do: cnets_llamagen.py:229, cnets_lumina_mgpt.py:411-430)
        cfg = kw.pop("config", None) or getattr(ref, "config", None) or ref.layers[0].self_attn.config
        has_bias = getattr(ref.fc, "bias", None) is not None
        m = cls(cfg, bias=has_bias, total_tokens=int(ref.total_tokens) + 1, depth=int(ref.depth), top_k=int(ref.top_k), model_type=model_type,
                embed_upscale=float(getattr(ref, "embed_upscale", 1.0)), **kw)
        m.threshold = float(getattr(ref, "threshold", m.threshold))
        missing, unexpected = m.load_state_dict(ref.state_dict(), strict=False)
        bad = [k for k in list(missing) + list(unexpected) if "rotary_emb" not in k]
        if bad:
            raise ops._lib.LanternError(f"cnets.Model.from_reference: parameters that do not line up with the reference drafter: {bad[:8]}")
        p = next(ref.parameters())
        return m.to(device=p.device, dtype=p.dtype)

    # ------------------------------------------------------------------ tree state (cnets_lumina_mgpt.py:1000-1012)
    def init_tree(self, tree=None):
        dev = self.embed_tokens.weight.device
        if tree is not None:                                   # EAGLE v1: static drafter buffers
            self.tree = tree
            tb = ops.tree_drafter_build(tree, TOPK)
            t = lambda a: torch.from_numpy(a).to(dev)
            self.tree_buffer = dict(attn_mask=[t(m)[None, None] for m in tb["attn_mask"]], tree_indices=[t(x) for x in tb["tree_indices"]],
                                    position_ids=[t(x) for x in tb["position_ids"]], repeat_nums=tb["repeat_nums"])
        else:                                                  # EAGLE v2
            self.tree_mask_init = torch.eye(self.top_k, device=dev)[None, None]
            self.position_ids = torch.zeros(self.top_k, device=dev, dtype=torch.long)

    def init_tree_v1(self, tree_choices):                      # cnets_llamagen.py:914-916
        self.init_tree(tree_choices)

    def reset(self):
        self.tree_mask = None

    def reset_kv(self):
        self.stable_kv = None
        self._poll_mask_checks(wait=True)          # end of an image: every deferred left-padding verdict is in

    # ------------------------------------------------------------------ a5
    def _prepare_decoder_attention_mask(self, attention_mask, input_shape, inputs_embeds, past_key_values_length):
        B, T = input_shape
        tm = self.tree_mask if getattr(self, "tree_mask", None) is not None else None
        if attention_mask is None and T == 1 and tm is None:
            return None
        return ops.drafter_attention_mask(attention_mask, tm, B, T, past_key_values_length, device=inputs_embeds.device)

    def _packed_weight(self, name, weight, row_lo=0, n_rows=None):
        """ops.pack_linear_weight of weight[row_lo : row_lo + n_rows], made once per weight version (the brick layout the stream-K kernel streams)."""
        cache = self.__dict__.setdefault("_packed_weights", {})
        key = (weight.data_ptr(), weight._version, row_lo, n_rows)
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            rows = weight if n_rows is None else weight[row_lo:row_lo + n_rows]
            hit = cache[name] = (key, ops.pack_linear_weight(rows.contiguous()))
        return hit[1]

    def _input_stage(self, hidden_states, input_ids):
        """embed_tokens(ids) -> cast -> x upscale -> fc(cat(embeds, hidden)) (cnets_lumina_mgpt.py:1071,1095-1098)."""
        B, T, H = hidden_states.shape
        w = self.fc.weight
        if hidden_states.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and H % 16 == 0:
            emb = self.embed_tokens.weight
            emb = emb if emb.dtype == torch.bfloat16 else emb.to(torch.bfloat16)
            ids, hid = input_ids.reshape(-1), hidden_states.reshape(B * T, H)
            out = torch.empty_like(hid)
            pk = self._packed_weight("fc", w) if (B * T <= 32 and H % 64 == 0) else None          # the drafting shape streams the packed weight
            for s in range(0, B * T, 128):                     # one drafter call is <= ~120 rows; prefills go in slices
                out[s:s + 128] = ops.drafter_fc(ids[s:s + 128], hid[s:s + 128], emb, w, self.fc.bias,
                                                embed_scale=float(self.embed_upscale) if self.embed_upscale > 1.0 else 1.0, packed=pk)
            return out.view(B, T, H)
        # other dtypes (f32 checkpoints in tests): the same arithmetic in torch on the device
        e = self.embed_tokens(input_ids).to(hidden_states.dtype)
        if self.embed_upscale > 1.0:
            e = e * self.embed_upscale
        return self.fc(torch.cat((e, hidden_states), dim=-1))

    def forward(self, hidden_states, input_ids, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, std=None):
        B, T, _ = hidden_states.shape
        # ops.mask_left_padding(attention_mask) when the caller of this forward already took it (topK_generate): left on the module for THIS mask object
        # (the signature stays the reference's: subclasses override forward)
        pend = self.__dict__.pop("_pending_mask_stats", None)
        mask_stats = pend[1] if pend is not None and pend[0] is attention_mask else None
        past = past_key_values[0][0].shape[2] if past_key_values is not None else 0
        if position_ids is None:
            position_ids = torch.arange(past, T + past, dtype=torch.long, device=hidden_states.device)[None].view(-1, T)
        else:
            position_ids = position_ids.view(-1, T).long()
        if attention_mask is None:
            attention_mask = torch.ones((B, T + past), dtype=torch.bool, device=hidden_states.device)
        elif attention_mask.is_cuda:
            # the ancestor-word attention path describes padding by ONE first-visible-key index per row (kv_start = argmax of the mask): never assumed.
            # At a prompt's prefill (past == 0) the mask is checked on the spot -- left padding only, ones contiguous up to the end; on the calls
            # behind it (past > 0: the accepted tokens, tree rows) the same test is enqueued without stalling the launch queue and read back on a
            # LATER call (`_poll_mask_checks`): a mask with holes, or one that stops being left padding, raises one call late instead of being
            # silently reduced to kv_start (the reference's additive mask would honour it)
            self._poll_mask_checks()
            # (one launch: per row the first non-zero index, the count and "a zero behind a one" -- the to(int64) / cummax / != / any / argmax chain was
            # seven launches on every drafting call)
            if mask_stats is None:
                mask_stats = ops.mask_left_padding(attention_mask.to(hidden_states.device))
            bad = mask_stats[2]
            if past == 0:
                if bool(bad.any()):
                    raise ops._lib.LanternError("cnets.Model.forward: attention_mask with zeros behind a one (not left padding): the tree-attention path "
                                                "takes one first-visible-key index per row")
            else:
                self._defer_mask_check(bad)
        extra = self.layer_kwargs(position_ids) if self.layer_kwargs is not None else {}
        first_key = (lambda: mask_stats[0]) if mask_stats is not None else (lambda: attention_mask.to(hidden_states.device).to(torch.int64).argmax(dim=1))
        self._mask_stats = mask_stats
        tm = getattr(self, "tree_mask", None)
        bits_ok = (hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16 and all(getattr(l, "supports_tree_bits", False) for l in self.layers))
        have_bits = False
        if (tm is not None and bits_ok and tm.shape[-1] <= 64 and tm.shape[-2] >= T and tm.shape[-1] >= T and past + T >= tm.shape[-1]):
            # the tree block of the mask as ancestor words + the left padding as a first-visible-key index: the layer's attention
            # runs on lantern_tree_attention (the additive mask still rides along for layers / shapes that want it)
            bits, t1 = ops.drafter_tree_bits(tm.to(hidden_states.device), T)
            extra = dict(extra, tree_bits=bits, tree_keys=t1, kv_start=first_key())
            have_bits = True
        elif tm is None and bits_ok and past > 0 and B * T <= 32:
            # the accepted tokens of a drafting call behind the cached prefix (a few rows): causal among themselves = a chain-shaped "tree"
            # (row i sees the new keys 0..i), so this call's attention runs on lantern_tree_attention too -- no additive mask, no torch attention
            chain = self.__dict__.setdefault("_chain_bits", {}).get((T, hidden_states.device))
            if chain is None:
                chain = self._chain_bits[(T, hidden_states.device)] = ((torch.ones(T, dtype=torch.int64, device=hidden_states.device) << (torch.arange(T, dtype=torch.int64, device=hidden_states.device) + 1)) - 1)
            extra = dict(extra, tree_bits=chain, tree_keys=T, kv_start=first_key())
            have_bits = True
        elif tm is None and bits_ok:
            # a prompt prefill (a handful to hundreds of rows): causal among the new tokens behind the left padding -- the HIP layers run their GEMMs
            # (row-blocked above 32 rows) and block-causal lantern_tree_attention on this hint (the additive mask still rides along for a layer whose
            # shapes keep it on torch's ops)
            extra = dict(extra, causal=True, kv_start=first_key())
        # the additive mask only when some layer may still want it (the HIP layers at the drafting shape take the ancestor words)
        fast = have_bits and B * T <= 32 and all(getattr(l, "fused", False) and hasattr(l, "_fast") for l in self.layers)
        mask = None if fast else self._prepare_decoder_attention_mask(attention_mask, (B, T), hidden_states, past)
        hidden_states = self._input_stage(hidden_states, input_ids.to(hidden_states.device))
        cache = () if use_cache else None
        for idx, layer in enumerate(self.layers):
            pkv = past_key_values[idx] if past_key_values is not None else None
            outs = layer(hidden_states, attention_mask=mask, position_ids=position_ids, past_key_value=pkv, output_attentions=output_attentions,
                         use_cache=use_cache, **extra)
            hidden_states = outs[0]
            if use_cache:
                cache += (outs[2 if output_attentions else 1],)
        return (hidden_states, cache) if use_cache else hidden_states

    # ------------------------------------------------------------------ deferred checks of the left-padding assumption (past > 0 calls)
    _MASK_RING = 8

    def _defer_mask_check(self, bad):
        """`bad`: 0-dim bool device tensor.  Copied to a pinned flag behind the work already enqueued; read by a later `_poll_mask_checks`."""
        st = self.__dict__.setdefault("_mask_chk", None)
        if st is None:
            st = self.__dict__["_mask_chk"] = dict(pins=torch.zeros((self._MASK_RING, 16), dtype=torch.int64).pin_memory(), events=[None] * self._MASK_RING, n=0)
        i = st["n"] % self._MASK_RING
        if st["events"][i] is not None:          # the ring came round: that slot's verdict is due now
            st["events"][i].synchronize()
            self._raise_if_bad(st, i)
        nb = min(int(bad.numel()), 16)
        if bad.numel() > 16:          # (more rows than the ring holds per slot: reduce on the device)
            bad, nb = bad.any().to(torch.int64).reshape(1), 1
        st["pins"][i, :nb].copy_(bad.reshape(-1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["events"][i] = ev
        st["n"] += 1

    def _raise_if_bad(self, st, i):
        st["events"][i] = None
        if bool(st["pins"][i].any()):
            st["pins"][i] = 0
            raise ops._lib.LanternError("cnets.Model.forward: an EARLIER call's attention_mask (behind a cached prefix) had zeros behind a one -- not left "
                                        "padding: the tree-attention path takes one first-visible-key index per row and would have ignored the holes")

    def _poll_mask_checks(self, wait: bool = False):
        st = self.__dict__.get("_mask_chk")
        if st is None:
            return
        for i, ev in enumerate(st["events"]):
            if ev is None:
                continue
            if wait:
                ev.synchronize()
            if ev.query():
                self._raise_if_bad(st, i)

    # ------------------------------------------------------------------ helpers of the tree loops
    def repeat_hidden(self, hidden_states, num_repeat):      # cnets_lumina_mgpt.py:930-934
        reps = torch.as_tensor(num_repeat, device=hidden_states.device)
        return torch.repeat_interleave(hidden_states[:, :len(num_repeat)], reps, dim=1)

    def sample(self, logits, logits_processor=None, k=1):
        """k draws without replacement + their conditional probabilities p_i / (1 - sum_{j<i} p_j) (cnets_lumina_mgpt.py:936-955)."""
        if logits_processor is not None and not isinstance(logits_processor, (list, tuple)):
            logits = logits_processor(None, logits)
        logits = logits.float()
        probs = torch.softmax(logits.view(-1, logits.shape[-1]), dim=-1)
        idx = torch.multinomial(probs, k, replacement=False)
        return idx, ops.sample_static(probs, idx), probs

    def _head(self, head, hidden):
        """`head(hidden)` -> [..., V] logits.  For Lumina-mGPT / Anole every drafted row is masked to the image-token ids right
        after the head, so when `head` is a bf16 nn.Linear only those rows of its weight are multiplied (lantern_linear_rows:
        64 MiB instead of 512 MiB of weights per call); the other columns of the returned tensor are never read (the mask
        kernels overwrite them).  Any other head (a callable, another dtype, LlamaGen's V == K) is simply called."""
        w = getattr(head, "weight", None)
        if (self.model_type in ("lumina_mgpt", "anole") and isinstance(head, nn.Linear) and w is not None and w.is_cuda
                and w.dtype == torch.bfloat16 and hidden.dtype == torch.bfloat16 and w.shape[1] % 16 == 0):
            V, H = w.shape
            lead = hidden.shape[:-1]
            A = hidden.reshape(-1, H)
            key = (A.shape[0], V)
            buf = self.__dict__.setdefault("_head_buf", {})
            if key not in buf:            # the non-window columns stay whatever they are (finite zeros): never read downstream
                buf[key] = torch.zeros((A.shape[0], V), dtype=torch.bfloat16, device=A.device)
            out = ops.linear_rows(A, w, self.image_lo, self.image_hi - self.image_lo, bias=head.bias, out=buf[key])
            return out.view(*lead, V)
        return head(hidden)

    def _expand_depth(self, head, hidden, proc, pos_ids, scores, k):
        """One expansion depth: head -> CFG -> processors -> log-softmax -> top-k (+ parents' scores) -> best k of n*k.  With a bf16 nn.Linear head
        every model takes the fused path (lantern_head_expand: the head's [2, n, V] logits never reach HBM) -- Lumina on its image window with
        the grammar rows, Anole on the same window (non-image ids at finfo.min never survive the log-softmax), LlamaGen on its whole 16384-id
        vocabulary; the HF processors the reference defaults to (top-k; temperature 1, top_p 1: generate_images.py:47-55) are the kernel's
        threshold, any other processor list takes the three-step composition (lantern_linear_rows -> lantern_cfg_mask_topk_window ->
        lantern_expand_dynamic).  hidden [2, n, H] or [2, H] (cond row(s), then uncond)."""
        n = hidden.shape[1] if hidden.dim() == 3 else 1
        hx = self._head_fusion(head, proc, n, k, pos_ids is not None) if hidden.dtype == torch.bfloat16 else None
        if hx is not None:
            return ops.head_expand(hidden.reshape(2 * n, -1), head.weight, hx["row_lo"], hx["n_cols"], float(self.cfg_scale), bias=head.bias, model=hx["model"],
                                   pos_ids=pos_ids.reshape(-1) if hx["model"] == ops.MODEL_LUMINA else None, pos_base=2, top_k_filter=hx["top_k_filter"],
                                   scores_in=scores, top_k=k, packed=hx["packed"])
        ho = self._head(head, hidden)
        if hidden.dim() == 2:
            half = ho.shape[0] // 2
            rows = self._post_head(ho[:half], ho[half:], proc, pos_ids=pos_ids)
        else:
            rows = self._post_head(ho[0], ho[1], proc, pos_ids=pos_ids)
        return ops.expand_dynamic(rows[None], scores, k)

    def _head_fusion(self, head, proc, n, k, with_pos):
        """What lantern_head_expand needs for this head / processor list, or None when the three-step composition has to run (another head type, a
        temperature or nucleus processor, shapes outside the kernel)."""
        w = getattr(head, "weight", None)
        if not (isinstance(head, nn.Linear) and w is not None and w.is_cuda and w.dtype == torch.bfloat16 and w.shape[1] % 16 == 0 and n <= 16 and n * k <= 256):
            return None
        V = w.shape[0]
        if self.model_type == "lumina_mgpt":
            if not with_pos:
                return None
            top_k = min(int(proc[1].image_top_k), V) if (proc is not None and len(proc) > 1) else 0
            lo, nc, model = self.image_lo, self.image_hi - self.image_lo, ops.MODEL_LUMINA
        else:
            from ..verify import ProcessorSpec
            spec = ProcessorSpec.from_hf(proc) or ProcessorSpec()
            lo, nc = (self.image_lo, self.image_hi - self.image_lo) if self.model_type == "anole" else (0, V)
            if spec.temperature != 1.0 or (1e-8 <= spec.top_p < 1.0) or nc % 8 or nc > 16384:
                return None
            top_k, model = min(spec.top_k, V), (ops.MODEL_ANOLE if self.model_type == "anole" else ops.MODEL_PLAIN)
        pk = self._packed_weight("head", w, lo, nc) if w.shape[1] % 64 == 0 else None
        return dict(row_lo=lo, n_cols=nc, model=model, top_k_filter=top_k, packed=pk)

    def _depth_plan(self, head, proc, k):
        """The DraftPlan of this model / head / processor list (cached), or None when the depth loop has to stay in Python: injected or torch layers,
        more than one layer, a head / processor list outside the fused expansion, more tree keys than one ancestor word holds."""
        from .decoder_layer import DecoderLayer
        if not self.use_depth_plan or len(self.layers) != 1 or not isinstance(self.layers[0], DecoderLayer):
            return None
        layer, w = self.layers[0], self.fc.weight
        if not (layer.fused and layer.inplace_cache and w.is_cuda and w.dtype == torch.bfloat16 and layer.self_attn.q_proj.weight.dtype == torch.bfloat16
                and layer.self_attn.head_dim in (64, 128) and w.shape[0] % 64 == 0 and 2 * k <= 32 and (self.depth + 1) * k <= DraftPlan.ROWS
                and self.depth >= 1 and getattr(layer.mlp, "act_fn", None) is torch.nn.functional.silu):
            return None
        hx = self._head_fusion(head, proc, k, k, True)
        if hx is None:
            return None
        # every tensor the plan snapshots a pointer (or a packed / converted copy) of: an in-place update of any of them -- or a new storage -- rebuilds it
        snap = [head.weight, head.bias, w, self.fc.bias, self.embed_tokens.weight] + [p_ for p_ in layer.parameters()]
        key = (id(head), k, self.depth, hx["top_k_filter"], hx["model"]) + tuple((t.data_ptr(), t._version) for t in snap if t is not None)
        hit = self.__dict__.get("_plan")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_plan"] = (key, DraftPlan(self, head, k, hx))
        return hit[1]

    def _static_plan(self, head, proc, k):
        """The StaticDraftPlan of this model / head / processor list / tree (cached), or None when the static loop has to stay in Python (the same
        conditions as _depth_plan; levels of at most 16 rows, at most 64 tree nodes)."""
        from .decoder_layer import DecoderLayer
        tb = getattr(self, "tree_buffer", None)
        if not self.use_depth_plan or tb is None or len(self.layers) != 1 or not isinstance(self.layers[0], DecoderLayer):
            return None
        layer, w = self.layers[0], self.fc.weight
        levels = [int(len(t)) for t in tb["tree_indices"]]
        if not (layer.fused and layer.inplace_cache and w.is_cuda and w.dtype == torch.bfloat16 and layer.self_attn.q_proj.weight.dtype == torch.bfloat16
                and layer.self_attn.head_dim in (64, 128) and w.shape[0] % 64 == 0 and levels and max(levels) <= 16 and sum(levels) <= DraftPlan.ROWS
                and 1 <= k <= 16 and getattr(layer.mlp, "act_fn", None) is torch.nn.functional.silu and self.vocab_size % 4 == 0):
            return None
        hx = self._head_fusion(head, proc, max(levels), k, True)
        if hx is None or hx["packed"] is None:
            return None
        snap = [head.weight, head.bias, w, self.fc.bias, self.embed_tokens.weight] + [p_ for p_ in layer.parameters()]
        key = (id(head), k, id(tb), hx["top_k_filter"], hx["model"]) + tuple((t.data_ptr(), t._version) for t in snap if t is not None)
        hit = self.__dict__.get("_splan")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_splan"] = (key, StaticDraftPlan(self, head, k, hx, tb))
        return hit[1]

    use_depth_plan = True          # False: the depth loop stays in Python (one ctypes call per kernel): the form the plan is tested against
    static_draw_uniforms = None    # callable (rows, k) -> [rows, k] f64 device tensor: the uniforms behind the static plan's draws (default: torch.rand)
    static_draw_idx = None         # [rows, k] int64: the draws themselves, injected (tests, recorded runs)

    def _post_head(self, cond, uncond, proc, pos_ids=None, pos_base=2):
        """CFG combine + the model's mask + its processors on the head's rows -> processed logits [R,V] f32 (dense rows: the tree
        ops and the verify side index them by token id)."""
        V = cond.shape[-1]
        cond, uncond = cond.reshape(-1, V).contiguous(), uncond.reshape(-1, V).contiguous()
        if self.model_type == "lumina_mgpt":
            # MultiModalLogitsProcessor (position-dependent) + InterleavedTopKLogitsWarper, cnets_lumina_mgpt.py:1216-1224,1291-1298
            top_k = min(int(proc[1].image_top_k), V) if (proc is not None and len(proc) > 1) else 0
            if pos_ids is None:
                return ops.cfg_mask_topk(cond, uncond, float(self.cfg_scale), model=ops.MODEL_PLAIN, top_k=top_k)
            return ops.cfg_mask_topk(cond, uncond, float(self.cfg_scale), model=ops.MODEL_LUMINA, pos_ids=pos_ids.reshape(-1),
                                     pos_base=pos_base, img_lo=self.image_lo, img_hi=self.image_hi, top_k=top_k)
        from ..verify import ProcessorSpec
        spec = ProcessorSpec.from_hf(proc) or ProcessorSpec()
        if self.model_type == "anole":                         # non-image ids -> finfo.min (cnets_anole.py:837,878), then the HF list
            lo, W = self.image_lo, self.image_hi - self.image_lo
            win, _ = ops.cfg_mask_topk_window(cond, uncond, float(self.cfg_scale), lo, W, model=ops.MODEL_ANOLE, img_lo=lo, img_hi=lo + W,
                                              top_k=min(spec.top_k, V), temperature=spec.temperature, top_p=spec.top_p)
            out = torch.full((win.shape[0], V), float("-inf"), dtype=torch.float32, device=win.device)
            out[:, lo:lo + W] = win
            return out
        return ops.cfg_mask_topk_window(cond, uncond, float(self.cfg_scale), 0, V, model=ops.MODEL_PLAIN, top_k=min(spec.top_k, V),
                                        temperature=spec.temperature, top_p=spec.top_p)[0]

    def _finalize_dynamic_plan(self, plan, cu0, ti0, sample_token, sort_rows):
        """_finalize_dynamic on the plan's contiguous logs (no torch.cat of the per-depth lists)."""
        sc, tk, pa = plan.finalize_inputs(cu0, ti0)
        draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(sc, tk, pa, sample_token.reshape(-1)[:1], self.top_k, self.total_tokens, sort_rows=sort_rows)
        nl, md = int(nl[0]), int(md[0])
        return draft, ret[0, :nl, :md].contiguous(), mask[:, None], pos[0]

    def _finalize_dynamic(self, scores_list, ss_token, parents_list, sample_token, sort_rows):
        draft, mask, pos, ret, nl, md = ops.tree_dynamic_finalize(torch.cat(scores_list)[None], torch.cat(ss_token)[None],
                                                                  torch.cat(parents_list)[None], sample_token.reshape(-1)[:1], self.top_k,
                                                                  self.total_tokens, sort_rows=sort_rows)
        nl, md = int(nl[0]), int(md[0])
        return draft, ret[0, :nl, :md].contiguous(), mask[:, None], pos[0]

    # ------------------------------------------------------------------ LlamaGen / Anole: cnets_llamagen.py:732-912
    @torch.no_grad()
    def _prefill(self, hidden_states, input_ids, input_position_diff, attention_mask):
        """First drafter forward of a call: everything past the cached prefix.  Anole (`input_position_diff` given,
        cnets_anole.py:815-831) runs the uncond row at positions clamp(pos - diff, 0) and passes the padding mask."""
        dev = hidden_states.device
        kv_len = self.stable_kv[0][0].shape[2] if self.stable_kv is not None else 0
        pos = torch.arange(kv_len, input_ids.shape[1], device=dev)[None]
        kw = {}
        if input_position_diff is not None:
            pos = torch.cat([pos, torch.clamp(pos - input_position_diff, 0)])
            kw["attention_mask"] = attention_mask
        elif self.stable_kv is None:
            pos = None
        out_hidden, pkv = self(hidden_states, input_ids=input_ids[:, kv_len:], past_key_values=self.stable_kv, use_cache=True,
                               position_ids=pos, **kw)
        self.stable_kv = pkv
        return out_hidden, pkv

    def _first_visible_key(self, attention_mask, dev):
        """argmax of the (left-padded) mask per row: from the statistics the prefill's forward() just took of the same mask, else computed."""
        st = getattr(self, "_mask_stats", None)
        if st is not None and st.shape[1] == attention_mask.shape[0] and st.device == dev:
            return st[0]
        return attention_mask.to(dev).to(torch.int64).argmax(dim=1)

    def _tree_positions(self, len_posi, offsets, input_position_diff):
        pos = len_posi + offsets
        if input_position_diff is None:
            return pos
        pos = pos[None]
        return torch.cat([pos, pos - input_position_diff])          # no clamp inside the loop (cnets_anole.py:858-862)

    @torch.no_grad()
    def topK_genrate(self, hidden_states, input_ids, head, logits_processor, cfg_scale, input_position_diff=None, attention_mask=None):
        self.cfg_scale = cfg_scale
        dev = hidden_states.device
        input_ids = input_ids.to(dev)
        sample_token = input_ids[:, -1]
        input_ids = input_ids[:, 1:]
        len_posi = input_ids.shape[1]
        k = self.top_k
        self.reset()
        akw = {} if input_position_diff is None else {"attention_mask": attention_mask}
        out_hidden, pkv = self._prefill(hidden_states, input_ids, input_position_diff, attention_mask)
        last_hidden = out_hidden[:, -1]
        ti, cu, ci, scores = self._expand_depth(head, last_hidden, logits_processor, None, None, k)
        scores_list, ss_token = [cu.reshape(-1)], [ti.reshape(-1)]
        parents_list = [torch.zeros(1, dtype=torch.long, device=dev)]
        cur = ti.reshape(1, -1)
        input_ids = torch.cat([cur, cur])
        plan = self._depth_plan(head, logits_processor, k)
        input_hidden = last_hidden[:, None].expand(-1, k, -1) if plan is not None else last_hidden[:, None].repeat(1, k, 1)
        if plan is not None:          # the depth loop: one lantern_draft_depth call per depth
            # positions: depth i at len_posi + i, the unconditional row at - input_position_diff (no clamp inside the loop, cnets_anole.py:858-862)
            # (input_position_diff None: the reference does not hand the mask to its depth forwards either -- `akw` above, cnets_llamagen.py:783-790 --
            # so no padding is described; with it, the mask went through forward()'s left-padding check at the prefill)
            start = None if attention_mask is None or input_position_diff is None else self._first_visible_key(attention_mask, dev)
            plan.begin(pkv, input_hidden, cur.reshape(-1), scores, None, kv_start=start, len_posi=len_posi, position_diff=input_position_diff)
            for i in range(self.depth):
                plan.run(i)
            return self._finalize_dynamic_plan(plan, cu, ti, sample_token, logits_processor is not None)
        tree_mask = self.tree_mask_init
        cs = torch.arange(k, device=dev)
        for i in range(self.depth):
            self.tree_mask = tree_mask
            position_ids = self._tree_positions(len_posi, self.position_ids, input_position_diff)
            out_hidden, pkv = self(input_hidden, input_ids=input_ids, past_key_values=pkv, position_ids=position_ids, use_cache=True, **akw)
            len_posi += 1
            parents_list.append(cs + (1 + k * k * max(0, i - 1) + (k if i > 0 else 0)))
            ti, cu, ci, scores = self._expand_depth(head, out_hidden, logits_processor, None, scores, k)
            cs = ci[0]
            out_ids = cs // k
            input_hidden = out_hidden[:, out_ids]
            cur = ti.reshape(-1)[cs][None]
            input_ids = torch.cat([cur, cur])
            ss_token.append(ti.reshape(-1))
            scores_list.append(cu.reshape(-1))
            tree_mask = torch.cat((tree_mask[:, :, out_ids], self.tree_mask_init), dim=3)
        return self._finalize_dynamic(scores_list, ss_token, parents_list, sample_token, logits_processor is not None)

    # ------------------------------------------------------------------ LlamaGen / Anole static: cnets_llamagen.py:944-1023
    @torch.no_grad()
    def topK_genrate_v1(self, hidden_states, input_ids, head, logits_processor, cfg_scale, input_position_diff=None, attention_mask=None):
        self.cfg_scale = cfg_scale
        dev = hidden_states.device
        input_ids = input_ids[:, 1:].to(dev)
        ss_token, ss_prob, ss_op = [], [], []
        len_posi = input_ids.shape[1]
        self.reset()
        akw = {} if input_position_diff is None else {"attention_mask": attention_mask}
        out_hidden, pkv = self._prefill(hidden_states, input_ids, input_position_diff, attention_mask)
        plan = self._static_plan(head, logits_processor, TOPK) if out_hidden.shape[0] == 2 else None
        if plan is not None:          # the level loop: one lantern_draft_depth call per level (StaticDraftPlan)
            start = None if attention_mask is None or input_position_diff is None else self._first_visible_key(attention_mask, dev)
            return plan.draft(pkv, out_hidden[:, -1], len_posi, kv_start=start, position_diff=input_position_diff)
        ho = self._head(head, out_hidden[:, -1])
        half = ho.shape[0] // 2
        rows = self._post_head(ho[:half], ho[half:], logits_processor)
        tb = self.tree_buffer
        for i in range(len(tb["tree_indices"])):
            idx, prob, op = self.sample(rows, k=TOPK)
            ss_token.append(idx); ss_prob.append(prob); ss_op.append(op)
            sel = idx.view(-1)[tb["tree_indices"][i]]
            cur = sel[None]
            input_ids = torch.cat([cur, cur])
            input_hidden = self.repeat_hidden(out_hidden[:, -1:] if i == 0 else out_hidden, tb["repeat_nums"][i])
            self.tree_mask = tb["attn_mask"][i]
            position_ids = self._tree_positions(len_posi, tb["position_ids"][i], input_position_diff)
            out_hidden, pkv = self(input_hidden, input_ids=input_ids, past_key_values=pkv, position_ids=position_ids, use_cache=True, **akw)
            len_posi += 1
            ho = self._head(head, out_hidden)
            half = ho.shape[0] // 2
            rows = self._post_head(ho[:half], ho[half:], logits_processor)
        idx, prob, op = self.sample(rows, k=TOPK)
        ss_token.append(idx); ss_prob.append(prob); ss_op.append(op)
        return torch.cat(ss_token), torch.cat(ss_prob), ss_op

    # ------------------------------------------------------------------ Lumina-mGPT: cnets_lumina_mgpt.py:1148-1392
    @torch.no_grad()
    def topK_generate(self, hidden_states, uncond_hidden_states, input_ids, head, logits_processors, attention_mask=None,
                      tree_type="static"):
        assert uncond_hidden_states is not None, "uncond_hidden_states should not be None since we always use CFG."
        assert tree_type in ("static", "dynamic"), "tree_type should be 'static' for EAGLE v1 or 'dynamic' for EAGLE v2."
        dev = hidden_states.device
        input_ids = input_ids[:, 1:].to(dev)
        k = self.top_k
        if tree_type == "dynamic":
            sample_token = input_ids[:, -1]
        first = self.stable_kv is None
        if first:
            if hidden_states.shape[1] > uncond_hidden_states.shape[1]:          # sequential CFG: left-pad the uncond stream
                pad = torch.zeros((1, hidden_states.shape[1] - uncond_hidden_states.shape[1], uncond_hidden_states.shape[2]),
                                  dtype=uncond_hidden_states.dtype, device=dev)
                uncond_hidden_states = torch.cat((pad, uncond_hidden_states), dim=1)
            if tree_type == "dynamic" and self.tree_mask_init.shape[0] == 1:
                self.tree_mask_init = torch.cat([self.tree_mask_init, self.tree_mask_init], dim=0)
        hidden_states = torch.cat((hidden_states, uncond_hidden_states), dim=0)
        input_ids = input_ids.repeat(2, 1)
        attention_mask = attention_mask.to(dev)
        if attention_mask.shape[1] < input_ids.shape[1]:
            attention_mask = torch.nn.functional.pad(attention_mask, (0, input_ids.shape[1] - attention_mask.shape[1]), "constant", True)
        # position_ids = attention_mask.long().cumsum(-1) - 1; len_posi = position_ids[:, -1] + 1 (cnets_lumina_mgpt.py:1180-1186).  One launch gives the
        # first visible key, the count of visible keys and the left-padding flag per row; behind a cached prefix the new tokens' positions are then
        # (column - first key) -- what the cumsum says for a left-padded mask; a mask that is not raises through forward()'s deferred check
        stats = ops.mask_left_padding(attention_mask) if attention_mask.is_cuda else None
        self._pending_mask_stats = (attention_mask, stats) if stats is not None else None
        self.reset()
        if not first and stats is not None:
            kv_len = self.stable_kv[0][0].shape[2]
            S = attention_mask.shape[1]
            ar = self.__dict__.get("_col_ids")
            if ar is None or ar.numel() < S or ar.device != dev:
                ar = self.__dict__["_col_ids"] = torch.arange(max(2 * S, 4096), dtype=torch.long, device=dev)
            len_posi = stats[1][:, None]                                       # [2,1]: cond / uncond stream lengths
            out_hidden, pkv = self(hidden_states, input_ids[:, kv_len:], attention_mask=attention_mask, position_ids=ar[kv_len:S][None] - stats[0][:, None],
                                   past_key_values=self.stable_kv, use_cache=True)
        else:
            position_ids = attention_mask.long().cumsum(-1) - 1
            len_posi = (position_ids[:, -1] + 1)[:, None]
            if not first:
                kv_len = self.stable_kv[0][0].shape[2]
                out_hidden, pkv = self(hidden_states, input_ids[:, kv_len:], attention_mask=attention_mask, position_ids=position_ids[:, kv_len:],
                                       past_key_values=self.stable_kv, use_cache=True)
            else:
                out_hidden, pkv = self(hidden_states, input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, use_cache=True)
        self.stable_kv = pkv
        last_hidden = out_hidden[:, -1]
        if tree_type == "static":
            plan = self._static_plan(head, logits_processors, k)
            if plan is not None:      # the level loop: one lantern_draft_depth call per level (StaticDraftPlan)
                return plan.draft(pkv, last_hidden, len_posi, head_len=len_posi[1], kv_start=(stats[0] if stats is not None else attention_mask.to(torch.int64).argmax(dim=1)))
            ho = self._head(head, last_hidden)                                                  # [2,V]
            rows = self._post_head(ho[0:1], ho[1:2], logits_processors, pos_ids=len_posi[1])
            tb = self.tree_buffer
            ss_token, ss_prob, ss_op = [], [], []
            for i in range(len(tb["tree_indices"])):
                idx, prob, op = self.sample(rows, k=k)
                ss_token.append(idx); ss_prob.append(prob); ss_op.append(op)
                sel = idx.view(-1)[tb["tree_indices"][i]]
                input_ids = sel[None].repeat(2, 1)
                input_hidden = self.repeat_hidden(out_hidden[:, -1:] if i == 0 else out_hidden, tb["repeat_nums"][i])
                position_ids = len_posi + tb["position_ids"][i]
                tm = tb["attn_mask"][i]
                self.tree_mask = torch.cat((tm, tm), dim=0)
                out_hidden, pkv = self(input_hidden, input_ids=input_ids, attention_mask=attention_mask, past_key_values=pkv,
                                       position_ids=position_ids, use_cache=True)
                len_posi = len_posi + 1
                ho = self._head(head, out_hidden)
                rows = self._post_head(ho[0], ho[1], logits_processors, pos_ids=position_ids[1] + 1)
            idx, prob, op = self.sample(rows, k=k)
            ss_token.append(idx); ss_prob.append(prob); ss_op.append(op)
            return torch.cat(ss_token), torch.cat(ss_prob), ss_op
        # dynamic (EAGLE-2)
        ti, cu, ci, scores = self._expand_depth(head, last_hidden, logits_processors, len_posi[1], None, k)
        scores_list, ss_token = [cu.reshape(-1)], [ti.reshape(-1)]
        parents_list = [torch.zeros(1, dtype=torch.long, device=dev)]
        input_ids = ti.reshape(1, -1)
        plan = self._depth_plan(head, logits_processors, k)
        input_hidden = last_hidden[:, None].expand(-1, k, -1) if plan is not None else last_hidden[:, None].repeat(1, k, 1)
        if plan is not None:          # the depth loop: one lantern_draft_depth call per depth
            # [depth, 2, k] positions: the cond / uncond streams' lengths + the depth; the head's grammar positions follow the unconditional stream + 1
            plan.begin(pkv, input_hidden, input_ids.reshape(-1), scores, None, len_posi=len_posi, head_len=len_posi.reshape(-1)[1],
                       kv_start=(stats[0] if stats is not None else attention_mask.to(torch.int64).argmax(dim=1)))
            for i in range(self.depth):
                plan.run(i)
            return self._finalize_dynamic_plan(plan, cu, ti, sample_token, logits_processors is not None)
        tree_mask = self.tree_mask_init
        cs = torch.arange(k, device=dev)
        for i in range(self.depth):
            position_ids = len_posi + self.position_ids
            self.tree_mask = tree_mask
            out_hidden, pkv = self(input_hidden, input_ids=torch.cat((input_ids, input_ids), dim=0), attention_mask=attention_mask,
                                   past_key_values=pkv, position_ids=position_ids, use_cache=True)
            len_posi = len_posi + 1
            parents_list.append(cs + (1 + k * k * max(0, i - 1) + (k if i > 0 else 0)))
            ti, cu, ci, scores = self._expand_depth(head, out_hidden, logits_processors, position_ids[1] + 1, scores, k)
            cs = ci[0]
            out_ids = cs // k
            input_hidden = out_hidden[:, out_ids]
            input_ids = ti.reshape(-1)[cs][None]
            ss_token.append(ti.reshape(-1))
            scores_list.append(cu.reshape(-1))
            tree_mask = torch.cat((tree_mask[:, :, out_ids], self.tree_mask_init), dim=-1)
        return self._finalize_dynamic(scores_list, ss_token, parents_list, sample_token, logits_processors is not None)
