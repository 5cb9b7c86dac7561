"""Host mirror of models/ea_model_llamagen.py (and, through two class attributes, of
models/ea_model_anole.py) for the verify/accept path.

`EaModel` keeps the reference's method names and argument order: `generate_tree_buffers`,
`generate_candidates`, `tree_decoding`, `evaluate_posterior` (dynamic tree), `evaluate_posterior_v1`
(static tree), `update_inference_inputs`, `generate` (alias `eagenerate`).  The base model, its T5
text encoder and the drafter network are supplied by the caller.

Reference: models/ea_model_llamagen.py:26-29 (CFG), :283-420, :423-461, :464-669, :676-787,
:908-999, :1002-1170; models/ea_model_anole.py same structure (+ image-token offset 4,
`non_image_tokens` masking :931, cond/uncond position ids :915-918).
"""
from __future__ import annotations

import time
import types
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .drafters.choices import mc_sim_7b_63, naive_extend_57  # noqa: F401
from .drafters.kv_cache import initialize_past_key_values
from .verify import (NodeLogits, ProcessorSpec, WindowRows, UniformFifo, as_rows, concat_original_prob, generate_tree_buffers,
                     prepare_logits_processor)


def cfg_logit_process(combined_logits, cfg_scale=4.0):
    """uncond + (cond - uncond) * cfg_scale on the [cond; uncond] batch (ea_model_llamagen.py:26-29), HIP."""
    half = len(combined_logits) // 2
    cond, uncond = combined_logits[:half], combined_logits[half:]
    out = ops.cfg_mask_topk(cond, uncond, float(cfg_scale), model=ops.MODEL_PLAIN)
    return out.to(combined_logits.dtype)


class EaModel(nn.Module):
    # per-model constants: LlamaGen has V == K and no offset
    image_token_offset = 0
    image_lo, image_hi = 0, 2 ** 31 - 1
    mask_non_image = False
    uniform_window = 4096            # uniforms staged per refill (verify.UniformFifo)
    # "window": inside generate() the tree rows leave tree_decoding as probabilities over the image-token window with the HF
    # processors (Temperature -> TopP -> TopK) already applied to every row (one workgroup per row), and evaluate_posterior keeps
    # the residual in LDS.  "dense": full-vocabulary logit rows, processors (top-p included) inside evaluate_posterior per visited row (the
    # reference's order).  Greedy decoding always takes the dense rows (it returns raw logits).
    kernel_set = "window"
    _active_proc = None              # ProcessorSpec of the running generate() call (tree_decoding has no processor argument)
    prefix_pad = 120                 # input_ids carries 120 leading zero ids (ea_model_llamagen.py:437,1107)

    def __init__(self, base_model, ea_layer, nearest_latents):
        super().__init__()
        self.base_model = base_model
        self.ea_layer = ea_layer
        self.config = getattr(base_model, "config", None)
        dev = base_model.lm_head.weight.device
        if isinstance(nearest_latents, np.ndarray):
            nearest_latents = torch.from_numpy(np.ascontiguousarray(nearest_latents.astype(np.uint16)).view(np.int16))
        self.nearest_latents = nearest_latents.to(dev)
        self.vocab_size = base_model.lm_head.weight.shape[0]
        self._fifo: Optional[UniformFifo] = None

    def forward(self, cond_idx=None, input_ids=None, attention_mask=None, past_key_values=None, output_orig=False,
                position_ids=None):
        with torch.inference_mode():
            outputs = self.base_model.model(cond_idx=cond_idx, input_ids=input_ids, attention_mask=attention_mask,
                                            past_key_values=past_key_values, position_ids=position_ids)
            if output_orig:
                orig = self.base_model.lm_head(outputs[0])
            hidden_states = outputs[0]
        return (outputs, orig, hidden_states) if output_orig else (outputs, hidden_states)

    def generate_tree_buffers(self, tree_choices, device="cuda"):
        return generate_tree_buffers(tree_choices, device=device)

    def reset_tree_mode(self):
        self.base_model.model.tree_mode = True
        self.base_model.model.tree_mask = None

    def _uniforms(self) -> UniformFifo:
        if self._fifo is None:
            self._fifo = UniformFifo(self.nearest_latents.device, window=self.uniform_window)
        return self._fifo

    def _ep_config(self, static: bool, proc: ProcessorSpec, lantern, lantern_k, lantern_delta) -> ops.EpConfig:
        return ops.EpConfig(mode=ops.MODE_STATIC_LG if static else ops.MODE_DYNAMIC, tok_offset=self.image_token_offset,
                            img_lo=self.image_lo, img_hi=self.image_hi, lantern=bool(lantern), k=int(lantern_k),
                            delta=float(lantern_delta), temperature=proc.temperature, top_p=proc.top_p, top_k=proc.top_k)

    # ------------------------------------------------------------------ O6, :676-706
    def generate_candidates(self, tree_logits, tree_indices, retrieve_indices, sample_token, logits_processor):
        dev = tree_indices.device
        prob = tree_logits[1].to(dev).float()[None] if logits_processor is not None else None
        cand, cprob, tcand = ops.gather_candidates(tree_logits[0].to(dev)[None], prob, sample_token.to(dev).reshape(-1)[:1],
                                                   tree_indices, retrieve_indices)
        return cand[0], (cprob[0] if cprob is not None else None), tcand

    # ------------------------------------------------------------------ :908-932
    def _tree_forward(self, tree_candidates, past_key_values, tree_position_ids, input_ids, attention_mask=None, input_position_diff=0):
        """The target forward over the tree tokens (the first half of tree_decoding, ea_model_llamagen.py:908-925): (outputs, [2,N,V] logits, hidden)."""
        position_ids = tree_position_ids + input_ids.shape[1]
        if self.mask_non_image:      # Anole: separate cond / uncond position ids (ea_model_anole.py:915-918)
            position_ids = position_ids.unsqueeze(0)
            position_ids = torch.cat([position_ids, position_ids - input_position_diff], dim=0)
        if attention_mask is not None:
            remaining = input_ids.shape[1] + tree_candidates.shape[1] - attention_mask.shape[1]
            attention_mask = torch.cat([attention_mask, torch.ones((attention_mask.shape[0], remaining), dtype=torch.long,
                                                                    device=attention_mask.device)], dim=1)
        return self(input_ids=tree_candidates, output_orig=True, past_key_values=past_key_values, position_ids=position_ids, attention_mask=attention_mask)

    def tree_decoding(self, tree_candidates, past_key_values, tree_position_ids, input_ids, retrieve_indices, cfg_scale,
                      attention_mask=None, input_position_diff=0):
        position_ids = tree_position_ids + input_ids.shape[1]
        if self.mask_non_image:      # Anole: separate cond / uncond position ids (ea_model_anole.py:915-918)
            position_ids = position_ids.unsqueeze(0)
            position_ids = torch.cat([position_ids, position_ids - input_position_diff], dim=0)
        if attention_mask is not None:
            remaining = input_ids.shape[1] + tree_candidates.shape[1] - attention_mask.shape[1]
            attention_mask = torch.cat([attention_mask, torch.ones((attention_mask.shape[0], remaining), dtype=torch.long,
                                                                    device=attention_mask.device)], dim=1)
        outputs, tree_logits, hidden_state = self(input_ids=tree_candidates, output_orig=True, past_key_values=past_key_values,
                                                  position_ids=position_ids, attention_mask=attention_mask)
        half = tree_logits.shape[0] // 2
        proc = self._active_proc
        if self.kernel_set == "window" and proc is not None:
            V = tree_logits.shape[-1]
            lo, W = (self.image_lo, self.image_hi - self.image_lo) if self.mask_non_image else (0, V)
            win, hot = ops.cfg_mask_topk_window(tree_logits[0], tree_logits[half], float(cfg_scale), lo, W,
                                                model=ops.MODEL_ANOLE if self.mask_non_image else ops.MODEL_PLAIN,
                                                img_lo=self.image_lo if self.mask_non_image else 0,
                                                img_hi=self.image_hi if self.mask_non_image else V, top_k=min(proc.top_k, V),
                                                temperature=proc.temperature, top_p=proc.top_p, probs=True)
            wr = WindowRows(win, hot, retrieve_indices, V, lo)
            wr.dense_source = lambda: NodeLogits(self._dense_node_logits(tree_logits, half, cfg_scale), retrieve_indices)
            return wr, hidden_state, outputs
        return NodeLogits(self._dense_node_logits(tree_logits, half, cfg_scale), retrieve_indices), hidden_state, outputs

    def _dense_node_logits(self, tree_logits, half, cfg_scale):
        return ops.cfg_mask_topk(tree_logits[0], tree_logits[half], float(cfg_scale),
                                 model=ops.MODEL_ANOLE if self.mask_non_image else ops.MODEL_PLAIN,
                                 img_lo=self.image_lo if self.mask_non_image else 0,
                                 img_hi=self.image_hi if self.mask_non_image else tree_logits.shape[-1])

    # ------------------------------------------------------------------ O8 dynamic, :709-787 / greedy :789-905
    def evaluate_posterior(self, logits, candidates, logits_processor=None, lantern=False, lantern_k=1000, lantern_delta=0.1):
        rows, row_index = (None, None) if isinstance(logits, WindowRows) else as_rows(logits)
        if logits_processor is None:
            return self._evaluate_posterior_greedy(rows, row_index, candidates, lantern, lantern_k, lantern_delta)
        proc = ProcessorSpec.from_hf(logits_processor)
        cfg = self._ep_config(False, proc, lantern, lantern_k, lantern_delta)
        fifo = self._uniforms()
        fifo.reserve(candidates.shape[0] * candidates.shape[1])
        if isinstance(logits, WindowRows):
            return self._evaluate_posterior_window(logits, cfg, candidates, fifo, lantern, lantern_k, None)
        best, alen, sample_p, counters = ops.evaluate_posterior(cfg, rows.float()[None], row_index, candidates[None], fifo.buf,
                                                                table=self.nearest_latents if lantern else None,
                                                                cursor=fifo.cursor)
        ops.raise_on_status(counters)
        return best[0].to(torch.int64), int(alen[0]), sample_p[0]

    def _evaluate_posterior_window(self, logits, cfg, candidates, fifo, lantern, lantern_k, aux):
        import copy
        cfg_w = copy.copy(cfg)
        cfg_w.temperature, cfg_w.top_p, cfg_w.top_k = 1.0, 1.0, 0     # the rows are final probabilities (tree_decoding applied the processors)
        cur0 = fifo.cursor.clone()
        out = ops.evaluate_posterior_window(cfg_w, logits.V, logits.win[None], logits.win_lo, logits.row_index(), candidates[None], fifo.buf,
                                            row_hot=logits.row_hot[None], table=self._packed_table(int(lantern_k)) if lantern else None,
                                            aux=aux, cursor=fifo.cursor, want_dense=True, want_window=False, rows_probs=True)
        if int(out["counters"][0, 5]) in (2, 6, 7, 8) and logits.dense_source is not None:
            # a state only the dense kernel represents (the residual vanished, staging limits): the same step on the dense HIP
            # kernel -- processors applied per visited row there -- from the same position of the uniform stream
            fifo.cursor.copy_(cur0)
            rows, row_index = as_rows(logits.dense_rows())
            best, alen, sample_p, counters = ops.evaluate_posterior(cfg, rows.float()[None], row_index, candidates[None], fifo.buf,
                                                                    table=self.nearest_latents if lantern else None, aux=aux, cursor=fifo.cursor)
            ops.raise_on_status(counters)
            return best[0].to(torch.int64), int(alen[0]), sample_p[0]
        ops.raise_on_status(out["counters"])
        return out["best"][0].to(torch.int64), int(out["accept_len"][0]), out["sample_p"][0]

    def _packed_table(self, k: int) -> torch.Tensor:
        cols = -(-(k + 1) // 8) * 8
        if cols > min(1024, self.nearest_latents.shape[1]):
            return self.nearest_latents
        cache = self.__dict__.setdefault("_packed_tables", {})
        if cols not in cache:
            cache[cols] = ops.pack_vq_table(self.nearest_latents, cols)
        return cache[cols]

    # ------------------------------------------------------------------ O8 static, :464-669
    def evaluate_posterior_v1(self, logits, candidates, logits_processor, cart_candidates_prob, op, p_indices, tree_candidates,
                              b_indices, lantern=False, lantern_k=1000, lantern_delta=0.1):
        rows, row_index = (None, None) if isinstance(logits, WindowRows) else as_rows(logits)
        if logits_processor is None:
            return self._evaluate_posterior_greedy(rows, row_index, candidates, lantern, lantern_k, lantern_delta)
        proc = ProcessorSpec.from_hf(logits_processor)
        cfg = self._ep_config(True, proc, lantern, lantern_k, lantern_delta)
        hip = self.tree_buffers["_hip"]
        aux = ops.StaticAux(cart_prob=cart_candidates_prob.to(logits.device).float()[None], orig_prob=concat_original_prob(op),
                            op_off=hip["op_off"], p_idx=hip["p_idx"], b_off=hip["b_off"], b_idx=hip["b_idx"],
                            tree_cand=tree_candidates[:1].reshape(1, -1)[:, :hip["N"]])
        fifo = self._uniforms()
        fifo.reserve(candidates.shape[0] * candidates.shape[1])
        if isinstance(logits, WindowRows):
            return self._evaluate_posterior_window(logits, cfg, candidates, fifo, lantern, lantern_k, aux)
        best, alen, sample_p, counters = ops.evaluate_posterior(cfg, rows.float()[None], row_index, candidates[None], fifo.buf,
                                                                table=self.nearest_latents if lantern else None, aux=aux,
                                                                cursor=fifo.cursor)
        ops.raise_on_status(counters)
        return best[0].to(torch.int64), int(alen[0]), sample_p[0]

    def _evaluate_posterior_greedy(self, rows, row_index, candidates, lantern, lantern_k, lantern_delta):
        best, alen, out_row = ops.evaluate_posterior_greedy(rows.float()[None], row_index, candidates[None], lantern=bool(lantern),
                                                            k=int(lantern_k), delta=float(lantern_delta),
                                                            tok_offset=self.image_token_offset,
                                                            table=self.nearest_latents if lantern else None,
                                                            win_lo=self.image_lo if self.mask_non_image else 0,
                                                            win_len=(self.image_hi - self.image_lo) if self.mask_non_image else None)
        return best[0].to(torch.int64), alen[0].to(torch.int64), out_row[0]

    # ------------------------------------------------------------------ O9 + O10, :935-999
    def update_inference_inputs(self, input_ids, candidates, best_candidate, accept_length, retrieve_indices, logits_processor,
                                new_token, past_key_values_data_list, current_length_data, hidden_state_new, sample_p, cfg_scale,
                                input_position_diff=None, attention_mask=None, static_tree=False, u=None):
        """The reference's method, host-integer form (best_candidate / accept_length arrive as Python ints or 0-d tensors and are read
        here).  generate() does not come through here: its steps keep the verdict on the device (_verify_step)."""
        prev_input_len = input_ids.shape[1]
        n = int(accept_length) + 1
        dev = retrieve_indices.device
        best = torch.as_tensor([int(best_candidate)], dtype=torch.int32, device=dev)
        alen = torch.as_tensor([n - 1], dtype=torch.int32, device=dev)
        input_ids = torch.cat([input_ids, candidates[None, int(best_candidate), :n].to(input_ids.device)], dim=-1)
        for data in past_key_values_data_list:
            ops.kv_gather([data], torch.zeros(1, dtype=torch.int32, device=data.device),
                          torch.tensor([prev_input_len], dtype=torch.int64, device=data.device), retrieve_indices.to(data.device),
                          best.to(data.device), alen.to(data.device))
        current_length_data.fill_(prev_input_len + n)
        if u is None and logits_processor is not None:
            u = torch.rand(1, dtype=torch.float64, device=dev)
        out_h, _, token = ops.accept_gather(hidden_state_new[None], retrieve_indices, None, best, alen,
                                            sample_p=sample_p[None].float(), u=u)
        accept_hidden_state_new = out_h[0, :, :n]
        token = token.reshape(1, 1)
        out = self._draft_next(input_ids, accept_hidden_state_new, token, logits_processor, cfg_scale, input_position_diff, attention_mask, static_tree)
        new_token += n
        if static_tree:
            return input_ids, out, new_token, None, token
        return (input_ids, *out, new_token, None, token)

    def _draft_next(self, input_ids, accept_hidden, token, logits_processor, cfg_scale, input_position_diff, attention_mask, static_tree):
        """The drafter call that ends a step (ea_model_llamagen.py:984-999): static -> tree_logits, dynamic -> (draft_tokens,
        retrieve_indices, tree_mask, tree_position_ids)."""
        ea_input_ids = torch.cat((input_ids, token.to(input_ids.device)), dim=1).repeat(2, 1)
        kw = {}
        if self.mask_non_image:
            kw = dict(input_position_diff=input_position_diff, attention_mask=attention_mask)
        fn = self.ea_layer.topK_genrate_v1 if static_tree else self.ea_layer.topK_genrate
        return fn(accept_hidden, input_ids=ea_input_ids, head=self.base_model.lm_head, logits_processor=logits_processor, cfg_scale=cfg_scale, **kw)

    # ------------------------------------------------------------------ the decode driver of generate(), :1109-1169
    # Same observable behaviour as the reference's loop (tests/golden/generate_lg.npz pins ids, accept lengths, KV length, drafter and
    # target calls and both RNG positions), organised for the device: a step's verdict -- best path, accept length, status -- stays in
    # HBM and feeds the KV move, the accepted-hidden gather and the bonus-token draw there; the host reads ONE packed 4-int record per
    # step (it needs the accept length to slice the drafter's inputs), where the reference syncs on every tried candidate.
    _RETRY_DENSE = (2, 6, 7, 8)

    def _posterior_on_device(self, rows, candidates, logits_processor, aux, static, lantern, lantern_k, lantern_delta, u):
        """O8 (+ the bonus-token draw when the windowed kernel does it) with every result left on the device:
        dict(best [1] i32, accept_len [1] i32, status [1] i32, token [1] i64 or None, sample_p [V] f32 or None)."""
        dev = candidates.device
        if logits_processor is None:
            r, ri = as_rows(rows)
            best, alen, out_row = ops.evaluate_posterior_greedy(r.float()[None], ri, candidates[None], lantern=bool(lantern), k=int(lantern_k),
                                                                delta=float(lantern_delta), tok_offset=self.image_token_offset,
                                                                table=self.nearest_latents if lantern else None,
                                                                win_lo=self.image_lo if self.mask_non_image else 0,
                                                                win_len=(self.image_hi - self.image_lo) if self.mask_non_image else None)
            return dict(best=best, accept_len=alen, status=torch.zeros(1, dtype=torch.int32, device=dev), token=None, sample_p=out_row[0], counters=None)
        proc = ProcessorSpec.from_hf(logits_processor)
        cfg = self._ep_config(static, proc, lantern, lantern_k, lantern_delta)
        fifo = self._uniforms()
        if isinstance(rows, WindowRows):
            import copy
            cfg_w = copy.copy(cfg)
            cfg_w.temperature, cfg_w.top_p, cfg_w.top_k = 1.0, 1.0, 0     # the rows are final probabilities (tree_decoding applied the processors)
            out = ops.evaluate_posterior_window(cfg_w, rows.V, rows.win[None], rows.win_lo, rows.row_index(), candidates[None], fifo.buf,
                                                row_hot=rows.row_hot[None], table=self._packed_table(int(lantern_k)) if lantern else None,
                                                aux=aux, cursor=fifo.cursor, u_bonus=u, want_dense=False, want_window=False, rows_probs=True)
            return dict(best=out["best"], accept_len=out["accept_len"], status=out["counters"][:, 5], token=out["token"], sample_p=None,
                        counters=out["counters"])
        r, ri = as_rows(rows)
        best, alen, sample_p, counters = ops.evaluate_posterior(cfg, r.float()[None], ri, candidates[None], fifo.buf,
                                                                table=self.nearest_latents if lantern else None, aux=aux, cursor=fifo.cursor)
        return dict(best=best, accept_len=alen, status=counters[:, 5], token=None, sample_p=sample_p[0], counters=counters)

    # ------------------------------------------------------------------ the static-tree step through ONE lantern_verify_step call
    native_step = True          # False: every kernel its own ctypes call with fresh tensors (the form the one-call step is tested against)

    def _native_ctx(self, st, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta):
        """What a generate() call's static-tree steps share, built once (ea_model_lumina_mgpt.EaLumina_mGPT._native_ctx is the same idea): the
        lantern_step_group with preallocated outputs.  None when this configuration stays on the per-kernel path (dynamic trees, greedy decoding,
        the dense kernel set, a model spread over devices, slabs of different shapes)."""
        import ctypes as C
        from . import _lib
        greedy = logits_processor is None          # temperature <= 1e-5: the greedy / TVD accept as the call's O8 stage (lantern_step_greedy)
        if not (self.native_step and (self.kernel_set == "window" or greedy) and not st.multi_device):
            return None
        s0 = st.slabs[0]
        if any(x.shape != s0.shape or x.dtype != s0.dtype or x.device != s0.device or not x.is_contiguous() for x in st.slabs):
            return None
        dev = s0.device
        if st.static:
            tb, hip = self.tree_buffers, self.tree_buffers["_hip"]
            N, P, D, R = hip["N"], hip["P"], hip["D"], hip["R"]
        else:          # EAGLE-2: the drafter hands over a tree per step (token list, paths); P and D are that step's, N = its token count
            N, P, D, R = int(st.draft_tokens.shape[1]), 0, 0, 0
        V = self.vocab_size
        lo, W = (self.image_lo, self.image_hi - self.image_lo) if self.mask_non_image else (0, V)
        if D > 8 or N > 64 or W % 8 or W > 16384:
            return None
        proc = ProcessorSpec.from_hf(logits_processor) if not greedy else ProcessorSpec()
        nx = types.SimpleNamespace(C=C, L=_lib.lib(), dev=dev, N=N, P=P, D=D, R=R, lo=lo, W=W, static=bool(st.static), greedy=greedy)
        z = lambda *shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        nx.cand, nx.cart, nx.tcand = z(1, max(P, 1), max(D, 1), dt=torch.int64), z(1, max(P, 1), max(D, 1), dt=torch.float32), z(1, N, dt=torch.int64)
        nx.win, nx.hot = z(N, W, dt=torch.float32), z(N, dt=torch.int32)
        # the verdict record, double-buffered: best, accept_len, counters[6], bonus token (int64 at words 8-9)
        nx.recs = [z(16, dt=torch.int32), z(16, dt=torch.int32)]
        nx.toks = [r_[8:10].view(torch.int64) for r_ in nx.recs]
        nx.otok, nx.omass = z(1, dt=torch.int32), z(1, dt=torch.float32)
        nx.out_hs, nx.acc = [None, None], z(1, 8, dt=torch.int64)
        nx.stream = torch.cuda.current_stream().cuda_stream
        if st.static:
            nx.tree_indices = tb["tree_indices"].to(dev).contiguous()
            nx.retrieve = tb["retrieve_indices_head"].to(dev).contiguous()
            ri = nx.retrieve.clone()
            ri[ri < 0] += N
            nx.row_index = ri.to(torch.int32).contiguous()
        n_sl = len(st.slabs)
        nx.prev = [torch.zeros(n_sl, dtype=torch.int64, device=dev), torch.zeros(n_sl, dtype=torch.int64, device=dev)]
        nx.parity = 0
        g = nx.group = (_lib.StepGroup * 1)()
        a = g[0]
        if st.static:
            a.tree_indices, a.retrieve = nx.tree_indices.data_ptr(), nx.retrieve.data_ptr()
        a.B, a.n_flat, a.N, a.P, a.D = 1, 0, N, P, D          # (n_flat: set per step from the drafter's sample list; dynamic trees: P, D per step)
        a.tree_cand, a.cand, a.cart_prob = nx.tcand.data_ptr(), nx.cand.data_ptr(), nx.cart.data_ptr()
        a.V, a.cfg, a.model = V, float(cfg_scale), (ops.MODEL_ANOLE if self.mask_non_image else ops.MODEL_PLAIN)
        a.img_lo, a.img_hi = (self.image_lo, self.image_hi) if self.mask_non_image else (0, V)
        a.top_k, a.win_lo, a.win_len, a.out_kind = min(proc.top_k, V), lo, W, ops.ROWS_PROBS
        a.out_win, a.row_hot, a.temperature, a.top_p = nx.win.data_ptr(), nx.hot.data_ptr(), float(proc.temperature), float(proc.top_p)
        if greedy:
            # the dense f32 rows of the O7 stage, the accepted row, the scratch of the (path, depth) cells (sized for any tree the context takes)
            nx.g_logits, nx.g_out, nx.g_ok = z(max(N, 64), V, dt=torch.float32), z(1, V, dt=torch.float32), z(64 * 8, dt=torch.int32)
            nx.g_table = self.nearest_latents.contiguous() if lantern else None
            q = nx.gq = _lib.StepGreedy()
            q.logits, q.out_row, q.ok_scratch = nx.g_logits.data_ptr(), nx.g_out.data_ptr(), nx.g_ok.data_ptr()
            q.lantern, q.k, q.delta, q.tok_offset = int(bool(lantern)), int(lantern_k), float(lantern_delta), int(self.image_token_offset)
            if nx.g_table is not None:
                q.nn_table = nx.g_table.data_ptr()
                q.table_rows, q.table_cols = nx.g_table.shape
            q.win_lo, q.win_len = lo, W
            if st.static:
                q.row_index = nx.row_index.data_ptr()
            a.greedy = C.pointer(q)
            a.out_win, a.top_k = None, 0
        cfg = self._ep_config(bool(st.static), proc, lantern, lantern_k, lantern_delta)
        p = a.ep
        p.B, p.P, p.D, p.V, p.rows_per_seq = 1, P, D, V, N
        p.mode, p.syntax_shortcut, p.tok_offset = cfg.mode, int(cfg.syntax_shortcut), cfg.tok_offset
        p.img_lo, p.img_hi, p.n_syntax = cfg.img_lo, min(cfg.img_hi, 2 ** 31 - 1), len(cfg.syntax)
        for i, sx in enumerate(cfg.syntax):
            p.syntax[i] = int(sx)
        p.lantern, p.k, p.delta = int(cfg.lantern), int(cfg.k), float(cfg.delta)
        p.top_k, p.temperature, p.top_p = 0, 1.0, 1.0          # the rows are final probabilities (the processors ran in the O7 stage)
        fifo = None if greedy else self._uniforms()          # (greedy decoding draws nothing: the Python RNG stays where the reference leaves it)
        p.n_uniforms, p.R, p.N, p.row_index_per_seq = (0 if greedy else fifo.buf.shape[1]), R, N, 0
        nx.table = self._packed_table(int(lantern_k)) if lantern else None
        if nx.table is not None:
            p.table_rows, p.table_cols = nx.table.shape
        b = a.ep_buf
        b.logits = nx.win.data_ptr()
        if st.static:
            b.row_index, b.cand, b.cart_prob = nx.row_index.data_ptr(), nx.cand.data_ptr(), nx.cart.data_ptr()
            b.op_off, b.p_idx, b.b_off, b.b_idx = hip["op_off"].data_ptr(), hip["p_idx"].data_ptr(), hip["b_off"].data_ptr(), hip["b_idx"].data_ptr()
            b.tree_cand = nx.tcand.data_ptr()
        b.nn_table = nx.table.data_ptr() if nx.table is not None else None
        if not greedy:
            b.uniforms, b.cursor = fifo.buf.data_ptr(), fifo.cursor.data_ptr()
        w = a.ep_win
        w.win_lo, w.win_len, w.row_hot, w.rows_kind = lo, W, nx.hot.data_ptr(), ops.ROWS_PROBS
        w.orig_prob_stride, w.orig_prob_offset = V, lo
        w.out_tok, w.out_mass = nx.otok.data_ptr(), nx.omass.data_ptr()
        a.slab_ptrs, a.slab_seq = st.slab_ptrs.data_ptr(), st.slab_seq.data_ptr()
        S, d = s0.shape[-2], s0.shape[-1]
        a.n_slabs, a.elem_bytes, a.outer, a.S_max, a.d = n_sl, s0.element_size(), s0.numel() // (S * d), S, d
        a.accepted_tokens = nx.acc.data_ptr()
        a.hid_groups = 2
        return nx

    def _verify_step_native(self, st, nx, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta):
        """One static-tree step: generate_candidates (one call: the target forward needs the tree tokens), the forward, then ONE lantern_verify_step
        call -- Temperature -> TopP -> TopK + softmax of all rows, evaluate_posterior with the bonus draw, the KV /
        hidden / token commit (only where the walk reported no status) -- on preallocated buffers, and one host read of the verdict record.  Same
        kernels, same uniforms, same results as the per-kernel path (tests/test_gpu_generate_lg.py runs the reference-recorded cases through both)."""
        if not nx.static:
            return self._verify_step_native_dynamic(st, nx, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta)
        C, L, a = nx.C, nx.L, nx.group[0]
        tl = st.tree_logits
        ss_token = tl[0].to(nx.dev).contiguous()
        ss_prob = None
        if not nx.greedy:
            ss_prob = tl[1].to(nx.dev)
            ss_prob = (ss_prob if ss_prob.dtype == torch.float32 else ss_prob.float()).contiguous()
        sample = st.sample_token.to(nx.dev).reshape(-1)[:1].contiguous()
        par = nx.parity
        rec, tokbuf = nx.recs[par], nx.toks[par]
        a.ep_buf.best, a.ep_buf.accept_len, a.ep_buf.counters = rec.data_ptr(), rec.data_ptr() + 4, rec.data_ptr() + 8
        a.ep_win.token = tokbuf.data_ptr()
        a.stream, a.ss_token = nx.stream, None          # (ss_token NULL: lantern_verify_step takes the candidates as the call below leaves them)
        a.flags = ops._lib.STEP_CANDIDATES_READY
        ops.check(L.lantern_gather_candidates(C.c_void_p(ss_token.data_ptr()), C.c_void_p(ss_prob.data_ptr() if ss_prob is not None else None), C.c_void_p(sample.data_ptr()),
                                              C.c_void_p(a.tree_indices), C.c_void_p(a.retrieve), 1, ss_token.numel(), nx.N, nx.P, nx.D, C.c_void_p(a.tree_cand),
                                              C.c_void_p(a.cand), C.c_void_p(a.cart_prob if ss_prob is not None else None), C.c_void_p(nx.stream)), "gather_candidates")
        kw = dict(input_position_diff=st.input_position_diff) if self.mask_non_image else {}
        tree_candidates = torch.cat([nx.tcand, nx.tcand])
        _, tree_logits, hidden_new = self._tree_forward(tree_candidates, self.base_model.past_key_values, self.tree_buffers["tree_position_ids"], st.input_ids,
                                                        st.attention_mask, **kw)
        half = tree_logits.shape[0] // 2
        cl, ul = tree_logits[0], tree_logits[half]
        if cl.dtype not in (torch.bfloat16, torch.float32):
            cl, ul = cl.float(), ul.float()
        cl, ul = cl.contiguous(), ul.contiguous()
        a.cond, a.uncond, a.dtype = cl.data_ptr(), ul.data_ptr(), int(cl.dtype == torch.bfloat16)
        if not nx.greedy:
            orig = concat_original_prob(tl[2])
            a.ep_buf.orig_prob = orig.data_ptr()
        hid = hidden_new.contiguous()[None]                                                # [1, 2, N, H]
        out_h = nx.out_hs[par]
        if out_h is None or out_h.dtype != hid.dtype or out_h.shape[-1] != hid.shape[-1]:
            out_h = nx.out_hs[par] = torch.zeros((1, 2, nx.D, hid.shape[-1]), dtype=hid.dtype, device=nx.dev)
        a.hidden, a.out_hidden, a.hid_elem_bytes, a.H = hid.data_ptr(), out_h.data_ptr(), hid.element_size(), hid.shape[-1]
        if nx.greedy:          # no uniforms, no counters: the record's counter words stay zero
            a.ep_buf.counters = None
            nx.gq.token = tokbuf.data_ptr()
        else:
            fifo = self._uniforms()
            fifo.reserve(nx.P * nx.D)
            u = torch.rand(1, dtype=torch.float64, device=nx.dev)
            a.ep_win.u_bonus = u.data_ptr()
        prev = st.input_ids.shape[1]
        cur, nxt = nx.prev[par], nx.prev[par ^ 1]
        cur.fill_(prev)
        a.slab_prev, a.new_len = cur.data_ptr(), nxt.data_ptr()
        ops.check(L.lantern_verify_step(nx.group, 1), "verify_step")
        r = rec.tolist()                                                                   # the step's one host read
        best, alen, n_used, status, tok = r[0], r[1], r[5], r[7], r[8]
        nx.parity ^= 1
        if status != 0:
            if status in self._RETRY_DENSE:
                # a state only the dense kernel represents: nothing was committed (lantern_verify_step commits only walks without a status); the same
                # step on the dense HIP kernel, from the same uniforms, through the host-integer path (rare: once in millions of steps)
                fifo.cursor.sub_(n_used)
                hip = self.tree_buffers["_hip"]
                aux = ops.StaticAux(cart_prob=nx.cart, orig_prob=orig, op_off=hip["op_off"], p_idx=hip["p_idx"], b_off=hip["b_off"], b_idx=hip["b_idx"],
                                    tree_cand=nx.tcand[:, :hip["N"]])
                rows = NodeLogits(self._dense_node_logits(tree_logits, half, cfg_scale), nx.retrieve)
                ep = self._posterior_on_device(rows, nx.cand[0], logits_processor, aux, True, lantern, lantern_k, lantern_delta, u)
                ops.raise_on_status(ep["counters"])
                al, bst = int(ep["accept_len"][0]), int(ep["best"][0])
                out = self.update_inference_inputs(st.input_ids, nx.cand[0], bst, al, nx.retrieve, logits_processor, 0, st.slabs, self.base_model.current_length_data,
                                                   hidden_new, ep["sample_p"], cfg_scale, st.input_position_diff, st.attention_mask, True, u=u)
                st.input_ids = out[0]
                self._take_draft(st, out[1], out[-1])
                return bst, al
            ops.raise_on_status(rec[2:8].reshape(1, 6))
        n = alen + 1
        self.base_model.current_length_data.fill_(prev + n)
        st.input_ids = torch.cat([st.input_ids, nx.acc[:, :n].to(st.input_ids.device)], dim=-1)
        token = tokbuf.reshape(1, 1)
        self._take_draft(st, self._draft_next(st.input_ids, out_h[0, :, :n], token, logits_processor, cfg_scale, st.input_position_diff, st.attention_mask, True),
                         token)
        return best, alen

    def _verify_step_native_dynamic(self, st, nx, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta):
        """One EAGLE-2 step (the reference's default for LlamaGen / Anole): the tree arrives from the drafter with its token list, so the candidates are
        one gather here (ea_model_llamagen.py:1125-1131) and lantern_verify_step runs the row post-process, evaluate_posterior with the bonus draw and
        the commit on this step's paths (`retrieve` [P, D]; P and D change from step to step, the group's fields follow them).  None: a tree outside
        what the group holds (more rows than the context was built for, depth > 8) -- the caller takes the per-kernel path for this step."""
        C, L, a = nx.C, nx.L, nx.group[0]
        retrieve = st.retrieve_indices.to(nx.dev)
        if retrieve.dtype != torch.int64:
            retrieve = retrieve.to(torch.int64)
        retrieve = retrieve.contiguous()
        P, D = retrieve.shape
        N = int(st.draft_tokens.shape[1])
        if D > 8 or P < 1 or N < 1:
            return None
        if N > nx.win.shape[0] and not nx.greedy:          # (a drafter whose trees grow: the row buffers follow)
            nx.win = torch.zeros((N, nx.W), dtype=torch.float32, device=nx.dev)
            nx.hot = torch.zeros(N, dtype=torch.int32, device=nx.dev)
            a.out_win, a.row_hot, a.ep_buf.logits, a.ep_win.row_hot = nx.win.data_ptr(), nx.hot.data_ptr(), nx.win.data_ptr(), nx.hot.data_ptr()
        nx.N = N
        a.N, a.ep.rows_per_seq, a.ep.N = N, N, N
        self.base_model.model.tree_mask = st.tree_mask
        candidates = torch.cat((st.draft_tokens, st.padding), dim=1)[0, retrieve].to(nx.dev).contiguous()          # [P, D] i64, -1 behind a path's end
        row_index = torch.where(retrieve < 0, retrieve + nx.N, retrieve).to(torch.int32)
        kw = dict(input_position_diff=st.input_position_diff) if self.mask_non_image else {}
        _, tree_logits, hidden_new = self._tree_forward(torch.cat([st.draft_tokens, st.draft_tokens]), self.base_model.past_key_values, st.tree_position_ids,
                                                        st.input_ids, st.attention_mask, **kw)
        half = tree_logits.shape[0] // 2
        cl, ul = tree_logits[0], tree_logits[half]
        if cl.dtype not in (torch.bfloat16, torch.float32):
            cl, ul = cl.float(), ul.float()
        cl, ul = cl.contiguous(), ul.contiguous()
        par = nx.parity
        rec, tokbuf = nx.recs[par], nx.toks[par]
        a.stream, a.ss_token, a.retrieve = nx.stream, None, retrieve.data_ptr()
        a.flags = ops._lib.STEP_CANDIDATES_READY
        a.P, a.D, a.ep.P, a.ep.D = P, D, P, D
        a.cand = candidates.data_ptr()
        a.cond, a.uncond, a.dtype = cl.data_ptr(), ul.data_ptr(), int(cl.dtype == torch.bfloat16)
        b = a.ep_buf
        b.best, b.accept_len, b.counters = rec.data_ptr(), rec.data_ptr() + 4, rec.data_ptr() + 8
        b.row_index, b.cand = row_index.data_ptr(), candidates.data_ptr()
        a.ep_win.token = tokbuf.data_ptr()
        hid = hidden_new.contiguous()[None]                                                # [1, 2, N, H]
        flat = nx.out_hs[par]
        if flat is None or flat.dtype != hid.dtype or flat.numel() != 2 * 8 * hid.shape[-1]:
            flat = nx.out_hs[par] = torch.zeros(2 * 8 * hid.shape[-1], dtype=hid.dtype, device=nx.dev)
        out_h = flat[:2 * D * hid.shape[-1]].view(1, 2, D, hid.shape[-1])          # (the kernel's [B, G, D, H] layout for THIS step's depth)
        a.hidden, a.out_hidden, a.hid_elem_bytes, a.H = hid.data_ptr(), out_h.data_ptr(), hid.element_size(), hid.shape[-1]
        if nx.greedy:
            if N > nx.g_logits.shape[0] or P * max(D - 1, 1) > nx.g_ok.numel():
                return None
            b.counters = None
            nx.gq.token, nx.gq.row_index = tokbuf.data_ptr(), row_index.data_ptr()
        else:
            fifo = self._uniforms()
            fifo.reserve(P * D)
            u = torch.rand(1, dtype=torch.float64, device=nx.dev)
            a.ep_win.u_bonus = u.data_ptr()
        prev = st.input_ids.shape[1]
        cur, nxt = nx.prev[par], nx.prev[par ^ 1]
        cur.fill_(prev)
        a.slab_prev, a.new_len = cur.data_ptr(), nxt.data_ptr()
        ops.check(L.lantern_verify_step(nx.group, 1), "verify_step")
        r = rec.tolist()                                                                   # the step's one host read
        best, alen, n_used, status, tok = r[0], r[1], r[5], r[7], r[8]
        nx.parity ^= 1
        if status != 0:
            if status in self._RETRY_DENSE:
                fifo.cursor.sub_(n_used)          # nothing was committed: the same step on the dense HIP kernel, from the same uniforms (host-integer path)
                rows = NodeLogits(self._dense_node_logits(tree_logits, half, cfg_scale), retrieve)
                ep = self._posterior_on_device(rows, candidates, logits_processor, None, False, lantern, lantern_k, lantern_delta, u)
                ops.raise_on_status(ep["counters"])
                al, bst = int(ep["accept_len"][0]), int(ep["best"][0])
                out = self.update_inference_inputs(st.input_ids, candidates, bst, al, retrieve, logits_processor, 0, st.slabs, self.base_model.current_length_data,
                                                   hidden_new, ep["sample_p"], cfg_scale, st.input_position_diff, st.attention_mask, False, u=u)
                st.input_ids = out[0]
                self._take_draft(st, out[1:5], out[-1])
                return bst, al
            ops.raise_on_status(rec[2:8].reshape(1, 6))
        n = alen + 1
        self.base_model.current_length_data.fill_(prev + n)
        st.input_ids = torch.cat([st.input_ids, nx.acc[:, :n].to(st.input_ids.device)], dim=-1)
        token = tokbuf.reshape(1, 1)
        self._take_draft(st, self._draft_next(st.input_ids, out_h[0, :, :n], token, logits_processor, cfg_scale, st.input_position_diff, st.attention_mask, False),
                         token)
        return best, alen

    def _verify_step(self, st, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta):
        """One step of the loop: O6, target forward + O7 (tree_decoding), O8, O9 + O10, the bonus token, the next draft."""
        nx = getattr(st, "native", None)
        if nx is None and not getattr(st, "native_tried", False):
            st.native_tried = True
            nx = st.native = self._native_ctx(st, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta)
        if nx is not None:
            done = self._verify_step_native(st, nx, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta)
            if done is not None:
                return done
        pkv = self.base_model.past_key_values
        kw = dict(input_position_diff=st.input_position_diff) if self.mask_non_image else {}
        aux = None
        if st.static:
            tb = self.tree_buffers
            candidates, cart_prob, tree_candidates = self.generate_candidates(st.tree_logits, tb["tree_indices"], tb["retrieve_indices"],
                                                                              st.sample_token, logits_processor)
            tree_candidates = torch.cat([tree_candidates, tree_candidates])
            retrieve = tb["retrieve_indices_head"]
            rows, hidden_new, _ = self.tree_decoding(tree_candidates, pkv, tb["tree_position_ids"], st.input_ids, retrieve, cfg_scale,
                                                     st.attention_mask, **kw)
            if logits_processor is not None:
                hip = tb["_hip"]
                aux = ops.StaticAux(cart_prob=cart_prob.to(candidates.device).float()[None], orig_prob=concat_original_prob(st.tree_logits[2]),
                                    op_off=hip["op_off"], p_idx=hip["p_idx"], b_off=hip["b_off"], b_idx=hip["b_idx"],
                                    tree_cand=tree_candidates[:1].reshape(1, -1)[:, :hip["N"]])
        else:
            self.base_model.model.tree_mask = st.tree_mask
            retrieve = st.retrieve_indices
            rows, hidden_new, _ = self.tree_decoding(torch.cat([st.draft_tokens, st.draft_tokens]), pkv, st.tree_position_ids, st.input_ids,
                                                     retrieve, cfg_scale, st.attention_mask, **kw)
            candidates = torch.cat((st.draft_tokens, st.padding), dim=1)[0, retrieve]
        dev = candidates.device
        # ---- O8: uniforms staged first, then the bonus uniform (the order the host-integer path draws them in)
        u, cur0 = None, None
        if logits_processor is not None:
            fifo = self._uniforms()
            fifo.reserve(candidates.shape[0] * candidates.shape[1])
            cur0 = fifo.cursor.clone()
            u = torch.rand(1, dtype=torch.float64, device=dev)
        ep = self._posterior_on_device(rows, candidates, logits_processor, aux, st.static, lantern, lantern_k, lantern_delta, u)
        best, alen, status = ep["best"], ep["accept_len"], ep["status"]
        # ---- O9 + O10 from the device-side verdict; a failed walk (status != 0) commits nothing
        prev = st.input_ids.shape[1]
        alen_commit = torch.where(status == 0, alen, torch.full_like(alen, -1))
        slabs = st.slabs
        sdev = slabs[0].device
        prev_t = st.prev_buf.fill_(prev)
        _, out_h, acc = ops.update_inference_inputs(slabs[:1] if st.multi_device else slabs, st.slab_seq[:1] if st.multi_device else st.slab_seq,
                                                    prev_t[:1] if st.multi_device else prev_t, retrieve.to(sdev), best.to(sdev), alen_commit.to(sdev),
                                                    hidden_new[None], candidates[None], slab_ptrs=None if st.multi_device else st.slab_ptrs)
        for data in (slabs[1:] if st.multi_device else ()):            # a model spread over devices: the other devices' slabs, same verdict
            ops.kv_gather([data], torch.zeros(1, dtype=torch.int32, device=data.device), torch.tensor([prev], dtype=torch.int64, device=data.device),
                          retrieve.to(data.device), best.to(data.device), alen_commit.to(data.device))
        token = ep["token"]
        if token is None:
            _, _, token = ops.accept_gather(None, retrieve, None, best, alen, sample_p=ep["sample_p"][None].float(), u=u)
        # ---- the step's one host read
        a, bst, stt, tok = torch.cat((alen.to(torch.int64), best.to(torch.int64), status.to(torch.int64), token.to(torch.int64).reshape(-1)[:1])).tolist()
        if stt != 0:
            if stt in self._RETRY_DENSE and isinstance(rows, WindowRows) and rows.dense_source is not None:
                # a state only the dense kernel represents (the residual vanished, staging limits): the same step on the dense HIP kernel --
                # processors applied per visited row there -- from the same position of the uniform stream, through the host-integer path
                self._uniforms().cursor.copy_(cur0)
                ep = self._posterior_on_device(rows.dense_rows(), candidates, logits_processor, aux, st.static, lantern, lantern_k, lantern_delta, u)
                ops.raise_on_status(ep["counters"])
                a, bst = int(ep["accept_len"][0]), int(ep["best"][0])
                out = self.update_inference_inputs(st.input_ids, candidates, bst, a, retrieve, logits_processor, 0, st.slabs, self.base_model.current_length_data,
                                                   hidden_new, ep["sample_p"], cfg_scale, st.input_position_diff, st.attention_mask, st.static, u=u)
                st.input_ids = out[0]
                self._take_draft(st, out[1] if st.static else out[1:5], out[-1])
                return bst, a
            ops.raise_on_status(ep["counters"])
        n = a + 1
        self.base_model.current_length_data.fill_(prev + n)
        st.input_ids = torch.cat([st.input_ids, acc[:, :n].to(st.input_ids.device)], dim=-1)
        token = torch.full((1, 1), tok, dtype=torch.long, device=dev)
        self._take_draft(st, self._draft_next(st.input_ids, out_h[0, :, :n], token, logits_processor, cfg_scale, st.input_position_diff,
                                              st.attention_mask, st.static), token)
        return bst, a

    def _take_draft(self, st, draft, token):
        st.sample_token = token
        if st.static:
            st.tree_logits = draft
        else:
            st.draft_tokens, st.retrieve_indices, st.tree_mask, st.tree_position_ids = draft

    def _decode_loop(self, st, max_length, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta):
        """generate()'s loop (ea_model_llamagen.py:1109-1169): steps until more than max_length tokens are out."""
        slabs = list(self.base_model.past_key_values_data)
        st.slabs, st.multi_device = slabs, len({x.device for x in slabs}) > 1
        sdev = slabs[0].device
        st.slab_ptrs = torch.tensor([x.data_ptr() for x in slabs], dtype=torch.int64, device=sdev)
        st.slab_seq = torch.zeros(len(slabs), dtype=torch.int32, device=sdev)
        st.prev_buf = torch.zeros(len(slabs), dtype=torch.int64, device=sdev)
        st.padding = torch.full((1, 1), -1, dtype=torch.long, device=st.input_ids.device)
        st.new_token, accept_lengths, self.last_steps = 0, [], []
        self._uniforms().begin()          # this prompt's acceptance uniforms start at random's current position
        try:
            for _ in range(max_length):
                best, alen = self._verify_step(st, logits_processor, cfg_scale, lantern, lantern_k, lantern_delta)
                self.last_steps.append((best, alen))
                accept_lengths.append(alen + 1)
                st.new_token += alen + 1
                if st.new_token > max_length:
                    break
        finally:
            self._uniforms().end()        # unconsumed staged draws go back to the module-level stream (also when a step raises)
        return accept_lengths

    # ------------------------------------------------------------------ :423-461
    @torch.no_grad()
    def initialize_tree(self, cond_combined, past_key_values, logits_processor, cfg_scale, attention_mask=None, static_tree=False,
                        tree_attn_mask=None, tree_choices=None):
        outputs, orig, hidden_states = self(cond_idx=cond_combined, past_key_values=past_key_values, output_orig=True,
                                            attention_mask=attention_mask)
        logits = cfg_logit_process(orig[:, -1], cfg_scale)
        if logits_processor is not None:
            logits = logits_processor(None, logits)
            token = torch.multinomial(torch.nn.functional.softmax(logits.float(), dim=1), 1)
        else:
            token = torch.argmax(logits)[None, None]
        token = torch.cat([token, token], dim=0)
        zero_padding = torch.zeros((token.shape[0], self.prefix_pad), dtype=torch.long, device=token.device)
        input_ids = torch.cat((zero_padding, token.to(cond_combined.device)), dim=1)
        if static_tree:
            self.ea_layer.init_tree_v1(tree_choices)
            tree_logits = self.ea_layer.topK_genrate_v1(hidden_states, input_ids, self.base_model.lm_head, logits_processor, cfg_scale)
            self.base_model.model.tree_mask = tree_attn_mask
            return tree_logits, logits, token
        out = self.ea_layer.topK_genrate(hidden_states, input_ids, self.base_model.lm_head, logits_processor, cfg_scale)
        return (*out, orig, hidden_states, token)

    def initialize_tree_v1(self, cond_combined, tree_attn_mask, past_key_values, logits_processor, cfg_scale, attention_mask=None,
                           tree_choices=mc_sim_7b_63):
        """ea_model_llamagen.py:442-461 (same argument order)."""
        return self.initialize_tree(cond_combined, past_key_values, logits_processor, cfg_scale, attention_mask, static_tree=True,
                                    tree_attn_mask=tree_attn_mask, tree_choices=tree_choices)

    # ------------------------------------------------------------------ :1002-1170
    def _encode_prompt(self, prompt, cfg):
        """The prompt block of the reference's generate (models/ea_model_llamagen.py:1018-1056): T5 caption embeddings rotated from
        right- to left-padding, zeroed under the mask, the unconditional class embedding appended for CFG.  The T5 encoder itself
        stays the reference's (`base_model.t5_model.get_text_embeddings`); a base model that brings its own `encode_prompt(prompt,
        cfg) -> (cond_combined, attention_mask)` is used as is."""
        bm = self.base_model
        if hasattr(bm, "encode_prompt"):
            return bm.encode_prompt(prompt, cfg)
        emb, mask = bm.t5_model.get_text_embeddings(prompt)                  # [B,T,C], [B,T] (valid tokens first)
        T = emb.shape[1]
        valid = mask.sum(dim=-1).to(torch.long)                              # rotate every row so that its valid tokens come last
        src = (torch.arange(T, device=emb.device)[None, :] + valid[:, None]) % T
        rot = torch.gather(emb, 1, src[:, :, None].expand(-1, -1, emb.shape[2]))
        lmask = torch.flip(mask, dims=[-1])
        cond = rot * lmask[:, :, None]
        if cfg is not None:
            null = torch.zeros_like(cond) + bm.model.cls_embedding.uncond_embedding.to(cond.device)
            cond, lmask = torch.cat([cond, null]), torch.cat([lmask, lmask])
        return cond.to(bm.dtype), lmask

    def _check_processors(self, temperature, top_p):
        """Every sampling configuration of the reference's generate() is built in both kernel sets (nucleus filtering -- TopPLogitsWarper,
        drafters/utils.py:36-52 -- per row in lantern_cfg_mask_topk_window for "window", per visited row inside lantern_evaluate_posterior for
        "dense"); what is left to refuse is a value outside the warper's own domain."""
        if top_p is not None and not (0.0 <= top_p <= 1.0):
            from ._lib import LanternError
            raise LanternError(f"generate: top_p={top_p} outside [0, 1] (TopPLogitsWarper raises the same)")

    @torch.no_grad()
    def generate(self, prompt: Optional[List[str]] = None, max_length: Optional[int] = None, temperature: Optional[float] = None,
                 top_k: Optional[int] = None, top_p: Optional[float] = None, cfg: Optional[float] = None,
                 lantern: Optional[bool] = None, lantern_k: Optional[int] = None, lantern_delta: Optional[float] = None,
                 static_tree: Optional[bool] = None, tree_choices: Optional[List[List[int]]] = naive_extend_57, **model_kwargs):
        self._check_processors(temperature, top_p)
        cond_combined, attention_mask = self._encode_prompt(prompt, cfg)
        st = time.time()
        if not hasattr(self.base_model, "past_key_values"):
            (self.base_model.past_key_values, self.base_model.past_key_values_data,
             self.base_model.current_length_data) = initialize_past_key_values(self.base_model, 2)
        past_key_values = self.base_model.past_key_values
        past_key_values_data = self.base_model.past_key_values_data
        current_length_data = self.base_model.current_length_data
        current_length_data.zero_()
        dev = cond_combined.device
        padding = (torch.zeros(1, 1, dtype=torch.long) - 1).to(dev)
        self.ea_layer.reset_kv()
        logits_processor = prepare_logits_processor(temperature=temperature, top_k=top_k, top_p=top_p) if temperature > 1e-5 else None
        self._active_proc = ProcessorSpec.from_hf(logits_processor)
        if static_tree:
            if not (hasattr(self, "tree_choices") and self.tree_choices == tree_choices):
                self.tree_buffers = self.generate_tree_buffers(tree_choices, device=dev)
                self.tree_buffers["retrieve_indices_head"] = self.tree_buffers["retrieve_indices"]
                self.tree_choices = tree_choices
            tree_buffers = self.tree_buffers
        self.reset_tree_mode()
        if static_tree:
            tree_logits, logits, sample_token = self.initialize_tree(cond_combined, past_key_values, logits_processor, cfg,
                                                                     attention_mask, static_tree=True,
                                                                     tree_attn_mask=tree_buffers["tree_attn_mask"],
                                                                     tree_choices=tree_choices)
        else:
            draft_tokens, retrieve_indices, tree_mask, tree_position_ids, logits, hidden_state, sample_token = self.initialize_tree(
                cond_combined, past_key_values, logits_processor, cfg, attention_mask)
        input_ids = torch.zeros((cond_combined.shape[0] // (2 if cfg is not None else 1), self.prefix_pad), dtype=torch.long).to(dev)
        st_ = types.SimpleNamespace(input_ids=input_ids, static=bool(static_tree), attention_mask=attention_mask, input_position_diff=None)
        if static_tree:
            self._take_draft(st_, tree_logits, sample_token)
        else:
            self._take_draft(st_, (draft_tokens, retrieve_indices, tree_mask, tree_position_ids), sample_token)
        accept_length_list = self._decode_loop(st_, max_length, logits_processor, cfg, lantern, lantern_k, lantern_delta)
        return (st_.input_ids[:, self.prefix_pad:self.prefix_pad + max_length], sum(accept_length_list) / len(accept_length_list),
                time.time() - st)

    eagenerate = generate

    def decode_ids(self, ids):
        """Token ids -> image: the base model's VQGAN decoder (out of scope), as the reference delegates."""
        return self.base_model.decode_ids(ids)

    @classmethod
    def from_pretrained(cls, Type="LLaMA", base_model_path=None, ea_model_path=None, total_token=59, depth=4, top_k=10, threshold=1.0, **kwargs):
        """The reference's constructor surface (models.ea_model_llamagen.EaModel.from_pretrained, ea_model_llamagen.py:154-227): the reference's own loader
        reads the checkpoints (base model, drafter weights, vq_distances table); the loaded model is wrapped so that generate() /
        eagenerate() run this package's accept loop.  generate_images.py:127-128 constructs the model with exactly this call."""
        from .verify import reference_loader
        ref = reference_loader("models.ea_model_llamagen", "EaModel").from_pretrained(Type=Type, base_model_path=base_model_path, ea_model_path=ea_model_path,
                                                                 total_token=total_token, depth=depth, top_k=top_k, threshold=threshold, **kwargs)
        return cls.from_reference(ref)

    @classmethod
    def from_reference(cls, ref, **kw):
        """Wrap a model the reference's own `from_pretrained` loaded (checkpoint loading stays there): same base model, drafter
        and neighbour table, this package's accept loop."""
        return cls(ref.base_model, ref.ea_layer, ref.nearest_latents, **kw)
