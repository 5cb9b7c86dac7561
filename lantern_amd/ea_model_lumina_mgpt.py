"""Host mirror of models/ea_model_lumina_mgpt.py for the verify/accept path.

Same class / method names, argument meaning and return values as the reference
(`EaLumina_mGPT.generate` -- also exported as `eagenerate` -- `initialize_tree`,
`generate_candidates`, `tree_decoding`, `evaluate_posterior`, `update_inference_inputs`,
module-level `generate_tree_buffers`, `MultiModalLogitsProcessor`, `InterleavedTopKLogitsWarper`),
with every hot torch-op sequence / Python loop replaced by one HIP entry point.  The target model
and the drafter network are NOT re-implemented: `base_model` and `ea_layer` are whatever the caller
built (the reference's own modules).

Reference: models/ea_model_lumina_mgpt.py:25-112 (processors), :140-277 (tree buffers),
:458-1017 (spec-decode driver).
"""
from __future__ import annotations

from typing import Optional

import types

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .drafters.choices import mc_sim_7b_63
from .drafters.kv_cache import initialize_past_key_values
from .verify import WindowRows, NodeLogits, UniformFifo, as_rows, concat_original_prob, generate_tree_buffers  # noqa: F401

TOPK = 10
IMAGE_LO, IMAGE_HI = 4, 8196          # image tokens are 4..8195 (ea_model_lumina_mgpt.py:37,322-323)
SYNTAX_TOKENS = (8196, 8197, 8803, 8828)


class MultiModalLogitsProcessor:
    """Position-dependent Lumina mask (ea_model_lumina_mgpt.py:25-86) on the HIP kernel: grid
    position -> only image ids stay finite; row end -> only 8803; after the last row -> only 8196."""

    def __init__(self, image_next_line_token_id=8803, image_end_token_id=8196, voc_size=65536):
        self.image_next_line_token_id = image_next_line_token_id
        self.image_end_token_id = image_end_token_id
        self.voc_size = voc_size

    def __call__(self, scores, h_latent_dim=48, w_latent_dim=48, image_start_token_id_index=None, position_ids=None):
        if position_ids is None:
            return scores
        base = 2 if image_start_token_id_index is None else image_start_token_id_index + 1 + 2
        pos = position_ids.reshape(-1).to(torch.int64)
        out = ops.cfg_mask_topk(scores, None, 1.0, model=ops.MODEL_LUMINA, pos_ids=pos, pos_base=int(base), w=w_latent_dim,
                                h=h_latent_dim, img_lo=IMAGE_LO, img_hi=IMAGE_HI, newline_id=self.image_next_line_token_id,
                                eos_id=self.image_end_token_id, top_k=0)
        return out.to(scores.dtype)


class InterleavedTopKLogitsWarper:
    """`scores < kth-largest -> filter_value` (ea_model_lumina_mgpt.py:88-112) on the HIP kernel."""

    def __init__(self, image_top_k: int = 2000, filter_value: float = -float("Inf"), min_tokens_to_keep: int = 1):
        if not isinstance(image_top_k, int) or image_top_k <= 0:
            raise ValueError(f"`image_top_k` has to be a strictly positive integer, but is {image_top_k}")
        assert filter_value == -float("Inf"), "the kernel filters with -inf like the reference default"
        self.image_top_k = max(image_top_k, min_tokens_to_keep)
        self.filter_value = filter_value

    def __call__(self, scores):
        out = ops.cfg_mask_topk(scores, None, 1.0, model=ops.MODEL_PLAIN, top_k=min(self.image_top_k, scores.size(-1)))
        return out.to(scores.dtype)


class EaLumina_mGPT(nn.Module):
    uniform_window = 4096     # uniforms staged per refill (verify.UniformFifo)
    # "window": tree rows cross HBM as 8192-wide image-token windows of probabilities, evaluate_posterior keeps the residual
    # distribution in LDS.  "dense": full-vocabulary rows, the reference's intermediate tensors exactly.
    kernel_set = "window"

    def __init__(self, base_model, ea_layer, nearest_latents, cfg_mode: str = "sequential", eagle_version: int = 1,
                 dtype=torch.bfloat16):
        super().__init__()
        self.base_model = base_model
        self.ea_layer = ea_layer
        self.config = getattr(base_model, "config", None)
        self.dtype = dtype
        self.cfg_mode = cfg_mode
        self.eagle_version = eagle_version
        self.vocab_size = base_model.lm_head.weight.shape[0]
        self.hidden_size = base_model.lm_head.weight.shape[-1]
        dev = base_model.lm_head.weight.device
        # uint16 [K, K-1] table (ckpts/lumina_mgpt/vq_distances/top_8191_indices.npy, :321) staged once in HBM
        if isinstance(nearest_latents, np.ndarray):
            nearest_latents = torch.from_numpy(np.ascontiguousarray(nearest_latents.astype(np.uint16)).view(np.int16))
        self.nearest_latents = nearest_latents.to(dev)
        self.image_token_offset = 4
        self.image_tokens = torch.arange(IMAGE_LO, IMAGE_HI, device=dev)
        self.image_syntax_tokens = torch.tensor(SYNTAX_TOKENS, device=dev)
        self.image_start_token_id = 8197
        self.internal_logits_processors = [MultiModalLogitsProcessor()]
        self.drafter_logits_processors = [MultiModalLogitsProcessor()]
        self._fifo: Optional[UniformFifo] = None
        self.w_latent_dim = self.h_latent_dim = 48

    # ------------------------------------------------------------------ plumbing (unchanged semantics)
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, output_orig=False, position_ids=None):
        with torch.inference_mode():
            outputs = self.base_model.model(input_ids=input_ids, attention_mask=attention_mask, past_key_values=past_key_values,
                                            position_ids=position_ids)
            if output_orig:
                orig = self.base_model.lm_head(outputs[0])
            hidden_states = outputs[0]
        return (outputs, orig, hidden_states) if output_orig else (outputs, hidden_states)

    def reset_tree_mode(self):
        self.base_model.model.tree_mode = True
        self.base_model.model.tree_mask = None

    def _packed_table(self, k: int) -> torch.Tensor:
        """Neighbour table in the hot-path layout (16-byte aligned rows of ceil8(k+1) ids, lantern_pack_vq_table), built once per k;
        tables too narrow to pad (k + 1 > 1024 staged ids is the kernel's limit anyway) are passed through."""
        cols = -(-(k + 1) // 8) * 8
        if cols > min(1024, self.nearest_latents.shape[1]):
            return self.nearest_latents
        cache = self.__dict__.setdefault("_packed_tables", {})
        if cols not in cache:
            cache[cols] = ops.pack_vq_table(self.nearest_latents, cols)
        return cache[cols]

    def _uniforms(self) -> UniformFifo:
        if self._fifo is None:
            self._fifo = UniformFifo(self.nearest_latents.device, window=self.uniform_window)
        return self._fifo

    # ------------------------------------------------------------------ :458-522
    def initialize_tree(self, input_ids, past_key_values, logits_processors, attention_mask=None, tree_attn_mask=None):
        if self.cfg_mode == "parallel":
            _, logits, hidden_states = self(input_ids=input_ids, attention_mask=attention_mask, past_key_values=past_key_values,
                                            output_orig=True)
            logits, uncond_logits = torch.split(logits, [1, 1])
            hidden_states, uncond_hidden_states = torch.split(hidden_states, [1, 1])
            input_ids = input_ids[:-1]
        else:
            _, logits, hidden_states = self(input_ids=input_ids, past_key_values=past_key_values["cond"], output_orig=True)
            self.image_start_token_id_index = torch.where(input_ids[0] == 8197)[0][-1].item()
            uncond_input_ids = input_ids[:, self.image_start_token_id_index:]
            _, uncond_logits, uncond_hidden_states = self(input_ids=uncond_input_ids, past_key_values=past_key_values["uncond"],
                                                          output_orig=True)
        cfg_logits = uncond_logits[:, -1] + self.cfg_scale * (logits[:, -1] - uncond_logits[:, -1])
        for logits_processor in (logits_processors or [])[1:]:
            cfg_logits = logits_processor(input_ids, cfg_logits)
        probabilities = torch.nn.functional.softmax(cfg_logits.float(), dim=-1)
        token = torch.multinomial(probabilities, 1)
        input_ids = torch.cat((input_ids, token.to(input_ids.device)), dim=1)
        if self.eagle_version == 1:
            self.ea_layer.init_tree(self.tree_choices)
            self.base_model.model.tree_mask = tree_attn_mask
        else:
            self.ea_layer.init_tree()
        output = self.ea_layer.topK_generate(hidden_states=hidden_states, uncond_hidden_states=uncond_hidden_states,
                                             input_ids=input_ids, attention_mask=attention_mask, head=self.base_model.lm_head,
                                             logits_processors=self.drafter_logits_processors,
                                             tree_type="static" if self.eagle_version == 1 else "dynamic")
        return (output, token) if self.eagle_version == 1 else output

    # ------------------------------------------------------------------ O6, :525-554
    def generate_candidates(self, tree_logits, tree_indices, retrieve_indices, sample_token):
        ss_token, ss_prob = tree_logits[0], tree_logits[1]
        dev = tree_indices.device
        cand, cprob, tcand = ops.gather_candidates(ss_token.to(dev)[None], ss_prob.to(dev).float()[None],
                                                   sample_token.to(dev).reshape(-1)[:1], tree_indices, retrieve_indices)
        return cand[0], cprob[0], tcand        # tree_candidates keeps its leading batch dim of 1

    # ------------------------------------------------------------------ :556-608 (O7 replaces :597-607)
    def tree_decoding(self, tree_candidates, attention_mask, past_key_values, tree_position_ids, input_ids, retrieve_indices):
        tree_logits, uncond_tree_logits, hidden_states, uncond_hidden_states, position_ids = self._tree_forward(
            tree_candidates, attention_mask, past_key_values, tree_position_ids, input_ids)
        return self._tree_postprocess(tree_logits, uncond_tree_logits, hidden_states, uncond_hidden_states, position_ids, retrieve_indices)

    def _tree_forward(self, tree_candidates, attention_mask, past_key_values, tree_position_ids, input_ids):
        """The target model on the tree tokens (cond and uncond passes: ea_model_lumina_mgpt.py:556-595)."""
        position_ids = tree_position_ids + input_ids.shape[1]
        if self.cfg_mode == "parallel":
            tree_candidates = torch.cat((tree_candidates, tree_candidates), dim=0)
            position_ids = torch.cat((position_ids[None], position_ids[None] - self.image_start_token_id_index), dim=0)
            _, tree_logits, hidden_states = self(input_ids=tree_candidates, attention_mask=attention_mask, output_orig=True,
                                                 past_key_values=past_key_values, position_ids=position_ids)
            tree_logits, uncond_tree_logits = torch.split(tree_logits, [1, 1])
            hidden_states, uncond_hidden_states = torch.split(hidden_states, [1, 1])
            position_ids = position_ids[0]
        else:
            _, tree_logits, hidden_states = self(input_ids=tree_candidates, output_orig=True,
                                                 past_key_values=past_key_values["cond"], position_ids=position_ids)
            _, uncond_tree_logits, uncond_hidden_states = self(input_ids=tree_candidates, output_orig=True,
                                                               past_key_values=past_key_values["uncond"],
                                                               position_ids=position_ids - self.image_start_token_id_index)
        return tree_logits, uncond_tree_logits, hidden_states, uncond_hidden_states, position_ids

    def _tree_postprocess(self, tree_logits, uncond_tree_logits, hidden_states, uncond_hidden_states, position_ids, retrieve_indices):
        # one kernel: CFG combine + MultiModalLogitsProcessor + InterleavedTopKLogitsWarper; no [P,D,V] gather
        top_k = self.internal_logits_processors[1].image_top_k if len(self.internal_logits_processors) > 1 else 0
        kw = dict(model=ops.MODEL_LUMINA, pos_ids=(position_ids + 1).reshape(-1), pos_base=self.image_start_token_id_index + 3,
                  w=self.w_latent_dim, h=self.h_latent_dim, img_lo=IMAGE_LO, img_hi=IMAGE_HI, newline_id=8803, eos_id=8196,
                  top_k=min(top_k, tree_logits.shape[-1]))
        if self.kernel_set == "window":
            win, hot = ops.cfg_mask_topk_window(tree_logits[0], uncond_tree_logits[0], float(self.cfg_scale), IMAGE_LO,
                                                IMAGE_HI - IMAGE_LO, probs=True, **kw)
            wr = WindowRows(win, hot, retrieve_indices, tree_logits.shape[-1], IMAGE_LO)
            # what the dense kernel set would need for the same step (only built if the windowed kernel asks for it)
            wr.dense_source = lambda: NodeLogits(ops.cfg_mask_topk(tree_logits[0], uncond_tree_logits[0], float(self.cfg_scale), **kw), retrieve_indices)
            return wr, hidden_states, uncond_hidden_states
        node_logits = ops.cfg_mask_topk(tree_logits[0], uncond_tree_logits[0], float(self.cfg_scale), **kw)
        return NodeLogits(node_logits, retrieve_indices), hidden_states, uncond_hidden_states

    # ------------------------------------------------------------------ O8, :610-729
    def _ep_config(self, lantern, lantern_k, lantern_delta) -> ops.EpConfig:
        # the id ranges live in device tensors (as in the reference); read them back once per tensor, not once per step
        key = (id(self.image_tokens), id(self.image_syntax_tokens))
        if getattr(self, "_range_key", None) != key:
            self._range_key = key
            self._img_range = (int(self.image_tokens[0]), int(self.image_tokens[-1]) + 1)
            self._syntax = tuple(int(x) for x in self.image_syntax_tokens.tolist())
        cfg = ops.EpConfig.lumina(self.eagle_version == 1, lantern=bool(lantern), k=int(lantern_k), delta=float(lantern_delta))
        cfg.img_lo, cfg.img_hi = self._img_range
        cfg.syntax, cfg.tok_offset = self._syntax, self.image_token_offset
        return cfg

    # static trees, windowed rows: "nodes" = one workgroup per internal tree node + the walk (33 vs 39 us per verify step at one sequence),
    # "chain" = one workgroup per sequence.  A node launch that meets duplicate sibling tokens reports NEEDS_CHAIN and the step re-runs on the chain.
    ep_form = "nodes"

    def _posterior_on_device(self, logits, candidates, cart_candidates_prob, original_prob, tree_candidates, lantern, lantern_k,
                             lantern_delta, u_bonus=None, force_dense=False, force_chain=False, reserve=True):
        """evaluate_posterior with every result left on the device: dict(best, accept_len, counters, sample_p | None, token | None).
        Windowed rows + u_bonus: the bonus token is drawn inside the kernel and sample_p never exists.
        `reserve=False`: the caller has already reserved this step's draws (it snapshots the cursor for a retry, and the window
        must not be refilled -- shifted, cursor zeroed -- between that snapshot and the retry)."""
        static = self.eagle_version == 1
        windowed = isinstance(logits, WindowRows) and not force_dense
        if isinstance(logits, WindowRows) and force_dense:
            logits = logits.dense_rows()
        rows, row_index = (logits.win, logits.row_index()) if windowed else as_rows(logits)
        cfg = self._ep_config(lantern, lantern_k, lantern_delta)
        aux = None
        if static:
            hip = self.tree_buffers["_hip"]
            aux = ops.StaticAux(cart_prob=cart_candidates_prob.to(rows.device).float()[None], orig_prob=concat_original_prob(original_prob),
                                op_off=hip["op_off"], p_idx=hip["p_idx"], b_off=hip["b_off"], b_idx=hip["b_idx"],
                                tree_cand=tree_candidates.reshape(1, -1)[:, :hip["N"]])
        fifo = self._uniforms()
        if reserve:
            fifo.reserve(candidates.shape[0] * candidates.shape[1])
        if windowed:
            nodes = None
            if static and self.ep_form == "nodes" and not force_chain and (not lantern or int(lantern_k) + 1 <= 1024):
                nodes = self.tree_buffers["_hip"].get("nodes")
            out = ops.evaluate_posterior_window(cfg, logits.V, rows[None], logits.win_lo, row_index, candidates[None], fifo.buf,
                                                row_hot=logits.row_hot[None], table=self._packed_table(int(lantern_k)) if lantern else None,
                                                aux=aux, cursor=fifo.cursor, u_bonus=u_bonus, want_dense=u_bonus is None, want_window=False,
                                                rows_probs=True, nodes=nodes)
            return dict(best=out["best"], accept_len=out["accept_len"], counters=out["counters"], sample_p=out["sample_p"], token=out["token"])
        best, alen, sample_p, counters = ops.evaluate_posterior(cfg, rows.float()[None], row_index, candidates[None], fifo.buf,
                                                                table=self.nearest_latents if lantern else None, aux=aux, cursor=fifo.cursor)
        return dict(best=best, accept_len=alen, counters=counters, sample_p=sample_p, token=None)

    # statuses of the windowed kernel the dense kernel does not have (the residual vanished, uniforms / tree beyond its LDS staging)
    _RETRY_DENSE = (2, 6, 7, 8)

    def evaluate_posterior(self, logits, candidates, cart_candidates_prob=None, original_prob=None, p_indices=None,
                           tree_candidates=None, b_indices=None, do_sample=True, lantern=False, lantern_k=1000,
                           lantern_delta=0.1):
        if not do_sample:
            raise NotImplementedError("Greedy decoding is not implemented yet")   # same as the reference (:728-729)
        if self.eagle_version == 1:
            assert cart_candidates_prob is not None, "Cartesian candidate probabilities are required for EAGLE v1"
            assert original_prob is not None, "Original probabilities are required for EAGLE v1"
            assert tree_candidates is not None, "Tree candidates are required for EAGLE v1"
            assert p_indices is not None, "Parent indices are required for EAGLE v1"
            assert b_indices is not None, "B indices are required for EAGLE v1"
        fifo = self._uniforms()
        fifo.reserve(candidates.shape[0] * candidates.shape[1])          # (begins the window if need be) BEFORE the snapshot: a refill moves the cursor
        cur0 = fifo.cursor.clone()
        out = self._posterior_on_device(logits, candidates, cart_candidates_prob, original_prob, tree_candidates, lantern, lantern_k, lantern_delta,
                                        reserve=False)
        status = int(out["counters"][0, 5])          # host sync: the B=1 caller wants accept_length as a Python int anyway
        if status == 8 and isinstance(logits, WindowRows):          # NEEDS_CHAIN: duplicate sibling tokens -> the chain kernel, same uniforms
            fifo.cursor.copy_(cur0)
            out = self._posterior_on_device(logits, candidates, cart_candidates_prob, original_prob, tree_candidates, lantern, lantern_k, lantern_delta,
                                            force_chain=True, reserve=False)
            status = int(out["counters"][0, 5])
        if status in self._RETRY_DENSE and isinstance(logits, WindowRows):
            # the windowed kernel reported a state only the dense kernel represents: same step again on the dense HIP kernel,
            # from the same position of the uniform stream (HIP -> HIP; there is no CPU path)
            fifo.cursor.copy_(cur0)
            out = self._posterior_on_device(logits, candidates, cart_candidates_prob, original_prob, tree_candidates, lantern, lantern_k,
                                            lantern_delta, force_dense=True, reserve=False)
        self._last = (out["best"], out["accept_len"], out["counters"])          # device copies for update_inference_inputs (no re-upload)
        ops.raise_on_status(out["counters"])
        return out["best"][0].to(torch.int64), int(out["accept_len"][0]), out["sample_p"][0]

    # ------------------------------------------------------------------ O9 + O10, :731-799
    def update_inference_inputs(self, input_ids, attention_mask, candidates, best_candidate, accept_length, retrieve_indices,
                                do_sample, new_token, past_key_values_data, current_length_data, hidden_states_new,
                                uncond_hidden_states_new, sample_p):
        dev = retrieve_indices.device
        best = torch.as_tensor([int(best_candidate)], dtype=torch.int32, device=dev)
        alen = torch.as_tensor([int(accept_length)], dtype=torch.int32, device=dev)
        n = int(accept_length) + 1
        if self.cfg_mode == "parallel":
            prev_input_len = input_ids.shape[1]
            slabs = list(past_key_values_data)
            prevs = [prev_input_len] * len(slabs)
            lens = [current_length_data] * len(slabs)
        else:
            slabs, prevs, lens = [], [], []
            for key in ["cond", "uncond"]:
                prev = input_ids.shape[1] if key == "cond" else input_ids.shape[1] - self.image_start_token_id_index
                for data in past_key_values_data[key]:
                    slabs.append(data)
                    prevs.append(prev)
                    lens.append(current_length_data[key])
        # slabs of equal geometry go in one launch
        groups = {}
        for s, p, l in zip(slabs, prevs, lens):
            groups.setdefault((tuple(s.shape), s.dtype, s.device), []).append((s, p, l))
        for (_, _, sdev), items in groups.items():
            ops.kv_gather([s for s, _, _ in items], torch.zeros(len(items), dtype=torch.int32, device=sdev),
                          torch.tensor([p for _, p, _ in items], dtype=torch.int64, device=sdev), retrieve_indices.to(sdev),
                          best.to(sdev), alen.to(sdev))
            for _, p, l in items:
                l.fill_(p + n)
        acc = candidates[None, int(best_candidate), :n].to(input_ids.device)
        if self.cfg_mode == "parallel":
            input_ids = torch.cat([input_ids[None, 0], acc], dim=-1)
        else:
            input_ids = torch.cat([input_ids, acc], dim=-1)
        # accepted hidden states (cond, uncond) + bonus token in one launch
        hid = torch.stack([hidden_states_new[0], uncond_hidden_states_new[0]])[None]        # [1,2,N,H]
        u = self._bonus_uniform(dev) if do_sample else None
        out_h, _, token = ops.accept_gather(hid, retrieve_indices, None, best, alen, sample_p=sample_p[None].float(), u=u)
        accept_hidden_states_new = out_h[:, 0, :n]
        accept_uncond_hidden_states_new = out_h[:, 1, :n]
        token = token.reshape(1, 1)
        output = self.ea_layer.topK_generate(hidden_states=accept_hidden_states_new,
                                             uncond_hidden_states=accept_uncond_hidden_states_new,
                                             input_ids=torch.cat((input_ids, token.to(input_ids.device)), dim=-1),
                                             attention_mask=attention_mask, head=self.base_model.lm_head,
                                             logits_processors=self.drafter_logits_processors,
                                             tree_type="static" if self.eagle_version == 1 else "dynamic")
        new_token += n
        return input_ids, output, new_token, token

    # ------------------------------------------------------------------ :801-1017
    # The decode driver.  Same observable behaviour as the reference's generate() (tests/golden/generate.npz pins ids, accept
    # lengths, KV lengths, drafter calls and the RNG position), organised for the device: a step's results -- best path, accept
    # length, status, accepted tokens, bonus token, new KV lengths -- stay in HBM and feed the KV gather / hidden gather there;
    # the host reads ONE packed 5-int record per step (it needs the accept length to slice the drafter's inputs), where the
    # reference's loop syncs on every tried candidate.
    def _prepare_generation(self, input_ids, cfg_scale, top_k, drafter_top_k, tree_choices):
        self.cfg_scale = self.ea_layer.cfg_scale = cfg_scale
        self.internal_logits_processors = [self.internal_logits_processors[0], InterleavedTopKLogitsWarper(image_top_k=top_k)]
        self.drafter_logits_processors = [self.drafter_logits_processors[0], InterleavedTopKLogitsWarper(image_top_k=drafter_top_k or top_k)]
        self.eval()
        self.ea_layer.reset_kv()
        dev = self.base_model.lm_head.weight.device
        static, parallel = self.eagle_version == 1, self.cfg_mode == "parallel"
        if static and getattr(self, "tree_choices", None) != tree_choices:
            tb = generate_tree_buffers(tree_choices, device=dev)
            tb["retrieve_indices_head"] = tb["retrieve_indices"]
            if parallel:
                tb["tree_attn_mask"] = torch.cat((tb["tree_attn_mask"], tb["tree_attn_mask"]), dim=0)
            self.tree_buffers, self.tree_choices = tb, tree_choices
        if not hasattr(self, "past_key_values"):
            if parallel:
                self.past_key_values, self.past_key_values_data, self.current_length_data = initialize_past_key_values(self.base_model, batch_size=2)
            else:
                self.past_key_values, self.past_key_values_data, self.current_length_data = {}, {}, {}
                for key in ("cond", "uncond"):
                    self.past_key_values[key], self.past_key_values_data[key], self.current_length_data[key] = \
                        initialize_past_key_values(self.base_model)
        for cl in ([self.current_length_data] if parallel else self.current_length_data.values()):
            cl.zero_()
        ids = torch.cat((input_ids, torch.tensor([[8197, 8828, 8828]], dtype=torch.long, device=input_ids.device)), dim=-1)
        self.reset_tree_mode()
        self.image_start_token_id_index = int(torch.where(ids[0] == self.image_start_token_id)[0][-1])
        L, P0 = ids.shape[1], self.image_start_token_id_index
        attn = torch.ones((2, L), dtype=torch.bool, device=ids.device)
        attn[1, :P0] = False                       # the unconditional row does not see the prompt
        st = types.SimpleNamespace(input_ids=ids.repeat(2, 1) if parallel else ids, attn_mask=attn, input_len=L, new_token=0, accept_lengths=[],
                                   static=static, parallel=parallel)
        # the KV slabs of this model, flat, with the prompt offset of each (cond slabs see the whole sequence, uncond slabs the image part)
        if parallel:
            st.slabs, st.offsets, st.len_tensors = list(self.past_key_values_data), [0] * len(self.past_key_values_data), None
        else:
            st.slabs = list(self.past_key_values_data["cond"]) + list(self.past_key_values_data["uncond"])
            st.offsets = [0] * len(self.past_key_values_data["cond"]) + [P0] * len(self.past_key_values_data["uncond"])
        sdev = st.slabs[0].device
        st.slab_ptrs = torch.tensor([x.data_ptr() for x in st.slabs], dtype=torch.int64, device=sdev)
        st.slab_seq = torch.zeros(len(st.slabs), dtype=torch.int32, device=sdev)
        st.slab_off = torch.tensor(st.offsets, dtype=torch.int64, device=sdev)
        return st

    def _first_draft(self, st, logits_processors):
        if st.static:
            tb = self.tree_buffers
            st.tree_logits, st.sample_token = self.initialize_tree(input_ids=st.input_ids, attention_mask=st.attn_mask,
                                                                   tree_attn_mask=tb["tree_attn_mask"], past_key_values=self.past_key_values,
                                                                   logits_processors=logits_processors)
            st.tree_position_ids, st.retrieve_indices = tb["tree_position_ids"], tb["retrieve_indices_head"]
        else:
            self._take_dynamic_draft(st, self.initialize_tree(input_ids=st.input_ids, attention_mask=st.attn_mask,
                                                              past_key_values=self.past_key_values, logits_processors=logits_processors))

    def _take_dynamic_draft(self, st, output):
        st.tree_candidates, st.retrieve_indices, st.tree_mask, st.tree_position_ids = output
        if st.parallel:
            st.tree_mask = st.tree_mask.repeat(2, 1, 1, 1)

    # ------------------------------------------------------------------ the step through ONE lantern_verify_step call
    native_step = True          # False: every kernel its own ctypes call with fresh tensors (the form the native step is tested against)

    def _native_ctx(self, st, lantern, lantern_k, lantern_delta):
        """Everything of a generate() call that does not change from step to step, built once: the lantern_step_group with its preallocated outputs
        (candidates, processed rows, verdict record, accepted hidden states / tokens, the KV length double buffer) and the node tables' workspace.
        None when this configuration stays on the per-kernel path (dynamic trees, the dense kernel set, slabs of different shapes)."""
        import ctypes as C
        from . import _lib
        if not (self.native_step and st.static and self.kernel_set == "window"):
            return None
        tb, hip = self.tree_buffers, self.tree_buffers["_hip"]
        s0 = st.slabs[0]
        if any(x.shape != s0.shape or x.dtype != s0.dtype or x.device != s0.device or not x.is_contiguous() for x in st.slabs):
            return None
        dev = s0.device
        N, P, D, R = hip["N"], hip["P"], hip["D"], hip["R"]
        if D > 8 or N - 1 > 64:
            return None
        W, V = IMAGE_HI - IMAGE_LO, self.vocab_size
        nx = types.SimpleNamespace(C=C, L=_lib.lib(), dev=dev, N=N, P=P, D=D, R=R)
        z = lambda *shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        nx.cand, nx.cart, nx.tcand = z(1, P, D, dt=torch.int64), z(1, P, D, dt=torch.float32), z(1, N, dt=torch.int64)
        nx.win, nx.hot = z(N, W, dt=torch.float32), z(N, dt=torch.int32)
        # the verdict record, double-buffered (the drafter keeps the previous step's bonus token / hidden rows while this step's are written):
        # best, accept_len, counters[6], bonus token (int64 at words 8-9)
        nx.recs = [z(16, dt=torch.int32), z(16, dt=torch.int32)]
        nx.toks = [r_[8:10].view(torch.int64) for r_ in nx.recs]
        nx.otok, nx.omass = z(1, dt=torch.int32), z(1, dt=torch.float32)
        nx.out_hs, nx.acc = [None, None], z(1, D, dt=torch.int64)
        nx.stream = torch.cuda.current_stream().cuda_stream          # (generate() runs on one stream)
        nx.hid = None
        nx.pos1 = (tb["tree_position_ids"].to(dev) + 1).to(torch.int64).contiguous()
        nx.tree_indices, nx.retrieve = tb["tree_indices"].to(dev).contiguous(), tb["retrieve_indices"].to(dev).contiguous()
        ri = nx.retrieve.clone()
        ri[ri < 0] += N
        nx.row_index = ri.to(torch.int32).contiguous()
        n_sl = len(st.slabs)
        L0 = st.input_ids.shape[1]
        nx.lens = [(L0 - st.slab_off).to(dev).contiguous(), torch.zeros(n_sl, dtype=torch.int64, device=dev)]          # KV lengths: [this step, next step], swapped
        nx.seq_len = [torch.full((1,), L0, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)]
        nx.parity = 0
        g = nx.group = (_lib.StepGroup * 1)()
        a = g[0]
        a.tree_indices, a.retrieve = nx.tree_indices.data_ptr(), nx.retrieve.data_ptr()
        a.B, a.n_flat, a.N, a.P, a.D = 1, R * TOPK, N, P, D
        a.tree_cand, a.cand, a.cart_prob = nx.tcand.data_ptr(), nx.cand.data_ptr(), nx.cart.data_ptr()
        a.V, a.cfg, a.model = V, float(self.cfg_scale), ops.MODEL_LUMINA
        a.pos_ids, a.pos_base = nx.pos1.data_ptr(), self.image_start_token_id_index + 3
        a.w_latent, a.h_latent, a.img_lo, a.img_hi, a.newline_id, a.eos_id = self.w_latent_dim, self.h_latent_dim, IMAGE_LO, IMAGE_HI, 8803, 8196
        top_k = self.internal_logits_processors[1].image_top_k if len(self.internal_logits_processors) > 1 else 0
        a.top_k, a.win_lo, a.win_len, a.out_kind = min(top_k, V), IMAGE_LO, W, ops.ROWS_PROBS
        a.out_win, a.row_hot, a.temperature, a.top_p = nx.win.data_ptr(), nx.hot.data_ptr(), 1.0, 1.0
        cfg = self._ep_config(lantern, lantern_k, lantern_delta)
        p = a.ep
        p.B, p.P, p.D, p.V, p.rows_per_seq = 1, P, D, V, N
        p.mode, p.syntax_shortcut, p.tok_offset = cfg.mode, int(cfg.syntax_shortcut), cfg.tok_offset
        p.img_lo, p.img_hi, p.n_syntax = cfg.img_lo, min(cfg.img_hi, 2 ** 31 - 1), len(cfg.syntax)
        for i, sx in enumerate(cfg.syntax):
            p.syntax[i] = int(sx)
        p.lantern, p.k, p.delta = int(cfg.lantern), int(cfg.k), float(cfg.delta)
        p.top_k, p.temperature, p.top_p = 0, 1.0, 1.0
        fifo = self._uniforms()
        p.n_uniforms, p.R, p.N, p.row_index_per_seq = fifo.buf.shape[1], R, N, 0
        nx.table = self._packed_table(int(lantern_k)) if lantern else None
        if nx.table is not None:
            p.table_rows, p.table_cols = nx.table.shape
        b = a.ep_buf
        b.logits, b.row_index, b.cand, b.cart_prob = nx.win.data_ptr(), nx.row_index.data_ptr(), nx.cand.data_ptr(), nx.cart.data_ptr()
        b.op_off, b.p_idx, b.b_off, b.b_idx = hip["op_off"].data_ptr(), hip["p_idx"].data_ptr(), hip["b_off"].data_ptr(), hip["b_idx"].data_ptr()
        b.tree_cand, b.nn_table = nx.tcand.data_ptr(), (nx.table.data_ptr() if nx.table is not None else None)
        b.uniforms, b.cursor = fifo.buf.data_ptr(), fifo.cursor.data_ptr()
        w = a.ep_win
        w.win_lo, w.win_len, w.row_hot, w.rows_kind = IMAGE_LO, W, nx.hot.data_ptr(), ops.ROWS_PROBS
        w.orig_prob_stride, w.orig_prob_offset = V, IMAGE_LO
        w.out_tok, w.out_mass = nx.otok.data_ptr(), nx.omass.data_ptr()
        nodes = hip.get("nodes") if (self.ep_form == "nodes" and (not lantern or int(lantern_k) + 1 <= 1024)) else None
        nx.nodes_struct = None
        if nodes is not None:
            nbytes = int(nx.L.lantern_evaluate_posterior_nodes_workspace(C.byref(p), C.byref(w), nodes.n_internal, 0))
            nx.nodes_ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            nx.nodes_struct = nodes.struct(nx.nodes_ws.data_ptr(), nbytes, -1)
            a.nodes = C.pointer(nx.nodes_struct)
        a.slab_ptrs, a.slab_seq = st.slab_ptrs.data_ptr(), st.slab_seq.data_ptr()
        S, d = s0.shape[-2], s0.shape[-1]
        a.n_slabs, a.elem_bytes, a.outer, a.S_max, a.d = n_sl, s0.element_size(), s0.numel() // (S * d), S, d
        a.accepted_tokens = nx.acc.data_ptr()
        a.hid_groups = 2
        # ---- host work taken out of the step (round 5): the verdict where the host can poll it, a block of bonus uniforms, input_ids appended in place
        nx.verdict = nx.verdict_np = None
        try:
            nx.verdict = torch.zeros(16, dtype=torch.int32).pin_memory()          # evaluate_posterior writes the record here as well (ep_win.verdict_host)
            nx.verdict_np = nx.verdict.numpy()
            w.verdict_host = nx.verdict.data_ptr()
        except RuntimeError:
            nx.verdict = nx.verdict_np = None
        nx.ids_buf = None
        if st.input_ids.device == dev and st.input_ids.dim() == 2:
            cap = max(int(getattr(st, "max_length", 4096)), L0) + 2 * D + 64
            nx.ids_buf = torch.zeros((1, cap), dtype=torch.int64, device=dev)
            nx.ids_buf[:, :L0] = st.input_ids[:1]
            a.ids_buf, a.ids_stride = nx.ids_buf.data_ptr(), cap
        return nx

    _UB_BLOCK = 4096          # bonus-draw uniforms generated per torch.rand call (one per verify step before)

    def _bonus_uniform(self, dev):
        """The next bonus-draw uniform as a 1-element float64 view: every path of the mirror (the one-call step, the per-kernel step, the public
        update_inference_inputs) takes it from ONE block of `_UB_BLOCK` values per torch.rand call, restarted by generate() -- so a torch seed gives
        the same image whichever path runs (the one-call step used to draw blocks while the per-kernel path drew torch.rand(1) per step)."""
        ub = getattr(self, "_ub", None)
        if ub is None or self._ub_i >= ub.shape[0] or ub.device != dev:
            self._ub, self._ub_i = torch.rand(self._UB_BLOCK, dtype=torch.float64, device=dev), 0
            ub = self._ub
        u = ub[self._ub_i:self._ub_i + 1]
        self._ub_i += 1
        return u

    def _verify_step_native(self, st, nx, lantern, lantern_k, lantern_delta):
        """One verify step: generate_candidates (one call: the target forward needs the tree tokens), the two target forwards, then ONE
        lantern_verify_step call -- the tree_decoding post-process of all rows, evaluate_posterior with the bonus
        draw, the KV / hidden / token commit (only where the walk reported no status) -- on preallocated buffers, and one host read of the 40-byte
        verdict record.  Same kernels, same uniforms (`_bonus_uniform`: one block shared by every path), same results as the per-kernel path
        (tests/test_gpu_generate_ref.py runs both, with blocks of 1 and of 8 recorded uniforms)."""
        C, L, a = nx.C, nx.L, nx.group[0]
        tl = st.tree_logits
        dev = nx.dev
        # (host time is what this path costs at a batch of one: no-op conversions are skipped, constant argument objects are built once)
        ss_token = tl[0] if (tl[0].device == dev and tl[0].is_contiguous()) else tl[0].to(dev).contiguous()
        ss_prob = tl[1] if (tl[1].device == dev and tl[1].dtype == torch.float32 and tl[1].is_contiguous()) else tl[1].to(dev).float().contiguous()
        sample = st.sample_token
        sample = sample.reshape(-1) if (sample.device == dev and sample.is_contiguous()) else sample.to(dev).reshape(-1)[:1].contiguous()
        stream = nx.stream
        par = nx.parity
        rec, tokbuf = nx.recs[par], nx.toks[par]
        pc = nx.__dict__.get("_per_parity")
        if pc is None:
            pc = nx._per_parity = [(r_.data_ptr(), t_.data_ptr()) for r_, t_ in zip(nx.recs, nx.toks)]
            nx._gc_const = tuple(C.c_void_p(x) for x in (a.tree_indices, a.retrieve, a.tree_cand, a.cand, a.cart_prob, stream))
        rp, tp = pc[par]
        eb, ew = a.ep_buf, a.ep_win
        eb.best, eb.accept_len, eb.counters = rp, rp + 4, rp + 8
        ew.token = tp
        p_tok, p_prob, p_smp = ss_token.data_ptr(), ss_prob.data_ptr(), sample.data_ptr()
        a.stream, a.ss_token = stream, None          # (ss_token NULL: lantern_verify_step takes the candidates as this call leaves them -- no second O6 launch)
        a.flags = ops._lib.STEP_CANDIDATES_READY
        g = nx._gc_const
        ops.check(L.lantern_gather_candidates(C.c_void_p(p_tok), C.c_void_p(p_prob), C.c_void_p(p_smp), g[0], g[1], 1, a.n_flat, nx.N, nx.P, nx.D, g[2], g[3], g[4], g[5]),
                  "gather_candidates")
        tree_logits, uncond_logits, hidden, uhidden, _pos = self._tree_forward(nx.tcand, st.attn_mask, self.past_key_values, st.tree_position_ids, st.input_ids)
        cl, ul = tree_logits[0], uncond_logits[0]
        if cl.dtype != ul.dtype or cl.dtype not in (torch.bfloat16, torch.float32):
            cl, ul = cl.float(), ul.float()
        if not cl.is_contiguous():
            cl = cl.contiguous()
        if not ul.is_contiguous():
            ul = ul.contiguous()
        a.cond, a.uncond, a.dtype = cl.data_ptr(), ul.data_ptr(), int(cl.dtype == torch.bfloat16)
        # the drafter's distributions, level by level: one [R, V] f32 block (no copy when the levels already sit back to back)
        ol = tl[2]
        ptrs = [o.data_ptr() for o in ol]
        if (all(o.dtype == torch.float32 and o.is_contiguous() and o.device == dev for o in ol) and
                all(ptrs[i] + 4 * ol[i].numel() == ptrs[i + 1] for i in range(len(ol) - 1))):
            orig = ol
            eb.orig_prob = ptrs[0]
        else:
            orig = concat_original_prob(ol)
            eb.orig_prob = orig.data_ptr()
        h0, h1 = hidden[0], uhidden[0]
        if h0.is_contiguous() and h1.is_contiguous() and h0.dtype == h1.dtype and h0.shape == h1.shape and h0.device == dev and h1.device == dev:
            hid, a.hidden, a.hidden_uncond = h0, h0.data_ptr(), h1.data_ptr()          # the two passes' rows as two pointers: nothing is stacked
        else:
            hid = torch.stack((h0, h1))[None]                                              # [1, 2, N, H]
            a.hidden, a.hidden_uncond = hid.data_ptr(), None
        out_h = nx.out_hs[par]
        if out_h is None or out_h.dtype != hid.dtype or out_h.shape[-1] != hid.shape[-1]:
            out_h = nx.out_hs[par] = torch.zeros((1, 2, nx.D, hid.shape[-1]), dtype=hid.dtype, device=nx.dev)
        a.out_hidden, a.hid_elem_bytes, a.H = out_h.data_ptr(), hid.element_size(), hid.shape[-1]
        fifo = self._uniforms()
        fifo.reserve(nx.N - 1)                                                             # one uniform per tried candidate, every non-root node at most once
        u = self._bonus_uniform(dev)                                                       # the bonus draws' uniforms: one torch.rand per _UB_BLOCK steps
        ew.u_bonus = u.data_ptr()
        cur, nxt = nx.lens[par], nx.lens[par ^ 1]
        a.slab_prev, a.new_len, a.seq_len = cur.data_ptr(), nxt.data_ptr(), cur.data_ptr()          # (slab 0 is a cond slab at offset 0: its length is len(input_ids))
        Lcur = st.input_ids.shape[1]
        inplace_ids = nx.ids_buf is not None and Lcur + nx.D + 2 <= nx.ids_buf.shape[1]
        a.ids_buf, a.ids_len = (nx.ids_buf.data_ptr() if inplace_ids else None), (cur.data_ptr() if inplace_ids else None)
        vh = nx.verdict_np
        if vh is not None:
            vh[10] = 0
        ops.check(L.lantern_verify_step(nx.group, 1), "verify_step")
        # the step's one host read: the verdict record, polled in pinned memory where evaluate_posterior wrote it (visible as soon as the walk is
        # done, while the commit still runs) -- no copy, no stream synchronisation; `rec.tolist()` when there is no pinned record
        r = None
        if vh is not None:
            spins = 0
            while vh[10] == 0:
                spins += 1
                if spins > 2_000_000:          # (seconds: something is badly wrong -- take the synchronising read)
                    break
            if vh[10] != 0:
                r = vh[:10].tolist()
                r[8] = (r[8] & 0xffffffff) | (r[9] << 32)
        if r is None:
            r = rec.tolist()
        best, alen, n_used, status, tok = r[0], r[1], r[5], r[7], r[8]
        fifo.consumed(n_used)
        if status != 0:
            if status in self._RETRY_DENSE:
                # a state only the dense kernel represents: nothing was committed (lantern_verify_step commits only walks without a status); the
                # same step on the dense HIP kernel, from the same uniforms, through the host-int path (rare: once in millions of steps)
                fifo.cursor.sub_(n_used)
                rows = WindowRows(nx.win, nx.hot, st.retrieve_indices, self.vocab_size, IMAGE_LO)
                top_k = self.internal_logits_processors[1].image_top_k if len(self.internal_logits_processors) > 1 else 0
                kw = dict(model=ops.MODEL_LUMINA, pos_ids=(nx.pos1 + Lcur).reshape(-1), pos_base=self.image_start_token_id_index + 3, w=self.w_latent_dim,
                          h=self.h_latent_dim, img_lo=IMAGE_LO, img_hi=IMAGE_HI, newline_id=8803, eos_id=8196, top_k=min(top_k, self.vocab_size))
                rows.dense_source = lambda: NodeLogits(ops.cfg_mask_topk(cl, ul, float(self.cfg_scale), **kw), st.retrieve_indices)
                bc, al, sample_p = self.evaluate_posterior(rows.dense_rows(), nx.cand[0], nx.cart[0], tl[2], self.tree_buffers["p_indices"], nx.tcand,
                                                           self.tree_buffers["b_indices"], True, lantern, lantern_k, lantern_delta)
                hit = self._commit_from_host(st, nx.cand[0], bc, al, hidden, uhidden, sample_p, u, None)
                nxt.copy_(cur + (int(al) + 1))
                nx.parity ^= 1
                if nx.ids_buf is not None and st.input_ids.shape[1] <= nx.ids_buf.shape[1]:          # keep the in-place copy of input_ids in step
                    nx.ids_buf[:, :st.input_ids.shape[1]] = st.input_ids[:1]
                return hit
            ops.raise_on_status(rec[2:8].reshape(1, 6))
        n = alen + 1
        nx.parity ^= 1
        done = set()
        for clen, off in zip(self._length_tensors(st), st.offsets):          # (one length tensor per cache, shared by its slabs)
            if (id(clen), off) not in done:
                clen.fill_(Lcur - off + n)
                done.add((id(clen), off))
        if inplace_ids:          # the commit appended the accepted tokens and, behind them, the bonus token: views, no torch.cat
            st.input_ids = nx.ids_buf[:, :Lcur + n]
            self._draft_next(st, out_h[:, 0, :n], out_h[:, 1, :n], tokbuf.reshape(1, 1), ids_with_token=nx.ids_buf[:, :Lcur + n + 1])
        else:
            accepted = nx.acc[:, :n]
            if accepted.device != st.input_ids.device:
                accepted = accepted.to(st.input_ids.device)
            st.input_ids = torch.cat([st.input_ids[None, 0] if st.parallel else st.input_ids, accepted], dim=-1)
            self._draft_next(st, out_h[:, 0, :n], out_h[:, 1, :n], tokbuf.reshape(1, 1))      # (views of this parity's buffers: the next step writes the other pair)
        st.new_token += n
        st.accept_lengths.append(n)
        return False

    def _verify_step(self, st, lantern, lantern_k, lantern_delta, eos_token_ids):
        if eos_token_ids is None:
            nx = getattr(st, "native", None)
            if nx is None and not getattr(st, "native_tried", False):
                st.native_tried = True
                nx = st.native = self._native_ctx(st, lantern, lantern_k, lantern_delta)
            if nx is not None:
                return self._verify_step_native(st, nx, lantern, lantern_k, lantern_delta)
        dev = st.retrieve_indices.device
        # ---- O6 + target forward + O7
        if st.static:
            tb = self.tree_buffers
            candidates, cart_prob, tree_candidates = self.generate_candidates(tree_logits=st.tree_logits, tree_indices=tb["tree_indices"],
                                                                              retrieve_indices=tb["retrieve_indices"], sample_token=st.sample_token)
            original_prob = st.tree_logits[2]
        else:
            self.base_model.model.tree_mask = st.tree_mask
            tree_candidates = st.tree_candidates.to(st.input_ids.device)
            cart_prob = original_prob = None
        rows, hidden, uhidden = self.tree_decoding(tree_candidates=tree_candidates, attention_mask=st.attn_mask, past_key_values=self.past_key_values,
                                                   tree_position_ids=st.tree_position_ids, input_ids=st.input_ids,
                                                   retrieve_indices=st.retrieve_indices)
        if not st.static:
            ext = torch.cat((tree_candidates, tree_candidates.new_full((1, 1), -1)), dim=1)
            candidates = ext[0, st.retrieve_indices]
            tree_candidates = ext
        # ---- O8 (+ bonus token), results on the device
        fifo = self._uniforms()
        fifo.reserve(candidates.shape[0] * candidates.shape[1])          # before the snapshot (see _posterior_on_device)
        cur0 = fifo.cursor.clone()
        u = self._bonus_uniform(dev)
        ep = self._posterior_on_device(rows, candidates, cart_prob, original_prob, tree_candidates, lantern, lantern_k, lantern_delta,
                                       u_bonus=u if isinstance(rows, WindowRows) else None, reserve=False)
        best, alen, status = ep["best"], ep["accept_len"], ep["counters"][:, 5]
        # ---- O9 + O10 for every slab in one launch, from the device-side (best, accept_len); a failed walk commits nothing
        L = st.input_ids.shape[1]
        alen_commit = torch.where(status == 0, alen, torch.full_like(alen, -1))
        hid = torch.stack([hidden[0], uhidden[0]])[None]                                     # [1, 2, N, H]
        prev = (L - st.slab_off).to(st.slabs[0].device)
        new_len, out_h, acc = ops.update_inference_inputs(st.slabs, st.slab_seq, prev, st.retrieve_indices.to(st.slabs[0].device),
                                                          best.to(st.slabs[0].device), alen_commit.to(st.slabs[0].device), hid,
                                                          candidates[None], slab_ptrs=st.slab_ptrs)
        token = ep["token"]
        if token is None:
            _, _, token = ops.accept_gather(None, st.retrieve_indices, None, best, alen, sample_p=ep["sample_p"].float(), u=u)
        eos_hit = torch.zeros(1, dtype=torch.int64, device=dev)
        if eos_token_ids is not None:
            live = torch.arange(acc.shape[1], device=acc.device)[None] <= alen_commit.to(acc.device)[:, None]
            eos_hit = ((acc == eos_token_ids) & live).any().to(torch.int64).reshape(1).to(dev)
        # ---- the step's one host read
        a, bst, stt, tok, eos = torch.cat((alen.to(torch.int64), best.to(torch.int64), status.to(torch.int64), token.to(torch.int64), eos_hit)).tolist()
        if stt != 0:
            if stt in self._RETRY_DENSE and isinstance(rows, WindowRows):
                # a state only the dense kernel represents: the same step on the dense HIP kernel, from the same uniforms,
                # through the host-int path (rare: once in millions of steps)
                fifo.cursor.copy_(cur0)
                bc, al, sample_p = self.evaluate_posterior(rows.dense_rows(), candidates, cart_prob, original_prob, self.tree_buffers["p_indices"] if st.static else None,
                                                           tree_candidates, self.tree_buffers["b_indices"] if st.static else None, True, lantern,
                                                           lantern_k, lantern_delta)
                return self._commit_from_host(st, candidates, bc, al, hidden, uhidden, sample_p, u, eos_token_ids)
            ops.raise_on_status(ep["counters"])
        n = a + 1
        for cl, off in zip(self._length_tensors(st), st.offsets):
            cl.fill_(L - off + n)
        accepted = acc[:, :n].to(st.input_ids.device)
        st.input_ids = torch.cat([st.input_ids[None, 0] if st.parallel else st.input_ids, accepted], dim=-1)
        self._draft_next(st, out_h[:, 0, :n], out_h[:, 1, :n], torch.tensor([[tok]], device=dev))
        st.new_token += n
        st.accept_lengths.append(n)
        return bool(eos)

    def _length_tensors(self, st):
        if st.parallel:
            return [self.current_length_data] * len(st.slabs)
        nc = len(self.past_key_values_data["cond"])
        return [self.current_length_data["cond"]] * nc + [self.current_length_data["uncond"]] * (len(st.slabs) - nc)

    def _draft_next(self, st, hidden, uhidden, token, ids_with_token=None):
        """`ids_with_token`: cat(input_ids, token) as a view of the step's in-place id buffer (the commit kernel appended the token) -- else built here."""
        ids = ids_with_token if ids_with_token is not None else torch.cat((st.input_ids, token.to(st.input_ids.device)), dim=-1)
        out = self.ea_layer.topK_generate(hidden_states=hidden, uncond_hidden_states=uhidden,
                                          input_ids=ids, attention_mask=st.attn_mask,
                                          head=self.base_model.lm_head, logits_processors=self.drafter_logits_processors,
                                          tree_type="static" if st.static else "dynamic")
        if st.static:
            st.tree_logits, st.sample_token = out, token
        else:
            self._take_dynamic_draft(st, out)

    def _commit_from_host(self, st, candidates, best, alen, hidden, uhidden, sample_p, u, eos_token_ids=None):
        """The fallback step's commit: host-side (best, accept_len) through the same kernels.  Returns whether an end-of-sequence
        token was accepted in this step (the reference re-scans the generated suffix every step: ea_model_lumina_mgpt.py:1007-1009)."""
        dev = st.retrieve_indices.device
        b = torch.as_tensor([int(best)], dtype=torch.int32, device=dev)
        a = torch.as_tensor([int(alen)], dtype=torch.int32, device=dev)
        L, n = st.input_ids.shape[1], int(alen) + 1
        sdev = st.slabs[0].device
        hid = torch.stack([hidden[0], uhidden[0]])[None]
        _, out_h, acc = ops.update_inference_inputs(st.slabs, st.slab_seq, (L - st.slab_off).to(sdev), st.retrieve_indices.to(sdev), b.to(sdev), a.to(sdev),
                                                    hid, candidates[None], slab_ptrs=st.slab_ptrs)
        _, _, token = ops.accept_gather(None, st.retrieve_indices, None, b, a, sample_p=sample_p[None].float(), u=u)
        for cl, off in zip(self._length_tensors(st), st.offsets):
            cl.fill_(L - off + n)
        st.input_ids = torch.cat([st.input_ids[None, 0] if st.parallel else st.input_ids, acc[:, :n].to(st.input_ids.device)], dim=-1)
        self._draft_next(st, out_h[:, 0, :n], out_h[:, 1, :n], token.reshape(1, 1))
        st.new_token += n
        st.accept_lengths.append(n)
        if eos_token_ids is None:
            return False
        return bool((acc[:, :n] == eos_token_ids).any())

    @torch.no_grad()
    def generate(self, input_ids, do_sample=True, max_new_tokens=2353, max_length=4096, cfg_scale=3.0, top_k=2000,
                 logits_processors=None, eos_token_ids=None, lantern=False, lantern_k=1000, lantern_delta=0.1,
                 tree_choices=mc_sim_7b_63, **kwargs):
        if not do_sample:
            raise NotImplementedError("Greedy decoding is not implemented yet")   # as the reference (:728-729)
        self._ub = None          # the bonus uniforms restart with every image (same torch seed -> same image on every path)
        st = self._prepare_generation(input_ids.clone(), cfg_scale, top_k, kwargs.get("drafter_top_k"), tree_choices)
        st.max_length = max_length
        self._first_draft(st, logits_processors)
        fifo = self._uniforms()
        fifo.begin()                       # the acceptance uniforms of this prompt start at random's current position
        try:
            while st.new_token < max_new_tokens:
                hit_eos = self._verify_step(st, lantern, lantern_k, lantern_delta, eos_token_ids)
                if hit_eos or st.input_ids.shape[1] > max_length:
                    break
        finally:
            fifo.end()                     # unconsumed staged draws go back to the module-level stream
        return st.input_ids, st.accept_lengths

    # BASELINE.json's north_star calls the entry point `eagenerate`; the reference names it `generate`
    eagenerate = generate

    def decode_ids(self, ids):
        """Token ids -> image: the base model's VQGAN decoder (out of scope), as the reference delegates."""
        return self.base_model.decode_ids(ids)

    @classmethod
    def from_pretrained(cls, base_model_path=None, ea_model_path=None, total_token=-1, depth=5, top_k=10, threshold=1.0, cfg_mode="sequential",
                        eagle_version=1, **kwargs):
        """The reference's constructor surface (EaLumina_mGPT.from_pretrained, models/ea_model_lumina_mgpt.py:347-416, as called by
        FlexARInferenceSolver.__init__, eagle_inference_solver.py:248-255): the reference's own loader reads the checkpoints; the
        loaded model is wrapped so that generate() runs this package's accept loop."""
        from .verify import reference_loader
        ref = reference_loader("models.ea_model_lumina_mgpt", "EaLumina_mGPT").from_pretrained(
            base_model_path=base_model_path, ea_model_path=ea_model_path, total_token=total_token, depth=depth, top_k=top_k, threshold=threshold,
            cfg_mode=cfg_mode, eagle_version=eagle_version, **kwargs)
        return cls.from_reference(ref, cfg_mode=cfg_mode, eagle_version=eagle_version)

    @classmethod
    def from_reference(cls, ref, **kw):
        """Wrap a model the reference's own `from_pretrained` loaded (checkpoint loading stays there): same base model, drafter
        and neighbour table, this package's accept loop."""
        return cls(ref.base_model, ref.ea_layer, ref.nearest_latents, **kw)
