"""Host mirror of models/ea_model_anole.py: the LlamaGen driver with Anole's per-model constants
(SURVEY 8a-bis): image tokens 4..8195 with table offset 4 (ea_model_anole.py:142-146), non-image
logits forced to finfo.min after CFG (:425,:931), separate cond/uncond position ids (:915-918), no
120-token zero prefix (input_ids is the cond row, :1088), and a text prompt that enters as token ids
(left-padded with id 1, :1029-1045) instead of T5 embeddings.

Everything per step (tree_decoding, evaluate_posterior[_v1], update_inference_inputs) is inherited: the
base class already switches on `mask_non_image`.  Only the prompt handling differs."""
import time
from typing import List, Optional, Sequence, Union

import torch

from . import ops
from .ea_model_llamagen import EaModel as _LlamaGenEaModel, cfg_logit_process  # noqa: F401
from .drafters.choices import mc_sim_7b_63, naive_extend_57
from .drafters.kv_cache import initialize_past_key_values
from .verify import ProcessorSpec, prepare_logits_processor

PAD_ID, BOS_ID, BOI_ID, SEP_ID = 1, 0, 8197, 8710        # ids used by ea_model_anole.py:1030-1033


def pad_nested_list_left(nested_list: Sequence[Sequence[int]]):
    """Left-pad every row with PAD_ID to the longest row (ea_model_anole.py:88-95)."""
    width = max(len(row) for row in nested_list)
    return [[PAD_ID] * (width - len(row)) + list(row) for row in nested_list], width


class EaModel(_LlamaGenEaModel):
    image_token_offset = 4
    image_lo, image_hi = 4, 8196
    mask_non_image = True
    prefix_pad = 0

    def __init__(self, base_model, ea_layer, nearest_latents, tokenizer=None):
        super().__init__(base_model, ea_layer, nearest_latents)
        self.tokenizer = tokenizer                       # needs .tokenize_text(str) -> List[int]; token-id prompts need none

    # ------------------------------------------------------------------ :306-336 (Chameleon has no cond_idx)
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, output_orig=False, position_ids=None):
        with torch.inference_mode():
            outputs = self.base_model.model(input_ids=input_ids, attention_mask=attention_mask, past_key_values=past_key_values,
                                            position_ids=position_ids)
            if output_orig:
                orig = self.base_model.lm_head(outputs[0])
            hidden_states = outputs[0]
        return (outputs, orig, hidden_states) if output_orig else (outputs, hidden_states)

    # ------------------------------------------------------------------ :419-436 / :438-459
    def _first_token(self, input_ids, past_key_values, logits_processor, cfg_scale, attention_mask, input_position_ids):
        outputs, orig, hidden_states = self(input_ids=input_ids, past_key_values=past_key_values, output_orig=True,
                                            attention_mask=attention_mask, position_ids=input_position_ids)
        half = orig.shape[0] // 2
        # CFG mix + non-image ids -> finfo.min: the same O7 kernel the tree rows go through
        logits = ops.cfg_mask_topk(orig[:half, -1].contiguous(), orig[half:, -1].contiguous(), float(cfg_scale), model=ops.MODEL_ANOLE,
                                   img_lo=self.image_lo, img_hi=self.image_hi)
        if logits_processor is not None:
            logits = logits_processor(None, logits)
            token = torch.multinomial(torch.nn.functional.softmax(logits.float(), dim=1), 1)
        else:
            token = torch.argmax(logits)[None, None]
        token = torch.cat([token, token], dim=0)
        input_ids = torch.cat((input_ids, token.to(input_ids.device)), dim=1)
        return input_ids, token, logits, orig, hidden_states, input_position_ids.shape[1] - 2

    @torch.no_grad()
    def initialize_tree(self, input_ids, past_key_values, logits_processor, cfg_scale, attention_mask=None, input_position_ids=None):
        input_ids, token, logits, orig, hidden_states, diff = self._first_token(input_ids, past_key_values, logits_processor, cfg_scale,
                                                                                 attention_mask, input_position_ids)
        out = self.ea_layer.topK_genrate(hidden_states, input_ids, self.base_model.lm_head, logits_processor, cfg_scale, diff,
                                         attention_mask)
        return (*out, orig, hidden_states, token)

    @torch.no_grad()
    def initialize_tree_v1(self, input_ids, tree_attn_mask, past_key_values, logits_processor, cfg_scale, attention_mask=None,
                           input_position_ids=None, tree_choices=mc_sim_7b_63):
        input_ids, token, logits, orig, hidden_states, diff = self._first_token(input_ids, past_key_values, logits_processor, cfg_scale,
                                                                                 attention_mask, input_position_ids)
        self.ea_layer.init_tree_v1(tree_choices)
        tree_logits = self.ea_layer.topK_genrate_v1(hidden_states, input_ids, self.base_model.lm_head, logits_processor, cfg_scale, diff,
                                                    attention_mask)
        self.base_model.model.tree_mask = tree_attn_mask
        return tree_logits, logits, token

    # ------------------------------------------------------------------ :1010-1155
    def _prompt_tokens(self, prompt: Sequence[Union[str, Sequence[int]]]) -> List[List[int]]:
        rows = []
        for p in prompt:
            if isinstance(p, str):
                if self.tokenizer is None:
                    raise RuntimeError("ea_model_anole.EaModel.generate: a text prompt needs a tokenizer (pass tokenizer= or token-id lists)")
                p = self.tokenizer.tokenize_text(p)
            rows.append([int(t) for t in p])
        return rows

    @torch.no_grad()
    def generate(self, prompt: Optional[List[str]] = None, max_length: Optional[int] = None, temperature: Optional[float] = None,
                 top_k: Optional[int] = None, top_p: Optional[float] = None, cfg: Optional[float] = None,
                 lantern: Optional[bool] = None, lantern_k: Optional[int] = None, lantern_delta: Optional[float] = None,
                 static_tree: Optional[bool] = None, tree_choices: Optional[List[List[int]]] = naive_extend_57, **model_kwargs):
        self._check_processors(temperature, top_p)
        dev = self.base_model.lm_head.weight.device
        cond_tokens, max_input_length = pad_nested_list_left([[BOS_ID] + row + [SEP_ID, BOI_ID] for row in self._prompt_tokens(prompt)])
        uncond_tokens = [[PAD_ID] * (max_input_length - 2) + [BOS_ID, BOI_ID] for _ in cond_tokens]
        input_tokens = torch.tensor(cond_tokens + uncond_tokens, dtype=torch.long, device=dev)
        n_rows = len(cond_tokens)
        input_mask = input_tokens != PAD_ID
        input_position_ids = torch.zeros_like(input_tokens)
        input_position_ids[:n_rows] = torch.arange(max_input_length, device=dev)
        input_position_ids[n_rows:, -1] = 1              # uncond row: <pad>.. <bos>@0 <boi>@1
        input_position_diff = max_input_length - 2
        self.ea_layer.reset_kv()
        logits_processor = prepare_logits_processor(temperature=temperature, top_k=top_k, top_p=top_p) if temperature > 1e-5 else None
        self._active_proc = ProcessorSpec.from_hf(logits_processor)
        st = time.time()
        if static_tree:
            if not (hasattr(self, "tree_choices") and self.tree_choices == tree_choices):
                self.tree_buffers = self.generate_tree_buffers(tree_choices, device=dev)
                self.tree_buffers["retrieve_indices_head"] = self.tree_buffers["retrieve_indices"]
                self.tree_choices = tree_choices
            tree_buffers = self.tree_buffers
        if not hasattr(self.base_model, "past_key_values"):
            (self.base_model.past_key_values, self.base_model.past_key_values_data,
             self.base_model.current_length_data) = initialize_past_key_values(self.base_model, 2)
        past_key_values = self.base_model.past_key_values
        past_key_values_data = self.base_model.past_key_values_data
        current_length_data = self.base_model.current_length_data
        current_length_data.zero_()
        self.reset_tree_mode()
        if static_tree:
            tree_logits, logits, sample_token = self.initialize_tree_v1(input_tokens, tree_buffers["tree_attn_mask"], past_key_values,
                                                                        logits_processor, cfg, input_mask, input_position_ids, tree_choices)
        else:
            draft_tokens, retrieve_indices, tree_mask, tree_position_ids, logits, hidden_state, sample_token = self.initialize_tree(
                input_tokens, past_key_values, logits_processor, cfg, input_mask, input_position_ids)
        import types
        st_ = types.SimpleNamespace(input_ids=input_tokens[:1], static=bool(static_tree), attention_mask=input_mask, input_position_diff=input_position_diff)
        if static_tree:
            self._take_draft(st_, tree_logits, sample_token)
        else:
            self._take_draft(st_, (draft_tokens, retrieve_indices, tree_mask, tree_position_ids), sample_token)
        accept_length_list = self._decode_loop(st_, max_length, logits_processor, cfg, lantern, lantern_k, lantern_delta)   # the LlamaGen mirror's loop
        return (st_.input_ids[:, max_input_length:max_input_length + max_length], sum(accept_length_list) / len(accept_length_list),
                time.time() - st)

    @classmethod
    def from_pretrained(cls, Type="LLaMA", base_model_path=None, ea_model_path=None, total_token=59, depth=4, top_k=10, threshold=1.0, **kwargs):
        """models.ea_model_anole.EaModel.from_pretrained (ea_model_anole.py:151-224) loads; this class runs the accept loop."""
        from .verify import reference_loader
        ref = reference_loader("models.ea_model_anole", "EaModel").from_pretrained(Type=Type, base_model_path=base_model_path, ea_model_path=ea_model_path,
                                                                                  total_token=total_token, depth=depth, top_k=top_k, threshold=threshold, **kwargs)
        return cls.from_reference(ref)

    eagenerate = generate
