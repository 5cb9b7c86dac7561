"""Host mirror of the caller of the Lumina accept loop: `FlexARInferenceSolver`
(models/base_models/lumina_mgpt/eagle_inference_solver.py:234-403), the object entrypoints/generate_images.py
constructs (:99-107) and calls (`generate(images, qas, max_gen_len, temperature, top_k, cfg_scale, lantern, …)`,
`decode_ids(tokens)`).

What the solver itself does around `EaLumina_mGPT.generate` is small and is kept: conversation -> prompt token ids
through the item processor, the `(tokens after the prompt, mean accept length, latency)` return value with a trailing
8710 dropped (:318-324), splitting generated ids into text and image spans for `decode_ids` (:326-354), and the three
logits processors of `create_logits_processor` (:372-403).  Checkpoint loading, the tokenizer and the VQGAN are out
of scope: the model (`lantern_amd.ea_model_lumina_mgpt.EaLumina_mGPT` around the reference's loaded
base model and drafter) and the reference's `FlexARItemProcessor` are handed in."""
import time
from typing import List

import torch

from ... import ops
from ...ea_model_lumina_mgpt import IMAGE_HI, IMAGE_LO

EOS_IDS = [8710, 8196]
RESOLUTION_BASE = 8804        # <|image start|> is followed by two size tokens 8804 + grids (eagle_inference_solver.py:166-170)


class MultiModalLogitsProcessor:
    """HF-signature `(input_ids, scores)` form of the Lumina image-grammar mask (eagle_inference_solver.py:100-182): where the
    sequence stands inside an open image is read off `input_ids` (host side, once per call), the masking is the O7 kernel.
    Outside an image the scores pass through."""

    def __init__(self, image_start_token_id=None, image_end_token_id=None, image_next_line_token_id=None, patch_size=None,
                 voc_size=None):
        self.image_start_token_id = image_start_token_id
        self.image_end_token_id = image_end_token_id
        self.image_next_line_token_id = image_next_line_token_id
        self.patch_size, self.voc_size = patch_size, voc_size
        self.image_start_token_id_index = None
        self.h_latent_dim = self.w_latent_dim = None

    def __call__(self, input_ids, scores, position_ids=None):
        row = input_ids[0]
        n_start = int((row == self.image_start_token_id).sum())
        n_end = int((row == self.image_end_token_id).sum())
        if n_start == n_end:
            self.h_latent_dim = self.w_latent_dim = self.image_start_token_id_index = None
            return scores
        if n_start != n_end + 1:
            return scores
        if self.image_start_token_id_index is None:
            self.image_start_token_id_index = int(torch.where(row == self.image_start_token_id)[0][-1])
        start = self.image_start_token_id_index
        if row.shape[0] - (start + 1) < 2:
            return scores
        if self.h_latent_dim is None or self.w_latent_dim is None:
            self.h_latent_dim = 2 * (int(row[start + 1]) - RESOLUTION_BASE)
            self.w_latent_dim = 2 * (int(row[start + 2]) - RESOLUTION_BASE)
        pos = torch.full((scores.shape[0],), row.shape[0], dtype=torch.int64, device=scores.device)
        out = ops.cfg_mask_topk(scores, None, 1.0, model=ops.MODEL_LUMINA, pos_ids=pos, pos_base=start + 3, w=self.w_latent_dim,
                                h=self.h_latent_dim, img_lo=IMAGE_LO, img_hi=IMAGE_HI, newline_id=self.image_next_line_token_id,
                                eos_id=self.image_end_token_id, top_k=0)
        return out.to(scores.dtype)


class InterleavedTopKLogitsWarper:
    """`(input_ids, scores)`: keep the `image_top_k` largest inside an open image, `text_top_k` otherwise
    (eagle_inference_solver.py:185-232); the k-th-largest filter is the O7 kernel."""

    def __init__(self, image_top_k: int, text_top_k: int, image_start_token_id=None, image_end_token_id=None,
                 filter_value: float = -float("Inf"), min_tokens_to_keep: int = 1):
        for name, v in (("text_top_k", text_top_k), ("image_top_k", image_top_k)):
            if not isinstance(v, int) or v <= 0:
                raise ValueError(f"`{name}` has to be a strictly positive integer, but is {v}")
        if filter_value != -float("Inf"):
            raise ValueError("the kernel filters with -inf (the reference's default)")
        self.image_top_k = max(image_top_k, min_tokens_to_keep)
        self.text_top_k = max(text_top_k, min_tokens_to_keep)
        self.filter_value = filter_value
        self.image_start_token_id, self.image_end_token_id = image_start_token_id, image_end_token_id

    def __call__(self, input_ids, scores, position_ids=None):
        row = input_ids[0]
        in_image = int((row == self.image_start_token_id).sum()) == int((row == self.image_end_token_id).sum()) + 1
        top_k = min(self.image_top_k if in_image else self.text_top_k, scores.size(-1))
        return ops.cfg_mask_topk(scores, None, 1.0, model=ops.MODEL_PLAIN, top_k=top_k).to(scores.dtype)


class ClassifierFreeGuidanceSlot:
    """Entry 0 of the reference's processor list is the unbatched CFG processor (:372-381).  `EaLumina_mGPT.generate` never calls
    it -- it skips entry 0 (`logits_processors[1:]`, ea_model_lumina_mgpt.py:493) and mixes cond / uncond itself with `cfg_scale` --
    so the slot only records the scale."""

    def __init__(self, guidance_scale):
        self.guidance_scale = guidance_scale


class FlexARInferenceSolver:
    def __init__(self, model_path=None, drafter_path=None, precision="bf16", target_size=512, cfg_mode="sequential", eagle_version=1, *,
                 model=None, item_processor=None):
        """The reference's constructor (eagle_inference_solver.py:243-257): `FlexARInferenceSolver(model_path, drafter_path, precision,
        target_size, cfg_mode, eagle_version)` as generate_images.py:103-110 calls it -- the checkpoints and the item processor
        (tokenizer + VQGAN) are loaded by the reference's own loaders, the model is wrapped in this package's EaLumina_mGPT.  With
        `model=` / `item_processor=` the solver wraps objects the caller already holds (tests)."""
        self.dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[precision]
        if model is None and not isinstance(model_path, (str, bytes)) and not hasattr(model_path, "__fspath__"):
            raise TypeError("FlexARInferenceSolver: model_path must be a path (the reference's positional form); a model object goes in as "
                            "`model=` together with `item_processor=`")
        if model is None:
            from ...ea_model_lumina_mgpt import EaLumina_mGPT
            model = EaLumina_mGPT.from_pretrained(base_model_path=model_path, ea_model_path=drafter_path, cfg_mode=cfg_mode,
                                                  eagle_version=eagle_version, dtype=self.dtype, device_map="cuda")
        if item_processor is None:
            from ...verify import reference_loader
            item_processor = reference_loader("models.base_models.lumina_mgpt.item_processor", "FlexARItemProcessor")(target_size=target_size)
        self.model = model
        self.item_processor = item_processor

    # ------------------------------------------------------------------ :263-324
    def prompt_ids(self, images, qas) -> List[int]:
        conversations = []
        for q, a in qas:
            conversations += [{"from": "human", "value": q}, {"from": "gpt", "value": a}]
        prompt: List[int] = []
        for value in self.item_processor.process_item({"image": images, "conversations": conversations}):
            if isinstance(value, int):
                prompt.append(value)
            else:
                prompt += value["input_ids"]
        return prompt

    @torch.no_grad()
    def generate(self, images, qas, max_gen_len, temperature, top_k, logits_processor=None, streamer=None, **kwargs):
        prompt = self.prompt_ids(images, qas)
        dev = self.model.base_model.lm_head.weight.device
        ids = torch.tensor(prompt, dtype=torch.int64, device=dev)[None]
        if logits_processor is None:
            logits_processor = self.create_logits_processor()
        with torch.amp.autocast("cuda", dtype=self.dtype):
            start = time.time()
            # `eos_token_id` (singular) lands in **kwargs exactly as in the reference (:311): the loop stops on length only
            result, accept_length_list = self.model.generate(ids, do_sample=temperature > 0, max_new_tokens=max_gen_len, top_k=top_k,
                                                             logits_processors=logits_processor, eos_token_id=EOS_IDS, **kwargs)
            latency = time.time() - start
        step_compression = float(torch.tensor(accept_length_list, dtype=torch.float32).mean())
        tokens = result[0][len(prompt):].tolist()
        if tokens and tokens[-1] == 8710:
            tokens = tokens[:-1]
        return tokens, step_compression, latency

    # ------------------------------------------------------------------ :326-357
    def decode_ids(self, tokens: List[int]):
        ip = self.item_processor
        boi, eoi = ip.token2id(ip.image_start_token), ip.token2id(ip.image_end_token)
        images, text_ids = [], []
        i = 0
        while i < len(tokens):
            if tokens[i] != boi:
                text_ids.append(tokens[i])
                i += 1
                continue
            span = []
            for j in range(i + 1, len(tokens)):
                if tokens[j] != eoi:
                    span.append(tokens[j])
                    i = j + 1
                else:
                    images.append(self.decode_image(span))
                    text_ids.append(ip.token2id("<|image|>"))
                    i = j + 1
                    break
        return ip.tokenizer.decode(text_ids), images

    def decode_image(self, tokens: List[int]):
        return self.item_processor.decode_image(tokens)

    # ------------------------------------------------------------------ :372-403
    def create_logits_processor(self, cfg=3.0, image_top_k=2000, text_top_k=10):
        """[CFG slot, MultiModal, InterleavedTopK] like the reference's list; entries 1.. shape the first token
        (`initialize_tree`), the per-step rows go through the fused O7 kernel inside `tree_decoding`."""
        ip = self.item_processor
        boi, eoi = ip.token2id(ip.image_start_token), ip.token2id(ip.image_end_token)
        return [ClassifierFreeGuidanceSlot(cfg),
                MultiModalLogitsProcessor(image_start_token_id=boi, image_end_token_id=eoi,
                                          image_next_line_token_id=ip.token2id(ip.new_line_token), patch_size=32,
                                          voc_size=getattr(getattr(self.model, "config", None), "vocab_size", 65536)),
                InterleavedTopKLogitsWarper(image_top_k=image_top_k, text_top_k=text_top_k, image_start_token_id=boi,
                                            image_end_token_id=eoi)]
