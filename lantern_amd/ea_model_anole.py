"""Host mirror of models/ea_model_anole.py: the LlamaGen driver with Anole's per-model constants
(SURVEY 8a-bis): image tokens 4..8195 with table offset 4 (ea_model_anole.py:142-146), non-image
logits forced to finfo.min after CFG (:931), separate cond/uncond position ids (:915-918), no
120-token zero prefix (input_ids is the cond row, :1088)."""
from .ea_model_llamagen import EaModel as _LlamaGenEaModel, cfg_logit_process  # noqa: F401


class EaModel(_LlamaGenEaModel):
    image_token_offset = 4
    image_lo, image_hi = 4, 8196
    mask_non_image = True
    prefix_pad = 0
