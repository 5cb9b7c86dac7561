"""Caller-side mirrors: the Lumina solver's HF-signature logits processors against vectors produced by the
reference's own classes (tests/golden/make_golden_solver.py), and `FlexARInferenceSolver.generate` / `decode_ids` end to end."""
import os
import random
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "solver.npz")
V, BOI, EOI, NL = 65536, 8197, 8196, 8803


def _scores(seed):
    return torch.from_numpy((4.0 * np.random.RandomState(int(seed)).standard_normal((1, V))).astype(np.float32))


def _check(out, scores, g, key):
    fin = np.unpackbits(g[f"{key}.finite"])[:V].astype(bool)
    got = out[0].float().cpu().numpy()
    assert np.array_equal(np.isfinite(got), fin), key
    want = scores[0].numpy().copy()
    want[g[f"{key}.forced_idx"]] = g[f"{key}.forced_val"]
    assert np.array_equal(got[fin], want[fin]), key
    assert np.all(got[~fin] == -np.inf), key


def test_solver_processors_match_the_reference_vectors():
    from lantern_amd.base_models.lumina_mgpt.eagle_inference_solver import InterleavedTopKLogitsWarper, MultiModalLogitsProcessor
    g = np.load(GOLD)
    for name in g["names"]:
        ids = torch.from_numpy(g[f"{name}.ids"])[None].cuda()
        s = _scores(g[f"{name}.seed"])
        mm = MultiModalLogitsProcessor(image_start_token_id=BOI, image_end_token_id=EOI, image_next_line_token_id=NL, patch_size=32, voc_size=V)
        tk = InterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10, image_start_token_id=BOI, image_end_token_id=EOI)
        a = mm(ids, s.cuda())
        _check(a, s, g, f"{name}.mm")
        b = tk(ids, a)
        _check(b, s, g, f"{name}.tk")


def test_solver_processor_keeps_its_state_across_calls():
    """The reference caches the image start index and the latent size on the object until the image closes."""
    from lantern_amd.base_models.lumina_mgpt.eagle_inference_solver import MultiModalLogitsProcessor
    g = np.load(GOLD)
    mm = MultiModalLogitsProcessor(image_start_token_id=BOI, image_end_token_id=EOI, image_next_line_token_id=NL, patch_size=32, voc_size=V)
    for name in ("first_image_token", "mid_row", "row_end", "second_row", "last_row_end", "image_end", "closed", "text_only"):
        ids = torch.from_numpy(g[f"{name}.ids"])[None].cuda()
        s = _scores(g[f"{name}.seed"])
        _check(mm(ids, s.cuda()), s, g, f"{name}.mm")
    assert mm.image_start_token_id_index is None and mm.h_latent_dim is None


class FakeItemProcessor:
    image_start_token, image_end_token, new_line_token = "<boi>", "<eoi>", "<nl>"
    _ids = {"<boi>": BOI, "<eoi>": EOI, "<nl>": NL, "<|image|>": 8711}

    def __init__(self):
        self.tokenizer = types.SimpleNamespace(decode=lambda ids: " ".join(map(str, ids)))

    def token2id(self, t):
        return self._ids[t]

    def process_item(self, item):
        assert [c["from"] for c in item["conversations"]] == ["human", "gpt"]
        return [0, {"input_ids": [9000 + (ord(c) % 50) for c in item["conversations"][0]["value"][:8]]}, 8710]

    def decode_image(self, tokens):
        return ("image", len(tokens))


def test_flexar_solver_generate_and_decode_ids():
    from test_gpu_generate import make_model
    from lantern_amd.base_models.lumina_mgpt import FlexARInferenceSolver
    random.seed(5)
    torch.manual_seed(5)
    mdl = make_model(1, "sequential")
    solver = FlexARInferenceSolver(model=mdl, item_processor=FakeItemProcessor(), precision="bf16")
    procs = solver.create_logits_processor(cfg=3.0, image_top_k=200)
    assert len(procs) == 3 and procs[0].guidance_scale == 3.0
    tokens, step_compression, latency = solver.generate([], [["Generate an image of a cat", None]], 40, 1.0, 200, logits_processor=procs,
                                                        cfg_scale=3.0, lantern=True, lantern_k=100, lantern_delta=0.1)
    assert isinstance(tokens, list) and tokens[:3] == [BOI, 8828, 8828] and len(tokens) >= 43
    assert 1.0 <= step_compression <= 7.0 and latency > 0
    first = tokens[3]
    assert 4 <= first < 8196                      # the solver's processors shaped the prefill token: image ids only
    text, images = solver.decode_ids([9001, BOI, 5, 6, 7, EOI, 9002])
    assert images == [("image", 3)] and text == "9001 8711 9002"
