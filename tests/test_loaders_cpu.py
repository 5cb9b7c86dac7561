"""CPU: the loader-side surface of the reference -- from_pretrained / FlexARInferenceSolver(model_path, drafter_path, ...)
delegate to the reference's own loaders and wrap what they return; the neighbour table's on-disk format round-trips."""
import sys
import types

import numpy as np
import pytest
import torch

from lantern_amd import ea_model_anole, ea_model_llamagen, ea_model_lumina_mgpt, ops
from lantern_amd._lib import LanternError
from lantern_amd.base_models.lumina_mgpt.eagle_inference_solver import FlexARInferenceSolver


def _ref_model():
    lm = types.SimpleNamespace(weight=torch.zeros(64, 8))
    base = types.SimpleNamespace(lm_head=lm, config=None, model=types.SimpleNamespace())
    table = (np.arange(16 * 15, dtype=np.uint16).reshape(16, 15) % 16)
    return types.SimpleNamespace(base_model=base, ea_layer=object(), nearest_latents=table)


@pytest.mark.parametrize("mod,cls,refmod,refcls", [(ea_model_llamagen, "EaModel", "models.ea_model_llamagen", "EaModel"),
                                                   (ea_model_anole, "EaModel", "models.ea_model_anole", "EaModel"),
                                                   (ea_model_lumina_mgpt, "EaLumina_mGPT", "models.ea_model_lumina_mgpt", "EaLumina_mGPT")])
def test_from_pretrained_delegates_to_the_reference_loader(monkeypatch, mod, cls, refmod, refcls):
    calls = []

    class RefCls:
        @classmethod
        def from_pretrained(c, **kw):
            calls.append(kw)
            return _ref_model()

    for name in ("models", refmod):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    setattr(sys.modules[refmod], refcls, RefCls)
    kw = dict(base_model_path="ckpts/base", ea_model_path="ckpts/drafter", total_token=59, depth=4, top_k=10, threshold=1.0)
    if refcls == "EaLumina_mGPT":
        m = getattr(mod, cls).from_pretrained(cfg_mode="parallel", eagle_version=2, **kw)
        assert m.cfg_mode == "parallel" and m.eagle_version == 2 and calls[0]["cfg_mode"] == "parallel"
    else:
        m = getattr(mod, cls).from_pretrained(Type="LLaMA", **kw)
    assert isinstance(m, getattr(mod, cls)) and calls[0]["base_model_path"] == "ckpts/base" and calls[0]["ea_model_path"] == "ckpts/drafter"
    assert calls[0]["total_token"] == 59 and m.nearest_latents.shape == (16, 15) and m.nearest_latents.dtype == torch.int16


def test_from_pretrained_without_the_reference_says_so(monkeypatch):
    for name in [n for n in sys.modules if n == "models" or n.startswith("models.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if "reference" not in p])
    with pytest.raises(LanternError, match="delegates model / checkpoint loading to the reference"):
        ea_model_llamagen.EaModel.from_pretrained(base_model_path="x", ea_model_path="y")
    with pytest.raises(LanternError, match="from_reference"):
        FlexARInferenceSolver("ckpts/lumina", "ckpts/drafter", "bf16", target_size=768, cfg_mode="sequential", eagle_version=1)


def test_solver_takes_the_reference_arguments(monkeypatch):
    """FlexARInferenceSolver(model_path, drafter_path, precision, target_size, cfg_mode, eagle_version), as generate_images.py:103-110
    constructs it: model and item processor come from the reference's loaders."""
    seen = {}

    class RefLumina:
        @classmethod
        def from_pretrained(c, **kw):
            seen["model"] = kw
            return _ref_model()

    class ItemProc:
        def __init__(self, target_size):
            seen["target_size"] = target_size

    for name in ("models", "models.ea_model_lumina_mgpt", "models.base_models", "models.base_models.lumina_mgpt", "models.base_models.lumina_mgpt.item_processor"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["models.ea_model_lumina_mgpt"].EaLumina_mGPT = RefLumina
    sys.modules["models.base_models.lumina_mgpt.item_processor"].FlexARItemProcessor = ItemProc
    s = FlexARInferenceSolver("ckpts/lumina", "ckpts/drafter", "bf16", target_size=768, cfg_mode="sequential", eagle_version=1)
    assert isinstance(s.model, ea_model_lumina_mgpt.EaLumina_mGPT) and s.dtype == torch.bfloat16 and seen["target_size"] == 768
    assert seen["model"]["base_model_path"] == "ckpts/lumina" and seen["model"]["ea_model_path"] == "ckpts/drafter"
    assert seen["model"]["device_map"] == "cuda" and seen["model"]["dtype"] == torch.bfloat16


def test_vq_table_file_round_trip(tmp_path):
    K = 32
    rs = np.random.RandomState(0)
    arr = np.stack([rs.permutation(K)[:K - 1] for _ in range(K)]).astype(np.uint16)
    path = ops.save_vq_table(torch.from_numpy(arr.view(np.int16)), str(tmp_path / "vq_distances"))
    assert path.endswith("top_31_indices.npy")
    raw = np.load(path)                                   # what the reference's np.load sees
    assert raw.dtype == np.uint16 and np.array_equal(raw, arr)
    back = ops.load_vq_table(path)
    assert back.dtype == torch.int16 and np.array_equal(back.numpy().view(np.uint16), arr)
    with pytest.raises(LanternError):
        ops.save_vq_table(torch.zeros((8, 5), dtype=torch.int16), str(tmp_path))


def test_llamagen_prompt_block_matches_the_reference_recipe():
    """EaModel._encode_prompt (reference: models/ea_model_llamagen.py:1018-1056): right-padded T5 rows become left-padded (each row
    rotated by its valid length), are zeroed under the flipped mask, and the unconditional embedding rows follow for CFG -- so the
    reference's base model needs no `encode_prompt` helper."""
    import types
    import torch
    from lantern_amd.ea_model_llamagen import EaModel
    g = torch.Generator().manual_seed(0)
    B, T, Cc = 3, 7, 5
    emb = torch.randn(B, T, Cc, generator=g)
    valid = [7, 3, 1]
    mask = torch.zeros(B, T, dtype=torch.int64)
    for i, v in enumerate(valid):
        mask[i, :v] = 1
    unc = torch.randn(T, Cc, generator=g)
    bm = types.SimpleNamespace(t5_model=types.SimpleNamespace(get_text_embeddings=lambda p: (emb, mask)), dtype=torch.float32,
                               model=types.SimpleNamespace(cls_embedding=types.SimpleNamespace(uncond_embedding=unc)))
    me = types.SimpleNamespace(base_model=bm)
    # the recipe, row by row
    rows = torch.stack([torch.cat([emb[i, v:], emb[i, :v]]) for i, v in enumerate(valid)])
    lm = torch.flip(mask, dims=[-1])
    want = rows * lm[:, :, None]
    cond, am = EaModel._encode_prompt(me, ["a", "b", "c"], cfg=1.5)
    assert torch.equal(cond[:B], want) and torch.equal(cond[B:], (torch.zeros_like(want) + unc))
    assert torch.equal(am, torch.cat([lm, lm]))
    cond1, am1 = EaModel._encode_prompt(me, ["a", "b", "c"], cfg=None)
    assert torch.equal(cond1, want) and torch.equal(am1, lm)
    bm.encode_prompt = lambda p, c: ("mine", None)           # a base model's own helper wins
    assert EaModel._encode_prompt(me, ["a"], 1.0)[0] == "mine"


def test_top_p_is_taken_by_both_kernel_sets_and_refused_only_outside_its_domain():
    """Nucleus filtering is built into the windowed kernel set (per row in tree_decoding's post-process) and the dense one (per visited row inside
    evaluate_posterior): the mirrors refuse only what TopPLogitsWarper itself refuses."""
    import pytest
    from lantern_amd import _lib
    from lantern_amd.ea_model_anole import EaModel as Anole
    from lantern_amd.ea_model_llamagen import EaModel as LlamaGen
    for cls in (LlamaGen, Anole):
        m = cls.__new__(cls)
        for ks in ("dense", "window"):
            m.kernel_set = ks
            m._check_processors(1.0, 0.9)
            m._check_processors(0.0, 0.9)          # greedy decoding has no processors
            m._check_processors(1.0, 1.0)
            with pytest.raises(_lib.LanternError, match="outside"):
                cls.generate(m, prompt=["x"], max_length=4, temperature=1.0, top_k=100, top_p=1.5, cfg=2.0)


@pytest.mark.parametrize("model_type", ["lumina_mgpt", "anole", "llamagen"])
def test_drafter_from_reference_rehosts_a_loaded_drafter(model_type):
    """cnets.Model.from_reference: a loaded drafter (here a stand-in with the reference Model's attributes -- no `config` of its own, the attention
    module keeps it, as in cnets_llamagen.py:229) re-hosted on this package's Model: same tree parameters, the state_dict loaded as is, the HIP decoder
    layer of the model family (constructed on the CPU; the kernels only run on a device)."""
    import types
    import torch
    from lantern_amd.drafters import cnets
    from lantern_amd.drafters.decoder_layer import DecoderLayer, LlamaDecoderLayer
    H, heads = 128, 2
    cfg = types.SimpleNamespace(vocab_size=512, hidden_size=H, pad_token_id=None, num_hidden_layers=1, num_attention_heads=heads, num_key_value_heads=heads,
                                intermediate_size=256, max_position_embeddings=128, rms_norm_eps=1e-5, model_parallel_size=1, input_type="t2i")
    torch.manual_seed(1)
    src = cnets.Model(cfg, total_tokens=40, depth=4, top_k=10, model_type=model_type)
    ref = types.SimpleNamespace(fc=src.fc, layers=src.layers, total_tokens=src.total_tokens, depth=src.depth, top_k=src.top_k, threshold=src.threshold,
                                state_dict=src.state_dict, parameters=src.parameters)
    assert not hasattr(ref, "config")
    src.layers[0].self_attn.config = cfg
    got = cnets.Model.from_reference(ref, model_type)
    assert isinstance(got.layers[0], LlamaDecoderLayer if model_type == "llamagen" else DecoderLayer)
    assert (got.total_tokens, got.depth, got.top_k) == (src.total_tokens, src.depth, src.top_k)
    a, b = src.state_dict(), got.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
