"""BASELINE.json config C1 (CPU plumbing): LlamaGen-B-sized class-conditional vanilla AR decode -> tokens -> the statistics file
the reference's driver writes (generate_images.py:296-306)."""
import json

import torch

from lantern_amd import sharding
from lantern_amd.base_models.llamagen.vanilla_ar import ClassCondLlamaGen, LlamaGenBConfig


def test_c1_class_conditional_vanilla_decode_and_statistics(tmp_path):
    torch.set_num_threads(8)
    cfg = LlamaGenBConfig()
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.vocab_size) == (768, 12, 12, 16384)     # LlamaGen-B
    model = ClassCondLlamaGen(cfg).eval()
    labels = [207, 360]
    g = torch.Generator().manual_seed(1)
    seq, mean_accept, seconds = model.generate(labels, max_length=256, temperature=1.0, top_k=2000, top_p=1.0, cfg=4.0, generator=g)
    assert seq.shape == (2, 256) and seq.dtype == torch.int64 and int(seq.min()) >= 0 and int(seq.max()) < 16384
    assert mean_accept == 1.0 and seconds > 0 and int(model.current_length_data[0]) == 256          # class token + 255 fed-back tokens
    assert len(set(seq[0].tolist())) > 64                              # a sampled sequence, not a collapsed one
    # same seed -> same image tokens (the KV cache is reused, reset by generate)
    g = torch.Generator().manual_seed(1)
    seq2, _, _ = model.generate(labels, max_length=256, temperature=1.0, top_k=2000, top_p=1.0, cfg=4.0, generator=g)
    assert torch.equal(seq, seq2)
    # the label matters, and so does the guidance scale
    g = torch.Generator().manual_seed(1)
    seq3, _, _ = model.generate([1, 2], max_length=32, temperature=1.0, top_k=2000, cfg=4.0, generator=g)
    assert not torch.equal(seq3, seq[:, :32])
    g = torch.Generator().manual_seed(1)
    nocfg, _, _ = model.generate(labels, max_length=32, temperature=1.0, top_k=2000, cfg=None, generator=g)
    assert nocfg.shape == (2, 32) and not torch.equal(nocfg, seq[:, :32])
    greedy, _, _ = model.generate(labels, max_length=8, temperature=0.0, cfg=1.0)
    greedy2, _, _ = model.generate(labels, max_length=8, temperature=0.0, cfg=None)
    assert torch.equal(greedy, greedy2)                                 # cfg = 1 is the conditional model itself
    entries = {f"prompt_{i}": sharding.statistics_entry(f"class {c}", mean_accept, seconds / len(labels)) for i, c in enumerate(labels)}
    path = sharding.write_global_statistics(str(tmp_path), entries, 0, len(labels))
    with open(path) as f:
        stats = json.load(f)
    assert list(stats) == ["prompt_0", "prompt_1"] and stats["prompt_0"]["step_compression"] == 1.0 and stats["prompt_1"]["prompt"] == "class 360"
