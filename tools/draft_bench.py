"""One drafting cycle of the EAGLE-2 drafter (cnets.Model.topK_generate / topK_genrate: prefill of the accepted tokens, `depth` tree steps of
top_k tokens each, the head expansion after every step, the tree finalisation) at the reference's model sizes, from Python, wall clock:
microseconds per cycle and per drafting depth, with the GPU time of the same kernels (HIP events around the cycle) beside it.

`<model>_static`: the static-tree loop (EAGLE v1 -- Lumina's default eagle_version with mc_sim_7b_63, generate_images.py:59; Anole / LlamaGen LANTERN++
static drafting with naive_extend_57, BASELINE config 4) through StaticDraftPlan: one lantern_head_sample + one lantern_draft_depth per tree level.

usage: draft_bench.py [lumina|anole|llamagen][_static] [cached positions] [cycles]"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from lantern_amd.drafters import cnets

model = sys.argv[1] if len(sys.argv) > 1 else "lumina"
static = model.endswith("_static")
model = model[:-7] if static else model
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev, bf = torch.device("cuda"), torch.bfloat16
if model == "llamagen":
    cfg = types.SimpleNamespace(vocab_size=16384, hidden_size=1280, pad_token_id=None, num_hidden_layers=1, num_attention_heads=20, num_key_value_heads=20,
                                intermediate_size=3584, max_position_embeddings=2048, rms_norm_eps=1e-6, input_type="t2i")
    depth, mt, S0 = 4, "llamagen", min(S0, 300)
else:
    cfg = types.SimpleNamespace(vocab_size=65536, hidden_size=4096, pad_token_id=None, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                                intermediate_size=11008, max_position_embeddings=4096, rms_norm_eps=1e-5, model_parallel_size=1)
    depth, mt = (5, "lumina_mgpt") if model == "lumina" else (4, "anole")
torch.manual_seed(0)
mdl = cnets.Model(cfg, total_tokens=59, depth=depth, top_k=10, model_type=mt).to(dev).to(bf)
if static:
    from lantern_amd.drafters import choices
    tree_name = "mc_sim_7b_63" if model == "lumina" else "naive_extend_57"
    mdl.init_tree(getattr(choices, tree_name))
    depth = len(mdl.tree_buffer["tree_indices"])
else:
    mdl.init_tree()
head = torch.nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False).to(dev).to(bf)
H = cfg.hidden_size
from transformers.generation.logits_process import LogitsProcessorList, TopKLogitsWarper
proc = LogitsProcessorList([TopKLogitsWarper(2000)])
lum_proc = [None, types.SimpleNamespace(image_top_k=2000)]


def cycle(n_new, total):
    """one drafting call: `n_new` accepted tokens behind a cached prefix, `total` positions in all"""
    hid = torch.randn(2, n_new if mdl.stable_kv is not None else total, H, device=dev, dtype=bf)
    if model == "lumina":
        ids = torch.randint(4, 8000, (1, total + 1), device=dev)
        am = torch.ones(2, total, dtype=torch.bool, device=dev)
        return mdl.topK_generate(hid[:1], hid[1:], ids, head, lum_proc, attention_mask=am, tree_type="static" if static else "dynamic")
    ids = torch.randint(4, 8000, (2, total + 1), device=dev)
    gen = mdl.topK_genrate_v1 if static else mdl.topK_genrate
    if model == "anole":
        return gen(hid, ids, head, proc, 3.0, input_position_diff=torch.zeros((), dtype=torch.long, device=dev),
                   attention_mask=torch.ones(2, total, dtype=torch.bool, device=dev))
    return gen(hid, ids, head, proc, 3.0)


total = S0 if model != "llamagen" else 120 + 20
cycle(total, total)          # prompt prefill (untimed)
for _ in range(3):
    total += 3
    cycle(3, total)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(cycles):
    total += 3
    cycle(3, total)
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / cycles
plan = mdl.__dict__.get("_splan" if static else "_plan")
print(json.dumps({"model": model + ("_static" if static else ""), "tree": (tree_name if static else "EAGLE-2 dynamic"), "one_c_call_per_depth": plan is not None,
                  "levels": ([len(t) for t in mdl.tree_buffer["tree_indices"]] if static else None), "cached_positions": S0, "depth": depth, "us_per_cycle_wall": 1e6 * wall, "us_per_depth_wall": 1e6 * wall / (depth + 1),
                  "us_per_cycle_stream": 1e3 * e0.elapsed_time(e1) / cycles, "drafting_path": type(mdl.layers[0]).__name__}))
