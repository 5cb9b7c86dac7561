"""The drafter's prompt prefill (first call of topK_generate: the whole prompt through cnets.Model.forward) at the 7B layer size: the HIP path
(drafter_fc slices, lantern_linear_rows_packed, head stage, block-causal lantern_tree_attention) against the same model on torch's ops
(F.linear = hipBLASLt, eager softmax attention under the additive mask), milliseconds per call between HIP events.
usage: prefill_bench.py [prompt tokens=600] [reps=5]"""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lantern_amd.drafters import cnets, decoder_layer

T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev, bf = torch.device("cuda"), torch.bfloat16
cfg = types.SimpleNamespace(vocab_size=65536, hidden_size=4096, pad_token_id=None, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                            intermediate_size=11008, max_position_embeddings=4096, rms_norm_eps=1e-5, model_parallel_size=1)
torch.manual_seed(0)
mdl = cnets.Model(cfg, total_tokens=59, depth=5, top_k=10, model_type="lumina_mgpt").to(dev).to(bf)
mdl.init_tree()
x = torch.randn(2, T, 4096, device=dev, dtype=bf)
ids = torch.randint(4, 8000, (2, T), device=dev)
am = torch.ones(2, T, dtype=torch.bool, device=dev)
am[1, :9] = False


def run():
    mdl.reset_kv()
    mdl.tree_mask = None
    with torch.no_grad():
        return mdl(x, ids, attention_mask=am, use_cache=True)[0]


def timed():
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, y


ms_hip, y_hip = timed()
for l in mdl.layers:
    l.fused = False
    l.inplace_cache = False
decoder_layer._hip_ok = lambda a, b: False
ms_torch, y_torch = timed()
w_bytes = sum(p.numel() * 2 for p in mdl.layers[0].parameters())
flops = 2 * 2 * T * (w_bytes // 2)
print(json.dumps({"rows": 2 * T, "ms_hip": round(ms_hip, 3), "ms_torch_ops": round(ms_torch, 3), "layer_weight_MB": round(w_bytes / 1e6, 1),
                  "layer_gemm_TFLOPs_hip": round(flops / ms_hip / 1e9, 1), "max_abs_diff_valid_rows": float((y_hip[:, 9:].float() - y_torch[:, 9:].float()).abs().max())}))
