"""Drafter decoder layer at Lumina-mGPT-7B size (hidden 4096, 32 heads, intermediate 11008) for one drafting depth (2 x top_k = 20 rows
against a KV cache): lantern_amd.drafters.decoder_layer on the HIP skinny GEMM vs the same module on torch's bf16 nn.functional.linear.
Usage: python tools/layer_bench.py [rows_per_stream=10] [past=1200] [out.json]"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd.drafters import decoder_layer as DL

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
past = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
dev, bf = torch.device("cuda"), torch.bfloat16
cfg = types.SimpleNamespace(hidden_size=4096, intermediate_size=11008, num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096,
                            model_parallel_size=1, rope_theta=10000.0, rms_norm_eps=1e-5, attention_bias=False, mlp_bias=False, hidden_act="silu")
torch.manual_seed(0)
layer = DL.DecoderLayer(cfg, 0).eval().to(device=dev, dtype=bf)
with torch.no_grad():
    for p in layer.parameters():
        if p.dim() > 1:
            p.copy_(torch.randn_like(p) / p.shape[-1] ** 0.5)
x = torch.randn(2, n, 4096, device=dev, dtype=bf)
kv = (torch.randn(2, 32, past, 128, device=dev, dtype=bf), torch.randn(2, 32, past, 128, device=dev, dtype=bf))
pos = (past + torch.arange(n, device=dev))[None].expand(2, n).contiguous()
mask = torch.zeros(2, 1, n, past + n, device=dev)
mask[:, :, :, past:] = torch.where(torch.eye(n, device=dev) > 0, 0.0, torch.finfo(torch.float32).min)

def run():
    with torch.no_grad():
        return layer(x, attention_mask=mask, position_ids=pos, past_key_value=kv, use_cache=True)[0]

def timeit(k=50):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3

wbytes = sum(p.numel() * 2 for p in layer.parameters() if p.dim() > 1)
if os.environ.get("LAYER_BENCH_ONLY") == "tree":
    # the drafting call as cnets.Model.forward issues it (tree block as ancestor words, in-place cache) and nothing else: the profile of
    # this run shows the layer's own kernels only
    from lantern_amd import ops
    bits, t1 = ops.drafter_tree_bits(torch.eye(n, device=dev)[None, None], n)
    start = torch.zeros(2, dtype=torch.int64, device=dev)
    layer.inplace_cache = True
    with torch.no_grad():
        _, pres = layer(x, attention_mask=mask, position_ids=pos, past_key_value=kv, use_cache=True)
    prefix = (pres[0][:, :, :past], pres[1][:, :, :past])
    def run():
        with torch.no_grad():
            return layer(x, attention_mask=None, position_ids=pos, past_key_value=prefix, use_cache=True, tree_bits=bits, tree_keys=t1, kv_start=start)[0]
    t = timeit(200)
    print(json.dumps({"shape": f"rows 2x{n}, past {past}, hidden 4096, heads 32, intermediate 11008 (bf16)", "weight_bytes": wbytes,
                      "hip_tree_attention_inplace_cache_us": t, "weight_stream_GBps": wbytes / (t * 1e-6) / 1e9}))
    sys.exit(0)
y_hip = run().float()
t_hip = timeit()
real = DL._hip_ok
DL._hip_ok = lambda a, w: False
y_t = run().float()
t_torch = timeit()
DL._hip_ok = real
err = float((y_hip - y_t).abs().max() / y_t.abs().max())
out = {"shape": f"rows 2x{n}, past {past}, hidden 4096, heads 32, intermediate 11008 (bf16)", "weight_bytes": wbytes,
       "hip_skinny_gemm_us": t_hip, "torch_linear_us": t_torch, "speedup": t_torch / t_hip,
       "weight_stream_GBps_hip": wbytes / (t_hip * 1e-6) / 1e9, "frac_of_8TBps_hip": wbytes / (t_hip * 1e-6) / 1e9 / 8000.0,
       "weight_stream_GBps_torch": wbytes / (t_torch * 1e-6) / 1e9, "max_rel_diff_between_the_two": err,
       "note": "wall time per layer call between HIP events, launch gaps of the small torch ops included (norms, rotary, 20-row attention)"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

# in-place cache: the drafter's pattern (every depth call extends the previous present; here: the same prefix view again and again)
layer.inplace_cache = True
with torch.no_grad():
    _, pres = layer(x, attention_mask=mask, position_ids=pos, past_key_value=kv, use_cache=True)      # brings the external cache into the slab
prefix = (pres[0][:, :, :past], pres[1][:, :, :past])
def run_inplace():
    with torch.no_grad():
        return layer(x, attention_mask=mask, position_ids=pos, past_key_value=prefix, use_cache=True)[0]
_run = run
run = run_inplace
y_in = run().float()
t_in = timeit()
run = _run
layer.inplace_cache = False
out["hip_inplace_cache_us"] = t_in
out["speedup_inplace_cache"] = t_torch / t_in
out["max_rel_diff_inplace_vs_cat"] = float((y_in - y_hip).abs().max() / y_hip.abs().max())
print(json.dumps({k: out[k] for k in ("hip_inplace_cache_us", "speedup_inplace_cache", "max_rel_diff_inplace_vs_cat")}))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

# the drafting call with the tree block as ancestor words (what cnets.Model.forward hands over): attention on lantern_tree_attention
from lantern_amd import ops
bits, t1 = ops.drafter_tree_bits(torch.eye(n, device=dev)[None, None], n)
start = torch.zeros(2, dtype=torch.int64, device=dev)
def run_tree():
    with torch.no_grad():
        return layer(x, attention_mask=mask, position_ids=pos, past_key_value=kv, use_cache=True, tree_bits=bits, tree_keys=t1, kv_start=start)[0]
_run = run
run = run_tree
y_tr = run().float()
t_tr = timeit()
layer.inplace_cache = True
def run_tree_inplace():
    with torch.no_grad():
        return layer(x, attention_mask=mask, position_ids=pos, past_key_value=prefix, use_cache=True, tree_bits=bits, tree_keys=t1, kv_start=start)[0]
run = run_tree_inplace
run()
t_tri = timeit()
layer.inplace_cache = False
run = _run
out["hip_tree_attention_us"] = t_tr
out["hip_tree_attention_inplace_cache_us"] = t_tri
out["max_rel_diff_tree_vs_mask"] = float((y_tr - y_hip).abs().max() / y_hip.abs().max())
print(json.dumps({k: out[k] for k in ("hip_tree_attention_us", "hip_tree_attention_inplace_cache_us", "max_rel_diff_tree_vs_mask")}))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

# host time of the call sequence vs GPU completion (is the layer launch-bound?)
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): run()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"fused path: host loop {1e6*(t1-t0)/50:.0f} us/call, GPU done {1e6*(t2-t0)/50:.0f} us/call")
