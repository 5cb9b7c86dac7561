"""lantern_tree_attention at the drafting shape (2 batch rows x 32 heads, N tree rows, ~1200 cached keys): microseconds per call (HIP events around 200
back-to-back calls on rotating caches) for the split policy given in the environment (LANTERN_TA_SPLITS, LANTERN_TA_MIN_TILES).  Diagnostic.
usage: ta_draft_shape.py [N=10] [S=1210]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lantern_amd import ops
from lantern_amd import _lib as _L
_KNOBS = _L.tuning_from_env()          # LANTERN_<NAME>=<int> of this tool's environment -> explicit lantern_tuning_set calls
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1210
B, Hq, d, R = 2, 32, 128, 8
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(B, N, Hq, d, generator=g, device="cuda").to(torch.bfloat16)
ks = [torch.randn(B, Hq, S + 6, d, generator=g, device="cuda").to(torch.bfloat16) for _ in range(R)]
vs = [torch.randn(B, Hq, S + 6, d, generator=g, device="cuda").to(torch.bfloat16) for _ in range(R)]
bits = ops.tree_mask_bits(torch.tril(torch.ones(N, N, device="cuda")))
out = torch.empty(B, N, Hq * d, dtype=torch.bfloat16, device="cuda")
for i in range(20):
    ops.tree_attention(q, ks[i % R], vs[i % R], bits, max_kv_len=S, out=out)
torch.cuda.synchronize()
n = 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    ops.tree_attention(q, ks[i % R], vs[i % R], bits, max_kv_len=S, out=out)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"N": N, "S": S, "splits": os.environ.get("LANTERN_TA_SPLITS"), "min_tiles": os.environ.get("LANTERN_TA_MIN_TILES"),
                  "us_per_call": 1e3 * e0.elapsed_time(e1) / n}))
