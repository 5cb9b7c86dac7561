"""C2 (LlamaGen dynamic, 64 sequences in 4 stream groups): is the step host-bound?  Enqueue time of 200 steps (no synchronise inside) against their wall time, and
the same with 1 / 2 groups of 16."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from lantern_amd import harness as HN
dev = torch.device("cuda", 0)
out = {}
for model in ("llamagen", "llamagen", "lumina"):
    for groups, n_seq in ((1, 16), (2, 32), (4, 64)):
        kw = dict(depth=4, kv_layers=12, kv_heads=12, kv_dim=64) if model == "llamagen" else {}
        dc = HN.DynamicConfig(model=model, n_seq=n_seq, total_tokens=58, kv_smax=4096, with_kv=True, max_steps=700, plausible=8.0, n_groups=groups, fuse_o7=True, spec_rows=2, **kw)
        wl = HN.DynamicVerifyWorkload(dc, dev)
        for _ in range(8):
            wl.step()
        torch.cuda.synchronize()
        res = []
        NS = 30 if model == "llamagen" else 200          # (LlamaGen: 256 tokens per image, the harness's image-end bound synchronises from step 42 on)
        for rep in range(1 if model == "llamagen" else 2):
            t0 = time.perf_counter()
            for _ in range(NS):
                wl.step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.append((round(1e6 * (t1 - t0) / NS, 1), round(1e6 * (t2 - t0) / NS, 1)))
        out.setdefault(f"{model}_g{groups}", []).extend(res)
        print(model, groups, res, flush=True)
        wl.release_kv(); del wl; torch.cuda.empty_cache()
print(json.dumps(out))
