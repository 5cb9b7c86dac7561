import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lantern_amd import harness as HN
dev = torch.device("cuda")
def run(fused, groups=2, fuse=True, spec=1, n_seq=6, steps=3):
    torch.manual_seed(3)
    cfg = HN.DynamicConfig(n_seq=n_seq, pool_steps=2, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 2, lantern_k=300, fuse_o7=fuse, n_groups=groups,
                           spec_rows=spec, fused_commit=fused)
    wl = HN.DynamicVerifyWorkload(cfg, dev)
    for _ in range(steps):
        wl.step(); wl.sync()
    r = (wl.log_best[:steps].clone().cpu(), wl.log_alen[:steps].clone().cpu(), wl.log_cnt[:steps].clone().cpu() if hasattr(wl, "log_cnt") else None)
    wl.release_kv(); del wl; torch.cuda.empty_cache()
    return r
for tag, a, b in (("sep/sep", False, False), ("fused/fused", True, True), ("fused/sep", True, False)):
    x, y = run(a), run(b)
    print(tag, "best equal", torch.equal(x[0], y[0]), "alen equal", torch.equal(x[1], y[1]))
    if not torch.equal(x[0], y[0]):
        print(x[0].tolist()); print(y[0].tolist())
for g in (1, 2):
    x, y = run(True, groups=g, spec=0), run(False, groups=g, spec=0)
    print("groups", g, "spec 0 fused/sep", torch.equal(x[0], y[0]))
