import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lantern_amd import harness as HN
def run(fuse, spec, steps=12):
    cfg = HN.WorkloadConfig(n_seq=6, pool_steps=4, kv_layers=2, kv_heads=4, kv_smax=512, max_steps=steps + 4, sigma=5.0, ep_kernel="chain", fuse_o7=fuse, spec_rows=spec)
    wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
    for _ in range(steps): wl.step()
    torch.cuda.synchronize()
    return wl.log_best[:steps].clone(), wl.log_alen[:steps].clone(), wl.cond.float().sum().item(), wl.ss_token.sum().item()
a = run(False, 0); b = run(True, 0); c = run(True, 5); d = run(False, 5); e = run(False, 0)
for n, x in zip("abcde", (a, b, c, d, e)):
    print(n, x[0][0].tolist(), x[0][1].tolist(), x[2], x[3])
