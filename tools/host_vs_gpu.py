#!/usr/bin/env python3
"""Host enqueue time vs GPU completion time of the step loop (is a configuration launch-bound or GPU-bound?).
Usage: python tools/host_vs_gpu.py [groups] [ep] [fuse]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ep = sys.argv[2] if len(sys.argv) > 2 else "chain"
fuse = len(sys.argv) > 3 and sys.argv[3] == "1"
n = 64 - 64 % G
cfg = HN.WorkloadConfig(n_seq=n, pool_steps=8, n_groups=G, ep_kernel=ep, fuse_o7=fuse, max_steps=400)
wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
wl.prime()
for _ in range(20):
    wl.step()
wl.join(); torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    wl.step()
t1 = time.perf_counter()
wl.join(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"G={G} ep={ep} fuse={fuse}: host enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
