#!/usr/bin/env python3
"""Host enqueue time vs GPU completion time of the step loop (is a configuration launch-bound or GPU-bound?).
Usage: python tools/host_vs_gpu.py [groups] [ep] [fuse] [spec_rows] [n_seq]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lantern_amd import harness as HN

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ep = sys.argv[2] if len(sys.argv) > 2 else "chain"
fuse = len(sys.argv) > 3 and sys.argv[3] == "1"
spec = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = int(sys.argv[5]) if len(sys.argv) > 5 else 64
n -= n % G
cfg = HN.WorkloadConfig(n_seq=n, pool_steps=8, n_groups=G, ep_kernel=ep, fuse_o7=fuse, spec_rows=spec, max_steps=400)
wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
wl.prime()
for _ in range(20):
    wl.step()
wl.sync()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    wl.step()
t1 = time.perf_counter()
wl.sync()
t2 = time.perf_counter()
print(f"G={G} n={n} ep={ep} fuse={fuse} spec={spec}: python loop {1e6*(t1-t0)/K:.1f} us/step, GPU done {1e6*(t2-t0)/K:.1f} us/step", flush=True)
