#!/usr/bin/env python3
"""Phase stamps of the node-parallel evaluate_posterior (lantern_debug_epn_trace): per-workgroup cycle counts of the
prologue, every candidate's scan / rejection, and the bonus draw, on one verify step of the bench workload.
Usage: python tools/epn_trace.py [n_seq] [steps] [nodes|walk]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lantern_amd import harness as HN, _lib

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
ep = sys.argv[3] if len(sys.argv) > 3 else "nodes"
cfg = HN.WorkloadConfig(n_seq=n_seq, pool_steps=2, with_kv=False, max_steps=64, ep_kernel=ep)
wl = HN.LuminaVerifyWorkload(cfg, dev)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
L = _lib.lib()
N = wl.N
grid = n_seq * N if ep == "nodes" else n_seq
buf = torch.zeros((grid, 64), dtype=torch.int64, device=dev)
names = {0: "start", 1: "loads issued", 2: "row+tables in LDS", 3: "S_q barrier", 4: "ids staged", 10: "cand start", 11: "decision", 12: "residual pass", 13: "block sum", 14: "normalised", 20: "loop end", 21: "bonus done"}
allrows = []
for s in range(steps):
    buf.zero_()
    L.lantern_debug_epn_trace(C.c_void_p(buf.data_ptr()))
    wl.step()
    torch.cuda.synchronize()
    L.lantern_debug_epn_trace(None)
    a = buf.cpu().numpy().astype(np.uint64)
    for w in range(grid):
        n = int(a[w, 0] & 0xffffffff); node = int(a[w, 0] >> 32)
        st = [(int(v >> 56), int(v & ((1 << 56) - 1))) for v in a[w, 1:1 + n]]
        allrows.append((node, st))
# per-phase durations
tot = sorted(((st[-1][1] - st[0][1]), node, st) for node, st in allrows if len(st) > 1)
print("workgroups", len(tot), "total cycles: median", tot[len(tot) // 2][0], "p90", tot[int(len(tot) * .9)][0], "max", tot[-1][0])
seg = {}
for _, node, st in tot:
    for (i0, t0), (i1, t1) in zip(st[:-1], st[1:]):
        seg.setdefault((i0, i1), []).append(t1 - t0)
for k in sorted(seg):
    v = np.array(seg[k])
    print(f"{names.get(k[0], k[0]):>20} -> {names.get(k[1], k[1]):<20} n={len(v):6d} median {int(np.median(v)):6d} p90 {int(np.percentile(v, 90)):6d} max {int(v.max()):6d}")
print("slowest workgroup: node", tot[-1][1])
t0 = tot[-1][2][0][1]
print([(names.get(i, i), t - t0) for i, t in tot[-1][2]])
first = min(st[0][1] for _, st in allrows[-grid:] if st)
last = max(st[-1][1] for _, st in allrows[-grid:] if st)
print("last step: first start .. last end =", last - first, "cycles")
starts = sorted(st[0][1] - first for _, st in allrows[-grid:] if st)
print("start offsets: median", starts[len(starts) // 2], "p90", starts[int(len(starts) * .9)], "max", starts[-1])
