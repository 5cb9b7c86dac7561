"""Diagnostic: phase stamps of epw_kernel (workgroup 0) from a separate -DEPW_TRACE build.  Not a benchmark."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "liblantern_trace.so")
if not os.path.exists(so):
    src = os.path.join(ROOT, "lantern_amd", "csrc")
    files = [os.path.join(src, f) for f in ("evaluate_posterior.hip", "logits_post.hip", "window_kernels.hip", "gather_ops.hip", "tree_dynamic.hip", "greedy.hip", "drafter_fc.hip", "vq_table.hip", "harness_util.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-DEPW_TRACE",
                           "-o", so] + files + ["-x", "hip", os.path.join(src, "tree_static.cpp")])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
import torch
from lantern_amd import _lib
_lib.LIB_PATH = so
from lantern_amd import harness as HN
from lantern_amd._lib import check
B = 32
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=2, with_kv=False, max_steps=64, use_graph=False), torch.device("cuda"))
L = wl._L
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 4096)()
L.lantern_debug_epw_trace(buf, 2048)   # drop earlier stamps
wl.step(); torch.cuda.synchronize()
n = L.lantern_debug_epw_trace(buf, 2048)
names = {1: "staged(loads issued)", 2: "staged(barrier)", 10: "level: masks done", 11: "level: softmax done", 12: "  softmax: row loaded+local max", 13: "  softmax: block max", 14: "  softmax: exp+local sum", 15: "  softmax: block sum", 20: "cand: start", 21: "cand: scan done",
         22: "  scan: gathers done", 23: "  scan: dpp scan done", 24: "  scan: checks done", 25: "  scan: wave reduce done", 26: "  wave0 decision written", 30: "reject: residual done", 40: "epilogue start", 50: "epilogue done"}
t0 = buf[1]
prev = t0
for i in range(n):
    pid, t = buf[2 * i], buf[2 * i + 1]
    print(f"{names.get(pid, pid):28s} +{(t - prev):7d} cyc   t={(t - t0) / 100.0:8.2f} us(100MHz ticks?)")
    prev = t
print("counters seq0:", wl.st_cnt[0].tolist())
