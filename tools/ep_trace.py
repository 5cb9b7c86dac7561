"""Diagnostic: per-phase cycle stamps of epw_kernel from a separate -DEPW_TRACE=<1|2> build (1 = phase level, 2 = also
inside the phases).  Every workgroup stamps into LDS; prints the slowest and the median sequence.  Not a benchmark."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
level = os.environ.get("EPW_TRACE", "1")
so = os.path.join(ROOT, "tools", f"liblantern_trace{level}.so")
if not os.path.exists(so) or (len(sys.argv) > 1 and sys.argv[1] == "build"):
    src = os.path.join(ROOT, "lantern_amd", "csrc")
    files = [os.path.join(src, f) for f in ("evaluate_posterior.hip", "logits_post.hip", "window_kernels.hip", "epw_generic.hip", "epw_throughput.hip", "node_kernels.hip", "gather_ops.hip", "tree_dynamic.hip", "greedy.hip", "drafter_fc.hip", "vq_table.hip", "tree_attention.hip", "harness_util.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", f"-DEPW_TRACE={level}",
                           "-o", so] + files + ["-x", "hip", os.path.join(src, "tree_static.cpp"), os.path.join(src, "verify_step.cpp")])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
import numpy as np
import torch
from lantern_amd import _lib
_lib.LIB_PATH = so
from lantern_amd import harness as HN
B = int(os.environ.get("EPW_B", "48"))
MODE = os.environ.get("EPW_MODE", "chain")      # "chain": probability rows from O7; "raw": raw bf16 rows + 3 rows up front (the bench default)
kw = dict(ep_kernel="chain")
if MODE == "raw":
    kw.update(fuse_o7=True, spec_rows=int(os.environ.get("EPW_SPEC", "3")))
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=4, with_kv=False, max_steps=64, use_graph=False, **kw), torch.device("cuda"))
L = wl._L
NAMES = {0: "start", 1: "staged(loads issued)", 2: "staged(barrier)", 10: "level: masks+prefetch", 11: "level: softmax", 12: "  softmax: row loaded+local max", 13: "  softmax: block max",
         14: "  softmax: exp+local sum", 15: "  softmax: block sum", 20: "cand: start", 21: "cand: decision (all waves)", 22: "  scan: gathers", 23: "  scan: dpp scan", 24: "  scan: checks",
         25: "  scan: wave reduce", 26: "  wave0 decision written", 27: "  wave0 section entered", 31: "  rej: siblings+nb zeroing", 32: "  rej: q zero+sum (waits q)", 33: "  rej: block sum qs", 34: "  rej: residual pass", 35: "  rej: block sum tot", 16: "  level: loop head", 17: "  level: per-lane path data", 18: "  level: candidate list", 30: "reject: residual", 81: "  row: loaded + CFG mix", 82: "  row: hist cleared", 83: "  row: radix pass 0 (atomics)", 84: "  row: hist merged", 85: "  row: radix pass 1 (atomics)", 86: "  row: threshold applied", 88: "  row: softmax", 40: "epilogue start", 50: "epilogue done"}
agg = {}
for step in range(8):
    wl.step(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 256))(); cnt = (C.c_int * 64)()
    assert L.lantern_debug_epw_trace(buf, cnt) == 256
    a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 256)
    tot = []
    for b in range(min(B, 64)):
        n = cnt[b]
        ids = (a[b, :n] >> np.uint64(56)).astype(int); t = (a[b, :n] & np.uint64((1 << 56) - 1)).astype(np.int64)
        tot.append(int(t[-1] - t[0]))
        for i in range(1, n):
            agg.setdefault(ids[i], []).append(int(t[i] - t[i - 1]))
    order = np.argsort(tot)
    if step >= 6:
        for tag, b in (("slowest", order[-1]), ("median", order[len(order) // 2])):
            n = cnt[b]
            ids = (a[b, :n] >> np.uint64(56)).astype(int); t = (a[b, :n] & np.uint64((1 << 56) - 1)).astype(np.int64)
            print(f"--- step {step} {tag} seq {b}: {tot[b]} cycles, counters {wl.st_cnt[b].tolist()}")
            for i in range(1, n):
                print(f"   {NAMES.get(ids[i], ids[i]):32s} +{int(t[i] - t[i - 1]):6d}")
    print(f"step {step}: per-seq cycles min {min(tot)} median {int(np.median(tot))} max {max(tot)}")
print("mean cycles per stamp interval (all sequences, all steps):")
for i in sorted(agg):
    v = np.array(agg[i])
    print(f"   {NAMES.get(i, i):32s} n={len(v):5d} mean {v.mean():8.0f}  p50 {np.median(v):8.0f}  max {v.max():7d}   total share {v.sum():10d}")
