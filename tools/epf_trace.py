"""Diagnostic: per-phase cycle stamps of the fast-walk kernel (walk_kernel.hip) from a separate -DEPF_TRACE build.  Every workgroup
stamps into LDS; prints the slowest and the median sequence and the mean cost of every phase.  Not a benchmark.
env: EPW_B (sequences per launch, default 64), EPW_MODE = chain (probability rows from O7) | raw (raw bf16 rows + EPW_SPEC rows up front)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "liblantern_tracef.so")
if not os.path.exists(so) or (len(sys.argv) > 1 and sys.argv[1] == "build"):
    src = os.path.join(ROOT, "lantern_amd", "csrc")
    files = [os.path.join(src, f) for f in ("evaluate_posterior.hip", "logits_post.hip", "window_kernels.hip", "node_kernels.hip", "walk_kernel.hip", "gather_ops.hip", "tree_dynamic.hip", "greedy.hip", "drafter_fc.hip", "vq_table.hip", "tree_attention.hip", "harness_util.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-DEPF_TRACE=1",
                           "-o", so] + files + ["-x", "hip", os.path.join(src, "tree_static.cpp"), os.path.join(src, "verify_step.cpp")])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
import numpy as np
import torch
from lantern_amd import _lib
_lib.LIB_PATH = so
from lantern_amd import harness as HN
B = int(os.environ.get("EPW_B", "64"))
MODE = os.environ.get("EPW_MODE", "chain")
kw = dict(ep_kernel="fast")
if MODE == "raw":
    kw.update(fuse_o7=True, spec_rows=int(os.environ.get("EPW_SPEC", "3")))
wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=4, with_kv=False, max_steps=64, use_graph=False, **kw), torch.device("cuda"))
L = wl._L
NAMES = {0: "start", 1: "staged (barrier)", 2: "dup flags + root requests", 3: "  pro: root requests issued", 4: "  pro: uniforms requested", 5: "  pro: staging loads issued", 6: "  pro: staged in LDS (w0)", 10: "arrival: row in LDS", 20: "cand: start", 21: "cand: verdict (B3 passed)", 22: "  w0: gathers done, B1 passed", 23: "  w0: prefix chain, B2 passed", 24: "  w0: scan done, verdict written", 30: "rejection done", 40: "walk end", 50: "epilogue done"}
agg = {}
for step in range(8):
    wl.step(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 160))(); cnt = (C.c_int * 64)()
    assert L.lantern_debug_epf_trace(buf, cnt) == 160
    a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 160)
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
            print(f"--- step {step} {tag} seq {b}: {tot[b]} cycles")
            for i in range(1, n):
                print(f"   {NAMES.get(ids[i], ids[i]):32s} +{int(t[i] - t[i - 1]):6d}")
    print(f"step {step}: per-seq cycles min {min(tot)} median {int(np.median(tot))} max {max(tot)}")
# the load / pass waves' own timelines of the last step, one sequence (absolute cycles from wave 0's first stamp)
RN = {60: "L: at cmd barrier", 61: "L: cmd read", 62: "L: loads issued", 63: "L: ids staged", 64: "L: dma landed", 65: "L: verdict passed", 66: "L: neighbours zeroed",
      70: "P: at cmd barrier", 71: "P: cmd read", 72: "P: pass done", 73: "P: verdict passed", 74: "P: residual written", 75: "P: rejection barriers passed"}
bsel = int(order[len(order) // 2])
n0 = cnt[bsel]
t00 = int(a[bsel, 0] & np.uint64((1 << 56) - 1))
print(f"--- wave 0 of seq {bsel} (absolute):")
for i in range(n0):
    print(f"   {NAMES.get(int(a[bsel, i] >> np.uint64(56)), '?'):32s} @{int(a[bsel, i] & np.uint64((1 << 56) - 1)) - t00:7d}")
for role in (0, 1):
    bufr = (C.c_ulonglong * (64 * 96))(); cntr = (C.c_int * 64)()
    assert L.lantern_debug_epf_trace_role(role, bufr, cntr) == 96
    ar = np.frombuffer(bufr, dtype=np.uint64).reshape(64, 96)
    print(f"--- role {role} of seq {bsel}:")
    for i in range(cntr[bsel]):
        print(f"   {RN.get(int(ar[bsel, i] >> np.uint64(56)), '?'):32s} @{int(ar[bsel, i] & np.uint64((1 << 56) - 1)) - t00:7d}")
print("mean cycles per stamp interval (all sequences, all steps):")
for i in sorted(agg):
    v = np.array(agg[i])
    print(f"   {NAMES.get(i, i):32s} n={len(v):5d} mean {v.mean():8.0f}  p50 {np.median(v):8.0f}  max {v.max():7d}   total share {v.sum():10d}")
