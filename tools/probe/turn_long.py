"""Diagnostic: host time of every step() of a long unsynchronised run (is a single call blocking for milliseconds?) and the total.
usage: turn_long.py [steps] [commit_window]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lantern_amd import harness as HN
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 440
cw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = HN.WorkloadConfig(n_seq=64, pool_steps=16, n_groups=4, ep_kernel="chain", fuse_o7=True, spec_rows=3, commit_window=cw, max_steps=steps + 600)
wl = HN.LuminaVerifyWorkload(cfg, torch.device("cuda"))
wl.prime(0.3)
for _ in range(20):
    wl.step()
if os.environ.get("PROBE_JOIN") == "1":
    wl.join()
torch.cuda.synchronize()
ts = [time.perf_counter()]
for _ in range(steps):
    wl.step()
    ts.append(time.perf_counter())
torch.cuda.synchronize()
t_end = time.perf_counter()
d = [1e6 * (b - a) for a, b in zip(ts, ts[1:])]
big = [(i, round(x)) for i, x in enumerate(d) if x > 500]
marks = [50, 100, 200, 300, 360, 380, 400, 420, 440]
print("host clock at step marks (ms):", {m: round(1e3 * (ts[m] - ts[0]), 2) for m in marks if m < len(ts)})
print(f"commit_window {cw}: total {1e6 * (t_end - ts[0]) / steps:.1f} us/step; host per step median {sorted(d)[len(d) // 2]:.1f} us; steps whose call took > 500 us: {big[:20]}", flush=True)
wl.check_status(0, steps + 20)
# the image-end handling on its own, with a deep queue in front of it: join() + .item() on the current stream against a device-wide synchronise
for form in ("join + item", "device synchronize + item"):
    wl.reset_state()
    for _ in range(20):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        wl.step()
    t1 = time.perf_counter()
    for g in range(wl.G):
        s0, B = g * wl.Bg, wl.Bg
        nxt, base = wl.lens[0][2 * s0:2 * s0 + 2 * B], wl.len_base[2 * s0:2 * s0 + 2 * B]
        with torch.cuda.stream(wl.streams[g]):
            torch.where(nxt - base >= 10 ** 9, base, nxt, out=nxt)
    if form.startswith("join"):
        wl.join()
        wl._forked = True
    else:
        torch.cuda.synchronize()
    v = int((wl.lens[0] - wl.len_base).max().item())
    t4 = time.perf_counter()
    print(f"{form}: 300 steps enqueued in {1e3 * (t1 - t0):.2f} ms, read-back done after {1e3 * (t4 - t0):.2f} ms = {1e6 * (t4 - t0) / 300:.1f} us/step")
