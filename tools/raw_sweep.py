"""evaluate_posterior on RAW cond / uncond bf16 rows (LANTERN_ROWS_RAW_BF16: the row post-process of the visited rows inside the chain kernel) at
batch sizes beyond the CU count -- the throughput instances of epw_throughput.hip (LANTERN_EPW_TP_RAW=512 selects the 512-thread form,
LANTERN_EPW_TP=0 the generic two-per-CU instance).  Kernel-only durations from the launch's own HIP events.
usage: raw_sweep.py <batches, e.g. 512,2048> [steps] [spec_rows]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from lantern_amd import harness as HN
from lantern_amd import _lib as _L
_KNOBS = _L.tuning_from_env()          # LANTERN_<NAME>=<int> of this tool's environment -> explicit lantern_tuning_set calls

batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,2048").split(",") if x]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
spec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = []
for B in batches:
    S = 2 if B <= 2048 else 1
    cfg = HN.WorkloadConfig(n_seq=B, pool_steps=S, with_kv=False, max_steps=3 * steps + 8, ep_kernel="chain", fuse_o7=True, spec_rows=spec, n_groups=1)
    wl = HN.LuminaVerifyWorkload(cfg, dev)
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    names = bench.event_names(wl)
    evs = bench.make_events(names, steps, dev)
    for i in range(steps):
        wl.step(evs[i], serial=True)
    torch.cuda.synchronize()
    wl.check_status(0, 2 * steps)
    ms = float(np.median([e["evaluate_posterior"][0].elapsed_time(e["evaluate_posterior"][1]) for e in evs]))
    cnt = wl.log_cnt[steps:2 * steps]
    needed = wl.ep_window_bytes_from(cnt) / steps
    out.append({"sequences_per_launch": B, "pool_steps": S, "prepared_rows": wl.n_spec, "launch_ms": ms, "us_per_sequence": 1e3 * ms / B,
                "needed_bytes_per_launch_upper": needed, "achieved_GBps_upper": needed / (ms * 1e-3) / 1e9,
                "per_step": {k: float(cnt[..., j].float().mean()) for j, k in enumerate(("levels", "tried", "rejected"))}})
    del wl
    torch.cuda.empty_cache()
print(json.dumps({"knobs": {k: os.environ.get(k) for k in ("LANTERN_EPW_TP", "LANTERN_EPW_TP_RAW", "LANTERN_EPW_SPEC")}, "raw_sweep": out}))
