"""evaluate_posterior alone for LlamaGen's standard verify (BASELINE config 2: V = window = 16384 ids, EAGLE-2 trees of 59 nodes, LANTERN off) at batches beyond the
CU count: the two-per-CU throughput instance `epw_kernel<512, 8, 1, 4, true, false, 5, ..>` (lantern_tuning_set("epw_tp_lg", 1), the default: rows by LDS-DMA) against its
variants (2: + second LDS pass for the residual, 3: rows through registers, 4: + raised priority) and the generic one-per-CU instance (0), alternating inside one process.  Probability rows (O7 over all 59 rows of every
sequence, its own launch) -- 3.9 MB per sequence and step, so one step's rows (16 GB at 4096 sequences) never sit in a cache.  HIP events around the launch
(lantern_profile_next_launch); `frac` = (visited levels + fresh final rows) x 64 KB / time / 8 TB/s.

usage: lg_sweep.py <batches, e.g. 512,4096> [steps per variant] [variants, e.g. 1,2,0]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from lantern_amd import _lib as _L
from lantern_amd import harness as HN

batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,4096").split(",") if x]
KE = int(sys.argv[2]) if len(sys.argv) > 2 else 12
variants = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,3,2,4,0").split(",")]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = []
for B in batches:
    dc = HN.DynamicConfig(model="llamagen", n_seq=B, depth=4, total_tokens=58, kv_layers=12, kv_heads=12, kv_dim=64, with_kv=False, pool_steps=2,
                          max_steps=2 * len(variants) * (KE + 2) + 16, plausible=8.0, n_groups=1, fuse_o7=False, native_step=False)
    wl = HN.DynamicVerifyWorkload(dc, dev)
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize(dev)
    names = bench.event_names(wl)
    row = {"sequences_per_launch": B, "window": wl.W, "nodes": wl.N, "variants": {}}
    step = 2
    for rep in range(2):
        for v in variants:
            _L.set_tuning("epw_tp_lg", v)
            wl.step()
            step += 1
            evs = bench.make_events(names, KE, dev)
            for i in range(KE):
                wl.step(evs[i])
            torch.cuda.synchronize(dev)
            cnt = wl.log_cnt[step:step + KE, :wl.Bg].double()
            step += KE
            ms = float(np.median([e["evaluate_posterior"][0].elapsed_time(e["evaluate_posterior"][1]) for e in evs]))
            needed = float(((cnt[..., 0] + (1.0 - cnt[..., 4])) * wl.W * 4).sum() / KE)
            o7 = float(np.median([e["cfg_mask_topk"][0].elapsed_time(e["cfg_mask_topk"][1]) for e in evs]))
            row["variants"].setdefault(str(v), []).append({"launch_us": 1e3 * ms, "needed_bytes_per_launch": needed, "achieved_GBps": needed / (ms * 1e-3) / 1e9,
                                                           "frac": needed / (ms * 1e-3) / 1e9 / 8000.0, "levels_per_sequence": float(cnt[..., 0].mean()),
                                                           "cfg_mask_topk_us": 1e3 * o7, "cfg_mask_topk_frac": B * wl.N * wl.W * (2 * 2 + 4) / (o7 * 1e-3) / 8e12})
    _L.set_tuning("epw_tp_lg", 1)
    wl.check_status(0, step)
    out.append(row)
    print(json.dumps(row), flush=True)
    del wl
    torch.cuda.empty_cache()
print(json.dumps({"lg_sweep": out}))
