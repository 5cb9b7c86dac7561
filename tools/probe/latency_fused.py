"""Step latency at the reference's own batch sizes (1 and 8 sequences, one stream group): the node-parallel form, the chain on raw rows with three launches
(3 rows prepared) and the two-launch chain step with 0 / 2 / 4 helper rows; us per step of 100 timed steps, twice."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch
from lantern_amd import harness as HN
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = {}
for rep in range(2):
    for B in (1, 8):
        for name, kw in (("nodes", dict(ep_kernel="nodes")), ("chain_3_launches_spec3", dict(ep_kernel="chain", fuse_o7=True, spec_rows=3)),
                         ("chain_2_launches_no_helper", dict(ep_kernel="chain", fuse_o7=True, spec_rows=1, fused_prepare=True)),
                         ("chain_2_launches_spec3", dict(ep_kernel="chain", fuse_o7=True, spec_rows=3, fused_prepare=True)),
                         ("chain_2_launches_spec5", dict(ep_kernel="chain", fuse_o7=True, spec_rows=5, fused_prepare=True))):
            wl = HN.LuminaVerifyWorkload(HN.WorkloadConfig(n_seq=B, pool_steps=4, sigma=5.0, max_steps=140, **kw), dev)
            wl.prime(0.2)
            for _ in range(10):
                wl.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                wl.step()
            torch.cuda.synchronize()
            out.setdefault(f"B{B}_{name}", []).append(round(1e6 * (time.perf_counter() - t0) / 100, 1))
            wl.check_status(0, 110)
            wl.release_kv(); del wl; torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)
