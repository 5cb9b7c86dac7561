"""C4 (Anole static tree naive_extend_57, 56 sequences in 4 stream groups, lambda = 5 / k = 10): three launches per group and step against the two-launch step with
helper rows (LANTERN_STEP_FUSED_PREPARE, 3 listed rows) and without helpers (the root alone), alternating in one process; us per step of 100 timed steps."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from lantern_amd import harness as HN
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
base = HN.WorkloadConfig(n_seq=56, pool_steps=8, sigma=5.0, n_groups=4)
out = {}
for rep in range(2):
    for name, kw in (("three_launches_spec3", dict(fused_prepare=False, spec_rows=3)), ("two_launches_spec3", dict(fused_prepare=True, spec_rows=3)),
                     ("two_launches_spec2", dict(fused_prepare=True, spec_rows=2)), ("two_launches_no_helper", dict(fused_prepare=True, spec_rows=1))):
        r = bench.side_run(dev, base, 100, model="anole", tree="naive_extend_57", lantern_k=10, lantern_delta=5.0, fuse_o7=True, ep_kernel="chain", seed_base=4000, n_seq=56, n_groups=4, **kw)
        out.setdefault(name, []).append(round(1e3 * r["ms_per_step"], 2))
        print(name, out[name], flush=True)
print(json.dumps(out))
