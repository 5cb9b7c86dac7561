"""evaluate_posterior alone at given batch sizes, on ROTATING inputs (bench.py's ep_batch_sweep, stand-alone): the target of the
rocprofv3 --kernel-trace --stats / --pmc passes at the saturating batch (tools/run/ep_sweep_prof.sh) and of tuning runs
(LANTERN_EPW_TP / LANTERN_EPW_OCC2 select the kernel instance).

usage: ep_sweep.py <batches, e.g. 512,4096> [iters] [kernels, e.g. chain or chain,nodes]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from lantern_amd import harness as HN
from lantern_amd import _lib as _L
_KNOBS = _L.tuning_from_env()          # LANTERN_<NAME>=<int> of this tool's environment -> explicit lantern_tuning_set calls

batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,4096").split(",") if x]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 24
kernels = tuple((sys.argv[3] if len(sys.argv) > 3 else "chain").split(","))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
base = HN.WorkloadConfig()
res = bench.ep_batch_sweep(batches, dev, base, iters=iters, kernels=kernels)
print(json.dumps({"knobs": {k: os.environ.get(k) for k in ("LANTERN_EPW_TP", "LANTERN_EPW_OCC2", "LANTERN_EPW_SPEC")}, "sweep": res}))
