#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/nodes
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_nodes.py tests/test_gpu_generate_ref.py tests/test_gpu_generate.py tests/test_gpu_mirror.py tests/test_gpu_loop.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -40 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for sp in 2 0; do
LANTERN_EPW_SPEC=$sp timeout -k 10 300 python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, json, os, torch
sys.path.insert(0, ".")
import bench
from lantern_amd import harness as HN
from lantern_amd import _lib
_lib.tuning_from_env()
dev = torch.device("cuda")
base = HN.WorkloadConfig(n_seq=63, n_groups=3)
r = bench.step_latency(dev, base, (1, 8), 200)
print("SPEC", os.environ.get("LANTERN_EPW_SPEC"), {k: round(v["us_per_step"], 1) for k, v in r.items()})
PY
done
