#!/bin/bash
# dynamic-tree step through lantern_verify_step: tests, then the dynamic bench legs alone
cd $GRAFT_REPO_ROOT
O=gpurun_out/dyn
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_loop.py -x -q -m gpu > $O/t_loop.txt 2>&1 || { tail -40 $O/t_loop.txt; exit 1; }
tail -2 $O/t_loop.txt
timeout -k 10 600 python3 - <<'PY' > $O/dyn.txt 2>&1
import sys, json, torch
sys.path.insert(0, ".")
import bench
from lantern_amd import harness as HN
dev = torch.device("cuda")
base = HN.WorkloadConfig(n_seq=63, n_groups=3)
for fuse, g, spec in ((True, 3, 2), (True, 3, 1), (True, 3, 0), (True, 1, 2), (False, 3, 0)):
        r = bench.dynamic_run(dev, base, 200, 63, fuse_o7=fuse, groups=g, spec_rows=spec)
        print(json.dumps({k: r[k] for k in ("workload", "value", "ms_per_step", "kernel_ms", "tree_decoding_rows")}), flush=True)
PY
cat $O/dyn.txt
