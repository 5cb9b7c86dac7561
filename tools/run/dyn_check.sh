#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2dy}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_loop.py tests/test_gpu_window.py -x -q -m gpu -k "dynamic or raw or fused" > $OUT/tests.log 2>&1
rc=$?; tail -4 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python - <<PY
import torch, bench, json
from lantern_amd import harness as HN
dev = torch.device("cuda")
cfg = HN.WorkloadConfig(n_seq=64)
for fuse in (True, False):
    r = bench.dynamic_run(dev, cfg, 100, 64, fuse_o7=fuse)
    print(fuse, round(r["value"]), round(r["ms_per_step"]*1e3,1), r["mean_accept_length"], r["per_step"], {k: round(v*1e3,1) for k,v in r["kernel_ms"].items()}, flush=True)
PY
