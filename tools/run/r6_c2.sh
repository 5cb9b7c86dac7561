#!/bin/bash
# C2 (LlamaGen KV geometry: 288 row groups x 8 chunks per slab) by the commit kernel's form: slab blocks (kv_ks 4 / 1) against the tiled mover (kv_ks 0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6c2}; mkdir -p $O
for ks in 4 0 1 4 0; do
timeout -k 10 400 python3 - $ks <<'PY' > $O/c2_$ks.txt 2>&1 || { tail -20 $O/c2_$ks.txt; exit 1; }
import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
sys.path.insert(0, ".")
import torch
import bench
from lantern_amd import harness as HN, _lib
_lib.set_tuning("kv_ks", int(sys.argv[1]))
r = bench.other_configs(torch.device("cuda"), HN.WorkloadConfig(n_seq=64, n_groups=4), 100, 64, only="C2")["C2"]
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "mean_accept_length", "kernel_ms")}))
PY
echo "kv_ks=$ks $(tail -1 $O/c2_$ks.txt)"
done
