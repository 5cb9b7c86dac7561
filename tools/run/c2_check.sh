#!/bin/bash
# BASELINE config 2 (LlamaGen dynamic trees): the loop tests, then the C2 bench object with 2 / 0 / 1 rows prepared beside the tree build
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-c2}
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_loop.py tests/test_gpu_generate_lg.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -40 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for sp in 2 0 1 2; do
LANTERN_C2_SPEC_ROWS=$sp timeout -k 10 400 python3 - <<'PY' > $O/c2_$sp.txt 2>&1 || { tail -20 $O/c2_$sp.txt; exit 1; }
import sys, json, torch
sys.path.insert(0, ".")
import bench
from lantern_amd import harness as HN
r = bench.other_configs(torch.device("cuda"), HN.WorkloadConfig(n_seq=63, n_groups=3), 200, 63, only="C2")["C2"]
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "mean_accept_length", "kernel_ms", "tree_decoding_rows")}))
PY
echo "spec=$sp $(tail -1 $O/c2_$sp.txt)"
done
