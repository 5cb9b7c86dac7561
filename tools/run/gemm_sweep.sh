#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/gemm
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_drafter_layer.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -40 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for nb in 2 3 4; do for g in 256 512; do
  echo "== packed nbuf=$nb groups=$g"; GEMM_FORM=packed LANTERN_SK_NBUF=$nb LANTERN_SK_GROUPS=$g timeout -k 10 300 python3 tools/gemm_bench.py 20 2>&1 | grep -v amdgpu
done; done
