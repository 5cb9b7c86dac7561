#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/gemm
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_drafter_layer.py tests/test_gpu_drafter.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -40 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for g in ${SK_GROUPS:-0}; do
  echo "== stream-K packed groups=$g"; GEMM_FORM=packed LANTERN_SK_GROUPS=$g timeout -k 10 300 python3 tools/gemm_bench.py 20 $O/packed_$g.json 2>&1 | tee $O/packed_$g.txt
done
echo "== stream-K row-major"; GEMM_FORM=streamk timeout -k 10 300 python3 tools/gemm_bench.py 20 $O/streamk.json 2>&1 | tee $O/streamk.txt
echo "== per-tile"; GEMM_FORM=tile timeout -k 10 300 python3 tools/gemm_bench.py 20 $O/tile.json 2>&1 | tee $O/tile.txt
