#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2hg}
mkdir -p $OUT
for a in "1 chain 1 3 0" "1 chain 1 3 1" "3 chain 1 3 0" "3 chain 1 3 3" "4 chain 1 3 0" "4 chain 1 3 4" "4 chain 1 3 2" "4 chain 1 3 4 32" "4 chain 1 3 4 16" "2 chain 1 3 2 32" "1 chain 1 3 0 16" "1 chain 1 3 0 8"; do
  timeout -k 10 200 python tools/host_vs_gpu.py $a 2>/dev/null | tee -a $OUT/hostgpu.txt
done
