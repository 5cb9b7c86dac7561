#!/bin/bash
# wider randomized soaks (static / dynamic kernel sets, node kernels) + long step loops of the optional launch paths against the CPU oracle
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2soak2}
mkdir -p $OUT
timeout -k 10 400 python tests/fuzz_soak.py 60 > $OUT/static_soak.txt 2>&1; tail -1 $OUT/static_soak.txt
timeout -k 10 400 python tests/fuzz_soak.py 60 dynamic > $OUT/dynamic_soak.txt 2>&1; tail -1 $OUT/dynamic_soak.txt
timeout -k 10 400 python tests/fuzz_soak.py 80 nodes > $OUT/nodes_soak.txt 2>&1; tail -1 $OUT/nodes_soak.txt
timeout -k 10 300 python tests/fuzz_soak.py 300 streamk > $OUT/streamk_soak.txt 2>&1; tail -1 $OUT/streamk_soak.txt
