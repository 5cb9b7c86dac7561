#!/bin/bash
# a longer randomized soak against the CPU oracle on the final library (tests/fuzz_soak.py; ~10 minutes): every line prints cases / fails
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak_r06_long; mkdir -p $O
{
timeout -k 10 400 python tests/fuzz_soak.py 240 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 160 dynamic 2>&1 | tail -1
timeout -k 10 200 python tests/fuzz_soak.py 3000 o7 2>&1 | tail -1
timeout -k 10 200 python tests/fuzz_soak.py 6000 o3 2>&1 | tail -1
timeout -k 10 200 python tests/fuzz_soak.py 40 top_p 2>&1 | tail -1
timeout -k 10 200 python tests/fuzz_soak.py 3000 draws 2>&1 | tail -1
} | tee $O/soak.txt
