#!/bin/bash
# a longer randomized soak on the final library (progress lines keep the call alive)
O=gpurun_out/${1:-soak_long}; mkdir -p $O
{
for base in 1000 1100 1200 1300; do FUZZ_SEED0=$base timeout -k 10 400 python tests/fuzz_soak.py 100 2>&1 | tail -1; echo "  (seeds $base..+100 done)"; done
timeout -k 10 500 python tests/fuzz_soak.py 150 dynamic 2>&1 | tail -1
timeout -k 10 300 python tests/fuzz_soak.py 60 nodes 2>&1 | tail -1
} | tee $O/soak.txt
