#!/bin/bash
# round 5's randomized soaks against the CPU oracle on the final library (tests/fuzz_soak.py): static trees through both kernel sets, EAGLE-2 trees,
# O7 shapes, top-p inside the dense kernel, the static drafter's draws
O=gpurun_out/${1:-soak_r05}; mkdir -p $O
{
timeout -k 10 500 python tests/fuzz_soak.py 120 2>&1 | tail -3
timeout -k 10 400 python tests/fuzz_soak.py 80 dynamic 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 1500 o7 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 3000 o3 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 25 top_p 2>&1 | tail -2
timeout -k 10 300 python tests/fuzz_soak.py 1500 draws 2>&1 | tail -1
} | tee $O/soak.txt
