#!/bin/bash
O=gpurun_out/${1:-r5z}; mkdir -p $O
timeout -k 10 600 python tests/fuzz_soak.py 1500 draws 2>&1 | tail -6 | tee $O/draws_soak.txt
timeout -k 10 600 python -m pytest tests/test_gpu_more.py -m gpu -q -x -k "head_sample or mask_left" 2>&1 | tail -3
