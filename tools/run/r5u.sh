#!/bin/bash
O=gpurun_out/${1:-r5u}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_generate_lg.py tests/test_gpu_generate.py tests/test_gpu_mirror.py tests/test_gpu_loop.py -m gpu -q -x > $O/test.log 2>&1; echo "tests rc=$?"; tail -25 $O/test.log
