#!/bin/bash
O=gpurun_out/${1:-r5d2}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_properties.py tests/test_gpu_loop.py  -m gpu -q -x > $O/test.log 2>&1; echo "tests rc=$?"; tail -3 $O/test.log
