#!/bin/bash
O=gpurun_out/tp_try3
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_window.py tests/test_gpu_configs.py tests/test_gpu_loop.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
bash tools/run/tp_try2.sh "$@"
