#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_generate_ref.py tests/test_gpu_generate.py tests/test_gpu_mirror.py tests/test_gpu_generate_lg.py tests/test_gpu_loop.py tests/test_gpu_solver.py -m gpu -q > $O/test.log 2>&1; echo "tests rc=$?"; tail -25 $O/test.log
for i in 1 2 3; do timeout -k 10 300 python tools/mirror_bench.py 300 > $O/mirror_$i.json 2> $O/mirror_$i.err || tail -5 $O/mirror_$i.err; cat $O/mirror_$i.json; done
