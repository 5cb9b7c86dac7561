#!/bin/bash
# the LlamaGen / Anole / Lumina generate() mirrors against the reference-recorded runs
cd $GRAFT_REPO_ROOT
O=gpurun_out/gen
mkdir -p $O
timeout -k 10 800 python3 -m pytest tests/test_gpu_generate_lg.py tests/test_gpu_generate_ref.py tests/test_gpu_generate.py tests/test_gpu_mirror.py tests/test_gpu_configs.py -x -q -m gpu > $O/t_gen.txt 2>&1 || { tail -60 $O/t_gen.txt; exit 1; }
tail -3 $O/t_gen.txt
