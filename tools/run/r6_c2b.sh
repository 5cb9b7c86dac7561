#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6c2b}; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_loop.py tests/test_gpu_generate_lg.py tests/test_gpu_parity.py tests/test_gpu_fullsize_properties.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras-out $O/bench_full.json > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['saturating']['frac'], d['extras'])
f=json.load(open('$O/bench_full.json')); c2=f['configs']['C2']; print('C2', c2['value'], c2['ms_per_step'], c2['kernel_ms'])"
