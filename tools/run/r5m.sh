#!/bin/bash
O=gpurun_out/${1:-r5m}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_loop.py tests/test_gpu_fullsize_properties.py tests/test_gpu_canary.py tests/test_gpu_configs.py -m gpu -q -x > $O/test.log 2>&1; echo "tests rc=$?"; tail -5 $O/test.log
for i in 1 2; do timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 8 --extras-out "" 2> $O/bench_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), round(1e3*d['ms_per_step'],2), 'ep_us', round(1e3*d['roofline']['avg_launch_ms'],1), d['kernels'], d['cpu_baseline'].get('matches_gpu_token_stream'))"; done
timeout -k 10 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out "" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench200', round(d['value']), round(1e3*d['ms_per_step'],2))"
