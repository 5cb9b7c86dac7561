#!/bin/bash
O=gpurun_out/${1:-r5s}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_loop.py tests/test_gpu_fullsize_properties.py -m gpu -q -x > $O/test.log 2>&1; echo "tests rc=$?"; tail -5 $O/test.log
run() { timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 8 --extras-out "" "$@" 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(1e3*d['ms_per_step'],2), 'ep_us', round(1e3*d['roofline']['avg_launch_ms'],1), d['cpu_baseline']['matches_gpu_token_stream'])"; }
run
run
run --no-defer-kv
timeout -k 10 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out "" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench200', round(d['value']), round(1e3*d['ms_per_step'],2))"
