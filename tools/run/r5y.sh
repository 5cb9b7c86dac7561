#!/bin/bash
O=gpurun_out/${1:-r5y}; mkdir -p $O
timeout -k 10 600 python tests/fuzz_soak.py 400 draws 2>&1 | tail -6 | tee $O/draws_soak.txt
run() { timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out "" "$@" 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(1e3*d['ms_per_step'],2), 'ep_us', round(1e3*d['roofline']['avg_launch_ms'],1))"; }
run
GPU_MAX_HW_QUEUES=16 run --groups 8
run --groups 2
run
GPU_MAX_HW_QUEUES=16 run --groups 8
