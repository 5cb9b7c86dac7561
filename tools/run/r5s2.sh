#!/bin/bash
O=gpurun_out/${1:-r5s2}; mkdir -p $O
run() { timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out "" "$@" 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=$LANTERN_GROUP_STREAM_SKIP hwq=$GPU_MAX_HW_QUEUES', round(d['value']), round(1e3*d['ms_per_step'],2), 'ep_us', round(1e3*d['roofline']['avg_launch_ms'],1))"; }
for s in 0 1 2 3 4 5 6 7; do export LANTERN_GROUP_STREAM_SKIP=$s; run; done
export LANTERN_GROUP_STREAM_SKIP=0
for q in 4 5 6 12 16; do export GPU_MAX_HW_QUEUES=$q; run; done
