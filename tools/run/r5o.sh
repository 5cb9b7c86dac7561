#!/bin/bash
O=gpurun_out/${1:-r5o}; mkdir -p $O
for sr in 3 6 10 16 26; do
timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 20 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out "" --spec-rows $sr 2> $O/b_$sr.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec_rows $sr', round(d['value']), round(1e3*d['ms_per_step'],2), 'ep_us', round(1e3*d['roofline']['avg_launch_ms'],1), 'prep_us', round(1e3*d['kernels']['cfg_mask_topk']['avg_launch_ms'],1), 'kv_us', round(1e3*d['kernels']['kv_gather']['avg_launch_ms'],1))"
done
