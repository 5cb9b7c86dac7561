#!/bin/bash
O=gpurun_out/${1:-r5r}; mkdir -p $O
run() { timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 0 --extras-out "" "$@" 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(1e3*d['ms_per_step'],2), 'per-seq-us', round(1e3*d['ms_per_step']/d['config']['seqs_per_gpu'],3))"; }
for i in 1 2 3; do
run --groups 3 --seqs-per-gpu 63
run --groups 4 --seqs-per-gpu 64
done
