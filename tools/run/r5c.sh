#!/bin/bash
O=gpurun_out/${1:-r5c}; mkdir -p $O
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --ep-sweep "" --cpu-seconds 0 --extras-out $O/bench_full.json > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_full.json')); c=d['configs']
print(d['value'], d['ms_per_step'])
print('C2', c['C2']['workload'][-120:], c['C2']['value'], c['C2']['ms_per_step'])
for x in c['C4']: print('C4', x['workload'][-60:], x['value'], x['ms_per_step'])
print('dyn', d['dynamic_tree']['value'], d['dynamic_tree']['ms_per_step']); print(d.get('mirror_generate',{}).get('us_per_verify_step'))"
