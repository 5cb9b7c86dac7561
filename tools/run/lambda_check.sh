#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2t}
mkdir -p $OUT
i=0
for args in "--groups 3" "--groups 1" "--groups 1 --no-fuse-o7" "--groups 3 --no-fuse-o7"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --lantern-delta 5 $args > $OUT/b_$i.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$OUT/b_$i.json').read().strip().splitlines()[-1]); print('$args', round(d['value']), round(d['ms_per_step']*1e3,1), d['mean_accept_length'], d['per_step'])"
done
