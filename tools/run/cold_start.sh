#!/bin/bash
# first thing on a fresh box: does the driver's invocation reach steady state on a GPU that was idle?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2cs}
mkdir -p $OUT
for su in ${2:-0.5} 0.5 0.0; do
  sleep 20
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --spin-up $su > $OUT/b_$su.$RANDOM.json 2>> $OUT/err.txt
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/b_*.json'), key=os.path.getmtime):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1))
PY
