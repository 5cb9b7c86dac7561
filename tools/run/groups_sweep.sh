#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2p}
mkdir -p $OUT
for ep in chain nodes; do for g in 2 3 4; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --ep $ep --groups $g $2 > $OUT/b_${ep}_g$g.json 2> $OUT/b_${ep}_g$g.err || tail -3 $OUT/b_${ep}_g$g.err
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1))
    except Exception as e: print(f,'ERR',e)
PY
