#!/bin/bash
# stream groups x GPU_MAX_HW_QUEUES: is the G >= 4 cliff the runtime's stream -> hardware-queue mapping?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2h}
mkdir -p $OUT
for q in 4 8 16; do for g in 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --groups $g > $OUT/b_q${q}_g$g.json 2> $OUT/b_q${q}_g$g.err || tail -3 $OUT/b_q${q}_g$g.err
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), d['config']['seqs_per_gpu'])
    except Exception as e: print(f,'ERR',e)
PY
