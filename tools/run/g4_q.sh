#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2g4}
mkdir -p $OUT
for q in 5 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --groups 4 > $OUT/b_q$q.json 2> $OUT/b_q$q.err || tail -3 $OUT/b_q$q.err
done
GPU_MAX_HW_QUEUES=6 timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --groups 3 > $OUT/b_g3_q6.json 2> $OUT/b_g3_q6.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), d['config']['seqs_per_gpu'], {k: round(v['ms_per_step']*1e3,1) for k,v in d['other_groupings'].items()}, round(d['lambda_mode']['ms_per_step']*1e3,1), round(d['dynamic_tree']['ms_per_step']*1e3,1))
    except Exception as e: print(f,'ERR',e)
PY
