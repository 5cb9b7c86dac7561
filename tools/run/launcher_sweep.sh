#!/bin/bash
# worker-thread launches: stream groups x threads (GPU_MAX_HW_QUEUES follows the group count)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2l}
mkdir -p $OUT
QQ=${2:-16}
for cfgs in "4 0" "4 4" "6 0" "6 3" "6 6" "8 0" "8 4" "8 8" "12 6" "12 12" "16 8"; do set -- $cfgs; g=$1; t=$2
  q=$QQ
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --groups $g --launch-threads $t > $OUT/b_g${g}_t$t.json 2> $OUT/b_g${g}_t$t.err || tail -3 $OUT/b_g${g}_t$t.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), d['config']['seqs_per_gpu'])
    except Exception as e: print(f,'ERR',e)
PY
