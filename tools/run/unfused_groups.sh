#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2ug}
mkdir -p $OUT
for cfgs in "3 chain" "3 nodes" "2 chain" "4 chain"; do set -- $cfgs; g=$1; ep=$2
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --groups $g --no-fuse-o7 --spec-rows 0 --ep $ep > $OUT/b_g${g}_$ep.json 2> $OUT/b_g${g}_$ep.err || tail -3 $OUT/b_g${g}_$ep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), d['config']['seqs_per_gpu'])
    except Exception as e: print(f,'ERR',e)
PY
