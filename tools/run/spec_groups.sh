#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2sg}
mkdir -p $OUT
for g in 3 4; do for k in 3 5 8 12 16; do
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --groups $g --spec-rows $k > $OUT/b_g${g}_s$k.json 2> $OUT/b_g${g}_s$k.err || tail -3 $OUT/b_g${g}_s$k.err
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d.get('roofline',{}).get('avg_launch_ms',0)*1e3,1), {k:round(v['avg_launch_ms']*1e3,1) for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f,'ERR',e)
PY
