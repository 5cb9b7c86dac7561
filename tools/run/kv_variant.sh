#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2kv}
mkdir -p $OUT
for v in 0 10 40 21 22 23 43; do
  timeout -k 10 300 python bench.py --tuning kv_variant=$v --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras > $OUT/b_v$v.json 2> $OUT/b_v$v.err || tail -3 $OUT/b_v$v.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d.get('roofline',{}).get('avg_launch_ms',0)*1e3,1), {k:round(v['avg_launch_ms']*1e3,1) for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f,'ERR',e)
PY
