#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2q}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_loop.py -x -q -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for k in 0 1 2 3 5 8 12; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --fuse-o7 --spec-rows $k > $OUT/b_spec$k.json 2> $OUT/b_spec$k.err || tail -3 $OUT/b_spec$k.err
done
for k in 3 5; do for g in 2 3; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --fuse-o7 --spec-rows $k --groups $g > $OUT/b_spec${k}_g$g.json 2>/dev/null
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d.get('roofline',{}).get('avg_launch_ms',0)*1e3,1), {k:round(v['avg_launch_ms']*1e3,1) for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f,'ERR',e)
PY
