#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2o}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_loop.py tests/test_gpu_window.py -x -q -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for g in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --fuse-o7 --groups $g > $OUT/b_fuse_g$g.json 2> $OUT/b_fuse_g$g.err || tail -3 $OUT/b_fuse_g$g.err
done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 6 --ep-sweep "" --no-extras --fuse-o7 --groups 2 --no-events > $OUT/b_fuse_g2_cpu.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras > $OUT/b_plain_g1.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d.get('roofline',{}).get('avg_launch_ms',0)*1e3,1), {k:round(v['avg_launch_ms']*1e3,1) for k,v in d.get('kernels',{}).items()}, d.get('cpu_baseline',{}).get('matches_gpu_token_stream'))
    except Exception as e: print(f,'ERR',e)
PY
