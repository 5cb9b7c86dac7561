#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2wk}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_nodes.py tests/test_gpu_loop.py -x -q -m gpu -k "nodes or serial" > $OUT/tests.log 2>&1
rc=$?; tail -6 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
for ep in chain walk nodes; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --groups 1 --no-fuse-o7 --spec-rows 0 --ep $ep > $OUT/b_$ep.json 2> $OUT/b_$ep.err || tail -3 $OUT/b_$ep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d.get('roofline',{}).get('avg_launch_ms',0)*1e3,2), round(d['roofline']['frac'],3), {k:round(v['avg_launch_ms']*1e3,1) for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f,'ERR',e)
PY
