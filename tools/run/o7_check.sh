#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2o7}
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_window.py tests/test_gpu_parity.py tests/test_gpu_loop.py tests/test_gpu_fullsize_properties.py tests/test_gpu_nodes.py -x -q -m gpu > $OUT/tests.log 2>&1
rc=$?; tail -4 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tests/fuzz_soak.py 1500 o7 > $OUT/o7_soak.txt 2>&1; tail -2 $OUT/o7_soak.txt
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --groups 1 --no-fuse-o7 --spec-rows 0 > $OUT/b_unfused.json 2> $OUT/b_unfused.err || tail -3 $OUT/b_unfused.err
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 5 --ep-sweep "" --no-extras > $OUT/b_default.json 2> $OUT/b_default.err || tail -3 $OUT/b_default.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d.get('roofline',{}).get('avg_launch_ms',0)*1e3,2), round(d['roofline']['frac'],3), {k:(round(v['avg_launch_ms']*1e3,1), round(v['frac'],3)) for k,v in d.get('kernels',{}).items()}, d.get('cpu_baseline',{}).get('matches_gpu_token_stream'))
    except Exception as e: print(f,'ERR',e)
PY
