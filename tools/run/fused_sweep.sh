#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2fa}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "fused_accept" > $OUT/tests.log 2>&1
rc=$?; tail -4 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
for cfgs in "1 0" "1 128" "2 0" "2 160" "3 0" "3 128" "4 0"; do set -- $cfgs; g=$1; w=$2
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --no-events --groups $g --fused-accept --fused-workers $w > $OUT/b_g${g}_w$w.json 2> $OUT/b_g${g}_w$w.err || tail -3 $OUT/b_g${g}_w$w.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), d['config']['seqs_per_gpu'], d['cpu_baseline'] if 'cpu_baseline' in d else '')
    except Exception as e: print(f,'ERR',e)
PY
