#!/bin/bash
# harness tests + bench over evaluate_posterior kernel x stream groups (scratch output under gpurun_out/)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2k}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_loop.py tests/test_gpu_fullsize_properties.py tests/test_gpu_nodes.py -x -q -m gpu > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
for ep in nodes chain; do for g in 1 2 4; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --ep $ep --groups $g --no-events > $OUT/b_${ep}_g$g.json 2> $OUT/b_${ep}_g$g.err || tail -3 $OUT/b_${ep}_g$g.err
done; done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --ep chain --groups 2 --no-events --python-launch > $OUT/b_chain_g2_py.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 8 --ep-sweep "" --no-extras --ep nodes --groups 2 > $OUT/b_nodes_g2_cpu.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), d.get('cpu_baseline',{}).get('matches_gpu_token_stream'))
    except Exception as e: print(f,'ERR',e)
PY
