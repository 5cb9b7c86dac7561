#!/bin/bash
# node-kernel parity tests + a short bench of both evaluate_posterior forms (scratch output under gpurun_out/)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_nodes.py -x -q -m gpu > $OUT/nodes_tests.log 2>&1
tail -4 $OUT/nodes_tests.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 5 --ep-sweep "" --no-extras > $OUT/bench_nodes.json 2> $OUT/bench_nodes.err || tail -5 $OUT/bench_nodes.err
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --seqs-per-gpu 8 > $OUT/bench_nodes_8.json 2> /dev/null
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --ep-sweep "" --no-extras --ep chain > $OUT/bench_chain.json 2> $OUT/bench_chain.err
python - <<PY
import json
for f in ['nodes','nodes_8','chain']:
    try:
        d=json.loads(open('$OUT/bench_%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d['roofline']['avg_launch_ms']*1e3,1), {k:round(v['avg_launch_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('cpu_baseline',{}).get('matches_gpu_token_stream'))
    except Exception as e: print(f,'ERR',e)
PY
