#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2pn}
mkdir -p $OUT
for nt in 512 1024; do for rep in a b; do
  timeout -k 10 300 python bench.py --tuning prep_nt=$nt --steps 100 --warmup 10 --cpu-seconds 5 --ep-sweep "" --no-extras > $OUT/b_${nt}_$rep.json 2> $OUT/b_${nt}_$rep.err || tail -3 $OUT/b_${nt}_$rep.err
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1), 'ep', round(d.get('roofline',{}).get('avg_launch_ms',0)*1e3,1), {k:round(v['avg_launch_ms']*1e3,1) for k,v in d.get('kernels',{}).items()}, d['cpu_baseline']['matches_gpu_token_stream'])
    except Exception as e: print(f,'ERR',e)
PY
