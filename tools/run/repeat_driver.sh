#!/bin/bash
# the driver's invocation five times in fresh processes: run-to-run spread of the headline
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2rep}
mkdir -p $OUT
for i in 1 2 3 4 5; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --ep-sweep "" --cpu-seconds 0 --no-events > $OUT/b_$i.json 2>/dev/null
done
python - <<PY
import json,glob
v=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob('$OUT/b_*.json'))]
print("accepted tokens/s:", [round(d['value']) for d in v]); print("us/step:", [round(d['ms_per_step']*1e3,1) for d in v])
PY
