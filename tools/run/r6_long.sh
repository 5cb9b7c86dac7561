#!/bin/bash
# long timed runs (cross the image-end bound of the harness at ~390 steps): the step rate must not depend on the run length
cd $GRAFT_REPO_ROOT
for cfg in "400 1" "400 0" "300 1" "800 1"; do set -- $cfg
  timeout -k 10 300 python3 bench.py --steps $1 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --commit-window $2 > gpurun_out/long_$1_$2.json 2> gpurun_out/long_$1_$2.err
  python3 -c "import json; d=json.load(open('gpurun_out/long_$1_$2.json')); print('steps $1 window $2: us/step %.2f' % (1e3*d['ms_per_step']))"
done
