#!/bin/bash
# the driver's form (20 timed steps behind 5 warm-up steps), alternating: the prepare stage inside the chain launch (default) against its own launch
O=gpurun_out/r6_fused; mkdir -p $O
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M  walk %.1f us' % (1e3*d['ms_per_step'], d['value']/1e6, 1e3*d['roofline']['avg_launch_ms']))"; }
for rep in 1 2 3 4; do
  echo -n "three launches (--fused-prepare 0): "; run --fused-prepare 0
  echo -n "two launches (default): "; run
done
