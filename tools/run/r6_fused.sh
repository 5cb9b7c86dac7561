#!/bin/bash
# the prepare stage inside the chain launch (bench.py --fused-prepare 1) against three launches, with and without commit turn-taking, alternating on one box
O=gpurun_out/r6_fused; mkdir -p $O
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps ${STEPS:-200} --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M  walk %.1f us' % (1e3*d['ms_per_step'], d['value']/1e6, 1e3*d['roofline']['avg_launch_ms']))"; }
for rep in 1 2; do
  for w in 1 0; do
    for f in 0 1; do echo -n "rep $rep window $w fused $f: "; run --commit-window $w --fused-prepare $f; done
  done
done
STEPS=20
for rep in 1 2 3; do for f in 0 1; do echo -n "driver form, fused $f: "; timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" --fused-prepare $f 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step' % (1e3*d['ms_per_step']))"; done; done
