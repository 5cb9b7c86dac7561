#!/bin/bash
O=gpurun_out/r6_fused; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_loop.py -x -q -k "prepare_stage_inside" > $O/t.txt 2>&1; tail -3 $O/t.txt
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps ${STEPS:-200} --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M' % (1e3*d['ms_per_step'], d['value']/1e6))"; }
for rep in 1 2; do
  for w in 1 0; do
    for f in 0 1; do echo -n "rep $rep window $w fused $f: "; run --commit-window $w --fused-prepare $f; done
  done
done
