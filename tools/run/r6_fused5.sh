#!/bin/bash
O=gpurun_out/r6_fused; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_loop.py -x -q -k "prepare_stage_inside" > $O/t.txt 2>&1; tail -2 $O/t.txt
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps ${STEPS:-200} --warmup ${WARM:-20} --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" --commit-window 1 --fused-prepare 1 "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M' % (1e3*d['ms_per_step'], d['value']/1e6))"; }
for rep in 1 2; do
  for spec in 2 3 4 6; do echo -n "helpers 1 spec $spec: "; run --spec-rows $spec; done
  for spec in 3 5; do echo -n "helpers 2 spec $spec: "; run --spec-rows $spec --tuning epw_fused_helpers=2; done
done
