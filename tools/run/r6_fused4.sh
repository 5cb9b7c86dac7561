#!/bin/bash
O=gpurun_out/r6_fused; mkdir -p $O
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps ${STEPS:-200} --warmup ${WARM:-20} --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M' % (1e3*d['ms_per_step'], d['value']/1e6))"; }
for rep in 1 2; do
  echo -n "baseline (three launches, spec3, w1): "; run --commit-window 1 --fused-prepare 0
  echo -n "three launches, spec2, w1: "; run --commit-window 1 --fused-prepare 0 --spec-rows 2
  echo -n "fused spec2 w1: "; run --commit-window 1 --fused-prepare 1 --spec-rows 2
  echo -n "fused spec2 w2: "; run --commit-window 2 --fused-prepare 1 --spec-rows 2
  echo -n "fused spec2 g2 w1: "; run --groups 2 --commit-window 1 --fused-prepare 1 --spec-rows 2
  echo -n "fused spec2 w0: "; run --commit-window 0 --fused-prepare 1 --spec-rows 2
done
STEPS=20; WARM=5
for rep in 1 2 3; do
  echo -n "driver form baseline: "; run --fused-prepare 0
  echo -n "driver form fused spec2: "; run --fused-prepare 1 --spec-rows 2
done
