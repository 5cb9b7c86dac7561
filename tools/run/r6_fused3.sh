#!/bin/bash
O=gpurun_out/r6_fused; mkdir -p $O
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps ${STEPS:-200} --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M' % (1e3*d['ms_per_step'], d['value']/1e6))"; }
for rep in 1 2; do
  echo -n "g4 w1 fused: "; run --commit-window 1 --fused-prepare 1
  echo -n "g4 w2 fused: "; run --commit-window 2 --fused-prepare 1
  echo -n "g2 w1 fused: "; run --groups 2 --commit-window 1 --fused-prepare 1
  echo -n "g2 w0 fused: "; run --groups 2 --commit-window 0 --fused-prepare 1
  echo -n "g4 w1 fused spec2: "; run --commit-window 1 --fused-prepare 1 --spec-rows 2
  echo -n "g4 w1 fused spec4: "; run --commit-window 1 --fused-prepare 1 --spec-rows 4
  echo -n "g4 w1 fused spec6: "; run --commit-window 1 --fused-prepare 1 --spec-rows 6
done
