#!/bin/bash
O=gpurun_out/r6_fused; mkdir -p $O
run() { timeout -k 10 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras --extras-out "" "$@" 2>$O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step  %.3f M' % (1e3*d['ms_per_step'], d['value']/1e6))"; }
for rep in 1 2; do
  echo -n "64 in 4 groups (default): "; run
  echo -n "63 in 3 groups w1: "; run --seqs-per-gpu 63 --groups 3 --commit-window 1
  echo -n "66 in 3 groups w1: "; run --seqs-per-gpu 66 --groups 3 --commit-window 1
  echo -n "63 in 3 groups w0: "; run --seqs-per-gpu 63 --groups 3 --commit-window 0
  echo -n "60 in 5 groups w1: "; run --seqs-per-gpu 60 --groups 5 --commit-window 1
done
