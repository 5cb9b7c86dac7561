#!/bin/bash
# C5's per-GPU share (8 sequences): which evaluate_posterior form / grouping is fastest; then the 2-rank one-device control-flow runs
O=gpurun_out/c5
mkdir -p $O
for cfg in "--ep chain --groups 1" "--ep chain --groups 2" "--ep nodes --groups 1" "--ep nodes --groups 2" "--ep chain --groups 1 --spec-rows 0"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 200 --warmup 20 --seqs-per-gpu 8 --cpu-seconds 0 --ep-sweep "" --no-extras $cfg > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
print("$cfg", round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,3), "M tok/s")
PY
done
