#!/bin/bash
O=gpurun_out/groups
mkdir -p $O
for rep in 1 2 3; do
for cfg in "63 3" "64 4"; do
  set -- $cfg
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --seqs-per-gpu $1 --groups $2 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/d$1g$2.json 2> $O/d$1g$2.err || { tail -5 $O/d$1g$2.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$O/d$1g$2.json").read().strip().splitlines()[-1])
print("20/5 seqs $1 groups $2:", round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,3), "M tok/s")
PY
done
done
