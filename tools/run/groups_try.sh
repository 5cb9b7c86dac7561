#!/bin/bash
O=gpurun_out/groups
mkdir -p $O
for rep in 1 2 3; do
for intr in 1 0; do
  HSA_ENABLE_INTERRUPT=$intr timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/i$intr.json 2> $O/i$intr.err || { tail -5 $O/i$intr.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$O/i$intr.json").read().strip().splitlines()[-1])
print("20/5 HSA_ENABLE_INTERRUPT=$intr:", round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,3), "M tok/s")
PY
done
done
