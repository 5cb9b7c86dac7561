#!/bin/bash
O=gpurun_out/commit
mkdir -p $O
for cfg in "2 8" "2 4" "2 16" "0 16"; do
  set -- $cfg
  LANTERN_COMMIT_DEBUG=$1 LANTERN_COMMIT_TEAM=$2 timeout -k 10 400 python3 bench.py --gpus 1 --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/u$1t$2.json 2> $O/u$1t$2.err || { tail -5 $O/u$1t$2.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$O/u$1t$2.json").read().strip().splitlines()[-1])
print("dbg $1 team $2:", round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,3), "M tok/s")
PY
done
