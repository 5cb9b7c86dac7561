#!/bin/bash
O=gpurun_out/groups
mkdir -p $O
for cfg in "63 3 nodes" "64 4 nodes" "64 4 chain" "48 3 nodes" "48 3 chain"; do
  set -- $cfg
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 200 --warmup 20 --seqs-per-gpu $1 --groups $2 --ep $3 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/n$1g$2$3.json 2> $O/n$1g$2$3.err || { tail -5 $O/n$1g$2$3.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$O/n$1g$2$3.json").read().strip().splitlines()[-1])
print("seqs $1 groups $2 ep $3:", round(d["ms_per_step"]*1e3,1), "us/step", round(d["value"]/1e6,3), "M tok/s", round(d["roofline"]["avg_launch_ms"]*1e3,1), {k: round(v["avg_launch_ms"]*1e3,1) for k,v in d["kernels"].items()})
PY
done
