#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/kvu
mkdir -p $O
for u in 1 2 4; do
  timeout -k 10 300 python3 bench.py --tuning kv_u=$u --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/b_$u.json 2> $O/b_$u.err || tail -3 $O/b_$u.err
  python3 - <<PY
import json
d=json.loads(open("$O/b_$u.json").read().strip().splitlines()[-1])
print("KV_U=$u", round(d["value"]), round(1e3*d["ms_per_step"],2), "us/step; epw", round(1e3*d["roofline"]["avg_launch_ms"],2), "us;", {k:round(1e3*v["avg_launch_ms"],1) for k,v in d.get("kernels",{}).items() if isinstance(v,dict) and "avg_launch_ms" in v})
PY
done
