#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/cmp
mkdir -p $O
i=0
IFS=";" read -ra FL <<< "${CMP_FLAGS:---spec-rows 3}"; for flags in "${FL[@]}"; do
  i=$((i+1))
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras $flags > $O/b_$i.json 2> $O/b_$i.err || tail -3 $O/b_$i.err
  python3 - <<PY
import json
d=json.loads(open("$O/b_$i.json").read().strip().splitlines()[-1])
print("[$flags]", round(d["value"]), round(1e3*d["ms_per_step"],2), "us/step; epw", round(1e3*d["roofline"]["avg_launch_ms"],2), "us;", {k:round(1e3*v["avg_launch_ms"],1) for k,v in d.get("kernels",{}).items() if isinstance(v,dict) and "avg_launch_ms" in v})
PY
done
