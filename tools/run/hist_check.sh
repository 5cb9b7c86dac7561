#!/bin/bash
# the 16-bit / 32-copy first-pass histogram of the top-k select: every test that goes through it, then O7 alone and the bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-hist}
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/t.txt 2>&1 || { tail -30 $O/t.txt; exit 1; }
tail -2 $O/t.txt
timeout -k 10 200 python3 tools/o7_parts.py 64 2>&1 | grep -v amdgpu | tee $O/parts.txt
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --ep-sweep "" --no-extras > $O/b_$i.json 2> $O/b_$i.err || { tail -5 $O/b_$i.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$O/b_$i.json").read().strip().splitlines()[-1])
print("run $i:", round(d["value"]), round(1e3*d["ms_per_step"],2), "us/step; epw", round(1e3*d["roofline"]["avg_launch_ms"],2), "us;", {k:round(1e3*v["avg_launch_ms"],1) for k,v in d.get("kernels",{}).items() if isinstance(v,dict) and "avg_launch_ms" in v})
PY
done
