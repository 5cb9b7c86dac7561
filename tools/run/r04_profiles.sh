#!/bin/bash
# the round's profile set: rocprofv3 stats + FETCH / WRITE passes of the bench command (raw rows, default config), the saturating-batch passes at
# 4096 and 512 sequences per launch, and the driver's invocation
O=gpurun_out/r04prof
mkdir -p $O
bash tools/run/prof_default.sh r04p > $O/prof_default.txt 2>&1 || { tail -20 $O/prof_default.txt; exit 1; }
tail -12 $O/prof_default.txt
for B in 4096 512; do
  mkdir -p gpurun_out/r04_sw$B
  bash tools/run/ep_sweep_prof.sh r04_sw$B $B 5 > $O/sw$B.txt 2>&1 || { tail -20 $O/sw$B.txt; exit 1; }
  head -3 gpurun_out/r04_sw$B/summary.txt
done
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("saturating",{}).get("frac"))
print("mirror", d.get("mirror_generate")); print("cycle", d.get("drafter_cycle"))
PY
